#!/usr/bin/env python3
"""bench.py -- end-to-end Mpix/s of the hot path (dense flow -> semi-Lagrangian Sobel edge field ->
marker-controlled watershed) on 5424 x 5424 GOES-16 full-disk-sized frames, one process per GPU.

  python bench.py --gpus N --steps K --warmup W [--config F|V|F3|C|window]
  N > 1 under a launcher (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...): the
  ranks come from RANK / LOCAL_RANK / WORLD_SIZE.  N > 1 WITHOUT a launcher: this process stays off the GPU, starts N
  fresh rank processes itself (one per GPU, rendezvous on 127.0.0.1) and relays rank 0's JSON line.

A step is one pass of the hot path over one STACK resident in HBM -- by default BASELINE.json's config F, 144 frames of
5424 x 5424 -- processed the production way (scripts/dcc_detect_goes.py:153): twelve time windows sharing four frames:
    create_flow(Farneback, vr_steps=1, smoothing_passes=1, interp_method="cubic") over the stack's 143 frame pairs, once
    per window:  Flow.window_view (= the Flow create_flow(window) would give, bit for bit: only the end frames differ)
                 seeds (linearise_field -> binary_erosion -> label: window-local component ids; -1 background)
                 Flow.sobel(uphill, cubic, float64) -> combined edge field
                 Flow.watershed(connectivity 1)
and then the label ids of all windows stitched by the reference's overlap rule (linking.py:49-161), all inside the
timed region.  `value` counts the DELIVERED frames (144 per step), not the 188 window frames flooded.
Under --gpus N every rank holds its own 144-frame segment of ONE synthetic sequence (weak scaling; consecutive segments
share four frames bit for bit); the stitch then runs over all windows of all ranks with one neighbour message per rank
boundary and all-gathers of the pair lists (tobac_flow_amd/parallel.py: stitch_rank_windows).
--config window --frames 12 is the round-1/2 sub-report (one 12-frame window per step).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E peak (MI355X_MICROARCH.md); the library's own 16-byte-per-lane copy timed in this run (roofline.practical_peak) is the practical ceiling


# library profile name -> kernel symbol prefix in the rocprofv3 counter files
_KERNEL_SYMBOL = {"fb_iteration_fused": "void k_fb_iter<", "vr_sor": "void k_vr_sor_tile<false", "vr_system": "void k_vr_system<true, false>",
                  "sobel": "void k_sobel27<2, double, 2, true>", "fb_polyexp": "k_fb_polyexp", "vr_prepare": "k_vr_prepare",
                  "smooth_flow": "void k_smooth<", "ws_relax_sweep": "k_ws_sweep", "fb_gaussian_blur": "k_fb_blur", "fb_resize": "k_fb_resize",
                  "binary_morph": "k_binary_morph", "ws_labels": "k_ws_labels", "to8bit_pair": ("k_minmax", "k_to8bit"),
                  "ws_setup": ("k_ws_relevant", "k_ws_compact", "k_ws_cid_tiles", "k_ws_count_tiles", "k_ws_classify", "k_ws_edge_masks")}
def _latest(*names):
    """the newest committed recording that exists (a round whose counter passes could not be re-recorded keeps the last one)"""
    for n in names:
        if os.path.exists(os.path.join(ROOT, n)):
            return n
    return names[-1]


_TRAFFIC_FILE = _latest("profiles/round6_pmc_traffic_bench.json", "profiles/round5_pmc_traffic_bench.json", "profiles/round4_pmc_traffic_bench.json")
_VALU_FILE = _latest("profiles/round6_pmc_valu_bench.json", "profiles/round5_pmc_valu_bench.json", "profiles/round4_pmc_valu_bench.json")
_CLOCK_GHZ_DEFAULT = 2.1       # GRBM_GUI_ACTIVE / 8 / duration under k_fb_iter; used when a pass has no timestamps


def _matching(doc, profile_name):
    """the recorded kernels of one profile name: every kernel whose symbol starts with (one of) its prefix(es) -- a profile
    name can stand for several kernels (the flood's sweeps: k_ws_sweep_a + k_ws_sweep_chain; the blur's variants)"""
    pre = _KERNEL_SYMBOL[profile_name]
    pre = pre if isinstance(pre, tuple) else (pre,)
    pre = pre + tuple("void " + q for q in pre if not q.startswith("void "))
    return [v for name, v in doc["kernels"].items() if name.startswith(pre)]


def _traffic_ratio(profile_name, doc):
    """measured HBM bytes (FETCH_SIZE corrected + WRITE_SIZE) over ALGORITHMIC bytes of one profile name in the recorded
    counter passes -- a property of the kernel(s), independent of the size of the run (None if not recorded)"""
    try:
        ks = _matching(doc, profile_name)
        rec = doc["recorded_on"]
        alg = rec["algorithmic_bytes_per_launch"][profile_name] * rec["launches"][profile_name]
        moved = sum(((k["fetch_bytes_corrected_per_launch"] or 0.0) + (k["write_bytes_per_launch"] or 0.0)) * k["launches"] for k in ks)
        return moved / alg if (ks and alg > 0) else None
    except (KeyError, TypeError, ZeroDivisionError):
        return None


def _valu_floor(profile_name, alg_bytes_total, doc):
    """VALU ISSUE FLOOR of one kernel over this run, in ms: the wave-level VALU instructions the committed counter pass
    (`rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE -- python3 bench.py --frames 62 ...`, tools/pmc_valu_json.py)
    counted per ALGORITHMIC byte of that kernel, times this run's algorithmic bytes, times 4 cycles per wave64 instruction,
    over 1024 SIMDs at the clock the pass measured under that kernel.  float64 instructions occupy a SIMD for eight cycles,
    so the real floor of the float64-heavy kernels (Sobel, polynomial expansion) is higher.  None if the pass lacks the kernel."""
    try:
        ks = _matching(doc, profile_name)
        rec_bytes = doc["recorded_on"]["algorithmic_bytes_per_launch"][profile_name] * doc["recorded_on"]["launches"][profile_name]
        # (several kernels under one name: each one's instructions at the clock it ran at)
        cycles_over_clock = sum(k["valu_wave_instructions_per_launch"] * k["launches"] / ((k.get("clock_ghz_under_this_kernel") or _CLOCK_GHZ_DEFAULT) * 1e9) for k in ks)
        if not ks or rec_bytes <= 0:
            return None
        return cycles_over_clock / rec_bytes * alg_bytes_total * doc.get("cycles_per_wave64_instruction", 4) / doc.get("simds", 1024) * 1e3
    except (KeyError, TypeError, ZeroDivisionError):
        return None
_TRAFFIC_WORKLOAD = ("F", 144, 5424, 5424, 1)      # (config, frames, height, width, vr_steps) the counter passes were recorded on


def _traffic(profile_name, alg_bytes_per_launch):
    """HBM bytes per launch of one kernel from the committed rocprofv3 counter passes (`rocprofv3 --pmc FETCH_SIZE -- python3
    bench.py --frames 62 --n-windows 5 --steps 1 --warmup 0 --no-cpu-baseline` and the same with WRITE_SIZE: separate
    passes, FETCH_SIZE doubled for wide reads as MI355X_MICROARCH.md prescribes; summarised by tools/pmc_traffic_json.py).
    The passes run a 62-frame stack of the same frame size and stage settings (a full 144-frame step has more dispatches
    than rocprofiler's counter collection survives): the bytes of a launch are proportional to the frame pairs in it, so
    the recorded per-launch figure is scaled by the ratio of the ALGORITHMIC bytes per launch of this run and of the
    recorded one (both in the file / this run's profile).  The benchmark cannot run the profiler on itself, so the figure
    belongs to the build the file was recorded with; None if the file or the kernel is missing."""
    try:
        with open(os.path.join(ROOT, _TRAFFIC_FILE)) as fh:
            doc = json.load(fh)
        k = _matching(doc, profile_name)[0]                    # (the dominant kernel has one symbol)
        rec = doc.get("recorded_on") or {}
        scale = 1.0
        per_kernel = rec.get("algorithmic_bytes_per_launch")
        rec_alg = per_kernel.get(profile_name) if isinstance(per_kernel, dict) else None
        if rec_alg:
            scale = alg_bytes_per_launch / float(rec_alg)
        return {"bytes_per_launch": (k["fetch_bytes_corrected_per_launch"] + k["write_bytes_per_launch"]) * scale,
                "fetch_bytes_raw": k["fetch_bytes_raw_per_launch"] * scale, "fetch_bytes_corrected": k["fetch_bytes_corrected_per_launch"] * scale,
                "write_bytes": k["write_bytes_per_launch"] * scale, "launches_profiled": k["launches"], "source": _TRAFFIC_FILE,
                "scaled_by_pairs_per_launch": round(scale, 4), "recorded_on": rec.get("workload")}
    except (OSError, ValueError, KeyError, TypeError, AttributeError, StopIteration, IndexError):
        return None


def cpu_baseline(seed, vr_steps=1):
    """The oracle (CPU restatement of the reference's cv2/numpy/Cython path, kind = "port") timed on the host cores of
    this box on a bounded sample of the same workload: a 9 x 1536 x 1536 stack (round 5: 16 Farneback tasks -- the CPU share
    of a one-GPU box of this pool; rounds 1 - 4 timed a 5-frame stack with eight tasks.  17 frames / 32 threads were measured too:
    flow 5.6 s and Sobel 13.2 s on 32 threads, then 30.2 s of sequential heap flood on one: 0.82 Mpix/s in 49 s -- the flood, one
    thread in the reference too, dominates any wider sample and a 49-second baseline is outside the benchmark's time budget).  The order-independent stages run on a thread pool (Farneback and refinement: one task per frame pair and
    direction; Sobel: one task per frame -- the C restatements and numpy release the GIL); the heap flood is sequential by
    construction (one thread, like the reference's).  TF_BENCH_CPU_FRAMES overrides the number of frames."""
    import numpy as np
    import scipy.ndimage as ndi
    import ctypes
    import warnings
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import blob_sequence
    from oracle import _lib as ol, np_ops, ws_oracle
    T, H, W = int(os.environ.get("TF_BENCH_CPU_FRAMES", "9")), 1536, 1536
    rng = np.random.default_rng(seed)
    bt = blob_sequence(rng, T, H, W, n_blobs=36)
    L = ol.lib()
    L.oracle_farneback.restype = ctypes.c_int
    threads = max(1, min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 2 * (T - 1)))

    def fb(ab):
        a, b = ab
        out = np.zeros((H, W, 2), np.float32)
        L.oracle_farneback(ol.ptr(np.ascontiguousarray(a), ctypes.c_uint8), ol.ptr(np.ascontiguousarray(b), ctypes.c_uint8),
                           H, W, ol.ptr(out, ctypes.c_float), 5, ctypes.c_double(0.5), 13, 10, 5, ctypes.c_double(1.1))
        return out
    t0 = time.perf_counter()
    fw = np.full((T, H, W, 2), np.nan, np.float32)
    bw = np.full((T, H, W, 2), np.nan, np.float32)
    with warnings.catch_warnings(), ThreadPoolExecutor(threads) as pool:
        warnings.simplefilter("ignore")
        p8 = [np_ops.to_8bit(np_ops.linear_norm(bt[i:i + 2].copy()), 0, 1) for i in range(T - 1)]
        pairs = [(p[0], p[1]) for p in p8] + [(p[1], p[0]) for p in p8]
        raw = list(pool.map(fb, pairs))
        if vr_steps > 0:                                     # flow.py:513-519: one refinement per direction
            raw = list(pool.map(lambda k: np_ops.variational_refinement(pairs[k][0], pairs[k][1], raw[k]), range(len(pairs))))
        sm = list(pool.map(lambda i: np_ops.smooth_flow_step(raw[i], raw[T - 1 + i], "cubic"), range(T - 1)))
        for i, (f, b) in enumerate(sm):
            fw[i], bw[i + 1] = f, b
        fw[-1], bw[0] = -bw[-1], -fw[0]
        fw, bw = np.clip(fw, -20, 20), np.clip(bw, -20, 20)
        t_flow = time.perf_counter() - t0
        lin = np.clip((bt - 270.0) / (250.0 - 270.0), 0, 1).astype(np.float32)

        def sobel_frame(i):                                  # frame i needs frames i-1 .. i+1 (convolve.py:305-330)
            lo, hi = max(i - 1, 0), min(i + 2, T)
            return np_ops.sobel(lin[lo:hi], fw[lo:hi], bw[lo:hi], "cubic", None, np.nan, "uphill")[i - lo]
        edges = np.stack(list(pool.map(sobel_frame, range(T))))
        edges[edges > 0] += 1
        edges = edges - lin
        t_sobel = time.perf_counter() - t0 - t_flow
        s = ndi.generate_binary_structure(3, 1) * np.array([0, 1, 0])[:, None, None].astype(bool)
        markers = ndi.label(ndi.binary_erosion(lin >= 1, structure=s))[0].astype(np.int32)   # component-labelled seeds
        markers[ndi.binary_erosion(lin <= 0, structure=np.ones([3, 3, 3]), border_value=1)] = -1
        want_labels = ws_oracle.watershed(fw, bw, edges, markers.astype(np.int32), None, 1)
    dt = time.perf_counter() - t0
    # the oracle as the checker (outside its timed window): the library on the same sample, stage by stage, bit for bit
    check = None
    try:
        import tobac_flow_amd.flow as tf
        from tobac_flow_amd.detection import get_combined_edge_field
        flow = tf.create_flow(bt, model="Farneback", vr_steps=vr_steps, smoothing_passes=1, interp_method="cubic")
        got_e = get_combined_edge_field(flow, lin)
        got_l = flow.watershed(got_e, markers.astype(np.int32), connectivity=ndi.generate_binary_structure(3, 1))
        check = {"flow_bit_identical": bool(np.array_equal(flow.forward_flow, fw, equal_nan=True) and np.array_equal(flow.backward_flow, bw, equal_nan=True)),
                 "edge_field_bit_identical": bool(np.array_equal(np.asarray(got_e), edges, equal_nan=True)),
                 "labels_bit_identical": bool(np.array_equal(np.asarray(got_l), want_labels))}
    except Exception as e:                                   # the baseline figure stands on its own
        check = {"error": f"{type(e).__name__}: {e}"}
    return {"value": round(T * H * W / dt / 1e6, 4), "unit": "Mpix/s", "cores": threads, "kind": "port", "library_vs_oracle_on_the_sample": check,
            "cores_available": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), "host_cores": os.cpu_count(),
            "sample": f"{T}x{H}x{W} synthetic stack, same stage sequence, oracle (C/numpy restatement of the "
                      f"cv2+scipy+Cython path{', with the refinement' if vr_steps > 0 else ''}): flow {t_flow:.1f} s and Sobel {t_sobel:.1f} s on {threads} threads, flood "
                      f"{dt - t_flow - t_sobel:.1f} s on one (sequential heap), {dt:.1f} s in all; host has {os.cpu_count()} cores"}


def cv2_parity(seed=20240601):
    """SURVEY.md section 7 hard part 2 / section 8d: Farneback, refinement and remap values are pinned only where OpenCV exists.
    Whatever part of it this box has is compared with the library on one synthetic frame pair: cv2.optflow (contrib) for
    Farneback, cv2.VariationalRefinement (the main `video` module: checked independently of optflow) and cv2.remap;
    otherwise the status says so."""
    try:
        import cv2
    except Exception as e:                                   # ImportError on this image
        return {"status": "parity unpinned", "reason": f"cv2 unavailable on this box ({type(e).__name__})"}
    import numpy as np
    from tobac_flow_amd.utils.flow_utils import FarnebackFlow, warp_flow
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import blob_sequence
    rng = np.random.default_rng(seed)
    bt = blob_sequence(rng, 2, 768, 1024, n_blobs=24)
    lo, hi = np.nanmin(bt), np.nanmax(bt)
    p8 = np.clip((bt - lo) / (hi - lo) * 255, 0, 255).astype(np.uint8)
    got = FarnebackFlow().calc(p8[0], p8[1], None)
    out = {"status": "pinned", "cv2": cv2.__version__}
    try:
        ref = cv2.optflow.createOptFlow_Farneback().calc(p8[0], p8[1], None)
        out["farneback_max_abs_diff"] = float(np.abs(ref - got).max())
    except Exception as e:                                   # a cv2 build without the contrib modules
        ref = got                                            # the other comparisons take the library's own vectors as their input
        out["status"] = "partly pinned"
        out["farneback_max_abs_diff"] = None
        out["farneback_reason"] = f"cv2.optflow unavailable ({type(e).__name__})"
    img = bt[0].astype(np.float32)
    locs = ref.copy()                                        # the map as utils/flow_utils.py:84-87 builds it
    locs[:, :, 0] += np.arange(img.shape[1])
    locs[:, :, 1] += np.arange(img.shape[0])[:, np.newaxis]
    for name, inter in (("nearest", cv2.INTER_NEAREST), ("linear", cv2.INTER_LINEAR), ("cubic", cv2.INTER_CUBIC),
                        ("lanczos", cv2.INTER_LANCZOS4)):
        want = cv2.remap(img, locs, None, inter, None, cv2.BORDER_CONSTANT, np.nan)
        have = warp_flow(img, ref, method=name)
        both = np.isfinite(want) & np.isfinite(have)
        out[f"remap_{name}_max_abs_diff"] = float(np.abs(want - have)[both].max())
        out[f"remap_{name}_nan_mask_equal"] = bool(np.array_equal(np.isnan(want), np.isnan(have)))
    # cv2.VariationalRefinement (flow.py:359, 513-519): the same refinement of the same input flow
    vr = getattr(cv2, "VariationalRefinement", None) or getattr(getattr(cv2, "optflow", None), "VariationalRefinement", None)
    if vr is not None:
        from tobac_flow_amd.flow import VariationalRefinement
        want = (vr.create() if hasattr(vr, "create") else vr_create(cv2)).calc(p8[0], p8[1], ref.copy())
        have = VariationalRefinement.create().calc(p8[0], p8[1], ref.copy())
        out["varref_max_abs_diff"] = float(np.abs(np.asarray(want) - np.asarray(have)).max())
    else:
        out["varref_max_abs_diff"] = None
    return out


def vr_create(cv2):
    return cv2.VariationalRefinement_create()


def launch_ranks(a):
    """--gpus N > 1 without a launcher: start N rank processes (fresh interpreters; this parent never touches the GPU)
    and relay their output.  Rank 0 prints the JSON line."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    codes = [p.wait() for p in procs]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return max(abs(c) for c in codes)


CONFIGS = {     # BASELINE.json configs: frames, height, width, windows, channels
    "F": (144, 5424, 5424, 12, 1),      # GOES-16 ABI full-disk C13, 144 frames, 1 x MI355X  (the metric's configuration)
    "V": (288, 3712, 3712, 24, 1),      # SEVIRI full-disk, 288 frames
    "F3": (288, 5424, 5424, 24, 3),     # full-disk, three channels (offsets 0 / -2 / -4 K) sharing ONE Flow
    "C": (24, 1500, 2500, 2, 1),        # GOES-16 CONUS, 24 frames
    "window": (12, 5424, 5424, 1, 1),   # one window per step (sub-report; --frames)
}
CHANNEL_OFFSETS = (0.0, -2.0, -4.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="F", help="BASELINE.json configuration (default F = the metric's)")
    ap.add_argument("--frames", type=int, default=None, help="frames of the stack per GPU per step (default: the config's)")
    ap.add_argument("--n-windows", type=int, default=None, help="time windows the stack is processed in (default: the config's)")
    ap.add_argument("--vr-steps", type=int, default=1,
                    help="create_flow(vr_steps=...): 1 = the setting of the reference's drop-in scripts "
                         "(scripts/dcc_detect_goes.py:164-166), 0 = no variational refinement")
    ap.add_argument("--overlap", type=int, default=4,
                    help="frames consecutive windows (and ranks) share; the stitch compares all but the first and last of them")
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--tie-order", choices=["raster", "reference"], default="reference",
                    help="equal-valued markers whose pop order decides a label: the reference heap's own order (default = the "
                         "library's default, on_ambiguous='reference': labels bit for bit the reference's; a sequential host replay "
                         "per flood that has such ties, run on worker threads beside the next windows' device work) or raster order "
                         "(such voxels are counted in `watershed`)")
    ap.add_argument("--inflight", type=int, default=12,
                    help="floods in flight per rank at most (each owns ~8 GB of scratch at 16 x 5424^2; the first step lowers the "
                         "number to what the free device memory holds): window k's host replay runs on a worker thread while the "
                         "device floods windows k+1 ...; 1 = strictly one after the other")
    ap.add_argument("--no-stream-windows", dest="stream_windows", action="store_false",
                    help="begin the windows only after the whole stack's flow (default: a window is begun as soon as the flow batch "
                         "with its last frame pair is enqueued, so that its host replay overlaps the later batches' flow; one channel only)")
    ap.add_argument("--chain-depth", type=int, default=3,
                    help="tie-break levels the floods start with (the library deepens on its own where ties remain; scheduling only)")
    ap.add_argument("--rotate", type=int, default=3,
                    help="the timed steps visit this many different T-frame stacks of the synthetic sequence in turn (offsets "
                         "0, s, 2s, ... frames with s = --rotate-shift), all resident before the timed region: the data-dependent "
                         "memos of the host layer see changing input, as in a production sweep")
    ap.add_argument("--rotate-shift", type=int, default=7, help="frame offset between consecutive rotated stacks (not a multiple "
                    "of the window stride: no window of one step repeats a window of another)")
    ap.add_argument("--no-raster-subreport", action="store_true", help="skip the one extra (untimed) step in raster tie order")
    ap.add_argument("--single-label-seeds", action="store_true",
                    help="round-2 seeds (every positive seed = 1, detect_anvils(markers=None)) instead of component-labelled ones")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--no-pipeline-stacks", dest="pipeline_stacks", action="store_false",
                    help="time the steps one after the other, each through detect_stack_windows (default, with windows begun during "
                         "the flow and one channel: the timed steps are ONE detect_stack_sequence call over the rotated stacks -- the end "
                         "of a stack, i.e. its last windows' host replays, root phases and the stitch, runs beside the next stack's flow, "
                         "as in a sweep over many days; `step_ms` are then the intervals between the stacks' completions)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): every rank holds its own T-frame segment of one sequence.  strong: ONE T-frame stack -- the "
                         "configuration as BASELINE.json words V and F3 (\"288 frames, frame-sharded across 8 x MI355X\") -- whose windows "
                         "are dealt out to the ranks (rank r: windows [r W / N, (r + 1) W / N) and the frames they cover; "
                         "tobac_flow_amd.parallel.rank_windows); value = T H W per step over the wall time, whatever N")
    ap.add_argument("--no-hand-out", dest="hand_out", action="store_false",
                    help="several channels (config F3): keep every channel's stitched windows resident until `consume` (round 5) instead of "
                         "handing each window out as it is finished, with window-local ids and the relabelling tables at the end "
                         "(detect_stack_windows(on_window=...): the reference's own product structure -- one file per window job + linking.py "
                         "-- and what lets the windows of all three channels be begun during the flow on one device)")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="diagnostic: do not record HIP events around the library's launches in the timed region "
                         "(roofline = null); the difference to a default run is what the instrumentation costs")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a))

    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: the two must agree")
    if a.single_device:
        local = 0
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)

    import tobac_flow_amd.flow as tf
    from tobac_flow_amd import _lib
    from tobac_flow_amd.parallel import detect_stack_sequence, detect_stack_windows, rank_windows, window_bounds
    from tools.synth import anvil_seeds, blob_stack

    cT, cH, cW, cN, C = CONFIGS[a.config]
    T, H, W = a.frames or cT, a.height or cH, a.width or cW
    n_windows = a.n_windows or (cN if T == cT else max(1, round(T / 12)))
    full_size = (T, H, W, n_windows) == (cT, cH, cW, cN)
    bounds = window_bounds(T, n_windows, a.overlap) if n_windows > 1 else [(0, T)]
    # weak scaling: rank r holds frames r (T - overlap) ... of one sequence.  strong: ONE T-frame stack, rank r its share of the windows
    T_glob, bounds_glob, strong = T, bounds, a.scaling == "strong"
    frame0 = rank * (T - a.overlap)
    if strong:
        frame0, f_hi, bounds = rank_windows(bounds_glob, rank, world)
        T = f_hi - frame0
        n_windows = len(bounds)
    # rank r holds frames r * (T - overlap) ... of ONE sequence: its last `overlap` frames are rank r + 1's first ones.
    # Resident in HBM before the timed region: the brightness-temperature stack (generated in blocks of frames: torch's
    # own kernels index in 32 bits, tools/synth.py).  Everything else -- seeds included -- is computed inside the step.
    # The timed steps rotate over `--rotate` stacks cut from that sequence at offsets 0, s, 2s, ... (a step that
    # floods the identical stack every time keeps the host layer's memos perfectly warm): ONE buffer of T + (rotate - 1) s frames.
    n_rot = max(1, a.rotate)
    T_all = T + (n_rot - 1) * a.rotate_shift
    bt_all = torch.empty((T_all, H, W), dtype=torch.float32, device="cuda")
    for f0 in range(0, T_all, 12):
        f1 = min(f0 + 12, T_all)
        bt_all[f0:f1] = blob_stack(f1 - f0, H, W, seed=20240601, t0=frame0 + f0)
    ws_stats = []                                            # tf_watershed stats of every window of every step (warmup included)
    ref_order = []                                           # reference order: (detour microseconds, replay form, replay us, export us) per flood that needed it
    tie_mode = {"order": a.tie_order}
    inflight = {}
    timeline = os.environ.get("TF_BENCH_TIMELINE") is not None      # development aid: when each part of a step starts and ends
    info = {"floods": ws_stats, "reference_order": ref_order}

    def mark(what, ms):
        print("  t+%7.1f ms  %s" % (ms, what), file=sys.stderr, flush=True)

    def seeds_of(w, c):
        """the detect_anvils recipe for a window (linearised field; label(binary_erosion(field >= 1)), -1 where
        get_watershed_mask) of channel c (config F3: offsets 0 / -2 / -4 K)"""
        lin, seeds = anvil_seeds(w + CHANNEL_OFFSETS[c] if c else w)
        if a.single_label_seeds:
            seeds = torch.clamp(seeds, max=1)
        return lin, seeds

    def step(bt, vr_steps=None):
        """one pass over the stack `bt`: tobac_flow_amd.parallel.detect_stack_windows -- the flow of the stack once, the windows
        begun while the later frames' flow is computed, their host replays on worker threads, floods finished out of order on
        a second stream, the stitch (the scheduler lives in the package since round 5; this function only counts).
        Returns (stitched windows of the LAST channel, objects per channel)"""
        objects, keep = [], []

        def consume(c, wins):
            n_obj = int(max(int(w.max()) for w in wins))
            if dist is not None:                             # ids are global after the stitch: the count is the largest id on ANY rank
                tn = torch.tensor([n_obj], dtype=torch.int64, device="cpu" if a.backend == "gloo" else bt_all.device)
                dist.all_reduce(tn, op=dist.ReduceOp.MAX)
                n_obj = int(tn.item())
            objects.append(n_obj)
            if c == C - 1:
                keep[:] = [wins]                             # (one channel's labels resident at a time: only the last one's are kept)
            return None
        hand_out = C > 1 and a.hand_out
        res, _ = detect_stack_windows(bt, bounds, seeds_of, channels=C, consume=consume, overlap=a.overlap,
                                      vr_steps=a.vr_steps if vr_steps is None else vr_steps, smoothing_passes=1, interp_method="cubic",
                                      connectivity=1, chain_depth=a.chain_depth,
                                      on_ambiguous="reference" if tie_mode["order"] == "reference" else "ignore",
                                      max_in_flight=a.inflight, stream_windows=a.stream_windows,
                                      flow_workspace_gb=float(os.environ["TF_BENCH_FLOW_GB"]) if "TF_BENCH_FLOW_GB" in os.environ else None,
                                      info=info, mark=mark if timeline else None,
                                      on_window=(lambda c, k, lab: None) if hand_out else None)
        inflight["n"] = info.get("floods_in_flight", 1)
        inflight["flow_workspace_gb"] = info.get("flow_workspace_gb")
        inflight["channels_begun_during_the_flow"] = info.get("channels_begun_during_the_flow")
        if hand_out:                                         # the windows have left one by one; objects = the largest consistent id
            for r_ in res:
                n_obj = int(max(int(l.max()) for l in r_["luts"]))
                if dist is not None:
                    tn = torch.tensor([n_obj], dtype=torch.int64, device="cpu" if a.backend == "gloo" else bt_all.device)
                    dist.all_reduce(tn, op=dist.ReduceOp.MAX)
                    n_obj = int(tn.item())
                objects.append(n_obj)
            return None, objects
        return keep[0], objects

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def stack_of(i):
        off = (i % n_rot) * a.rotate_shift
        return bt_all[off:off + T]

    out_labels = None
    for i in range(a.warmup):                                # same memory pattern as the timed steps: the previous step's labels
        out_labels = None                                    # live until the next step starts (otherwise the first timed step
        out_labels, _ = step(stack_of(i))                    # pays fresh device allocations: +0.7 s measured)
    barrier()
    if os.environ.get("TF_BENCH_MEMDEBUG"):
        ms_ = torch.cuda.memory_stats()
        print("after warmup: device allocs %d frees %d retries %d, reserved %.1f GB" % (
            ms_.get("num_device_alloc", -1), ms_.get("num_device_free", -1), ms_.get("num_alloc_retries", -1),
            torch.cuda.memory_reserved() / 1e9), file=sys.stderr, flush=True)
    n_warm = len(ws_stats)
    _lib.profile_enable(not a.no_kernel_events)
    _lib.profile_collect()
    retries_t0 = torch.cuda.memory_stats().get("num_alloc_retries", 0)
    t0 = time.perf_counter()
    n_objects, step_ms = [], []
    objects_per_step = []
    pipelined = bool(a.pipeline_stacks and a.stream_windows and C == 1 and len(bounds) > 1 and a.steps > 1)
    if pipelined:
        # THE TIMED STEPS AS ONE SWEEP (round 5): detect_stack_sequence over the a.steps rotated stacks -- every stack is processed
        # exactly as a detect_stack_windows step is (same windows, same labels: tests/test_gpu_windows.py), but the calling thread
        # enqueues stack k + 1's flow as soon as stack k's last window has been SET UP, and stack k's last sweeps, host replays,
        # root phases and stitch run beside it on the flood thread instead of with an idle device.  All of it inside the timed
        # region; the previous stack's labels are released when a stack completes (one stack's labels resident, as before).
        out_labels = None
        done_at, kept = [], []

        def consume_seq(k, wins):
            n_obj = int(max(int(w.max()) for w in wins))
            if dist is not None:                             # ids are global after the stitch: the count is the largest id on ANY rank
                tn = torch.tensor([n_obj], dtype=torch.int64, device="cpu" if a.backend == "gloo" else bt_all.device)
                dist.all_reduce(tn, op=dist.ReduceOp.MAX)
                n_obj = int(tn.item())
            objects_per_step.append(n_obj)
            torch.cuda.current_stream().synchronize()
            done_at.append(time.perf_counter())
            if k == a.steps - 1:
                kept[:] = [wins]
            return None
        detect_stack_sequence((stack_of(a.warmup + i) for i in range(a.steps)), bounds, seeds_of, consume=consume_seq, overlap=a.overlap,
                              vr_steps=a.vr_steps, smoothing_passes=1, interp_method="cubic", connectivity=1, chain_depth=a.chain_depth,
                              on_ambiguous="reference" if tie_mode["order"] == "reference" else "ignore", max_in_flight=a.inflight,
                              flow_workspace_gb=float(os.environ["TF_BENCH_FLOW_GB"]) if "TF_BENCH_FLOW_GB" in os.environ else None,
                              info=info, mark=mark if timeline else None)
        inflight["n"] = info.get("floods_in_flight", 1)
        inflight["flow_workspace_gb"] = info.get("flow_workspace_gb")
        inflight["channels_begun_during_the_flow"] = info.get("channels_begun_during_the_flow")
        out_labels = kept[0]
        kept.clear()                                         # (out_labels is the one reference: released with it)
        n_objects = [objects_per_step[-1]]
        step_ms = [round((b_ - a_) * 1e3, 1) for a_, b_ in zip([t0] + done_at[:-1], done_at)]
    for i in range(0 if pipelined else a.steps):
        out_labels = None                                    # the previous step's labels are released before the next step's exist
        ts = time.perf_counter()
        out_labels, n_objects = step(stack_of(a.warmup + i))
        objects_per_step.append(n_objects[0] if len(n_objects) == 1 else n_objects)
        torch.cuda.synchronize()                             # (a step ends with the stitch's host-side union-find anyway)
        step_ms.append(round((time.perf_counter() - ts) * 1e3, 1))
        if os.environ.get("TF_BENCH_MEMDEBUG"):              # development aid: does a timed step still grow the allocator's pool?
            ms_ = torch.cuda.memory_stats()
            print("step %d: %.1f ms, device allocs %d frees %d retries %d, reserved %.1f GB" % (
                len(step_ms), step_ms[-1], ms_.get("num_device_alloc", -1), ms_.get("num_device_free", -1),
                ms_.get("num_alloc_retries", -1), torch.cuda.memory_reserved() / 1e9), file=sys.stderr, flush=True)
    barrier()
    dt = time.perf_counter() - t0
    retries_timed = torch.cuda.memory_stats().get("num_alloc_retries", 0) - retries_t0
    prof = _lib.profile_collect()
    _lib.profile_enable(False)
    del out_labels
    n_timed = len(ws_stats)
    alone_ms = None
    if pipelined and not a.no_raster_subreport:
        # sub-report, outside the timed region: ONE stack processed by itself (detect_stack_windows), its end with nothing beside it
        barrier()
        ts = time.perf_counter()
        out_labels, _ = step(stack_of(a.warmup + a.steps))
        barrier()
        alone_ms = (time.perf_counter() - ts) * 1e3
        del out_labels
    raster_ms = None
    if a.tie_order == "reference" and not a.no_raster_subreport:
        # sub-report, outside the timed region: ONE step with equal-valued markers in raster order (on_ambiguous="ignore")
        tie_mode["order"] = "raster"
        barrier()
        ts = time.perf_counter()
        out_labels, _ = step(stack_of(a.warmup + a.steps))
        barrier()
        raster_ms = (time.perf_counter() - ts) * 1e3
        del out_labels
        tie_mode["order"] = a.tie_order
    # Sub-report, outside the timed region: the dominant flow kernels with NOTHING beside them -- create_flow over the first 43
    # frames (one batch of 42 pairs finished in two parts, as in the timed steps) on an otherwise idle device.  Since the end of
    # round 5 the floods run on a thread and a stream of their own beside the whole flow, so the launch durations the library's
    # HIP events record inside the timed region (`roofline.frac`) include what the floods' kernels take from the flow's.
    prof_alone = None
    if world == 1 and full_size and not a.no_kernel_events and T >= 43:
        torch.cuda.synchronize()
        _lib.profile_enable(True)
        _lib.profile_collect()
        try:
            fl_alone = tf.create_flow(bt_all[:43], model="Farneback", vr_steps=a.vr_steps, smoothing_passes=1, interp_method="cubic",
                                      workspace_gb=inflight.get("flow_workspace_gb"), split_parts=2,
                                      on_frames_ready=(lambda fl, n: None) if (a.stream_windows and C == 1 and n_windows > 1) else None)   # (the same form of the kernel as in the timed steps)
            fl_alone.check()
            torch.cuda.synchronize()
            prof_alone = _lib.profile_collect()
            del fl_alone
        except torch.OutOfMemoryError:                           # (a configuration that fills the device: no sub-report)
            prof_alone = None
        _lib.profile_enable(False)
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=bt_all.device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank == 0:
        frames_computed = sum(hi - lo for lo, hi in bounds) if not strong else sum(hi - lo for lo, hi in bounds_glob) / world     # (per rank, on average)
        dom = max(prof.items(), key=lambda kv: kv[1][1]) if prof else None
        roof = None
        if dom:
            name, (calls, ms, by) = dom
            achieved = by / (ms * 1e-3) / 1e9
            # the counter passes were recorded on the default workload: their per-launch bytes say nothing about another size
            tr = _traffic(name, by / calls) if (a.config, T, H, W, a.vr_steps) == _TRAFFIC_WORKLOAD and full_size else None
            roof = {"bound": "hbm", "kernel": name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": tr["bytes_per_launch"] if tr else None,
                    "launches": calls, "avg_launch_us": round(ms * 1e3 / calls, 2),
                    "algorithmic_bytes_per_launch": round(by / calls, 1),
                    "share_of_step": round(ms / (dt * 1e3), 4),
                    "traffic_detail": tr,
                    "all_kernels": {k: {"ms_per_step": round(v[1] / a.steps, 3),
                                        "alg_GBps": round(v[2] / (v[1] * 1e-3) / 1e9, 1) if v[2] > 0 else None,
                                        "launches": v[0], "algorithmic_bytes_per_launch": round(v[2] / v[0], 1)}
                                    for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])}}
            # the right ceiling per kernel: next to the HBM figure, the VALU issue floor from the committed SQ
            # counter pass; `bound` = whichever floor is larger (frac stays the HBM fraction)
            try:
                with open(os.path.join(ROOT, _VALU_FILE)) as fh:
                    valu_doc = json.load(fh)
            except (OSError, ValueError):
                valu_doc = None
            try:
                with open(os.path.join(ROOT, _TRAFFIC_FILE)) as fh:
                    traffic_doc = json.load(fh)
            except (OSError, ValueError):
                traffic_doc = None
            for k, v in prof.items():
                e = roof["all_kernels"][k]
                if v[2] > 0:
                    e["hbm_floor_ms"] = round(v[2] / a.steps / (HBM_PEAK_GBS * 1e9) * 1e3, 3)
                    e["frac_hbm"] = round(v[2] / (v[1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                # measured HBM bytes over algorithmic bytes, from the committed FETCH_SIZE / WRITE_SIZE passes (a ratio: it carries
                # over to a run of another size); well above 1 = re-reads
                tr_ratio = _traffic_ratio(k, traffic_doc) if (traffic_doc and k in _KERNEL_SYMBOL and v[2] > 0) else None
                if tr_ratio is not None:
                    e["traffic_over_algorithmic"] = round(tr_ratio, 3)
                    e["traffic_GBps"] = round(tr_ratio * v[2] / (v[1] * 1e-3) / 1e9, 1)
                    e["traffic_frac_of_peak"] = round(e["traffic_GBps"] / HBM_PEAK_GBS, 4)
                vf = _valu_floor(k, v[2], valu_doc) if (valu_doc and k in _KERNEL_SYMBOL and v[2] > 0) else None
                if vf is not None:
                    e["valu_floor_ms"] = round(vf / a.steps, 3)
                    e["frac_valu"] = round(vf / v[1], 4)
                    e["bound"] = "valu" if vf / a.steps > e.get("hbm_floor_ms", 0.0) else "hbm"
            if "ws_relax_sweep" in roof["all_kernels"]:
                roof["all_kernels"]["ws_relax_sweep"]["note"] = ("algorithmic bytes = SURVEY 8(d)'s 29 B per voxel for ONE ideal sweep over the window, booked once "
                                                                 "per flood; a `launch` here is a batch of 32 sweep launches; the sweeps of all phases are timed")
            if valu_doc:
                roof["valu_floor_source"] = _VALU_FILE + " (instructions per algorithmic byte of a 62-frame run, scaled by this run's algorithmic bytes; 4 cycles per wave64 instruction, 1024 SIMDs)"
            if "bound" in roof["all_kernels"].get(name, {}):
                roof["bound_detail"] = roof["all_kernels"][name]["bound"]
            if tr:                                       # what the kernel really moves, at the measured launch time
                roof["traffic_GBps"] = round(tr["bytes_per_launch"] / (ms * 1e-3 / calls) / 1e9, 1)
                roof["traffic_frac_of_peak"] = round(roof["traffic_GBps"] / HBM_PEAK_GBS, 4)
            if prof_alone and name in prof_alone and prof_alone[name][1] > 0:
                c1, ms1, by1 = prof_alone[name]
                roof["uncontended"] = {"achieved": round(by1 / (ms1 * 1e-3) / 1e9, 1), "frac": round(by1 / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                       "launches": c1, "avg_launch_us": round(ms1 * 1e3 / c1, 2),
                                       "all_kernels_ms": {k: round(v[1], 2) for k, v in sorted(prof_alone.items(), key=lambda kv: -kv[1][1])},
                                       "note": "the same kernel with nothing beside it: create_flow over the first 43 frames (one batch of 42 pairs in two "
                                               "parts) after the timed region; `frac` above is measured inside the timed region, where the floods of the "
                                               "windows run beside the flow on a second stream"}
            if name == "fb_iteration_fused":
                # SURVEY.md 8(d) prices an iteration at 56 B per level pixel per DIRECTION; the fused launch serves both
                # directions of a pair and reads the two expansions R0, R1 once: 40 + 2 x 16 = 72 B per level pixel per
                # PAIR is what it must move at least.  Both fractions are reported.
                roof["achieved_compulsory"] = round(achieved * 72.0 / 112.0, 1)
                roof["frac_compulsory"] = round(achieved * 72.0 / 112.0 / HBM_PEAK_GBS, 4)
        names = {"F": "BASELINE config F: GOES-16 ABI full-disk-sized stack", "V": "BASELINE config V: SEVIRI full-disk-sized stack",
                 "F3": "BASELINE config F3: full-disk-sized stack, three channels sharing one Flow",
                 "C": "BASELINE config C: GOES-16 CONUS-sized stack", "window": "sub-report: ONE window per step"}
        what = ((f"{names[a.config]}: ONE {T_glob}x{H}x{W} float32 stack per step, its {len(bounds_glob)} windows dealt out to {world} GPU(s) "
                 f"(strong sharding: rank r holds the windows [r W / N, (r + 1) W / N) and the {T} or so frames they cover)" if strong else
                 f"{names[a.config]}: {T}x{H}x{W} float32 frames per GPU per step") + (f" x {C} channels" if C > 1 else "") +
                (f" ({n_rot} different stacks of the synthetic sequence visited in turn)" if n_rot > 1 else "") +
                (f", processed as {n_windows} time windows sharing {a.overlap} frames: flow of the {T - 1} frame pairs once for the stack, "
                 f"then per window Sobel edge field + seeds + watershed ({frames_computed} window frames for {T} delivered) + stitch of "
                 f"the label ids over all windows, all inside the timed region" if n_windows > 1 else ""))
        if not full_size:
            what = "REDUCED rehearsal of " + what
        out = {"metric": "Mpix/s end-to-end flow+sobel+watershed, 5424^2 frames" if (H, W) == (5424, 5424)
               else f"Mpix/s end-to-end flow+sobel+watershed, {H}x{W} frames",
               "value": round((T_glob if strong else world * T) * a.steps * H * W / dt / 1e6, 2),
               "unit": "Mpix/s", "n_gpus": world, "rccl_world_size": dist.get_world_size() if dist is not None else 1,
               "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(dt / a.steps * 1e3, 2), "step_ms": step_ms,
               "steps_pipelined": ({"on": True, "note": "the timed steps are ONE tobac_flow_amd.parallel.detect_stack_sequence call over the rotated stacks: the end of a "
                                    "stack (its last windows' host replays, root phases, the stitch) runs beside the next stack's flow; step_ms = intervals "
                                    "between the stacks' completions, the first one from the start of the timed region; --no-pipeline-stacks times them one by one"}
                                   if pipelined else {"on": False}),
               "one_stack_by_itself_ms": None if alone_ms is None else round(alone_ms, 1),
               "higher_is_better": True, "scaling": a.scaling,
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": what,
                          "frames_delivered_per_step": T_glob if strong else T, "window_frames_computed_per_step": frames_computed, "channels": C,
                          "delivery": ("every window handed out as it is finished, with window-local ids; relabelling tables (local id -> id consistent over all "
                                       "windows and ranks) at the end of the step -- the reference's product structure: one file per window job + linking.py"
                                       if (C > 1 and a.hand_out) else "stitched label windows resident at the end of the step"),
                          "seeds": ("every positive seed = 1 (detect_anvils(markers=None))" if a.single_label_seeds else
                                    "SURVEY 8(d): label(binary_erosion(field_lin >= 1)) per window on the device (component ids), "
                                    "-1 where get_watershed_mask(field_lin); computed inside the timed region"),
                          "stages": f"create_flow(Farneback, vr_steps={a.vr_steps}, smoothing_passes=1, cubic) + Flow.window_view + seeds + Flow.sobel(uphill, cubic, f64) "
                                    "+ edge field + Flow.watershed(connectivity 1)" + (" + stitch" if n_windows > 1 or world > 1 else ""),
                          "sharding": (f"STRONG: one {T_glob}-frame stack, rank r floods the windows [r W / N, (r + 1) W / N) and computes the flow of the frames they cover "
                                       f"(consecutive ranks share {a.overlap} frames); " if strong else
                                       f"one {T}-frame segment per GPU cut from one sequence, consecutive segments share {a.overlap} frames; ") +
                                      "label IDs stitched over all windows of all ranks by the reference's overlap rule "
                                      "(>= 5 px and >= 0.5, linking.py:49-161): one neighbour message per rank boundary + all-gathers of pair lists",
                          "input_rotation": f"{n_rot} stacks at frame offsets {[k * a.rotate_shift for k in range(n_rot)]} of the sequence, visited in turn (warm-up steps first)",
                          "objects_after_stitch": n_objects[0] if len(n_objects) == 1 else n_objects,
                          "objects_after_stitch_per_step": objects_per_step},
               "rate_over_computed_window_frames_Mpix_s": round(world * a.steps * frames_computed * H * W / dt / 1e6, 2),
               "peak_device_memory_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1),
               "allocator_retries_in_the_timed_region": int(retries_timed),      # (> 0: the device ran out and the caching allocator flushed its cache -- seconds)
               "roofline": roof}
        # which watershed schedule the timed windows ran: stats[5] = 1 / 0 probe (speculative root phase + conflict test,
        # conflict found / not found), -1 = root phase skipped on the conflict memo of watershed.py (identical labels)
        timed = np.array(ws_stats[n_warm:n_timed], np.int64)
        out["watershed"] = {"floods_timed": int(len(timed)),
                            "sweeps_per_phase_mean_per_flood": [round(float(v), 1) for v in timed[:, :5].mean(0)],
                            "sweeps_per_phase_last_flood": timed[-1][:5].tolist(), "relevant_pixels_last_flood": int(timed[-1][6]),
                            "chain_depth_used": int(timed[:, 8].max()), "chain_depth_used_min": int(timed[:, 8].min()),
                            "pixels_depending_on_equal_valued_marker_order": int(timed[:, 9].sum() // a.steps),
                            "marker_tie_points": int(timed[:, 10].sum() // a.steps), "ties_left_by_depth_cut_off": int(timed[:, 11].sum()),
                            "floods_probing": int((timed[:, 5] >= 0).sum()), "floods_skipping_root_phase": int((timed[:, 5] < 0).sum()),
                            "tie_order": a.tie_order,
                            # BY CONSTRUCTION: reference tie order and no tie left by the depth cut-off -- the configuration whose labels
                            # the tests compare voxel for voxel with the reference kernel's twin (tests/test_gpu_windows.py,
                            # test_gpu_reference_order.py); what THIS run checks against the oracle is cpu_baseline.library_vs_oracle_on_the_sample
                            "labels_bit_exact_with_the_reference_by_construction": a.tie_order == "reference" and int(timed[:, 11].sum()) == 0,
                            "floods_in_flight": inflight.get("n", 1), "flow_workspace_GB_chosen_by_the_scheduler": inflight.get("flow_workspace_gb"),
                            "windows_begun_during_the_flow": bool(a.stream_windows and n_windows > 1 and (inflight.get("channels_begun_during_the_flow") or 0) > 0),
                            "channels_begun_during_the_flow": inflight.get("channels_begun_during_the_flow")}
        if a.tie_order == "reference":
            ro = ref_order                                                          # warm-up floods included
            out["watershed"]["reference_order"] = {
                "replays": len(ro), "sparse_replays": sum(1 for r in ro if r[1] == "sparse"), "dense_replays": sum(1 for r in ro if r[1] == "dense"),
                "detour_ms_mean": round(float(np.mean([r[0] for r in ro])) / 1e3, 1) if ro else 0.0,
                "host_replay_ms_mean": round(float(np.mean([r[2] for r in ro])) / 1e3, 1) if ro else 0.0,
                "host_replay_ms_max": round(float(np.max([r[2] for r in ro])) / 1e3, 1) if ro else 0.0,
                "export_ms_mean": round(float(np.mean([r[3] for r in ro])) / 1e3, 1) if ro else 0.0,
                "exports_on_a_guessed_tie_value": sum(1 for r in ro if r[4]), "guesses_that_covered_the_tie": sum(1 for r in ro if r[5]),
                "root_phases_per_flood_mean": round(float(np.mean([r[6] for r in ro])), 2) if ro else 0.0,
                "replay_threads": __import__("tobac_flow_amd.parallel", fromlist=["x"])._REPLAY_POOL._max_workers,
                "note": "host replays run on worker threads beside the next windows' device work (tf_watershed_begin / _replay / _finish)"}
            if raster_ms is not None:
                out["watershed"]["raster_order_subreport"] = {
                    "ms_per_step": round(raster_ms, 1), "Mpix_per_s": round((T_glob if strong else world * T) * H * W / (raster_ms * 1e-3) / 1e6, 1),
                    "labels_bit_exact_with_the_reference_by_construction": False,
                    "note": "one extra step outside the timed region with on_ambiguous='ignore' (equal-valued markers in raster order)"}
        # per-stage rates, and the measured device-to-device copy rate as the practical HBM ceiling
        stage_of = {"to8bit_pair": "flow", "fb_gaussian_blur": "flow", "fb_resize": "flow", "fb_polyexp": "flow",
                    "fb_update_matrices": "flow", "fb_blur_solve": "flow", "fb_iteration_fused": "flow", "smooth_flow": "flow",
                    "vr_prepare": "refinement", "vr_system": "refinement", "vr_sor": "refinement",
                    "convolve": "sobel", "sobel": "sobel", "ws_setup": "watershed", "ws_relax_sweep": "watershed", "ws_labels": "watershed",
                    "binary_morph": "seeds"}
        stage_ms = {}
        for k, v in prof.items():
            stage_ms[stage_of.get(k, "other")] = stage_ms.get(stage_of.get(k, "other"), 0.0) + v[1] / a.steps
        out["stages"] = {k: {"kernel_ms_per_step": round(v, 2), "Mpix_per_s_over_computed_frames": round(frames_computed * H * W / v / 1e3, 1)}
                         for k, v in sorted(stage_ms.items(), key=lambda kv: -kv[1])}
        # the library's kernels run on two streams (the flow on the caller's, the floods on the flood thread's): their summed
        # launch durations can exceed the wall time.  Positive: kernel time hidden by the overlap; negative: wall time in which
        # none of the library's kernels ran (torch glue of the seeds, stitch, launches, syncs)
        out["stages"]["streams_overlap_ms"] = {"ms_per_step": round(sum(stage_ms.values()) - dt / a.steps * 1e3, 2),
                                               "note": "sum of the library's kernel time over both streams minus the step's wall time: > 0 = kernel time "
                                                       "hidden by running the floods beside the flow, < 0 = wall time outside the library's kernels"}
        if roof is not None:
            # the practical HBM ceiling: the library's own plain copy, 16 bytes per lane per access, four loads in flight per lane
            # (tf_copy16; round 5 -- rounds 2 - 4 timed torch's copy_, which reaches 4.9 TB/s where MI355X_MICROARCH.md's
            # float4 copy reaches 6.29: a flattering denominator)
            n_copy = 1 << 28                                                  # float32 elements: 1 GiB read + 1 GiB written
            src = torch.empty(n_copy, dtype=torch.float32, device=bt_all.device).normal_()
            dst = torch.empty_like(src)

            def copy_once():
                _lib.check(_lib.lib().tf_copy16(_lib.ptr(src), _lib.ptr(dst), 4 * n_copy, _lib.stream_ptr()), "tf_copy16")
            copy_once()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                copy_once()
            e1.record()
            torch.cuda.synchronize()
            copy_gbps = 10 * 2 * 4 * n_copy / (e0.elapsed_time(e1) * 1e-3) / 1e9
            assert torch.equal(src, dst)
            roof["practical_peak"] = round(copy_gbps, 1)
            roof["practical_peak_note"] = ("the library's 16-byte-per-lane copy kernel (tf_copy16) over 2 x %.2f GB, measured in this run (read + written "
                                           "bytes); MI355X_MICROARCH.md's float4 copy: 6290 GB/s" % (4 * n_copy / 1e9))
            roof["frac_practical"] = round(roof["achieved"] / copy_gbps, 4)
            del src, dst
        if not a.no_cpu_baseline and world == 1:             # reported baseline: rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(20240601, a.vr_steps)
            out["cv2_parity"] = cv2_parity()
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
