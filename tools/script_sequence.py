"""The path as the reference's drop-in scripts drive it (scripts/dcc_detect_goes.py:164-303), timed call by call:

    create_flow(bt, Farneback, vr_steps=1, smoothing_passes=1, cubic) -> detect_cores(use_wvd=False) ->
    get_anvil_markers(wvd - swd) -> detect_anvils (thick) -> relabel_anvils -> detect_anvils (thin)

on one synthetic window (default 16 x 5424 x 5424, the window the scripts process per file group; --config C = 24 x 1500 x
2500).  `--mode host` hands the entry points numpy arrays with a time coordinate, exactly what the scripts hand them
(DataArrays; xarray is not in this image), so every call pays its uploads and downloads; `--mode device` hands them
device tensors (what a device-resident pipeline gets).  Per call: wall time (device synchronised before and after) and
the summed duration of the GPU kernels it launched (torch.profiler's device activities: every HIP kernel of the process,
the library's included), `host_share` = 1 - kernels / wall.  Writes one JSON (VERDICT r4 item 1:
profiles/round5_script_sequence.json).

    python tools/script_sequence.py --frames 16 --size 5424 --out gpurun_out/script_sequence.json
"""
import argparse
import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def scene(T, H, W, minutes, device):
    """bt / wvd / swd of a window: the benchmark's translating cold blobs (tools/synth.blob_stack), their tops deepening
    over the window so that cores cool, WVD - SWD crossing the thick-anvil thresholds (-5 / -12.5) and WVD + SWD the thin
    ones (0 / -7.5) around them (the scene of tests/test_gpu_pipeline.py at scale)."""
    import torch
    from tools.synth import blob_stack
    bt0 = blob_stack(T, H, W, device=device)
    ramp = torch.linspace(0.2, 2.0, T, device=device, dtype=torch.float32)[:, None, None]     # (tops cool by 0.5 - 0.7 K / min: cores)
    cold = torch.clamp(250.0 - bt0, min=0)
    cold = torch.where(torch.isnan(bt0), torch.full_like(cold, float("nan")), cold)
    bt = (290.0 - ramp * cold).to(torch.float32)
    wvd = (0.8 * cold - 11).to(torch.float32)
    swd = torch.full_like(bt, 3.0)
    del bt0, cold
    return bt, wvd, swd


class Timer:
    def __init__(self, use_profiler):
        self.rows = []
        self.use_profiler = use_profiler

    def run(self, name, fn):
        import torch
        torch.cuda.synchronize()
        kernels_ms = n_kernels = copies_ms = top = None
        t0 = time.perf_counter()
        if self.use_profiler:
            from torch.profiler import ProfilerActivity, profile
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                out = fn()
                torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            try:
                evs = [e for e in prof.events() if getattr(e, "device_type", None) is not None and "cuda" in str(e.device_type).lower()]
                dur = lambda e: float(getattr(e, "device_time", 0.0) or getattr(e, "cuda_time", 0.0) or e.time_range.elapsed_us())   # noqa: E731
                is_copy = lambda e: e.name.lower().startswith(("memcpy", "memset")) or "copybuffer" in e.name.lower()             # noqa: E731
                kernels_ms = sum(dur(e) for e in evs if not is_copy(e)) / 1e3
                copies_ms = sum(dur(e) for e in evs if is_copy(e)) / 1e3
                n_kernels = sum(1 for e in evs if not is_copy(e))
                by_name = {}
                for e in evs:
                    by_name[e.name] = by_name.get(e.name, 0.0) + dur(e) / 1e3
                top = sorted(by_name.items(), key=lambda kv: -kv[1])[:6]
            except Exception as exc:                       # noqa: BLE001 -- the wall times are still worth having
                print("profiler events unreadable:", exc, flush=True)
        else:
            out = fn()
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
        row = {"call": name, "wall_ms": round(wall * 1e3, 2)}
        from tobac_flow_amd import _staging
        now = dict(_staging.stats)
        prev = getattr(self, "_staging_prev", {k: 0 for k in now})
        self._staging_prev = now
        delta = {k: now[k] - prev.get(k, 0) for k in now}
        if delta["uploads"] or delta["hits"] or delta["downloads"]:
            row["staging"] = {"uploads": delta["uploads"], "recognised_by_content": delta["hits"], "downloads": delta["downloads"],
                              "GB_up": round(delta["upload_bytes"] / 1e9, 2), "GB_down": round(delta["download_bytes"] / 1e9, 2),
                              "host_ms": {k[:-2]: round(delta[k] * 1e3, 1) for k in ("hash_s", "upload_s", "download_s", "pinned_alloc_s")}}
        if kernels_ms is not None and n_kernels:
            # device work = kernels + the copy engine's transfers (the uploads / downloads of host-mode calls: PCIe time, not
            # host compute); host_share = the part of the wall time in which the device did neither
            busy = kernels_ms + copies_ms
            row.update(kernels_ms=round(kernels_ms, 2), copies_ms=round(copies_ms, 2), n_kernels=n_kernels,
                       host_share=round(max(0.0, 1.0 - busy / (wall * 1e3)), 4),
                       wall_over_kernels=round(wall * 1e3 / max(kernels_ms, 1e-9), 3),
                       top_kernels=[[n[:70], round(ms, 2)] for n, ms in top])
        self.rows.append(row)
        print(json.dumps(row), flush=True)
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default=None, choices=[None, "C", "window"])
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--size", type=int, nargs="+", default=[5424])
    ap.add_argument("--minutes", type=int, default=None, help="cadence of the time coordinate (default: 10, the full-disk cadence; config C: 5)")
    ap.add_argument("--mode", default="host", choices=["host", "device"])
    ap.add_argument("--repeat", type=int, default=2, help="passes over the sequence; the last one is reported (the first pays allocator growth and lazy initialisation)")
    ap.add_argument("--no-profiler", action="store_true")
    ap.add_argument("--wall-only-pass", action="store_true", help="one more pass without the profiler: its wall times are `wall_ms_unprofiled`")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch
    import tobac_flow_amd.flow as tf
    from tobac_flow_amd.detection import detect_anvils, detect_cores, get_anvil_markers, relabel_anvils
    from tools.synth import field_with_time
    if args.minutes is None:
        args.minutes = 5 if args.config == "C" else 10
    if args.config == "C":
        T, H, W = 24, 1500, 2500
    else:
        T = args.frames
        H, W = (args.size[0], args.size[-1])
    dev = torch.device("cuda", 0)
    bt_d, wvd_d, swd_d = scene(T, H, W, args.minutes, dev)
    host = args.mode == "host"

    def give(x):
        return field_with_time(x.cpu().numpy() if host else x, minutes=args.minutes)

    bt, wvd, swd = give(bt_d), give(wvd_d), give(swd_d)
    wd, ws = give(wvd_d - swd_d), give(wvd_d + swd_d)
    if host:
        del bt_d, wvd_d, swd_d
        torch.cuda.empty_cache()
    t_offset = 3
    results = {}

    def sequence(timer):
        out = {}
        flow = timer.run("create_flow", lambda: tf.create_flow(bt, model="Farneback", vr_steps=1, smoothing_passes=1, interp_method="cubic"))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out["core"] = timer.run("detect_cores", lambda: detect_cores(flow, bt, wvd, swd, wvd_threshold=0.25, bt_threshold=0.5, overlap=0.5,
                                                                         absolute_overlap=4, subsegment_shrink=0.0, min_length=t_offset, use_wvd=False))
            out["markers"] = timer.run("get_anvil_markers", lambda: get_anvil_markers(flow, wd, threshold=-5, overlap=0.5, absolute_overlap=4,
                                                                                      subsegment_shrink=0.0, min_length=t_offset))
            out["thick0"] = timer.run("detect_anvils_thick", lambda: detect_anvils(flow, wd, markers=out["markers"], upper_threshold=-5, lower_threshold=-12.5,
                                                                                   erode_distance=2, min_length=t_offset))
            out["thick"] = timer.run("relabel_anvils", lambda: relabel_anvils(flow, out["thick0"], markers=out["markers"], overlap=0.5, absolute_overlap=4,
                                                                              min_length=t_offset))
            out["thin"] = timer.run("detect_anvils_thin", lambda: detect_anvils(flow, ws, markers=out["thick"], upper_threshold=0, lower_threshold=-7.5,
                                                                                erode_distance=2, min_length=t_offset))
        del flow
        return out

    rows = None
    first_pass_ms = None
    from tobac_flow_amd import _staging
    for k in range(max(1, args.repeat)):
        print(f"--- pass {k + 1} of {args.repeat}", flush=True)
        # every pass starts like a fresh file group: no result of the previous pass alive, no device twin remembered (a pass
        # would otherwise find every input of the previous one in HBM); the allocators' pools stay warm after the first pass
        results = None
        _staging.clear(trim=False)
        timer = Timer(use_profiler=not args.no_profiler)
        timer._staging_prev = dict(_staging.stats)
        results = sequence(timer)
        rows = timer.rows
        if k == 0:
            first_pass_ms = round(sum(r["wall_ms"] for r in rows), 1)
    if args.wall_only_pass and not args.no_profiler:
        print("--- pass without the profiler", flush=True)
        timer = Timer(use_profiler=False)
        sequence(timer)
        for r, r2 in zip(rows, timer.rows):
            r["wall_ms_unprofiled"] = r2["wall_ms"]

    def count(x):
        x = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.asarray(x).astype(np.int64))
        return int(x.max()), int((x != 0).sum())
    summary = {"workload": f"{T} x {H} x {W} f32 (bt, wvd, swd synthetic), the call sequence of scripts/dcc_detect_goes.py:164-303",
               "mode": args.mode, "calls": rows,
               "total_wall_ms": round(sum(r["wall_ms"] for r in rows), 1),
               "Mpix_per_s": round(T * H * W / 1e6 / max(sum(r["wall_ms"] for r in rows) / 1e3, 1e-9), 1),
               "objects": {k: dict(zip(("n", "voxels"), count(v))) for k, v in results.items()},
               "first_pass_total_wall_ms": first_pass_ms,
               "device": torch.cuda.get_device_name(0)}
    if all("kernels_ms" in r for r in rows):
        summary["total_kernels_ms"] = round(sum(r["kernels_ms"] for r in rows), 1)
        summary["total_copies_ms"] = round(sum(r["copies_ms"] for r in rows), 1)
    print(json.dumps(summary), flush=True)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(summary, f, indent=1)


if __name__ == "__main__":
    main()
