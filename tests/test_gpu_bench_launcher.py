"""bench.py's own multi-rank path (reference: scripts/linking_parallel.py:26-27 -- windows of one
sequence processed by several workers, label ids linked afterwards): the driver starts `bench.py --gpus N` on an 8-GPU node
at round end, the builder's box has one GPU, so the launcher, the rank set-up (RANK / LOCAL_RANK / WORLD_SIZE), the
per-rank segment of the synthetic sequence and stitch_rank_windows over the ranks' windows are rehearsed here with two ranks
on ONE device over gloo -- as a fresh child process each, like the driver's call -- and compared with the one-process run of
the same 44-frame sequence."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(*extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "C", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
           "--rotate", "1", "--no-raster-subreport", *extra]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, "bench.py %s failed:\n%s" % (" ".join(extra), r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly ONE JSON line on rank 0, got %d" % len(lines)
    return json.loads(lines[0])


def test_two_ranks_on_one_device_equal_the_one_process_run():
    """rank r holds frames 20 r .. 20 r + 23 of one sequence (two 14-frame windows each, consecutive ranks share four frames);
    one process holding frames 0 .. 43 as four windows floods the same four windows: same objects after the stitch."""
    two = _bench("--gpus", "2", "--single-device", "--backend", "gloo")
    one = _bench("--frames", "44", "--n-windows", "4")
    assert two["n_gpus"] == 2 and two["rccl_world_size"] == 2 and one["n_gpus"] == 1
    assert two["config"]["frames_delivered_per_step"] == 24 and two["scaling"] == "weak"
    assert two["watershed"]["tie_order"] == "reference" and two["watershed"]["labels_bit_exact_with_the_reference_by_construction"]
    assert two["config"]["objects_after_stitch"] == one["config"]["objects_after_stitch"] > 10
    # whole-job value: both ranks' frames over the slower rank's time
    assert two["value"] > 0 and abs(two["value"] - 2 * 24 * 1500 * 2500 / (two["ms_per_step"] * 1e-3) / 1e6) < 0.02 * two["value"]
    # round 5: several timed steps are ONE detect_stack_sequence call (the end of a stack beside the next stack's flow); under
    # --gpus N the stitch of every stack is a collective issued from the flood threads of the ranks, in the same order
    two_p = _bench("--gpus", "2", "--single-device", "--backend", "gloo", "--steps", "2")
    one_p = _bench("--frames", "44", "--n-windows", "4", "--steps", "2")
    assert two_p["steps_pipelined"]["on"] and one_p["steps_pipelined"]["on"] and not two["steps_pipelined"]["on"]
    n_obj = one["config"]["objects_after_stitch"]
    assert two_p["config"]["objects_after_stitch_per_step"] == [n_obj, n_obj] == one_p["config"]["objects_after_stitch_per_step"]
    assert len(two_p["step_ms"]) == 2 and two_p["n_gpus"] == 2


def test_strong_sharding_of_one_stack_equals_the_one_process_run():
    """bench.py --scaling strong (BASELINE configs 4 - 5 as worded: ONE stack "frame-sharded" over the GPUs; the reference's
    analogue: one window job per process, scripts/dcc_detect_seviri_nat.py:152 + scripts/linking_parallel.py:26-27): a 44-frame
    stack of four windows, rank r taking two of them and the 24 frames they cover -- the same objects after the stitch as the
    one-process run of the same stack, `value` = the ONE stack's pixels over the wall time (not times N), and N = 1 under
    --scaling strong is the plain run."""
    one = _bench("--frames", "44", "--n-windows", "4")
    two = _bench("--frames", "44", "--n-windows", "4", "--gpus", "2", "--single-device", "--backend", "gloo", "--scaling", "strong")
    solo = _bench("--frames", "44", "--n-windows", "4", "--scaling", "strong")
    assert two["scaling"] == "strong" and two["n_gpus"] == 2 and solo["scaling"] == "strong" and one["scaling"] == "weak"
    assert two["config"]["frames_delivered_per_step"] == 44 == solo["config"]["frames_delivered_per_step"]
    assert two["config"]["objects_after_stitch"] == one["config"]["objects_after_stitch"] == solo["config"]["objects_after_stitch"] > 10
    assert abs(two["value"] - 44 * 1500 * 2500 / (two["ms_per_step"] * 1e-3) / 1e6) < 0.02 * two["value"]
    assert "STRONG" in two["config"]["sharding"] and "ONE 44x1500x2500" in two["config"]["workload"]
    # two steps: every rank's share of the stack goes through detect_stack_sequence, the stitch of each step a collective
    two_p = _bench("--frames", "44", "--n-windows", "4", "--gpus", "2", "--single-device", "--backend", "gloo", "--scaling", "strong", "--steps", "2")
    n_obj = one["config"]["objects_after_stitch"]
    assert two_p["config"]["objects_after_stitch_per_step"] == [n_obj, n_obj] and two_p["steps_pipelined"]["on"]


def test_bench_pipeline_equals_the_plain_calls():
    """bench.py's step -- windows begun from create_flow's callback while the later frames' flow is computed, floods in parts
    with their replays on worker threads, finished on a second stream, stitched in place -- against the same stack processed
    with the plain calls one after the other (create_flow, Flow.window, seeds, edge field, Flow.watershed, stitch_window_list):
    the same number of objects after the stitch, for the streaming and the non-streaming schedule."""
    import numpy as np
    import torch
    import tobac_flow_amd.flow as tf
    from tobac_flow_amd.detection import get_combined_edge_field
    from tobac_flow_amd.parallel import stitch_window_list, window_bounds
    from tools.synth import anvil_seeds, blob_stack
    T, H, W, n_win, overlap = 44, 1500, 2500, 4, 4
    bt = blob_stack(T, H, W, seed=20240601, t0=0)
    flow = tf.create_flow(bt, model="Farneback", vr_steps=1, smoothing_passes=1, interp_method="cubic")
    labs = []
    for a, b in window_bounds(T, n_win, overlap):
        fl = flow.window(a, b)
        lin, seeds = anvil_seeds(bt[a:b])
        e = get_combined_edge_field(fl, lin, dtype=np.float32)
        labs.append(fl.watershed(e, seeds, connectivity=1))
    want = int(max(int(w.max()) for w in stitch_window_list(labs, overlap=overlap)))
    del flow, labs
    torch.cuda.empty_cache()
    streamed = _bench("--frames", "44", "--n-windows", "4")
    plain = _bench("--frames", "44", "--n-windows", "4", "--no-stream-windows")
    assert streamed["watershed"]["windows_begun_during_the_flow"] and not plain["watershed"]["windows_begun_during_the_flow"]
    assert streamed["config"]["objects_after_stitch"] == plain["config"]["objects_after_stitch"] == want > 10
