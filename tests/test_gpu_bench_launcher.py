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
    assert two["watershed"]["tie_order"] == "reference" and two["watershed"]["labels_bit_exact_with_the_reference"]
    assert two["config"]["objects_after_stitch"] == one["config"]["objects_after_stitch"] > 10
    # whole-job value: both ranks' frames over the slower rank's time
    assert two["value"] > 0 and abs(two["value"] - 2 * 24 * 1500 * 2500 / (two["ms_per_step"] * 1e-3) / 1e6) < 0.02 * two["value"]
