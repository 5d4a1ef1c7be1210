"""Stage timings of the hot path on the current GPU (development aid)."""
import argparse
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

from tools.synth import anvil_inputs, blob_stack
import tobac_flow_amd.flow as tf
from tobac_flow_amd import _lib
from tobac_flow_amd.watershed import neighbour_offsets, watershed_dev

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=6)
ap.add_argument("--H", type=int, default=5424)
ap.add_argument("--W", type=int, default=5424)
ap.add_argument("--smooth", type=int, default=1)
ap.add_argument("--depth", type=int, default=3)
a = ap.parse_args()
t0 = time.time()
bt = blob_stack(a.T, a.H, a.W)
torch.cuda.synchronize()
print("synth", time.time() - t0, flush=True)


def timed(name, fn):
    torch.cuda.synchronize()
    t = time.time()
    r = fn()
    torch.cuda.synchronize()
    dt = time.time() - t
    print(f"{name}: {dt * 1e3:.1f} ms  ({a.T * a.H * a.W / dt / 1e6:.1f} Mpix/s)", flush=True)
    return r


for rep in range(2):
    flow = timed("create_flow", lambda: tf.create_flow(bt, smoothing_passes=a.smooth, interp_method="cubic"))
    lin, markers = anvil_inputs(bt)
    edges64 = timed("sobel", lambda: flow.sobel(lin, direction="uphill", method="cubic"))

    def edge_field():
        e = edges64
        e = torch.where(e > 0, e + 1, e) - lin
        e = torch.where(torch.isnan(lin), torch.full_like(e, float("inf")), e)
        return e.to(torch.float32)
    edges = timed("edge_field", edge_field)
    st = {}
    nbr = neighbour_offsets(1)
    fw, bw = flow._dev_flows()
    labels = timed("watershed", lambda: watershed_dev(fw, bw, edges, markers, None, nbr, a.depth, st))
    print("  sweeps", st, "unlabelled", int((labels == 0).sum()), "active frac", float((markers == 0).float().mean()), flush=True)
