"""BASELINE config F on ONE MI355X in the EXACT mode of SURVEY 8(e): 144 frames of 5424 x 5424 through the whole hot path
with ONE watershed over the whole volume (4.24e9 voxels) instead of twelve stitched windows.  Prints wall-clock times.
(development aid / demonstration; ~200 GB of HBM)"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import tobac_flow_amd.flow as tf
from tobac_flow_amd import _lib
from tobac_flow_amd.detection import get_combined_edge_field
from tobac_flow_amd.watershed import neighbour_offsets, watershed_dev
from tools.synth import anvil_inputs, blob_stack

T = int(sys.argv[1]) if len(sys.argv) > 1 else 144
H = W = 5424


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


# one large allocation up front, handed back to torch's caching allocator: the stages below then carve their arrays out
# of it instead of paying hipMalloc for tens of GB each (which would dominate a cold single run)
x = torch.empty(int(min(250e9, 1.7e9 * T)), dtype=torch.uint8, device="cuda"); del x
t0 = sync()
# inputs are generated twelve frames at a time (torch's pooling kernels index in 32 bits: a 4e9-voxel call faults)
bt = torch.empty((T, H, W), dtype=torch.float32, device="cuda")
lin = torch.empty((T, H, W), dtype=torch.float32, device="cuda")
markers = torch.empty((T, H, W), dtype=torch.int32, device="cuda")
for a in range(0, T, 12):
    b = min(a + 12, T)
    bt[a:b] = blob_stack(b - a, H, W, seed=20240601, t0=a)
    lin[a:b], markers[a:b] = anvil_inputs(bt[a:b])
t1 = sync()
print("synthetic input: %.1f s, %.1f GB resident" % (t1 - t0, torch.cuda.memory_allocated() / 1e9), flush=True)
flow = tf.create_flow(bt, model="Farneback", vr_steps=1, smoothing_passes=1, interp_method="cubic")
del bt
_lib.release_workspaces()
t2 = sync()
print("create_flow (vr_steps=1, smoothing 1, cubic): %.2f s" % (t2 - t1), flush=True)
e = get_combined_edge_field(flow, lin, dtype=np.float32)
del lin
t3 = sync()
print("Sobel + edge field: %.2f s" % (t3 - t2), flush=True)
fw, bw = flow._dev_flows()
st = {}
labels = watershed_dev(fw, bw, e, markers, None, neighbour_offsets(1), stats=st, on_ambiguous="ignore")
t4 = sync()
print("ONE watershed over %d x %d x %d = %.3g voxels: %.2f s; relevant pixels %d, chain depth %d, sweeps %s" %
      (T, H, W, T * H * W, t4 - t3, st["sweeps"][6], st["chain_depth"], st["sweeps"][:5]), flush=True)
print("hot path total: %.2f s = %.0f Mpix/s; labels: %d distinct, %.1f %% of the volume labelled; peak HBM %.0f GB" %
      (t4 - t1, T * H * W / (t4 - t1) / 1e6, int(labels.max()), 100.0 * float((labels > 0).float().mean()),
       torch.cuda.max_memory_allocated() / 1e9), flush=True)
