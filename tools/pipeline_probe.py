"""Does the watershed of window k overlap the flow of window k + 1 when they run on two HIP streams?  (development aid)"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import tobac_flow_amd.flow as tf
from tobac_flow_amd.detection import get_combined_edge_field
from tobac_flow_amd.watershed import neighbour_offsets, watershed_dev
from tools.synth import anvil_inputs, blob_stack
T, H, W = 12, 5424, 5424
bt = blob_stack(T, H, W, seed=20240601)
lin, markers = anvil_inputs(bt)
nbr = neighbour_offsets(1)


def flow_stage():
    return tf.create_flow(bt, model="Farneback", vr_steps=1, smoothing_passes=1, interp_method="cubic")


def detect_stage(flow):
    e = get_combined_edge_field(flow, lin, dtype=np.float32)
    fw, bw = flow._dev_flows()
    return watershed_dev(fw, bw, e, markers, None, nbr, on_ambiguous="ignore")


for _ in range(2):
    detect_stage(flow_stage())
torch.cuda.synchronize()
K = 4
t0 = time.perf_counter()
th = time.perf_counter()
f = flow_stage()
print("host time of create_flow without a sync: %.1f ms" % ((time.perf_counter() - th) * 1e3))
torch.cuda.synchronize()
for _ in range(K - 1):
    detect_stage(f); f = flow_stage()
detect_stage(f)
torch.cuda.synchronize()
print("sequential: %.1f ms per window" % ((time.perf_counter() - t0) / K * 1e3))

sA, sB = torch.cuda.current_stream(), torch.cuda.Stream()
torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.cuda.stream(sB):
    f = flow_stage()
evs = []
for k in range(K):
    ev = torch.cuda.Event(); ev.record(sB); sA.wait_event(ev)
    nxt = None
    if k + 1 < K:
        with torch.cuda.stream(sB):
            nxt = flow_stage()
    detect_stage(f)
    f = nxt
torch.cuda.synchronize()
print("pipelined:  %.1f ms per window" % ((time.perf_counter() - t0) / K * 1e3))
