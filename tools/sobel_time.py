"""Time the fused Sobel + edge-field kernel on a 12 x 5424^2 window (development aid)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import tobac_flow_amd.flow as tf
from tobac_flow_amd.detection import get_combined_edge_field
from tools.synth import anvil_inputs, blob_stack
T = 12
bt = blob_stack(T, 5424, 5424)
flow = tf.create_flow(bt, smoothing_passes=1, interp_method="cubic")
lin, _ = anvil_inputs(bt)
for _ in range(2):
    e = get_combined_edge_field(flow, lin, dtype=np.float32)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    e = get_combined_edge_field(flow, lin, dtype=np.float32)
torch.cuda.synchronize()
print("sobel+edge ms per call", (time.perf_counter() - t0) / 5 * 1e3, float(e[3, 100, 100]))
