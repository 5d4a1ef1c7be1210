"""The fused Sobel + edge-field kernel on a 4 x 5424^2 stack, three launches (development aid; run under rocprofv3 --pmc,
see profiles/round3_sobel_pmc.txt for the counter sets)."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import tobac_flow_amd.flow as tf
from tobac_flow_amd.detection import get_combined_edge_field
from tools.synth import anvil_inputs, blob_stack
T = 4
bt = blob_stack(T, 5424, 5424)
flow = tf.create_flow(bt, vr_steps=0, smoothing_passes=0)
lin, _ = anvil_inputs(bt)
for _ in range(3):
    e = get_combined_edge_field(flow, lin, dtype=np.float32)
torch.cuda.synchronize()
print(float(e[1, 100, 100]))
