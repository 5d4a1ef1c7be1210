"""Run only create_flow on a small frame stack (profiling aid)."""
import sys
sys.path.insert(0, ".")
import torch
from tools.synth import blob_stack
import tobac_flow_amd.flow as tf
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2
H = int(sys.argv[2]) if len(sys.argv) > 2 else 5424
W = int(sys.argv[3]) if len(sys.argv) > 3 else H
bt = blob_stack(T, H, W, nan_every=0)
for _ in range(2):
    fl = tf.create_flow(bt, smoothing_passes=1, interp_method="cubic")
torch.cuda.synchronize()
print("done")
