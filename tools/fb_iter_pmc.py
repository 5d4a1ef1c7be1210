"""One full-resolution pair, num_levels=0: used under rocprofv3 --pmc to read HBM traffic of k_fb_iter."""
import sys
sys.path.insert(0, ".")
import torch
from tools.synth import blob_stack
from tobac_flow_amd.utils.flow_utils import FarnebackFlow
from tobac_flow_amd.utils.normalisation_utils import to_8bit_pair_dev
H = 5424
bt = blob_stack(2, H, H, nan_every=0)
a, b = to_8bit_pair_dev(bt[0], bt[1])
m = FarnebackFlow(num_levels=0)
for _ in range(2):
    m.calc_pair_dev(a, b)
torch.cuda.synchronize()
print("done")
