"""Time the fused Farnebaeck iteration kernel at one resolution (num_levels=0)."""
import sys
sys.path.insert(0, ".")
import torch
from tools.synth import blob_stack
from tobac_flow_amd import _lib
from tobac_flow_amd.utils.flow_utils import FarnebackFlow
from tobac_flow_amd.utils.normalisation_utils import to_8bit_pair_dev
H = int(sys.argv[1]) if len(sys.argv) > 1 else 5424
bt = blob_stack(2, H, H, nan_every=0)
a, b = to_8bit_pair_dev(bt[0], bt[1])
m = FarnebackFlow(num_levels=0)
m.calc_pair_dev(a, b)
torch.cuda.synchronize()
_lib.profile_enable(True); _lib.profile_collect()
for _ in range(3):
    m.calc_pair_dev(a, b)
torch.cuda.synchronize()
for k, (c, ms, by) in _lib.profile_collect().items():
    print(f"{k}: calls {c} avg {ms / c * 1e3:.1f} us  alg {by / ms / 1e6:.0f} GB/s")
