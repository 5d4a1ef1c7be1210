"""The Farnebaeck launches of one benchmark step (12 frames of 5424 x 5424: a batch of 8 pairs and a batch of 3, all six
pyramid resolutions, 10 iterations each = 120 k_fb_iter launches) without anything else around them: the target of the
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes behind bench.py's roofline.traffic (the full bench.py crashes under --pmc)."""
import sys
sys.path.insert(0, ".")
import torch
from tools.synth import blob_stack
from tobac_flow_amd.utils.flow_utils import FarnebackFlow
from tobac_flow_amd.utils.normalisation_utils import to_8bit_pair_dev
T, H = 12, 5424
bt = blob_stack(T, H, H, seed=20240601)
prev = torch.empty((T - 1, H, H), dtype=torch.uint8, device="cuda")
nxt = torch.empty_like(prev)
for i in range(T - 1):
    to_8bit_pair_dev(bt[i], bt[i + 1], out=(prev[i], nxt[i]))
fwd = torch.empty((T, H, H, 2), dtype=torch.float32, device="cuda")
bwd = torch.empty_like(fwd)
m = FarnebackFlow()
for i0, B in ((0, 8), (8, 3)):
    m.calc_batch_dev(prev[i0:i0 + B], nxt[i0:i0 + B], fwd[i0:i0 + B], bwd[i0 + 1:i0 + 1 + B])
torch.cuda.synchronize()
print("done", flush=True)
