"""Time the iteration kernel on a batch at one resolution (num_levels = 0): B pairs, both directions.
    python tools/fb_iter_batch_time.py [H [B [W]]]       env: TF_FB_ROW_SUMS_TREE=1 (round-3 kernel), TF_FBI_SEQ_ABLATE=<bits>"""
import sys
sys.path.insert(0, ".")
import torch
from tobac_flow_amd import _lib
from tobac_flow_amd.utils.flow_utils import FarnebackFlow
H = int(sys.argv[1]) if len(sys.argv) > 1 else 5424
B = int(sys.argv[2]) if len(sys.argv) > 2 else 21
W = int(sys.argv[3]) if len(sys.argv) > 3 else H
g = torch.Generator(device="cuda").manual_seed(3)
x = torch.randn((1, 1, H + 64, W + 64), device="cuda", generator=g)
for _ in range(4):
    x = torch.nn.functional.avg_pool2d(x, 7, stride=1, padding=3)
big = ((x - x.min()) / (x.max() - x.min()) * 255)[0, 0].to(torch.uint8)
frames = torch.stack([big[i:i + H, 2 * i:2 * i + W] for i in range(B + 1)]).contiguous()
m = FarnebackFlow(num_levels=0)
fwd = torch.empty((B, H, W, 2), dtype=torch.float32, device="cuda")
bwd = torch.empty_like(fwd)
m.calc_batch_dev(frames[:-1], frames[1:], fwd, bwd)
torch.cuda.synchronize()
_lib.profile_enable(True); _lib.profile_collect()
for _ in range(2):
    m.calc_batch_dev(frames[:-1], frames[1:], fwd, bwd)
torch.cuda.synchronize()
for k, (c, ms, by) in _lib.profile_collect().items():
    if "iter" in k:
        print(f"H {H} W {W} B {B} {k}: calls {c} avg {ms / c:.3f} ms  alg {by / ms / 1e6:.0f} GB/s", flush=True)
