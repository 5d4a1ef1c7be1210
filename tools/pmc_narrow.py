"""Narrowing aid for the counter-collection crash recorded in round 1 (rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE over the
benchmark's Farneback launches ended in a segmentation fault of the profiled process; one pair at num_levels = 0 worked).
One knob at a time: batch size B (blockIdx.z of k_fb_iter), pyramid levels, strided outputs into a bigger flow array,
frame size, and a control with no library call at all (torch kernels only).  Prints `done` when it got through.

  rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -- python3 -X faulthandler tools/pmc_narrow.py --B 8 --levels 5 --strided 1
"""
import argparse
import sys
sys.path.insert(0, ".")
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=1)
ap.add_argument("--levels", type=int, default=0)
ap.add_argument("--strided", type=int, default=0)
ap.add_argument("--size", type=int, default=5424)
ap.add_argument("--torch-only", type=int, default=0, help="control: this many plain torch kernels, no library call")
ap.add_argument("--repeat", type=int, default=1)
ap.add_argument("--vr", type=int, default=0, help="also run one variational refinement of the first pair")
a = ap.parse_args()
H = a.size
g = torch.Generator(device="cuda")
g.manual_seed(1)
if a.torch_only:
    x = torch.rand((1024, 1024), device="cuda", generator=g)
    for i in range(a.torch_only):
        x = x * 1.0001 + 0.5
    torch.cuda.synchronize()
    print("done torch-only", float(x[0, 0]), flush=True)
    sys.exit(0)
from tobac_flow_amd.utils.flow_utils import FarnebackFlow
F = torch.nn.functional
B = a.B
base = torch.rand((1, 1, H + 16, H + 16), device="cuda", generator=g)
for _ in range(3):
    base = F.avg_pool2d(base, 9, stride=1, padding=4)
base = ((base - base.min()) / (base.max() - base.min()) * 255).to(torch.uint8)[0, 0]
frames = torch.stack([base[8 + i:8 + i + H, 8 - i:8 - i + H] for i in range(B + 1)]).contiguous()   # drifting content
prev, nxt = frames[:-1].contiguous(), frames[1:].contiguous()
if a.strided:
    fwd_all = torch.empty((B + 3, H, H, 2), dtype=torch.float32, device="cuda")
    bwd_all = torch.empty_like(fwd_all)
    fwd, bwd = fwd_all[1:1 + B], bwd_all[2:2 + B]
else:
    fwd = torch.empty((B, H, H, 2), dtype=torch.float32, device="cuda")
    bwd = torch.empty_like(fwd)
m = FarnebackFlow(num_levels=a.levels)
for _ in range(a.repeat):
    m.calc_batch_dev(prev, nxt, fwd, bwd)
if a.vr:
    import tobac_flow_amd.flow as tf
    tf.vr_model.calc_dev(prev[0], nxt[0], fwd[0].contiguous())
torch.cuda.synchronize()
print("done", float(fwd[0, H // 2, H // 2, 0]), flush=True)
