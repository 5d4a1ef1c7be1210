"""Experiments on the fused Farnebaeck iteration kernel at one resolution (num_levels=0):
both directions vs one, batch size.  Strip height comes from the env TF_FBI_HS."""
import sys
sys.path.insert(0, ".")
import torch
from tools.synth import blob_stack
from tobac_flow_amd import _lib
from tobac_flow_amd.utils.flow_utils import FarnebackFlow
from tobac_flow_amd.utils.normalisation_utils import to_8bit_pair_dev
H = int(sys.argv[1]) if len(sys.argv) > 1 else 5424
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
bt = blob_stack(B + 2, H, H, nan_every=0)
imgs = torch.stack([to_8bit_pair_dev(bt[i], bt[i + 1])[0].clone() for i in range(B + 1)])
m = FarnebackFlow(num_levels=0)


def run(fwd, bwd, name):
    for rep in range(2):
        if rep == 1:
            _lib.profile_enable(True); _lib.profile_collect()
        if B == 1:
            m.calc_pair_dev(imgs[0], imgs[1], fwd, bwd)
        else:
            f = torch.empty((B, H, H, 2), dtype=torch.float32, device="cuda")
            b = torch.empty_like(f)
            m.calc_batch_dev(imgs[:-1].contiguous(), imgs[1:].contiguous(), f, b)
        torch.cuda.synchronize()
    for k, (c, ms, by) in _lib.profile_collect().items():
        if "iter" in k:
            print(f"{name}: {k}: calls {c} avg {ms / c * 1e3:.1f} us  alg {by / ms / 1e6:.0f} GB/s", flush=True)
    _lib.profile_enable(False)


run(True, True, f"H={H} B={B} both")
if B == 1:
    run(True, False, f"H={H} B={B} fwd only")
