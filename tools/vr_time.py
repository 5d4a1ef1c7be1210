"""Time tf_varref on one 5424^2 frame pair for a sweep of (fixedPointIterations, sorIterations) (development aid):
the differences separate the per-launch cost of prepare / weights+system / tile load+store / one half sweep."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import tobac_flow_amd.flow as tf
from tools.synth import blob_stack
H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 5424
bt = blob_stack(2, H, W)
bt = torch.as_tensor(bt).cuda().float()
lo, hi = bt.min(), bt.max()
a, b = [((bt[i] - lo) / (hi - lo) * 255).to(torch.uint8).contiguous() for i in (0, 1)]
g = torch.Generator(device="cuda").manual_seed(1)
flow0 = torch.randn((H, W, 2), device="cuda", generator=g)
vr = tf.VariationalRefinement.create()
for fp, sor in ((0, 0), (1, 0), (5, 0), (5, 1), (5, 2), (5, 3), (5, 4), (5, 5)):
    vr.fixedPointIterations, vr.sorIterations = fp, sor
    f = flow0.clone()
    for _ in range(2):
        vr.calc_dev(a, b, f)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        vr.calc_dev(a, b, f)
    torch.cuda.synchronize()
    print("fp %d sor %d  ms per call %.3f" % (fp, sor, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
