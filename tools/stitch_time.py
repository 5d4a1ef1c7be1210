"""Cost of the local (non-communication) parts of parallel.stitch_labels on one GPU at benchmark size."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from tobac_flow_amd.parallel import apply_global_lut
T, H, W = 12, 5424, 5424
g = torch.Generator(device="cuda").manual_seed(1)
labels = torch.randint(1, 5000, (T, H, W), dtype=torch.int32, device="cuda", generator=g)
lut = np.concatenate([[0], np.random.default_rng(0).permutation(np.arange(1, 5000))])
def timed(name, fn, n=3):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t) / n * 1e3:.2f} ms", flush=True)
    return r
timed("apply_global_lut (tf_apply_lut)", lambda: apply_global_lut(labels, lut))
def torch_path():
    lut_t = torch.from_numpy(lut.astype(np.int32)).cuda()
    pos = labels > 0
    out = labels.clone()
    out[pos] = lut_t[labels[pos].to(torch.int64)]
    return out
timed("boolean-mask torch path (previous)", torch_path, n=1)
a, b = labels[-1].reshape(-1).to(torch.int64), labels[0].reshape(-1).to(torch.int64)
def pairs():
    both = (a > 0) & (b > 0)
    base = 5001
    key, cnt = torch.unique(a[both] * base + b[both], return_counts=True)
    return key[cnt >= 1]
timed("boundary pairs (one frame, torch.unique)", pairs)
timed("labels.max()", lambda: torch.clamp(labels.max(), min=0))
