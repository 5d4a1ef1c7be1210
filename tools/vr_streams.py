"""Do independent refinements overlap when issued on several HIP streams?  (development aid)
N independent 5424^2 refinements on 1, 2, 3, 4 streams (own workspace per stream)."""
import ctypes, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import tobac_flow_amd.flow as tf
from tobac_flow_amd import _lib
from tools.synth import blob_stack
H = W = 5424
N = 8
bt = torch.as_tensor(blob_stack(2, H, W)).cuda().float()
lo, hi = bt.min(), bt.max()
a, b = [((bt[i] - lo) / (hi - lo) * 255).to(torch.uint8).contiguous() for i in (0, 1)]
g = torch.Generator(device="cuda").manual_seed(1)
flows = [torch.randn((H, W, 2), device="cuda", generator=g) for _ in range(N)]
L = _lib.lib()
p = _lib.VarRefParams(5, 5, 20.0, 5.0, 10.0, 1.6)
nb = L.tf_varref_workspace_bytes(H, W)
for ns in (1, 2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    wss = [torch.empty(nb, dtype=torch.uint8, device="cuda") for _ in range(ns)]
    def run():
        for i in range(N):
            k = i % ns
            _lib.check(L.tf_varref(_lib.ptr(a), _lib.ptr(b), H, W, ctypes.byref(p), _lib.ptr(flows[i]), _lib.ptr(wss[k]), nb,
                                   ctypes.c_void_p(streams[k].cuda_stream)), "tf_varref")
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    print("streams %d: %.3f ms per refinement" % (ns, (time.perf_counter() - t0) / 3 / N * 1e3), flush=True)
