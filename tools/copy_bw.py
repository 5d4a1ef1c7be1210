"""Device-to-device copy rate of the library's plain copy kernel in its development forms, and of torch's copy_ (the
practical HBM ceiling bench.py reports next to the 8 TB/s peak).  python tools/copy_bw.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tobac_flow_amd import _lib

L = _lib.lib()
for gib in (1, 4):
    n = gib << 28
    src = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
    dst = torch.empty_like(src)

    def rate(fn, reps=10):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return reps * 2 * 4 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9
    print("%d GiB: torch copy_ %.0f GB/s" % (gib, rate(lambda: dst.copy_(src))), flush=True)
    for v, name in ((0, "4 x 256 words per tile, 8 wg / CU, nontemporal (default)"), (7, "the same, plain loads / stores"), (1, "4 per tile, 16 wg / CU"), (2, "8 per tile"), (3, "4 per tile, nontemporal"),
                    (4, "8 per tile, nontemporal"), (5, "2 per tile"), (6, "1 per tile, 16 wg / CU")):
        r = rate(lambda: _lib.check(L.tf_copy16_variant(_lib.ptr(src), _lib.ptr(dst), 4 * n, _lib.stream_ptr(), v), "copy"))
        assert torch.equal(src, dst)
        print("       variant %d (%s): %.0f GB/s" % (v, name, r), flush=True)
    del src, dst
