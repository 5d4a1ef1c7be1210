"""Host <-> HBM rates of the staging path (csrc/staging.hip) beside torch's own copies: pageable upload through the pinned
ring, pinned upload, download into a pooled pinned block, host and device checksum, pinned allocation.
    python tools/staging_bw.py [--gb 1.88]"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=float, default=1.883)
    args = ap.parse_args()
    import torch
    from tobac_flow_amd import _lib, _staging
    L = _lib.lib()
    n = int(args.gb * 1e9) // 16 * 16
    print("cpus", len(os.sched_getaffinity(0)), "bytes", n, flush=True)
    a = np.empty(n, np.uint8)
    a[:] = 7                                                               # (touch the pages)
    a[::4099] = 3
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()

    def timed(label, fn, reps=3, sync=True):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            if sync:
                torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print("%-46s %8.1f ms  %6.1f GB/s" % (label, best * 1e3, n / best / 1e9), flush=True)
        return best

    timed("torch pageable H->D (copy_)", lambda: d.copy_(torch.from_numpy(a)))
    timed("tf_upload pageable (ring, no hash)", lambda: _lib.check(L.tf_upload(_lib.ptr(d), a.ctypes.data_as(_lib._P), n, None, _lib.stream_ptr()), "up"))
    h = np.zeros(2, np.uint64)
    timed("tf_upload pageable (ring + hash)", lambda: _lib.check(L.tf_upload(_lib.ptr(d), a.ctypes.data_as(_lib._P), n, h.ctypes.data_as(_lib._P), _lib.stream_ptr()), "up"))
    timed("tf_hash_host", lambda: L.tf_hash_host(a.ctypes.data_as(_lib._P), n, h.ctypes.data_as(_lib._P)), sync=False)
    h2 = np.zeros(2, np.uint64)
    timed("tf_hash_dev (incl. sync)", lambda: L.tf_hash_dev(_lib.ptr(d), n, h2.ctypes.data_as(_lib._P), _lib.stream_ptr()))
    print("hash host == dev:", bool((h == h2).all()), flush=True)
    t0 = time.perf_counter()
    pinned = _staging.empty_pinned((n,), np.uint8)
    print("%-46s %8.1f ms" % ("tf_host_alloc (first: hipHostMalloc)", (time.perf_counter() - t0) * 1e3), flush=True)
    timed("tf_download into pinned block", lambda: _lib.check(L.tf_download(ctypes.c_void_p(pinned.ctypes.data), _lib.ptr(d), n, _lib.stream_ptr()), "down"))
    print("round trip equal:", bool(np.array_equal(pinned, a)), flush=True)
    timed("tf_upload from pinned block", lambda: _lib.check(L.tf_upload(_lib.ptr(d), ctypes.c_void_p(pinned.ctypes.data), n, None, _lib.stream_ptr()), "up"))
    timed("torch D->H pageable (.cpu())", lambda: d.cpu())
    timed("numpy memcpy pageable -> pinned (1 thread)", lambda: np.copyto(pinned, a), sync=False)
    del pinned
    t0 = time.perf_counter()
    pinned = _staging.empty_pinned((n,), np.uint8)
    print("%-46s %8.1f ms" % ("tf_host_alloc (pooled)", (time.perf_counter() - t0) * 1e3), flush=True)
    # fresh arrays every time (nothing the runtime could have pinned before): what a script's one upload per array sees
    def fresh_upload():
        b = np.empty(n, np.uint8)
        b[:] = 5
        t0 = time.perf_counter()
        _lib.check(L.tf_upload(_lib.ptr(d), b.ctypes.data_as(_lib._P), n, None, _lib.stream_ptr()), "up")
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    ts = [fresh_upload() for _ in range(4)]
    print("%-46s %s ms" % ("tf_upload, a FRESH pageable array each time", [round(t * 1e3, 1) for t in ts]), flush=True)

    def fresh_torch():
        b = np.empty(n, np.uint8)
        b[:] = 5
        t0 = time.perf_counter()
        d.copy_(torch.from_numpy(b))
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    ts = [fresh_torch() for _ in range(4)]
    print("%-46s %s ms" % ("torch copy_, a FRESH pageable array each time", [round(t * 1e3, 1) for t in ts]), flush=True)


if __name__ == "__main__":
    main()
