"""Which stage is not bit-identical?  One pair, num_levels = 0 .. 5 and num_iters = 1 / 10, library vs oracle."""
import ctypes
import os
import sys

import numpy as np
import scipy.ndimage as ndi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def oracle(a, b, levels, iters):
    from oracle import _lib as ol
    L = ol.lib()
    h, w = a.shape
    out = np.zeros((h, w, 2), np.float32)
    L.oracle_farneback.restype = ctypes.c_int
    L.oracle_farneback(ol.ptr(a, ctypes.c_uint8), ol.ptr(b, ctypes.c_uint8), h, w, ol.ptr(out, ctypes.c_float), levels, ctypes.c_double(0.5), 13, iters, 5, ctypes.c_double(1.1))
    return out


def main():
    from tobac_flow_amd.utils.flow_utils import FarnebackFlow
    H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (700, 1100)
    rng = np.random.default_rng(5)
    img = ndi.gaussian_filter(rng.normal(size=(H + 8, W + 8)), 3)
    img = ((img - img.min()) / np.ptp(img) * 255).astype(np.uint8)
    a, b = np.ascontiguousarray(img[4:4 + H, 4:4 + W]), np.ascontiguousarray(img[2:2 + H, 7:7 + W])
    for levels in range(6):
        for iters in (1, 10):
            f = FarnebackFlow(num_levels=levels, num_iters=iters).calc(a, b)
            w = oracle(a, b, levels, iters)
            ne = f != w
            rows = np.flatnonzero(ne.any((1, 2)))
            cols = np.flatnonzero(ne.any((0, 2)))
            print("levels %d iters %2d: differing %7d, max %.3g%s" % (levels, iters, int(ne.sum()), np.abs(f - w).max(),
                  "" if not ne.any() else "  rows %d..%d cols %d..%d" % (rows[0], rows[-1], cols[0], cols[-1])), flush=True)


if __name__ == "__main__":
    main()
