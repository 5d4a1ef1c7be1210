"""levels = 0, iters = 1: where does the library's flow first differ from the oracle's?  Crops of ONE image, so that content
and geometry can be told apart."""
import os
import sys

import numpy as np
import scipy.ndimage as ndi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from fb_exact_levels import oracle


def main():
    from tobac_flow_amd.utils.flow_utils import FarnebackFlow
    H, W = 700, 1100
    rng = np.random.default_rng(5)
    img = ndi.gaussian_filter(rng.normal(size=(H + 8, W + 8)), 3)
    img = ((img - img.min()) / np.ptp(img) * 255).astype(np.uint8)
    A, B = np.ascontiguousarray(img[4:4 + H, 4:4 + W]), np.ascontiguousarray(img[2:2 + H, 7:7 + W])
    for (y0, y1, x0, x1) in [(0, 700, 0, 1100), (0, 333, 0, 517), (0, 700, 0, 517), (0, 333, 0, 1100), (100, 400, 200, 400), (0, 700, 232, 348), (0, 700, 116, 464)]:
        a, b = np.ascontiguousarray(A[y0:y1, x0:x1]), np.ascontiguousarray(B[y0:y1, x0:x1])
        f = FarnebackFlow(num_levels=0, num_iters=1).calc(a, b)
        w = oracle(a, b, 0, 1)
        ne = (f != w).any(-1)
        msg = "crop rows %d:%d cols %d:%d  differing px %d" % (y0, y1, x0, x1, int(ne.sum()))
        if ne.any():
            ys, xs = np.nonzero(ne)
            r0 = ys.min()
            msg += "; first row %d (image row %d) cols %s (image cols %s); col range %d..%d" % (r0, r0 + y0, xs[ys == r0], xs[ys == r0] + x0, xs.min() + x0, xs.max() + x0)
        print(msg, flush=True)


if __name__ == "__main__":
    main()
