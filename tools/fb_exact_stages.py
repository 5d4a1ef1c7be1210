"""blur + polynomial expansion of the library vs the oracle's, bit for bit, on the crop where the flows differ"""
import ctypes
import os
import sys

import numpy as np
import scipy.ndimage as ndi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from oracle import _lib as ol
    from tobac_flow_amd import _lib
    from tobac_flow_amd.utils.flow_utils import FarnebackFlow
    Lo = ol.lib()
    L = _lib.lib()
    H, W = 700, 1100
    rng = np.random.default_rng(5)
    img = ndi.gaussian_filter(rng.normal(size=(H + 8, W + 8)), 3)
    img = ((img - img.min()) / np.ptp(img) * 255).astype(np.uint8)
    A, B = np.ascontiguousarray(img[4:4 + H, 4:4 + W]), np.ascontiguousarray(img[2:2 + H, 7:7 + W])
    y0, y1, x0, x1 = 100, 400, 200, 400
    for name, im in (("prev", A[y0:y1, x0:x1]), ("next", B[y0:y1, x0:x1])):
        im = np.ascontiguousarray(im)
        h, w = im.shape
        f = im.astype(np.float32)
        bl = np.zeros_like(f)
        Lo.oracle_gaussian_blur(ol.ptr(f, ctypes.c_float), h, w, 3, ctypes.c_double(0.0), ol.ptr(bl, ctypes.c_float))
        want = np.zeros((h, w, 5), np.float32)
        Lo.oracle_poly_exp(ol.ptr(bl, ctypes.c_float), h, w, ol.ptr(want, ctypes.c_float), 5, ctypes.c_double(1.1))
        d_img = torch.from_numpy(im).cuda()
        d_bl = torch.empty((h, w), dtype=torch.float32, device="cuda")
        d_R = torch.empty(5 * h * w, dtype=torch.float32, device="cuda")
        m = FarnebackFlow()
        _lib.check(L.tf_farneback_expansion(_lib.ptr(d_img), h, w, ctypes.byref(m.params), _lib.ptr(d_bl), _lib.ptr(d_R), _lib.stream_ptr()), "expansion")
        torch.cuda.synchronize()
        got_bl = d_bl.cpu().numpy()
        R = d_R.cpu().numpy()
        got = np.concatenate([R[:4 * h * w].reshape(h, w, 4), R[4 * h * w:].reshape(h, w, 1)], -1)
        ne = got != want
        print(name, "blur differing", int((got_bl != bl).sum()), "; expansion differing", int(ne.sum()), "of", ne.size)
        if ne.any():
            ys, xs, cs = np.nonzero(ne)
            for k in range(min(12, len(ys))):
                print("   (%d, %d) ch %d (image %d, %d): got %r want %r" % (ys[k], xs[k], cs[k], ys[k] + y0, xs[k] + x0, got[ys[k], xs[k], cs[k]], want[ys[k], xs[k], cs[k]]))


if __name__ == "__main__":
    main()
