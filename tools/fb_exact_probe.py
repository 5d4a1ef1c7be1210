"""Raw Farneback flow of the library against the CPU oracle, bit for bit (sequential row sums, round 4): a few sizes that
exercise one strip / several strips / the ragged last strip / tiny levels, both directions; then timing of one batch.
    python tools/fb_exact_probe.py [--full]"""
import os
import sys
import time

import numpy as np
import scipy.ndimage as ndi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import tobac_flow_amd.flow as tf
    from test_gpu_parity import _oracle_farneback
    model = tf.select_of_model("Farneback")
    rng = np.random.default_rng(5)
    shapes = [(96, 128), (150, 250), (333, 517), (40, 64), (33, 35), (700, 1100), (260, 116), (260, 117), (260, 232), (260, 233)]
    if "--full" in sys.argv:
        shapes.append((5424, 5424))
    for H, W in shapes:
        img = ndi.gaussian_filter(rng.normal(size=(H + 8, W + 8)), 3)
        img = ((img - img.min()) / np.ptp(img) * 255).astype(np.uint8)
        a, b = np.ascontiguousarray(img[4:4 + H, 4:4 + W]), np.ascontiguousarray(img[2:2 + H, 7:7 + W])
        t0 = time.perf_counter()
        f, bk = tf.calculate_flow_frame(a, b, model)
        t1 = time.perf_counter()
        wf = _oracle_farneback(a, b)
        wb = _oracle_farneback(b, a) if H < 2000 else None
        t2 = time.perf_counter()
        df = np.abs(f - wf)
        msg = "%5d x %5d: fwd max %.3g, differing %d of %d" % (H, W, np.nanmax(df), int((f != wf).sum()), f.size)
        if wb is not None:
            msg += "; bwd max %.3g, differing %d" % (np.nanmax(np.abs(bk - wb)), int((bk != wb).sum()))
        print(msg + "   (gpu call %.2f s, oracle %.1f s)" % (t1 - t0, t2 - t1), flush=True)
        if not np.isfinite(f).all():
            print("   NON-FINITE flow: a chain gave up waiting"); break


if __name__ == "__main__":
    main()
