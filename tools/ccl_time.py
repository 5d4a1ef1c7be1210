"""Times tf_label (scipy.ndimage.label on the GPU) and binary_fill_holes on the masks the detection recipes hand them, at a
window's size: the cold-blob mask (sparse foreground, the seeds of the flood) and its complement (the background that
binary_fill_holes labels: one giant component per frame).  python tools/ccl_time.py [frames] [size]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.ndimage as ndi
import torch

from tobac_flow_amd import ndimage_dev as nd
from tools.synth import blob_stack

T = int(sys.argv[1]) if len(sys.argv) > 1 else 16
S = int(sys.argv[2]) if len(sys.argv) > 2 else 5424
bt = blob_stack(T, S, S)
fg = (bt < 262) & ~torch.isnan(bt)
plane = ndi.generate_binary_structure(3, 1)
plane[0] = 0
plane[2] = 0
full = ndi.generate_binary_structure(3, 1)


def timed(name, fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    print("%-44s %8.2f ms" % (name, (time.perf_counter() - t0) / reps * 1e3), flush=True)
    return out


print("foreground fraction %.3f" % float(fg.float().mean()))
for sname, st in (("plane", plane), ("3-D conn 1", full), ("3-D conn 3", np.ones((3, 3, 3), bool))):
    a = timed("label(foreground, %s)" % sname, lambda: nd.label(fg, st))
    b = timed("label(background, %s)" % sname, lambda: nd.label(~fg, st))
    print("   components:", a[1], b[1])
timed("binary_fill_holes(foreground, plane)", lambda: nd.binary_fill_holes(fg, plane))
