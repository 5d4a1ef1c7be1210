import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import tobac_flow_amd.flow as tf
from tobac_flow_amd import _lib
from tobac_flow_amd.watershed import neighbour_offsets, watershed_dev
import torch
z = np.load("tests/golden/watershed_ref.npz")
for name in ["F_zero_flow_c1", "A_cont_c1", "D_anvil_like_c1"]:
    c = {k.split("/")[1]: z[k] for k in z.files if k.startswith(name + "/")}
    for depth in (1, 2, 3):
        st = {}
        got = watershed_dev(_lib.to_dev(c["fwd"]), _lib.to_dev(c["bwd"]), _lib.to_dev(c["field"]), _lib.to_dev(c["markers"]),
                            None, neighbour_offsets(int(c["conn"])), depth, st).cpu().numpy()
        want = c["labels"]
        bad = got != want
        print(name, "depth", depth, "diff", int(bad.sum()), "unlabelled", int((got == 0).sum()), st)
