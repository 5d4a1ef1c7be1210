import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import tobac_flow_amd.flow as tf
from tobac_flow_amd.detection import get_combined_edge_field
from tobac_flow_amd.watershed import neighbour_offsets, watershed_dev
from tools.synth import anvil_seeds, blob_stack
for t0 in (0, 36, 100):
    bt = blob_stack(16, 5424, 5424, seed=20240601, t0=t0)
    fl = tf.create_flow(bt, vr_steps=1, smoothing_passes=1, interp_method="cubic")
    lin, seeds = anvil_seeds(bt)
    e = get_combined_edge_field(fl, lin, dtype=np.float32)
    fw, bw = fl._dev_flows()
    nbr = neighbour_offsets(1)
    ref = None
    for depth in (3, 2, 1):
        for rep in range(2):
            st = {}
            torch.cuda.synchronize(); t = time.perf_counter()
            lab = watershed_dev(fw, bw, e, seeds, None, nbr, depth, st, expect_conflict=True, on_ambiguous="ignore")
            torch.cuda.synchronize(); dt = time.perf_counter() - t
        if ref is None: ref = lab
        print("t0", t0, "start depth", depth, "-> used", st["chain_depth"], "sweeps", st["sweeps"][:5], "%.1f ms" % (dt * 1e3), "same labels", bool(torch.equal(lab, ref)), flush=True)
