"""decide TF_VR_FAST_DIVIDE on evidence.  Config S (16 x 512 x 512, vr_steps=1, smoothing_passes=1,
cubic) against the oracle, with the correctly rounded divisions of the default and with the hardware reciprocals
(VariationalRefinement.fastDivide): maximum / count beyond 1e-4 of the composed flow, and the same for the refinement
stage alone on the GPU's own raw vectors (where the default is bit-identical to the oracle)."""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import blob_sequence                                    # noqa: E402
from test_gpu_configs import _oracle_flow                            # noqa: E402
import tobac_flow_amd.flow as tf                                     # noqa: E402

rng = np.random.default_rng(20240601)
bt = blob_sequence(rng, 16, 512, 512, n_blobs=8)
bt[5, 100:150, 200:260] = np.nan
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    want_f, want_b = _oracle_flow(bt, 1, 1, "cubic")
for fast in (False, "sor", True):
    tf.vr_model.fastDivide = fast is True
    tf.vr_model.fastSor = fast == "sor"
    flow = tf.create_flow(bt, model="Farneback", vr_steps=1, smoothing_passes=1, interp_method="cubic")
    for name, got, want in (("forward", flow.forward_flow, want_f), ("backward", flow.backward_flow, want_b)):
        d = np.abs(np.nan_to_num(got) - np.nan_to_num(want))
        print("fastDivide=%s %s: composed flow vs oracle: mean %.3g, 99.9th percentile %.3g, max %.3g, %d of %d components beyond 1e-4"
              % (fast, name, d.mean(), np.percentile(d, 99.9), d.max(), int((d > 1e-4).sum()), d.size), flush=True)
    if fast:
        tf.vr_model.fastDivide = tf.vr_model.fastSor = False
        exact = tf.create_flow(bt, model="Farneback", vr_steps=1, smoothing_passes=1, interp_method="cubic")
        d = np.abs(np.nan_to_num(flow.forward_flow) - np.nan_to_num(exact.forward_flow))
        print("fastDivide vs exact division (GPU vs GPU, forward): mean %.3g, 99.9th %.3g, max %.3g, %d beyond 1e-4, %d differ at all of %d"
              % (d.mean(), np.percentile(d, 99.9), d.max(), int((d > 1e-4).sum()), int((d > 0).sum()), d.size))
