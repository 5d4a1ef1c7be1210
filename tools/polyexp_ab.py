"""the polynomial expansion on 64 x 32 tiles (TF_FB_POLYEXP_TH=32) against 64 x 16.  One process per
variant: the library's HIP-event time of fb_polyexp (and of its neighbours) over a batch of 5424^2 frame pairs, and the digest
of the raw flow (must not change).   python tools/polyexp_ab.py [size] [pairs]"""
import hashlib
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    import tobac_flow_amd.flow as tf
    from tobac_flow_amd import _lib
    from tools.synth import blob_stack
    H = W = int(sys.argv[2])
    B = int(sys.argv[3])
    bt = blob_stack(B + 1, H, W)
    fl = tf.create_flow(bt, model="Farneback")
    fl.check()
    torch.cuda.synchronize()
    digest = hashlib.sha1(fl.forward_flow.cpu().numpy().tobytes()).hexdigest()[:16]
    del fl
    _lib.profile_enable(True)
    _lib.profile_collect()
    reps = 4
    for _ in range(reps):
        fl = tf.create_flow(bt, model="Farneback")
        fl.check()
        del fl
    torch.cuda.synchronize()
    prof = _lib.profile_collect()
    out = {k: round(v[1] / reps / B, 3) for k, v in prof.items() if k in ("fb_polyexp", "fb_gaussian_blur", "fb_resize", "fb_iteration_fused")}
    print("TH=%s  ms per pair: %s  digest %s" % (os.environ.get("TF_FB_POLYEXP_TH", "16"), out, digest), flush=True)
    sys.exit(0)

size = sys.argv[1] if len(sys.argv) > 1 else "5424"
pairs = sys.argv[2] if len(sys.argv) > 2 else "8"
for th in ("16", "32", "16", "32"):
    subprocess.run([sys.executable, os.path.abspath(__file__), "child", size, pairs], env=dict(os.environ, TF_FB_POLYEXP_TH=th), check=False, timeout=300)
