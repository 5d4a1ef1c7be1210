"""k_vr_sor_tile on half-height tiles with two workgroups per CU (TF_VR_TILE=half) against the full tile.
One process per variant (the switch is read once): per-kernel time from the library's HIP events over N refinements of a
5424^2 frame pair, and bit identity of the refined flow with the default.
    python tools/vr_tile_ab.py [size] [batch]"""
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
    lo, hi = bt.min(), bt.max()
    u8 = ((bt - lo) / (hi - lo) * 255).to(torch.uint8).contiguous()
    g = torch.Generator(device="cuda").manual_seed(1)
    flow0 = torch.randn((B, H, W, 2), device="cuda", generator=g)
    vr = tf.VariationalRefinement.create()
    f = flow0.clone()
    vr.calc_batch_dev(u8[:B], u8[1:], f)
    torch.cuda.synchronize()
    digest = hashlib.sha1(f.cpu().numpy().tobytes()).hexdigest()[:16]
    _lib.profile_enable(True)
    _lib.profile_collect()
    reps = 6
    for _ in range(reps):
        f = flow0.clone()
        vr.calc_batch_dev(u8[:B], u8[1:], f)
    torch.cuda.synchronize()
    prof = _lib.profile_collect()
    out = {k: round(v[1] / reps / B, 3) for k, v in prof.items()}
    print("TILE=%s STAGGER=%s  ms per refinement: %s  sum %.3f  digest %s" % (os.environ.get("TF_VR_TILE", "full"), os.environ.get("TF_VR_STAGGER", "0"),
                                                                              out, sum(out.values()), digest), flush=True)
    sys.exit(0)

size = sys.argv[1] if len(sys.argv) > 1 else "5424"
batch = sys.argv[2] if len(sys.argv) > 2 else "2"
for tile, stagger in (("full", "0"), ("half", "0"), ("half", "200"), ("half", "600"), ("full", "0")):
    env = dict(os.environ, TF_VR_TILE=tile, TF_VR_STAGGER=stagger)
    subprocess.run([sys.executable, os.path.abspath(__file__), "child", size, batch], env=env, check=False, timeout=300)
