"""binary_morph16: the XCD-aware workgroup -> (tile, t) mapping (round 6) against the (x, y, t) order (TF_MORPH_GRID=plane), alone
on the device: library HIP events over erosions of a 16 x 5424^2 volume with the 3 x 3 x 3 and the in-plane structure."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    from scipy import ndimage as ndi
    from tobac_flow_amd import _lib, ndimage_dev as nd
    g = torch.Generator(device="cuda").manual_seed(1)
    x = (torch.rand((16, 5424, 5424), device="cuda", generator=g) > 0.3)
    full = np.ones((3, 3, 3)); plane = ndi.generate_binary_structure(3, 1); plane[0] = 0; plane[2] = 0
    out = {}
    for name, st in (("3x3x3", full), ("in-plane", plane)):
        nd.binary_erosion(x, st); torch.cuda.synchronize()
        _lib.profile_enable(True); _lib.profile_collect()
        for _ in range(10):
            r = nd.binary_erosion(x, st)
        torch.cuda.synchronize()
        p = _lib.profile_collect(); _lib.profile_enable(False)
        out[name] = (round(p["binary_morph"][1] / 10, 3), int(r.sum()))
    print(os.environ.get("TF_MORPH_GRID", "xcd"), out, flush=True)
    sys.exit(0)
for grid in ("xcd", "plane", "xcd", "plane"):
    subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, TF_MORPH_GRID=grid), check=False, timeout=200)
