"""Generate golden vectors for the semi-Lagrangian watershed from the REFERENCE ITSELF.

Run with the interpreter that has scikit-image (the reference's watershed.py imports skimage
private helpers):

    /opt/conda/bin/python3.9 tests/golden/make_watershed_golden.py

What it does:
  * compiles /root/reference/tobac_flow/_watershed.pyx for that interpreter (oracle/build_ref.py),
  * executes /root/reference/tobac_flow/watershed.py from a temp dir with ONE prepended line
    (``from __future__ import annotations`` -- python3.9 cannot evaluate its PEP-604 annotations),
  * runs it on seeded synthetic inputs and stores inputs + outputs as .npz under tests/golden/.

Only data (inputs / expected outputs / neighbour-order vectors) is written into the repository.
"""
import os
import sys
import tempfile
import warnings

warnings.filterwarnings("ignore")
import numpy as np
import scipy.ndimage as ndi

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import build_ref  # noqa: E402

so = build_ref.build(sys.executable)
tmp = tempfile.mkdtemp(prefix="tf_ref_")
pkg = os.path.join(tmp, "tobac_flow")
os.makedirs(pkg)
open(os.path.join(pkg, "__init__.py"), "w").close()
os.symlink(so, os.path.join(pkg, os.path.basename(so)))
with open("/root/reference/tobac_flow/watershed.py") as f:
    src = f.read()
with open(os.path.join(pkg, "watershed.py"), "w") as f:
    f.write("from __future__ import annotations\n" + src)
sys.path.insert(0, tmp)
from tobac_flow.watershed import watershed as ref_watershed  # noqa: E402
from skimage.morphology._util import _offsets_to_raveled_neighbors  # noqa: E402
from skimage.segmentation import watershed as sk_watershed  # noqa: E402


def rand_flow(rng, shape, amp, smooth=3.0):
    f = rng.normal(size=shape + (2,)).astype(np.float32)
    f = ndi.gaussian_filter(f, (0, smooth, smooth, 0)) * amp * 4
    return np.clip(f, -amp * 2, amp * 2).astype(np.float32)


def seeds(rng, shape, n, with_bg=True):
    m = np.zeros(shape, np.int32)
    for k in range(n):
        t, y, x = [rng.integers(0, s) for s in shape]
        m[t, max(y - 1, 0):y + 2, max(x - 1, 0):x + 2] = k + 1
    if with_bg:
        t, y, x = [rng.integers(0, s) for s in shape]
        m[t, y, x] = -1
    return m


cases = {}
rng = np.random.default_rng(20240601)
shape = (5, 36, 44)

# A: continuous field, conn 1, no mask, non-zero flow
field = ndi.gaussian_filter(rng.normal(size=shape), (0.7, 2, 2)).astype(np.float32)
cases["A_cont_c1"] = dict(field=field, markers=seeds(rng, shape, 12), mask=None,
                          fwd=rand_flow(rng, shape, 1.5), bwd=rand_flow(rng, shape, 1.5), conn=1)
# B: continuous, mask, conn 2 / 3
mask = ndi.gaussian_filter(rng.normal(size=shape), (0, 3, 3)) > -0.05
cases["B_cont_mask_c2"] = dict(field=field * 2, markers=seeds(rng, shape, 9) * mask, mask=mask,
                               fwd=rand_flow(rng, shape, 1.0), bwd=rand_flow(rng, shape, 1.0), conn=2)
cases["B_cont_mask_c3"] = dict(field=-field, markers=seeds(rng, shape, 9) * mask, mask=mask,
                               fwd=rand_flow(rng, shape, 1.0), bwd=rand_flow(rng, shape, 1.0), conn=3)
# C: quantised fields (heavy ties between non-marker pixels)
q = np.floor((field - field.min()) / (field.max() - field.min()) * 3.999).astype(np.float32)
cases["C_quant4_c1"] = dict(field=q, markers=seeds(rng, shape, 10), mask=None,
                            fwd=rand_flow(rng, shape, 1.0), bwd=rand_flow(rng, shape, 1.0), conn=1)
q = np.floor((field - field.min()) / (field.max() - field.min()) * 31.999).astype(np.float32)
cases["C_quant32_c1"] = dict(field=q, markers=seeds(rng, shape, 10), mask=None,
                             fwd=rand_flow(rng, shape, 1.0), bwd=rand_flow(rng, shape, 1.0), conn=1)
# D: production-like: [0,1]-clipped field with exact plateaus, edge field = grad + 1 - field,
#    markers = labelled eroded plateau, background -1 where field <= 0 (eroded), cf. detection.py:547-567
yy, xx = np.mgrid[0:shape[1], 0:shape[2]]
raw = np.zeros(shape, np.float32)
for k, (cy, cx) in enumerate([(8, 9), (9, 33), (26, 12), (27, 32)]):
    vy, vx = rng.uniform(-1.0, 1.0, 2)
    s = rng.uniform(2.5, 4.0)
    for t in range(shape[0]):
        raw[t] += 2.2 * np.exp(-((yy - cy - vy * t) ** 2 + (xx - cx - vx * t) ** 2) / (2 * s * s))
raw += ndi.gaussian_filter(rng.normal(size=shape), (0, 1, 1)).astype(np.float32) * 0.15
lin = np.clip((raw - 0.3) / (1.2 - 0.3), 0, 1).astype(np.float32)
g = np.sqrt(sum(np.maximum(d, 0) ** 2 for d in np.gradient(lin.astype(np.float64))))
edges = g.copy()
edges[edges > 0] += 1
edges = (edges - lin).astype(np.float32)
s_struct = ndi.generate_binary_structure(3, 1) * np.array([0, 1, 0])[:, None, None].astype(bool)
mk = ndi.label(lin >= 1, structure=ndi.generate_binary_structure(3, 1))[0].astype(np.int32)
mk = mk * ndi.binary_erosion(mk != 0, structure=s_struct)
bg = ndi.binary_erosion(lin <= 0, structure=np.ones([3, 3, 3]), border_value=1)
mk[bg] = -1
fw = np.zeros(shape + (2,), np.float32)
fw[..., 0] = 1.2
fw[..., 1] = -0.7
fw += rand_flow(rng, shape, 0.5)
cases["D_anvil_like_c1"] = dict(field=edges, markers=mk, mask=None, fwd=fw, bwd=-fw, conn=1)
# E: constant plateau, 8 single-pixel markers, zero flow (heap-internal marker order)
shape_e = (2, 24, 24)
me = np.zeros(shape_e, np.int32)
for k in range(8):
    t, y, x = [rng.integers(0, s) for s in shape_e]
    me[t, y, x] = k + 1
cases["E_const_plateau_c1"] = dict(field=np.zeros(shape_e, np.float32), markers=me, mask=None,
                                   fwd=np.zeros(shape_e + (2,), np.float32),
                                   bwd=np.zeros(shape_e + (2,), np.float32), conn=1)
# F: continuous, zero flow, must equal skimage's own watershed
cases["F_zero_flow_c1"] = dict(field=field, markers=seeds(rng, shape, 7, with_bg=False), mask=None,
                               fwd=np.zeros(shape + (2,), np.float32),
                               bwd=np.zeros(shape + (2,), np.float32), conn=1)
# G: larger flows (padding > 1), half-integer flows exercise round-half-to-even
fwg = np.zeros(shape + (2,), np.float32)
fwg[..., 0] = 2.5
fwg[..., 1] = -3.5
fwg[:, ::2] += 1.0
cases["G_big_flow_c1"] = dict(field=field, markers=seeds(rng, shape, 8), mask=None,
                              fwd=fwg, bwd=-fwg[:, ::-1].copy(), conn=1)

out = {}
for name, c in cases.items():
    labels = ref_watershed(c["fwd"], c["bwd"], c["field"], c["markers"], mask=c["mask"],
                           connectivity=c["conn"])
    assert labels.dtype == np.int32
    if name.startswith("F_"):
        assert np.array_equal(labels, sk_watershed(c["field"], c["markers"], connectivity=1))
    for k, v in c.items():
        if v is not None:
            out[name + "/" + k] = np.asarray(v)
    out[name + "/labels"] = labels
    print(name, "labels:", np.unique(labels).size, "unlabelled:", int((labels == 0).sum()))

# neighbour orders produced by skimage 0.18.3's non-stable argsort
pshape = (5, 30, 40)
S = np.array([pshape[1] * pshape[2], pshape[2], 1])
for conn in (1, 2, 3):
    selem = ndi.generate_binary_structure(3, conn)
    rav = _offsets_to_raveled_neighbors(pshape, selem, (1, 1, 1))
    offs = []
    for r in rav:
        found = [(a, b, c) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1)
                 if a * S[0] + b * S[1] + c * S[2] == r]
        assert len(found) == 1
        offs.append(found[0])
    out["neighbour_order/conn%d" % conn] = np.array(offs, np.int8)
np.savez_compressed(os.path.join(HERE, "watershed_ref.npz"), **out)
print("wrote", os.path.join(HERE, "watershed_ref.npz"),
      os.path.getsize(os.path.join(HERE, "watershed_ref.npz")), "bytes")
