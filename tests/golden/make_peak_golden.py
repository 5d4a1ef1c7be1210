"""Generate golden vectors for skimage.feature.peak_local_max from scikit-image itself.

The reference calls peak_local_max(field2d, min_distance=10) (tobac_flow/detection.py:154, 161); scikit-image is a
third-party dependency that is absent from the build image but present (0.18.3) in the conda interpreter:

    /opt/conda/bin/python3.9 tests/golden/make_peak_golden.py

Only data is written: tests/golden/peak_local_max_skimage.npz holds the input images and the coordinates
scikit-image returns (row order included) for seeded synthetic fields.
"""
import os
import warnings

warnings.filterwarnings("ignore")
import numpy as np
import scipy.ndimage as ndi
import skimage
from skimage.feature import peak_local_max

HERE = os.path.dirname(os.path.abspath(__file__))
out = {"skimage_version": np.array(skimage.__version__)}
rng = np.random.default_rng(20240607)
cases = []
for i, (shape, sigma) in enumerate([((64, 80), 2.0), ((120, 97), 3.0), ((200, 260), 4.0), ((45, 45), 1.0)]):
    img = ndi.gaussian_filter(rng.normal(size=shape), sigma).astype(np.float32)
    cases.append((f"smooth{i}", img))
    cases.append((f"smooth{i}_neg", -img))
# plateaus and ties: quantised field, saturated field, constant field, tiny image
q = ndi.gaussian_filter(rng.normal(size=(90, 110)), 3.0)
cases.append(("quantised", (np.round(q * 40) / 40).astype(np.float32)))
cases.append(("saturated", np.clip(q * 30, -1, 1).astype(np.float32)))
cases.append(("constant", np.full((40, 50), 2.5, np.float32)))
cases.append(("tiny", ndi.gaussian_filter(rng.normal(size=(9, 11)), 1.0).astype(np.float32)))
cases.append(("float64", ndi.gaussian_filter(rng.normal(size=(70, 75)), 2.5)))
names = []
for name, img in cases:
    out[f"{name}/image"] = img
    for d in (1, 3, 10):
        out[f"{name}/peaks_d{d}"] = np.asarray(peak_local_max(img, min_distance=d), np.int64).reshape(-1, 2)
    names.append(name)
out["names"] = np.array(names)
np.savez_compressed(os.path.join(HERE, "peak_local_max_skimage.npz"), **out)
for name in names:
    print(name, out[f"{name}/image"].shape, {d: len(out[f"{name}/peaks_d{d}"]) for d in (1, 3, 10)})
