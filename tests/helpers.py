"""Seeded synthetic inputs shared by the parity tests."""
import numpy as np
import scipy.ndimage as ndi


def rand_flow(rng, shape, amp, smooth=3.0):
    f = rng.normal(size=shape + (2,)).astype(np.float32)
    f = ndi.gaussian_filter(f, (0, smooth, smooth, 0)) * amp * 4
    return np.clip(f, -amp * 2, amp * 2).astype(np.float32)


def rand_field(rng, shape, smooth=(0.7, 2, 2), nan_frac=0.0):
    f = ndi.gaussian_filter(rng.normal(size=shape), smooth).astype(np.float32)
    if nan_frac:
        f[rng.random(shape) < nan_frac] = np.nan
    return f


def blob_sequence(rng, T, H, W, n_blobs=6, vmax=3.0, noise=1.0):
    """BT-like translating cold blobs on a 290 K background."""
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    out = np.full((T, H, W), 290.0, np.float32)
    for _ in range(n_blobs):
        cy, cx = rng.uniform(0, H), rng.uniform(0, W)
        vy, vx = rng.uniform(-vmax, vmax, 2)
        amp, sig = rng.uniform(20, 60), rng.uniform(6, max(7, min(H, W) / 8))
        for t in range(T):
            out[t] -= amp * np.exp(-((yy - cy - vy * t) ** 2 + (xx - cx - vx * t) ** 2) / (2 * sig * sig))
    for t in range(T):
        out[t] += ndi.gaussian_filter(rng.normal(size=(H, W)), 2).astype(np.float32) * noise * 4
    return out


def seeds(rng, shape, n, with_bg=True):
    m = np.zeros(shape, np.int32)
    for k in range(n):
        t, y, x = [rng.integers(0, s) for s in shape]
        m[t, max(y - 1, 0):y + 2, max(x - 1, 0):x + 2] = k + 1
    if with_bg:
        t, y, x = [rng.integers(0, s) for s in shape]
        m[t, y, x] = -1
    return m
