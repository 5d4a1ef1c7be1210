"""Seeded synthetic BT-like stacks generated ON THE GPU (torch) for bench.py and the large-size tests
(translating cold blobs on a 290 K background + band-limited noise + NaN patch)."""
import math

import numpy as np


def blob_stack(T, H, W, seed=20240601, device="cuda", nan_every=12, t0=0):
    """Frames t0 .. t0 + T - 1 of ONE endless synthetic sequence: frame content depends on (seed, global frame index)
    only, so two windows cut from the same sequence (e.g. the overlapping windows of adjacent ranks) share their common
    frames bit for bit.

    A Gaussian blob is separable, so a frame is ONE matrix product: 290 - (amp * Gy)^T @ Gx with Gy (K, H), Gx (K, W)
    the 1-D profiles of the K blobs -- a dozen launches per frame instead of several per blob (449 blobs per 5424^2
    frame; the counter-collection mode of rocprofv3 does not survive ~10 000 dispatches per process on this stack,
    profiles/README.md)."""
    import torch
    g = np.random.default_rng(seed)
    K = max(8, H * W // 65536)
    cy, cx = g.uniform(0, H, K), g.uniform(0, W, K)
    vy, vx = g.uniform(-3, 3, K), g.uniform(-3, 3, K)
    amp, sig = g.uniform(20, 60, K), g.uniform(6, 40, K)
    out = torch.empty((T, H, W), dtype=torch.float32, device=device)
    gen = torch.Generator(device=device)
    F = torch.nn.functional
    f64 = dict(dtype=torch.float64, device=device)
    ys, xs = torch.arange(H, **f64)[None, :], torch.arange(W, **f64)[None, :]
    inv = torch.tensor(1.0 / (2.0 * sig * sig), **f64)[:, None]
    a = torch.tensor(amp, **f64)[:, None]
    for ti in range(T):
        t = t0 + ti
        gen.manual_seed(seed * 1000003 + t)
        y0 = torch.tensor(cy + vy * t, **f64)[:, None]
        x0 = torch.tensor(cx + vx * t, **f64)[:, None]
        gy = (a * torch.exp(-(ys - y0) ** 2 * inv)).to(torch.float32)          # (K, H)
        gx = torch.exp(-(xs - x0) ** 2 * inv).to(torch.float32)                # (K, W)
        frame = out[ti]
        torch.matmul(gy.t(), gx, out=frame)
        frame.neg_().add_(290.0)
        # band-limited noise: white noise box-filtered three times (~ gaussian, sigma ~ 2 px)
        n = torch.randn((1, 1, H, W), generator=gen, device=device, dtype=torch.float32)
        for _ in range(3):
            n = F.avg_pool2d(n, 5, stride=1, padding=2, count_include_pad=False)
        frame += n[0, 0] * 16.0
        if nan_every and t % nan_every == nan_every // 2:
            hh, ww = max(H // 10, 1), max(W // 10, 1)
            gt = np.random.default_rng([seed, t])
            y1, x1 = int(gt.integers(0, H - hh + 1)), int(gt.integers(0, W - ww + 1))
            frame[y1:y1 + hh, x1:x1 + ww] = float("nan")
    return out


def anvil_inputs(bt, lower=270.0, upper=250.0):
    """linearised field, eroded markers (+1) and background seed (-1) as in detect_anvils with
    markers=None (reference detection.py:545-561), computed with torch on the GPU."""
    import torch
    F = torch.nn.functional
    lo, hi = min(lower, upper), max(lower, upper)
    lin = ((bt - lo) / (hi - lo)).clamp(0, 1)
    if lower > upper:
        lin = 1 - lin
    core = (lin >= 1)
    # erosion by the in-plane 4-neighbour cross (border_value = 0)
    c = core.float()[:, None]
    p = F.pad(c, (1, 1, 1, 1), value=0.0)
    er = c * p[:, :, :-2, 1:-1] * p[:, :, 2:, 1:-1] * p[:, :, 1:-1, :-2] * p[:, :, 1:-1, 2:]
    markers = er[:, 0].to(torch.int32)
    # background: (lin <= 0 or NaN) eroded by the full 3x3x3 cube, border_value = 1, NaNs kept
    nan = torch.isnan(lin)
    bg = ((lin <= 0) | nan).float()[None, None]
    bgp = F.pad(bg, (1, 1, 1, 1, 1, 1), value=1.0)
    bge = -F.max_pool3d(-bgp, 3, stride=1)
    bgm = (bge[0, 0] > 0.5) | nan
    markers[bgm] = -1
    return lin, markers
