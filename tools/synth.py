"""Seeded synthetic BT-like stacks generated ON THE GPU (torch) for bench.py and the large-size tests
(translating cold blobs on a 290 K background + band-limited noise + NaN patch)."""
import math

import numpy as np


def blob_stack(T, H, W, seed=20240601, device="cuda", nan_every=12, t0=0):
    """Frames t0 .. t0 + T - 1 of ONE endless synthetic sequence: frame content depends on (seed, global frame index)
    only, so two windows cut from the same sequence (e.g. the overlapping windows of adjacent ranks) share their common
    frames bit for bit."""
    import torch
    g = np.random.default_rng(seed)
    K = max(8, H * W // 65536)
    cy, cx = g.uniform(0, H, K), g.uniform(0, W, K)
    vy, vx = g.uniform(-3, 3, K), g.uniform(-3, 3, K)
    amp, sig = g.uniform(20, 60, K), g.uniform(6, 40, K)
    out = torch.full((T, H, W), 290.0, dtype=torch.float32, device=device)
    gen = torch.Generator(device=device)
    F = torch.nn.functional
    for ti in range(T):
        t = t0 + ti
        frame = out[ti]
        gen.manual_seed(seed * 1000003 + t)
        for k in range(K):
            y0, x0 = cy[k] + vy[k] * t, cx[k] + vx[k] * t
            rad = int(4 * sig[k]) + 1
            ya, yb = max(int(y0) - rad, 0), min(int(y0) + rad + 1, H)
            xa, xb = max(int(x0) - rad, 0), min(int(x0) + rad + 1, W)
            if ya >= yb or xa >= xb:
                continue
            yy = torch.arange(ya, yb, device=device, dtype=torch.float32)[:, None] - float(y0)
            xx = torch.arange(xa, xb, device=device, dtype=torch.float32)[None, :] - float(x0)
            frame[ya:yb, xa:xb] -= float(amp[k]) * torch.exp(-(yy * yy + xx * xx) / float(2 * sig[k] ** 2))
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
