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


_TORCH_INDEX_LIMIT = 2 ** 31 - 1       # torch's pooling / padding kernels index in 32 bits (a 4e9-voxel call faults the GPU)


def _padded_elements(frames, H, W):
    """elements of the largest tensor _anvil_inputs_block builds for `frames` frames: the background padded by one voxel
    on every side for the 3 x 3 x 3 pooling"""
    return (frames + 2) * (H + 2) * (W + 2)


def anvil_inputs(bt, lower=270.0, upper=250.0):
    """linearised field, eroded markers (+1) and background seed (-1) as in detect_anvils with
    markers=None (reference detection.py:545-561), computed with torch on the GPU.
    Volumes beyond 2^31 elements are processed in blocks of frames with a one-frame halo (the 3 x 3 x 3 erosion of the
    background reaches one frame): torch's pooling kernels index in 32 bits, and a larger call does not raise, it
    faults (the memory fault of round 2's `gpurun_out/config_f_exact.log`: 144 x 5424^2 = 4.2e9 voxels in one
    max_pool3d call).  A block is sized so that the PADDED tensor that is really built -- block + two halo frames + the
    padding ring -- stays below the limit, which is also what the guard in _anvil_inputs_block tests (the
    round-3 sizing left every block two frames over its own guard, so the blocked path could only raise)."""
    import torch
    T = bt.shape[0]
    H, W = (int(bt.shape[1]), int(bt.shape[2])) if T else (0, 0)
    if T == 0 or _padded_elements(T, H, W) <= _TORCH_INDEX_LIMIT:
        return _anvil_inputs_block(bt, lower, upper)
    block = _TORCH_INDEX_LIMIT // ((H + 2) * (W + 2)) - 4          # + 2 halo frames + 2 padding frames
    if block < 1:
        raise ValueError("anvil_inputs: three padded frames of %d x %d pixels are beyond what torch's 32-bit pooling kernels index" % (H, W))
    lin = torch.empty_like(bt)
    markers = torch.empty(bt.shape, dtype=torch.int32, device=bt.device)
    for a in range(0, T, block):
        b = min(a + block, T)
        lo, hi = max(a - 1, 0), min(b + 1, T)
        # the volume's own first / last frame see border_value = 1 beyond them; interior block edges are halo frames whose
        # own result is discarded
        l, m = _anvil_inputs_block(bt[lo:hi], lower, upper)
        lin[a:b], markers[a:b] = l[a - lo:b - lo], m[a - lo:b - lo]
    return lin, markers


def _anvil_inputs_block(bt, lower, upper):
    import torch
    if bt.shape[0] and _padded_elements(int(bt.shape[0]), int(bt.shape[1]), int(bt.shape[2])) > _TORCH_INDEX_LIMIT:
        raise ValueError("anvil_inputs: block beyond torch's 32-bit pooling kernels")
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


def anvil_seeds(bt, lower=270.0, upper=250.0, erode_distance=1):
    """SURVEY.md 8(d)'s marker recipe for one window, every step a library kernel (tobac_flow_amd/ndimage_dev.py):
    field_lin = linearise_field(bt, lower, upper); markers = label(binary_erosion(field_lin >= 1)) -- window-local
    component ids, as the drop-in scripts pass them (scripts/dcc_detect_goes.py:221-235) -- and -1 where
    get_watershed_mask(field_lin, erode_distance) (detection.py:547-561).  Returns (field_lin f32, seeds i32)."""
    import scipy.ndimage as ndi
    from tobac_flow_amd import ndimage_dev as nd
    lin = nd.linearise_field(bt, lower, upper)
    s = ndi.generate_binary_structure(3, 1) * np.array([0, 1, 0])[:, None, None].astype(bool)
    ge1, le0, isn = nd.field_masks(lin)                      # one pass instead of four elementwise ones
    comp = nd.label(nd.binary_erosion(ge1, s))[0]
    # get_watershed_mask (detection.py:590-617): (field <= 0 | NaN) eroded by the full cube, border_value 1, NaNs kept
    bg = nd.binary_erosion(le0, np.ones([3, 3, 3]), iterations=erode_distance, border_value=1)
    return lin, nd.merge_seeds(comp, bg, isn)


class _TimeCoord:
    """the part of an xarray time coordinate the detection recipes touch: `.values` / `.data` = the datetime64 array"""

    def __init__(self, values):
        self.values = self.data = values

    def __array__(self, dtype=None, copy=None):
        return self.values if dtype is None else self.values.astype(dtype)

    def __len__(self):
        return len(self.values)


class _TimedArray(np.ndarray):
    """ndarray with the two xarray attributes the recipes read: `.t` (time coordinate) and `.to_numpy()`"""

    def __array_finalize__(self, obj):
        self.t = getattr(obj, "t", None)

    def to_numpy(self):
        return np.asarray(self)


def field_with_time(data, minutes=10, start="2020-06-01T00:00"):
    """What the drop-in scripts hand the detection entry points, without xarray: a (t, y, x) numpy array with a `.t` time
    coordinate `minutes` apart -- or, for a device tensor, the package's DeviceField (the device-resident form)."""
    times = np.datetime64(start) + np.arange(data.shape[0]) * np.timedelta64(minutes, "m")
    if not isinstance(data, np.ndarray):
        from tobac_flow_amd.detection import DeviceField
        return DeviceField(data, times)
    out = np.asarray(data).view(_TimedArray)
    out.t = _TimeCoord(times)
    return out
