"""The reference's own known-answer tests, run against the oracle's restatement of the cv2-backed
operators (ports of /root/reference/tests/test_flow.py:53-194 and tests/test_detection.py:36-60).
These are the only golden values the reference holds at the cv2 boundary."""
import numpy as np
import pytest

from oracle import np_ops


def test_to_8bit_known_answers():
    assert np.all(np_ops.to_8bit(np.zeros(5)) == 0)
    assert np.all(np_ops.to_8bit(np.ones(5)) == 0)
    assert np.all(np_ops.to_8bit(np.ones(5), vmin=0, vmax=1) == 255)
    arr = np.arange(256)
    assert np.all(np_ops.to_8bit(arr) == arr)
    assert np.all(np_ops.to_8bit(arr + 10, vmin=10, vmax=10 + 255) == arr)


def _valid(w, ref):
    ok = ~np.isnan(w)
    return w[ok], ref[ok]


def test_warp_flow_known_answers():
    arr = np.arange(15, dtype=np.float32).reshape(3, 5)
    fl = np.zeros(arr.shape + (2,), np.float32)
    a, b = _valid(np_ops.warp_flow_single(arr, fl), arr)
    assert np.all(a == b)
    fl[..., 0] = 1
    a, b = _valid(np_ops.warp_flow_single(arr, fl)[:, :-1], arr[:, 1:])
    assert np.all(a == b)
    fl[:] = 0
    fl[..., 1] = 1
    a, b = _valid(np_ops.warp_flow_single(arr, fl)[:-1], arr[1:])
    assert np.all(a == b)
    fl[:] = 1
    a, b = _valid(np_ops.warp_flow_single(arr, fl)[:-1, :-1], arr[1:, 1:])
    assert np.all(a == b)
    fl[:] = 0
    fl[..., 0] = 0.5
    a, b = _valid(np_ops.warp_flow_single(arr, fl)[:, :-1], (arr[:, 1:] + arr[:, :-1]) * 0.5)
    assert np.all(a == b)


def test_smooth_flow_step_known_answers():
    z, one = np.zeros([3, 5, 2], np.float32), np.ones([3, 5, 2], np.float32)
    assert np.all(np.stack(list(np_ops.smooth_flow_step(z, z))) == 0)
    f, b = np_ops.smooth_flow_step(one, -one)
    assert np.all(f == 1) and np.all(b == -1)
    f, b = np_ops.smooth_flow_step(one, z)
    assert np.all(f[:1, :3] == 0.5) and np.all(b[:2, :4] == -0.5)


def test_combined_edge_field_known_answer():
    field = np.zeros([1, 5, 5], np.float32)
    field[:, 3:] = 1
    z = np.zeros([1, 5, 5, 2], np.float32)

    def edge_field(f):
        e = np_ops.sobel(f, z, z, method="cubic", dtype=None, direction="uphill")
        e[e > 0] += 1
        e = e - f
        e[np.isnan(f)] = np.inf
        return e
    r = edge_field(field)
    assert r.dtype == np.float64
    assert np.all(r[:, 2] > 0) and np.all(r[:, :2] == 0) and np.all(r[:, 3:] == -1)
    field[:, :, 0] = np.nan
    assert np.all(np.isnan(field) == np.isinf(edge_field(field)))


def test_farneback_oracle_recovers_translation():
    import ctypes
    import scipy.ndimage as ndi
    from oracle import _lib as ol
    rng = np.random.default_rng(0)
    img = ndi.gaussian_filter(rng.normal(size=(160, 200)), 4)
    img = ((img - img.min()) / (img.max() - img.min()) * 255).astype(np.uint8)
    nxt = np.roll(img, (1, -2), (0, 1))
    out = np.zeros(img.shape + (2,), np.float32)
    L = ol.lib()
    L.oracle_farneback.restype = ctypes.c_int
    n = L.oracle_farneback(ol.ptr(img, ctypes.c_uint8), ol.ptr(np.ascontiguousarray(nxt), ctypes.c_uint8), 160, 200,
                           ol.ptr(out, ctypes.c_float), 5, ctypes.c_double(0.5), 13, 10, 5, ctypes.c_double(1.1))
    assert n == 3          # 160 x 200 -> 40 x 50, 80 x 100, full
    c = out[40:120, 50:150].mean((0, 1))
    assert abs(c[0] + 2) < 0.05 and abs(c[1] - 1) < 0.05


# ----------------------------------------------------------------------------- Farnebaeck: identities of the method
# cv2 is not available, so the restated Farnebaeck stages (oracle/c/farneback.c) cannot be compared with OpenCV's
# output.  What CAN be checked without cv2 are the identities the method is derived from (Farnebaeck 2003): the
# polynomial expansion of an exactly quadratic image returns its coefficients, and for two quadratic images that differ
# by a translation d the very first iteration returns d.  They tie the tap tables, the inverse-metric constants, the
# channel order, the 1/2 factors of UpdateMatrices, the sign convention of the flow and the determinant regulariser
# to the mathematics rather than to a recollection of optflowgf.cpp.
def _fb_stage_lib():
    import ctypes
    from oracle import _lib as ol
    return ol, ol.lib(), ctypes


def _quadratic(h, w, c, p, q, r, s, u, dx=0.0, dy=0.0):
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    xx, yy = xx - dx, yy - dy
    return c + p * xx + q * yy + r * xx * xx + s * yy * yy + u * xx * yy


def _poly_exp(img, n=5, sigma=1.1):
    ol, L, ctypes = _fb_stage_lib()
    h, w = img.shape
    out = np.zeros((h, w, 5), np.float32)
    L.oracle_poly_exp(ol.ptr(np.ascontiguousarray(img, np.float32), ctypes.c_float), h, w, ol.ptr(out, ctypes.c_float), n,
                      ctypes.c_double(sigma))
    return out


@pytest.mark.parametrize("scale", [0.004, 0.05, 0.5, 3.0])
def test_farneback_polynomial_expansion_is_exact_on_quadratics(scale):
    h, w, n = 64, 80, 5
    c, p, q, r, s, u = 100.0, 0.8, -0.5, 1.0 * scale, -0.75 * scale, 0.5 * scale
    img = _quadratic(h, w, c, p, q, r, s, u)
    got = _poly_exp(img, n).astype(np.float64)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    # local expansion around (x0, y0); channel order [y, x, yy, xx, xy] as FarnebackUpdateMatrices reads it
    want = np.stack([q + 2 * s * yy + u * xx, p + 2 * r * xx + u * yy, np.full_like(xx, s), np.full_like(xx, r),
                     np.full_like(xx, u)], -1)
    inner = (slice(n, h - n), slice(n, w - n))
    # the vertical pass accumulates in float32: the error floor is the float32 spacing at the image's largest value
    # (measured: 0.4 - 0.7 of it over a 750x range of curvature); 4x leaves margin for other libm / compilers
    tol = 4 * np.spacing(np.float32(np.abs(img).max()))
    assert np.abs(got - want)[inner].max() <= tol
    # the replicate border breaks the polynomial: the check above is not satisfied trivially
    assert np.abs(got - want)[0].max() > 100 * tol


@pytest.mark.parametrize("scale,d", [(3.0, (0.7, -0.4)), (3.0, (-1.3, 0.9)), (0.5, (0.7, -0.4)), (0.5, (-1.3, 0.9)), (3.0, (0.0, 0.0))])
def test_farneback_first_iteration_returns_the_translation_of_a_quadratic(scale, d):
    ol, L, ctypes = _fb_stage_lib()
    F = ctypes.c_float
    h, w, win = 64, 80, 13
    r, s, u = 1.0 * scale, 0.75 * scale, 0.3 * scale
    r0 = _poly_exp(_quadratic(h, w, 50.0, 0.8, -0.5, r, s, u))
    r1 = _poly_exp(_quadratic(h, w, 50.0, 0.8, -0.5, r, s, u, dx=d[0], dy=d[1]))       # next(x) = prev(x - d)
    flow = np.zeros((h, w, 2), np.float32)
    m = np.zeros((h, w, 5), np.float32)
    L.oracle_update_matrices(ol.ptr(r0, F), ol.ptr(r1, F), ol.ptr(flow, F), ol.ptr(m, F), h, w)
    L.oracle_update_flow_blur(ol.ptr(r0, F), ol.ptr(r1, F), ol.ptr(flow, F), ol.ptr(m, F), h, w, win, 0)
    inner = (slice(win + 1, h - win - 1), slice(win + 1, w - win - 1))
    err = np.abs(flow[inner] - np.array(d, np.float32)).max()
    if d == (0.0, 0.0):
        assert err == 0.0
        return
    # flow = (G h) / (det G + 1e-3) with G = A^T A, A = [[s, u/2], [u/2, r]]: relative bias 1e-3 / det(A)^2
    bias = 1e-3 / (r * s - (u / 2) ** 2) ** 2 * max(abs(d[0]), abs(d[1]))
    assert err <= 2 * bias + 1e-5, (err, bias)          # measured: err / bias = 0.98 - 1.3
    if bias > 1e-3:
        assert err >= 0.5 * bias, (err, bias)           # ... and the regulariser really is 1e-3 on the window MEANS


# ----------------------------------------------------------------------------- remap: properties of the restated sampler
# Regression guards on oracle/c/remap.c.  They follow from the restated definitions (1/32-pixel coordinate table,
# bilinear weights, cubic kernel with A = -0.75); they do not add evidence about OpenCV itself.
def _grid_locs(h, w, dx, dy):
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    return np.stack([xx + np.float32(dx), yy + np.float32(dy)], -1)


@pytest.mark.parametrize("method", ["nearest", "linear", "cubic"])
def test_remap_reproduces_constants_and_integer_shifts(method):
    rng = np.random.default_rng(3)
    h, w = 20, 27
    const = np.full((h, w), 3.25, np.float32)
    out = np_ops.remap(const, _grid_locs(h, w, 0.37, -0.81), method, np.nan)
    inner = out[3:-3, 3:-3]
    assert np.all(np.abs(inner - 3.25) <= 2e-6)              # weights sum to one (float32 rounding of 16 products)
    img = rng.normal(size=(h, w)).astype(np.float32)
    out = np_ops.remap(img, _grid_locs(h, w, 2, -1), method, np.nan)
    assert np.array_equal(out[4:-4, 4:-4], img[3:-5, 6:-2])  # integer coordinates return the pixel itself
    assert np.isnan(out[0, -1])                              # outside the image: the NaN border value
    # x = w - 4 samples source column w - 2: inside for nearest / linear, but the 4-wide cubic patch reaches column w,
    # and a NaN border tap poisons the sum even at weight zero (the NaN rims of the reference's warped fields)
    assert np.isnan(out[8, w - 4]) == (method == "cubic")


def test_remap_linear_is_affine_exact_on_the_32nd_pixel_grid():
    """cv2.remap interpolates at coordinates rounded to 1/32 pixel: a ramp sampled at x + 0.37 returns
    x + round(0.37 * 32) / 32 = x + 0.375, not x + 0.37."""
    h, w = 16, 40
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    ramp = (2 * xx - 3 * yy + 5).astype(np.float32)
    out = np_ops.remap(ramp, _grid_locs(h, w, 0.37, 0.2), "linear", np.nan)
    qx, qy = round(0.37 * 32) / 32, round(0.2 * 32) / 32
    want = 2 * (xx + qx) - 3 * (yy + qy) + 5
    assert np.abs(out - want)[2:-2, 2:-2].max() <= 1e-5
    exact = 2 * (xx + 0.37) - 3 * (yy + 0.2) + 5
    assert np.abs(out - exact)[2:-2, 2:-2].min() > 5e-3      # ... and visibly not the unquantised value


def test_remap_cubic_kernel_has_unit_sum_but_no_linear_precision():
    """A = -0.75 (OpenCV's constant) gives a partition of unity but, unlike Keys' A = -0.5, does not reproduce
    ramps: at a fraction of 1/4 the first moment is 0.296875.  A sampler that returned x + 0.25 here would be a
    different kernel."""
    h, w = 12, 40
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    out = np_ops.remap(xx.copy(), _grid_locs(h, w, 0.25, 0.0), "cubic", np.nan)
    assert np.abs(out - (xx + 0.296875))[3:-3, 3:-3].max() <= 2e-5
    half = np_ops.remap(xx.copy(), _grid_locs(h, w, 0.5, 0.0), "cubic", np.nan)
    assert np.abs(half - (xx + 0.5))[3:-3, 3:-3].max() <= 2e-5   # symmetric at one half


# ---- variational refinement (oracle/c/varref.c; OpenCV restated, parity unpinned): the method's own identities ----
def _texture(rng, shape, smooth=2.5):
    import scipy.ndimage as ndi
    a = ndi.gaussian_filter(rng.normal(size=shape), smooth)
    return (a - a.min()) / np.ptp(a) * 255


def test_variational_refinement_leaves_a_perfect_flow_of_identical_frames_alone():
    """I0 == I1 and zero flow: Iz = 0, so b = 0 and dW = 0 is the exact solution of every SOR step."""
    from oracle import np_ops
    img = _texture(np.random.default_rng(1), (37, 53)).astype(np.uint8)
    out = np_ops.variational_refinement(img, img, np.zeros((37, 53, 2), np.float32))
    assert np.array_equal(out, np.zeros_like(out))


def test_variational_refinement_on_textureless_frames_only_smooths():
    """Constant images: every derivative vanishes, the data term is zeta^2 * I with b = 0, and what is left is the
    regulariser: the refined flow has less total variation, its range does not grow."""
    from oracle import np_ops
    rng = np.random.default_rng(2)
    img = np.full((40, 48), 90, np.uint8)
    flow = rng.normal(size=(40, 48, 2)).astype(np.float32)
    out = np_ops.variational_refinement(img, img, flow)
    tv = lambda f: np.abs(np.diff(f, axis=0)).sum() + np.abs(np.diff(f, axis=1)).sum()
    assert tv(out) < 0.5 * tv(flow)
    assert out.max() <= flow.max() + 1e-5 and out.min() >= flow.min() - 1e-5


@pytest.mark.parametrize("d", [(0.6, -0.3), (-1.25, 0.5)])
def test_variational_refinement_pulls_a_wrong_flow_towards_the_true_translation(d):
    """I1(x) = I0(x - d): starting from zero flow (an error of |d|), the default 5 x 5 iterations bring the interior flow
    closer to d, a second application closer still (each call linearises around its input); starting AT d it stays
    there (residual data term only from uint8 rounding)."""
    import scipy.ndimage as ndi
    from oracle import np_ops
    rng = np.random.default_rng(3)
    base = _texture(rng, (96, 128), 3.0)
    i0 = base.astype(np.uint8)
    i1 = ndi.shift(base, (d[1], d[0]), order=3, mode="nearest").astype(np.uint8)
    inner = (slice(12, -12), slice(12, -12))
    truth = np.array(d, np.float32)
    start = np.zeros((96, 128, 2), np.float32)
    out = np_ops.variational_refinement(i0, i1, start)
    err0 = np.linalg.norm(start[inner] - truth, axis=-1).mean()
    err1 = np.linalg.norm(out[inner] - truth, axis=-1).mean()
    again = np_ops.variational_refinement(i0, i1, out)
    err2 = np.linalg.norm(again[inner] - truth, axis=-1).mean()
    assert err1 < 0.8 * err0 and err2 < 0.8 * err1
    at = np.broadcast_to(truth, (96, 128, 2)).copy()
    stay = np_ops.variational_refinement(i0, i1, at)
    assert np.linalg.norm(stay[inner] - truth, axis=-1).mean() < 0.1


def test_variational_refinement_sor_solves_the_system_it_assembles():
    """Internal consistency of the restatement: with ONE fixed-point iteration and many SOR sweeps dW converges to the
    solution of the linear system (A - smoothness Laplacian) dW = b that the same code assembled -- recomputed here in
    numpy from the published formulas (weights from the input flow, replicated borders, edges absent at the image
    border)."""
    from oracle import np_ops
    rng = np.random.default_rng(4)
    H, W = 24, 31
    i0 = _texture(rng, (H, W)).astype(np.uint8)
    i1 = np.roll(i0, (1, -1), (0, 1))
    flow = (rng.normal(size=(H, W, 2)) * 0.3).astype(np.float32)
    out = np_ops.variational_refinement(i0, i1, flow, fixed_point_iterations=1, sor_iterations=400)
    dW = (out - flow).astype(np.float64)
    # --- the system in float64
    yy, xx = np.mgrid[0:H, 0:W]
    mx, my = xx + flow[..., 0].astype(np.float64), yy + flow[..., 1].astype(np.float64)
    fx, fy = np.rint(mx.astype(np.float32) * 32).astype(int), np.rint(my.astype(np.float32) * 32).astype(int)
    sx, sy, ax, ay = fx >> 5, fy >> 5, (fx & 31) / 32.0, (fy & 31) / 32.0
    g = lambda y, x: i1[np.clip(y, 0, H - 1), np.clip(x, 0, W - 1)].astype(np.float64)
    warped = (g(sy, sx) * (1 - ay) * (1 - ax) + g(sy, sx + 1) * (1 - ay) * ax + g(sy + 1, sx) * ay * (1 - ax)
              + g(sy + 1, sx + 1) * ay * ax)
    avg, Iz = (i0 + warped) / 2, warped - i0
    dx = lambda a: a[:, np.clip(np.arange(W) + 1, 0, W - 1)] - a[:, np.clip(np.arange(W) - 1, 0, W - 1)]
    dy = lambda a: a[np.clip(np.arange(H) + 1, 0, H - 1)] - a[np.clip(np.arange(H) - 1, 0, H - 1)]
    Ix, Iy, Ixz, Iyz = dx(avg), dy(avg), dx(Iz), dy(Iz)
    Ixx, Ixy, Iyy = dx(Ix), dy(Ix), dy(Iy)
    z2, e2 = 0.01, 1e-6
    n0 = Ix ** 2 + Iy ** 2 + z2
    w = 2.5 / np.sqrt(Iz ** 2 / n0 + e2)                       # dW = 0 at the start of the only fixed-point iteration
    A11, A12, A22 = w * Ix * Ix / n0 + z2, w * Ix * Iy / n0, w * Iy * Iy / n0 + z2
    b1, b2 = -w * Iz * Ix / n0, -w * Iz * Iy / n0
    n1, n2 = Ixx ** 2 + Ixy ** 2 + z2, Iyy ** 2 + Ixy ** 2 + z2
    w = 5.0 / np.sqrt(Ixz ** 2 / n1 + Iyz ** 2 / n2 + e2)
    A11 += w * (Ixx ** 2 / n1 + Ixy ** 2 / n2)
    A12 += w * (Ixx * Ixy / n1 + Ixy * Iyy / n2)
    A22 += w * (Ixy ** 2 / n1 + Iyy ** 2 / n2)
    b1 -= w * (Ixx * Ixz / n1 + Ixy * Iyz / n2)
    b2 -= w * (Ixy * Ixz / n1 + Iyy * Iyz / n2)
    u, v = flow[..., 0].astype(np.float64), flow[..., 1].astype(np.float64)
    fwd_x = lambda a: np.concatenate([a[:, 1:] - a[:, :-1], np.zeros((H, 1))], 1)
    fwd_y = lambda a: np.concatenate([a[1:] - a[:-1], np.zeros((1, W))], 0)
    wt = 5.0 / np.sqrt(fwd_x(u) ** 2 + fwd_x(v) ** 2 + fwd_y(u) ** 2 + fwd_y(v) ** 2 + e2)
    ex = wt.copy(); ex[:, -1] = 0                               # edge (p, right) exists except in the last column
    ey = wt.copy(); ey[-1] = 0                                  # edge (p, down) exists except in the last row
    exl = np.concatenate([np.zeros((H, 1)), ex[:, :-1]], 1)     # weight of the edge to the left / upper neighbour
    eyu = np.concatenate([np.zeros((1, W)), ey[:-1]], 0)
    diag = ex + exl + ey + eyu

    def lap(a):                                                 # sum over edges of weight * (neighbour - self)
        r = np.zeros_like(a)
        r[:, :-1] += ex[:, :-1] * (a[:, 1:] - a[:, :-1])
        r[:, 1:] += ex[:, :-1] * (a[:, :-1] - a[:, 1:])
        r[:-1] += ey[:-1] * (a[1:] - a[:-1])
        r[1:] += ey[:-1] * (a[:-1] - a[1:])
        return r
    # (A + diag) du + A12 dv - sum_w dW_nbr = b + lap(W)   <=>   A du + A12 dv - lap(du) = b + lap(u)
    r1 = A11 * dW[..., 0] + A12 * dW[..., 1] - lap(dW[..., 0]) - (b1 + lap(u))
    r2 = A22 * dW[..., 1] + A12 * dW[..., 0] - lap(dW[..., 1]) - (b2 + lap(v))
    scale = np.abs(b1 + lap(u)).max() + np.abs(b2 + lap(v)).max()
    assert diag.min() > 0 and max(np.abs(r1).max(), np.abs(r2).max()) < 2e-3 * scale


def test_remap_lanczos4_identities():
    """cv2.INTER_LANCZOS4 as restated in oracle/c/remap.c (parity unpinned): integer shifts are exact (the table entry
    for a zero fraction is a delta), constants are reproduced to float rounding (weights normalised to unit sum), the
    half-pixel kernel is symmetric, and a smooth image is interpolated several times better than by the bilinear kernel."""
    from oracle import np_ops
    rng = np.random.default_rng(9)
    img = rng.normal(size=(40, 52)).astype(np.float32)
    yy, xx = np.mgrid[0:40, 0:52].astype(np.float32)
    inner = (slice(8, -8), slice(8, -8))
    for dy, dx in ((0, 0), (3, -2), (-5, 4)):
        out = np_ops.remap(img, np.stack([xx + dx, yy + dy], -1), "lanczos", np.nan)
        assert np.array_equal(out[inner], np.roll(img, (-dy, -dx), (0, 1))[inner])
    const = np.full((40, 52), 7.25, np.float32)
    out = np_ops.remap(const, np.stack([xx + 0.37, yy - 0.81], -1), "lanczos", np.nan)
    assert np.abs(out[inner] - 7.25).max() < 1e-5
    half = np_ops.remap(img, np.stack([xx + 0.5, yy], -1), "lanczos", np.nan)
    mirror = np_ops.remap(img[:, ::-1].copy(), np.stack([xx + 0.5, yy], -1), "lanczos", np.nan)[:, ::-1]
    assert np.allclose(half[inner][:, :-1], mirror[inner][:, 1:], atol=2e-6)         # symmetric half-pixel kernel
    smooth = np.sin(xx / 5.0) * np.cos(yy / 7.0)
    truth = np.sin((xx + 0.4375) / 5.0) * np.cos((yy + 0.28125) / 7.0)            # shifts on the 1/32 grid
    locs = np.stack([xx + 0.4375, yy + 0.28125], -1)
    e_lan = np.abs(np_ops.remap(smooth, locs, "lanczos", np.nan) - truth)[inner].max()
    e_lin = np.abs(np_ops.remap(smooth, locs, "linear", np.nan) - truth)[inner].max()
    assert e_lan < 0.5 * e_lin
    # border: a patch that straddles the image takes cval + sum (S - cval) w; NaN poisons it, far outside is cval
    out = np_ops.remap(img, np.stack([xx + 0.5, yy], -1), "lanczos", np.nan)
    assert np.isnan(out[:, -4:]).all() and np.isfinite(out[8:-8, 8:-8]).all()
    assert np.isnan(np_ops.remap(img, np.stack([xx + 100, yy], -1), "lanczos", np.nan)).all()


def test_label_contract_oracle_known_answers():
    """oracle/np_dataset.py on a case small enough to work out by hand (the reference holds no fixture for dataset.py)."""
    from oracle import np_dataset as D
    core = np.zeros((2, 4, 4), np.int32); core[0, 1, 1] = 1; core[1, 1, 1] = 1; core[1, 3, 3] = 2
    thick = np.zeros_like(core); thick[:, 0:3, 0:3] = 5
    thin = thick.copy(); thin[1, 3, 2:4] = 6
    ds = {"core_label": core, "thick_anvil_label": thick, "thin_anvil_label": thin, "coords": {}}
    D.add_label_coords(ds)
    D.link_cores_and_anvils(ds, atol=1)
    # core 1 lies in anvil 5 (2 px), core 2 over background only -> no anvil
    assert ds["core_anvil_index"].tolist() == [5, 0] and ds["anvil_core_count"].tolist() == [1, 0]
    D.add_step_labels(ds)
    D.add_label_coords(ds)
    # step ids ascend by (step, label): core 1 @ t0 -> 1, core 1 @ t1 -> 2, core 2 @ t1 -> 3
    assert ds["core_step_label"][0, 1, 1] == 1 and ds["core_step_label"][1, 1, 1] == 2 and ds["core_step_label"][1, 3, 3] == 3
    assert ds["coords"]["core_step"].tolist() == [1, 2, 3] and ds["coords"]["anvil"].tolist() == [5, 6]
    D.link_step_labels(ds)
    assert ds["core_step_core_index"].tolist() == [1, 1, 2]
    assert ds["thick_anvil_step_anvil_index"].tolist() == [5, 5] and ds["thin_anvil_step_anvil_index"].tolist() == [5, 5, 6]
    D.flag_edge_labels(ds)
    assert ds["core_edge_label_flag"].tolist() == [False, True]          # core 2 sits in the corner
    assert ds["core_start_label_flag"].tolist() == [True, False] and ds["core_end_label_flag"].tolist() == [True, True]
    assert ds["thin_anvil_edge_label_flag"].tolist() == [True, True]
    # atol above the overlap: no link
    ds2 = {"core_label": core, "thick_anvil_label": thick.copy(), "thin_anvil_label": thin.copy(), "coords": {}}
    D.add_label_coords(ds2); D.link_cores_and_anvils(ds2, atol=3)
    assert ds2["core_anvil_index"].tolist() == [0, 0]
    # slice_labels: a step without labels leaves no gap in the numbering
    lab = np.zeros((3, 2, 2), np.int32); lab[0, 0, 0] = 7; lab[2, 1, 1] = 2; lab[2, 0, 1] = 7
    assert D.slice_labels(lab)[[0, 2, 2], [0, 1, 0], [0, 1, 1]].tolist() == [1, 2, 3]
