"""ORACLE (test infrastructure, never shipped or measured as the product).

Plain-numpy restatement of the reference's semi-Lagrangian operators, with the three cv2
calls replaced by the C restatement in oracle/c/remap.c:

  to_8bit / linear_norm        /root/reference/tobac_flow/utils/normalisation_utils.py:10-33,59-72
  warp_flow (single image)     /root/reference/tobac_flow/utils/flow_utils.py:80-99
  warp_flow (multi-offset)     /root/reference/tobac_flow/convolve.py:8-86
  convolve_same_step           /root/reference/tobac_flow/convolve.py:89-144
  convolve_step / convolve     /root/reference/tobac_flow/convolve.py:147-348
  sobel funcs                  /root/reference/tobac_flow/sobel.py:7-143
  Flow.diff                    /root/reference/tobac_flow/flow.py:159-191
  smooth_flow_step             /root/reference/tobac_flow/flow.py:530-568

Pinned by the reference's known-answer tests (tests/test_flow.py:53-194,
tests/test_detection.py:36-60), ported in tests/test_oracle_known_answers.py.
cv2-dependent numerics beyond those tests are "parity unpinned" (see oracle/c/remap.c).
"""
import ctypes
import warnings

import numpy as np
import scipy.ndimage as ndi

from . import _lib

_METHODS = {"nearest": 0, "linear": 1, "cubic": 2, "lanczos": 3}


def remap(img, locs, method, fill_value):
    """cv2.remap(img, locs(rows, cols, 2) f32, None, method, None, BORDER_CONSTANT, fill_value)."""
    if method not in ("nearest", "linear", "cubic", "lanczos"):
        raise ValueError("method must be one of ['nearest', 'linear', 'cubic', 'lanczos']")
    L = _lib.lib()
    locs = np.ascontiguousarray(locs, np.float32)
    rows, cols = locs.shape[:2]
    h, w = img.shape
    if np.issubdtype(img.dtype, np.integer):
        if method != "nearest":
            raise ValueError("integer images only support nearest")
        img = np.ascontiguousarray(img, np.int32)
        dst = np.empty((rows, cols), np.int32)
        L.oracle_remap_nearest_i32(_lib.ptr(img, ctypes.c_int32), h, w, _lib.ptr(locs, ctypes.c_float),
                                   ctypes.c_int64(rows), ctypes.c_int64(cols), ctypes.c_int32(int(fill_value)),
                                   _lib.ptr(dst, ctypes.c_int32))
        return dst
    img = np.ascontiguousarray(img, np.float32)
    dst = np.empty((rows, cols), np.float32)
    L.oracle_remap_f32(_lib.ptr(img, ctypes.c_float), h, w, _lib.ptr(locs, ctypes.c_float),
                       ctypes.c_int64(rows), ctypes.c_int64(cols), _METHODS[method],
                       ctypes.c_float(fill_value), _lib.ptr(dst, ctypes.c_float))
    return dst


# ----------------------------------------------------------------------------- normalisation
def to_8bit(array, vmin=None, vmax=None, fill_value=127):
    if vmin is None:
        vmin = np.nanmin(array)
    if vmax is None:
        vmax = np.nanmax(array)
    factor = 0 if vmin == vmax else 255 / (vmax - vmin)
    out = (array - vmin) * factor
    fin = np.isfinite(out)
    out[~fin] = fill_value
    if out.ndim >= 2:   # the reference's cross-frame patch (lines 29-31) raises TypeError on 1-D input
        out[0][~fin[0]] = out[1][~fin[0]]
        out[1][~fin[1]] = out[0][~fin[1]]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return out.astype("uint8")


def linear_norm(array, vmin=None, vmax=None):
    if vmin is None:
        vmin = np.nanmin(array)
    if vmax is None:
        vmax = np.nanmax(array)
    factor = 1 / (vmax - vmin) if vmax > vmin else 0
    out = (array - vmin) * factor
    return np.maximum(np.minimum(out, 1), 0)


# ----------------------------------------------------------------------------- warps
def warp_flow_single(img, flow, method="linear"):
    """utils.flow_utils.warp_flow: border constant NaN."""
    h, w = flow.shape[:2]
    locs = flow.copy()
    locs[:, :, 0] += np.arange(w)
    locs[:, :, 1] += np.arange(h)[:, np.newaxis]
    return remap(img, locs, method, np.nan)


def warp_flow_multi(img, flow, method="linear", fill_value=np.nan, offsets=np.array([[0, 0]]), grid_locs=None,
                    origin=(0, 0)):
    h, w = flow.shape[:2]
    locs = flow[np.newaxis, ...] + np.atleast_2d(offsets)[:, np.newaxis, np.newaxis, :].astype(np.float32)
    if grid_locs is None:
        locs[..., 0] += np.arange(w)
        locs[..., 1] += np.arange(h)[..., np.newaxis]
    else:
        locs += grid_locs
    if origin != (0, 0):
        # test aid: `img` is a crop whose pixel (0, 0) sits at `origin` (x, y) of the full frame and
        # grid_locs holds FULL-FRAME coordinates, so the float32 rounding of the coordinates is that of
        # the full frame; subtracting the integer origin afterwards is exact
        locs[..., 0] -= np.float32(origin[0])
        locs[..., 1] -= np.float32(origin[1])
    res = remap(img, locs.reshape([-1, locs.shape[-2], locs.shape[-1]]), method, fill_value)
    return res.reshape(locs.shape[:-1])


def convolve_same_step(img, offsets, fill_value, grid_locs):
    h, w = img.shape
    locs = grid_locs + np.atleast_2d(offsets)[:, np.newaxis, np.newaxis, :].astype(int)
    oob = np.logical_or.reduce([locs[..., 0] < 0, locs[..., 1] < 0, locs[..., 0] >= w, locs[..., 1] >= h])
    locs[..., 0][oob] = 0
    locs[..., 1][oob] = 0
    res = img[locs[..., 1], locs[..., 0]]
    return res, oob


def convolve_step(prev_step, same_step, next_step, fwd, bwd, structure, method, dtype, fill_value, grid_locs,
                  origin=(0, 0)):
    if len(structure.shape) != 3:
        raise ValueError("structure must have three dimensions")
    if structure.shape[0] != 3:
        raise ValueError("leading dimension of structure must have length 3")
    n_struct = np.count_nonzero(structure)
    res = np.full((n_struct,) + same_step.shape, fill_value, dtype=dtype)
    centre = np.array([structure.shape[1] // 2, structure.shape[2] // 2])
    nb, ns, nf = (np.count_nonzero(structure[k]) for k in range(3))
    if nb:
        offs = np.stack(np.where(structure[0]), -1)[..., ::-1] - centre
        res[:nb] = warp_flow_multi(prev_step, bwd, method, fill_value, offs, grid_locs, origin)
    if ns:
        offs = np.stack(np.where(structure[1]), -1)[..., ::-1] - centre
        vals, oob = convolve_same_step(same_step, offs, fill_value, grid_locs - np.array(origin))
        res[nb:nb + ns] = vals
        res[nb:nb + ns][oob] = fill_value
    if nf:
        offs = np.stack(np.where(structure[2]), -1)[..., ::-1] - centre
        res[nb + ns:] = warp_flow_multi(next_step, fwd, method, fill_value, offs, grid_locs, origin)
    return res


def convolve(data, fwd, bwd, structure=ndi.generate_binary_structure(3, 1), method="linear",
             dtype=np.float32, fill_value=np.nan, func=None, origin=(0, 0)):
    assert structure.shape == (3, 3, 3), "Structure input must be a 3x3x3 array"
    n_struct = np.count_nonzero(structure)
    if func is not None:
        res = np.full(data.shape, fill_value, dtype=dtype)
    else:
        res = np.full((n_struct,) + data.shape, fill_value, dtype=dtype)
    h, w = data.shape[1:]
    grid_locs = np.stack(np.meshgrid(np.arange(w) + origin[0], np.arange(h) + origin[1]), -1)
    T = data.shape[0]
    for i in range(T):
        prev_frame = np.full(data[i].shape, fill_value, dtype=dtype) if i == 0 else data[i - 1]
        next_frame = np.full(data[i].shape, fill_value, dtype=dtype) if i == T - 1 else data[i + 1]
        stack = convolve_step(prev_frame, data[i], next_frame, fwd[i], bwd[i], structure, method, dtype,
                              fill_value, grid_locs, origin)
        if func is not None:
            res[i] = func(stack)
        else:
            res[:, i] = stack
    if func is not None:
        res[np.isnan(data)] = fill_value
    return res


# ----------------------------------------------------------------------------- sobel / diff
def _sobel_matrix(ndims):
    m = np.array([-1, 0, 1])
    for _ in range(ndims - 1):
        m = np.multiply.outer(np.array([1, 2, 1]), m)
    return m


_S = _sobel_matrix(3)
_WX = _S.ravel()[:, None, None]
_WY = _S.transpose([1, 2, 0]).ravel()[:, None, None]
_WT = _S.transpose([2, 0, 1]).ravel()[:, None, None]


def _mag(x):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = np.nansum(x * _WX, 0) ** 2
        out += np.nansum(x * _WY, 0) ** 2
        out += np.nansum(x * _WT, 0) ** 2
        return out ** 0.5


def sobel_func(direction):
    if direction == "uphill":
        return lambda x: _mag(np.fmax(x - x[13], 0))
    if direction == "downhill":
        return lambda x: _mag(np.fmin(x - x[13], 0))
    return lambda x: _mag(x - x[13])


def sobel(data, fwd, bwd, method="linear", dtype=np.float32, fill_value=np.nan, direction=None, origin=(0, 0)):
    return convolve(data, fwd, bwd, ndi.generate_binary_structure(3, 3), method, dtype, fill_value,
                    sobel_func(direction), origin)


def diff(data, fwd, bwd, method="linear", dtype=np.float32):
    struct = np.zeros([3, 3, 3])
    struct[:, 1, 1] = 1

    def f(x):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return (np.nansum([x[2] - x[1], x[1] - x[0]], axis=0) * 1
                    / np.maximum(np.sum([np.isfinite(x[2]), np.isfinite(x[0])], 0), 1))
    return convolve(data, fwd, bwd, struct, method, dtype, np.nan, f)


# ----------------------------------------------------------------------------- flow smoothing
def smooth_flow_step(fwd, bwd, method="linear"):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        f2 = np.nanmean([fwd, np.stack([-warp_flow_single(bwd[..., 0], fwd, method),
                                        -warp_flow_single(bwd[..., 1], fwd, method)], -1)], 0)
        b2 = np.nanmean([bwd, np.stack([-warp_flow_single(fwd[..., 0], bwd, method),
                                        -warp_flow_single(fwd[..., 1], bwd, method)], -1)], 0)
    return f2, b2


def variational_refinement(I0, I1, flow, fixed_point_iterations=5, sor_iterations=5, alpha=20.0, delta=5.0, gamma=10.0,
                           omega=1.6):
    """cv2.VariationalRefinement.create().calc(I0, I1, flow) (flow.py:359, 513-519; oracle/c/varref.c, parity
    unpinned).  I0 / I1 uint8 (H, W); returns the refined copy of flow (H, W, 2) float32."""
    I0 = np.ascontiguousarray(I0, np.uint8)
    I1 = np.ascontiguousarray(I1, np.uint8)
    out = np.array(flow, np.float32, order="C", copy=True)
    H, W = I0.shape
    assert I1.shape == (H, W) and out.shape == (H, W, 2)
    L = _lib.lib()
    L.oracle_variational_refinement.restype = ctypes.c_int
    rc = L.oracle_variational_refinement(_lib.ptr(I0, ctypes.c_uint8), _lib.ptr(I1, ctypes.c_uint8), H, W,
                                         _lib.ptr(out, ctypes.c_float), int(fixed_point_iterations), int(sor_iterations),
                                         ctypes.c_float(alpha), ctypes.c_float(delta), ctypes.c_float(gamma),
                                         ctypes.c_float(omega))
    if rc:
        raise MemoryError("oracle_variational_refinement")
    return out
