"""Semi-Lagrangian convolution on the MI355X.

Mirrors /root/reference/tobac_flow/convolve.py (same names, arguments, defaults, exceptions);
the three cv2.remap-based stages (warp_flow :8-86, convolve_same_step :89-144, convolve_step
:147-245) run as ONE fused HIP gather (tf_convolve, include/tobac_flow_hip.h) and the callables the
reference itself passes as `func` are reduced inside that kernel.  Any other Python callable
still works: the (n_struct, H, W) stack is gathered on the GPU and `func` is applied on the host
per frame, exactly like convolve.py:305-347.
"""
import functools
from typing import Callable

import numpy as np
import scipy.ndimage as ndi

from tobac_flow_amd import _lib

_METHODS = ("nearest", "linear", "cubic", "lanczos")


def tag_func(code):
    """Mark a numpy callable as having a fused GPU implementation (tf_convolve `func` code)."""
    def deco(f):
        f._tf_func = code
        return f
    return deco


def _func_code(func):
    if func is None:
        return _lib.FUNC_STACK
    code = getattr(func, "_tf_func", None)
    if code is not None:
        return code
    if isinstance(func, functools.partial) and func.func is np.any and not func.args \
            and func.keywords == {"axis": 0}:
        return _lib.FUNC_ANY       # detection.py:313-320
    return None


def _check_method(method, structure):
    if method not in _METHODS:
        raise ValueError(f"method must be one of {list(_METHODS)}")


def _np_dtype(dtype):
    return np.dtype(np.float64 if dtype is None else dtype)


def convolve_dev(data, fwd, bwd, structure, method, dtype, fill_value, func_code, t0=0, t1=None, out=None):
    """Device-resident core: `data`, `fwd`, `bwd` are torch tensors on the GPU.

    Returns a torch tensor: (n_struct, T, H, W) for FUNC_STACK else (T, H, W) (frames outside
    [t0, t1) are left untouched / unspecified).
    """
    t = _lib.torch()
    L = _lib.lib()
    T, H, W = data.shape
    if t1 is None:
        t1 = T
    nd = _np_dtype(dtype)
    if data.dtype in (t.int32, t.int64, t.int16, t.int8, t.uint8, t.bool):
        data = data.to(t.int32)            # cv2 casts int64 images to CV_32S (remap: nearest only)
        dt_code = _lib.TF_I32
        if method != "nearest":
            raise ValueError("integer data can only be warped with method='nearest'")
    else:
        data = data.to(t.float32)
        dt_code = _lib.TF_F32
    if nd == np.float32:
        out_code, tdt = _lib.TF_F32, t.float32
    elif nd == np.float64:
        out_code, tdt = _lib.TF_F64, t.float64
    elif nd == np.int32:
        out_code, tdt = _lib.TF_I32, t.int32
    else:
        raise ValueError(f"dtype {nd} is not supported on the GPU path (float32, float64, int32)")
    struct = np.ascontiguousarray(np.asarray(structure) != 0, dtype=np.uint8)
    n_struct = int(struct.sum())
    shape = (n_struct, T, H, W) if func_code == _lib.FUNC_STACK else (T, H, W)
    if out is None:
        out = _lib.empty(shape, tdt)
    fill = float(fill_value)
    rc = L.tf_convolve(_lib.ptr(data), dt_code, T, H, W, _lib.ptr(fwd), _lib.ptr(bwd),
                       struct.ctypes.data_as(_lib._P), _lib.INTERP[method], fill, func_code,
                       _lib.ptr(out), out_code, t0, t1, _lib.stream_ptr())
    _lib.check(rc, "tf_convolve")
    return out


def convolve(
    data: np.ndarray,
    forward_flow: np.ndarray,
    backward_flow: np.ndarray,
    structure: np.ndarray = ndi.generate_binary_structure(3, 1),
    method: str = "linear",
    dtype: type = np.float32,
    fill_value: float = np.nan,
    func: Callable | None = None,
    _dev_flows=None,
) -> np.ndarray:
    """Convolve a sequence of images using optical flow vectors to offset adjacent elements in
    the leading dimension (reference: convolve.py:248-348)."""
    assert structure.shape == (3, 3, 3), "Structure input must be a 3x3x3 array"
    t = _lib.torch()
    if hasattr(data, "to_numpy") and not isinstance(data, np.ndarray) and not isinstance(data, t.Tensor):
        data = data.to_numpy()                                  # xr.DataArray
    _check_method(method, structure)
    on_device = isinstance(data, t.Tensor)
    d = _lib.to_dev(data)
    if d.dtype == t.float64:
        d = d.to(t.float32)        # documented deviation: cv2 would remap a float64 image in double
    if _dev_flows is not None:
        fwd, bwd = _dev_flows
    else:
        fwd, bwd = _lib.to_dev(forward_flow, t.float32), _lib.to_dev(backward_flow, t.float32)
    nd = _np_dtype(dtype)
    code = _func_code(func)
    T, H, W = d.shape
    if code is not None:
        out = convolve_dev(d, fwd, bwd, structure, method, nd, fill_value, code)
        return out if on_device else _lib.to_host(out)
    # arbitrary Python callable: gather the stack on the GPU frame by frame, reduce on the host
    data_np = d.cpu().numpy() if on_device else np.asarray(data)
    res = np.full(data_np.shape, fill_value, dtype=nd)
    for i in range(T):
        # a 3-frame window keeps the true sequence ends: frame -1 / T are all-fill (convolve.py:307-314)
        lo, hi = max(i - 1, 0), min(i + 2, T)
        st = convolve_dev(d[lo:hi], fwd[lo:hi], bwd[lo:hi], structure, method, nd, fill_value,
                          _lib.FUNC_STACK, t0=i - lo, t1=i - lo + 1)
        res[i] = func(st[:, i - lo].cpu().numpy())
    if np.issubdtype(data_np.dtype, np.floating):
        res[np.isnan(data_np)] = fill_value
    return _lib.to_dev(res) if on_device else res


__all__ = ("convolve", "convolve_dev", "tag_func")
