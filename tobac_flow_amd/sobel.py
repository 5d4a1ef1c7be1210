"""Semi-Lagrangian Sobel edge detection (mirrors /root/reference/tobac_flow/sobel.py).

The three reductions (_sobel_func, _sobel_func_uphill, _sobel_func_downhill, sobel.py:32-86)
are plain numpy callables tagged with their fused-kernel code, so `convolve(..., func=...)`
dispatches them to the GPU; called directly they behave like the reference's functions.
"""
import numpy as np
import scipy.ndimage as ndi

from tobac_flow_amd import _lib
from tobac_flow_amd.convolve import convolve, tag_func


def _sobel_matrix(ndims):
    """Sobel coefficient tensor: [1,2,1] (x) ... (x) [-1,0,1] (reference: sobel.py:7-26)."""
    sobel_matrix = np.array([-1, 0, 1])
    for _ in range(ndims - 1):
        sobel_matrix = np.multiply.outer(np.array([1, 2, 1]), sobel_matrix)
    return sobel_matrix


sobel_matrix = _sobel_matrix(3)
_W = [sobel_matrix.transpose(ax).ravel()[:, np.newaxis, np.newaxis] for ax in ([0, 1, 2], [1, 2, 0], [2, 0, 1])]


def _magnitude(x):
    out = np.nansum(x * _W[0], 0) ** 2
    out += np.nansum(x * _W[1], 0) ** 2
    out += np.nansum(x * _W[2], 0) ** 2
    return out ** 0.5


@tag_func(_lib.FUNC_SOBEL_UPHILL)
def _sobel_func_uphill(x):
    return _magnitude(np.fmax(x - x[13], 0))


@tag_func(_lib.FUNC_SOBEL_DOWNHILL)
def _sobel_func_downhill(x):
    return _magnitude(np.fmin(x - x[13], 0))


@tag_func(_lib.FUNC_SOBEL)
def _sobel_func(x):
    return _magnitude(x - x[13])


def sobel(data, forward_flow, backward_flow, method="linear", dtype=np.float32, fill_value=np.nan,
          direction=None, _dev_flows=None):
    """Sobel edge magnitude in a semi-Lagrangian frame (reference: sobel.py:89-143)."""
    if direction == "uphill":
        func = _sobel_func_uphill
    elif direction == "downhill":
        func = _sobel_func_downhill
    else:
        func = _sobel_func
    return convolve(data, forward_flow, backward_flow, structure=ndi.generate_binary_structure(3, 3),
                    method=method, dtype=dtype, fill_value=fill_value, func=func, _dev_flows=_dev_flows)
