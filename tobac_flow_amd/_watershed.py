"""`tobac_flow._watershed.watershed_raveled` on the MI355X -- the reference's only native seam
(/root/reference/tobac_flow/_watershed.pyx:222-233), with the same twelve positional arguments, the same dtypes
(1-D C-contiguous arrays; anything else is a ValueError, as Cython's typed memoryviews make it) and the same
contract: `output` is mutated in place, nothing is returned.  The flood itself is tf_watershed_raveled
(include/tobac_flow_hip.h): `compactness` must be 0 and `wsl` False, the only way watershed.py:151-164 calls it.
"""
import ctypes
import warnings

import numpy as np

from tobac_flow_amd import _lib
from tobac_flow_amd.watershed import (MAX_CHAIN_DEPTH, TF_EDEPTH, TF_WS_AMBIGUOUS, WatershedAmbiguityWarning,
                                      WatershedDepthError)


def _flat(a, dtype, name):
    a = np.asarray(a)
    if a.ndim != 1 or a.dtype != np.dtype(dtype) or not a.flags.c_contiguous:
        raise ValueError(f"Buffer dtype / shape mismatch for `{name}`: expected a 1-D C-contiguous {np.dtype(dtype).name} array")
    return a


def watershed_raveled(image, marker_locations, structure, forward_offset, backward_offset, forward_offset_locations,
                      backward_offset_locations, mask, strides, compactness, output, wsl, reference_order=True):
    """Perform the watershed on a raveled image and neighbourhood (reference: _watershed.pyx:222-344).
    `reference_order` (not in the reference; default True since round 4): equal-valued markers pop in the order the
    reference's heap gives them (TF_WS_REFERENCE_ORDER, include/tobac_flow_hip.h) -- the result of the reference's own
    call; False: marker_locations order + a warning when a label depends on it."""
    t = _lib.torch()
    L = _lib.lib()
    image = _flat(image, np.float32, "image")
    marker_locations = _flat(marker_locations, np.intp, "marker_locations")
    structure = _flat(structure, np.intp, "structure")
    forward_offset = _flat(forward_offset, np.int32, "forward_offset")
    backward_offset = _flat(backward_offset, np.int32, "backward_offset")
    floc = _flat(forward_offset_locations, np.int32, "forward_offset_locations")
    bloc = _flat(backward_offset_locations, np.int32, "backward_offset_locations")
    mask = _flat(mask, np.int8, "mask")
    strides = _flat(strides, np.int32, "strides")
    out = _flat(output, np.int32, "output")
    n = image.size
    if not (forward_offset.size == backward_offset.size == mask.size == out.size == n):
        raise ValueError("image, forward_offset, backward_offset, mask and output must have the same length")
    if not (structure.size == floc.size == bloc.size):
        raise ValueError("structure and the two offset-location arrays must have the same length")
    if n == 0:
        return None
    d_img, d_loc = _lib.to_dev(image), _lib.to_dev(marker_locations.astype(np.int64))
    d_fo, d_bo, d_mask, d_out = _lib.to_dev(forward_offset), _lib.to_dev(backward_offset), _lib.to_dev(mask), _lib.to_dev(out)
    st64 = np.ascontiguousarray(structure, np.int64)
    stats = np.zeros(16, np.int64)
    guess = min(n, int(((d_out == 0) & (d_mask != 0)).sum().item() * 1.5) + 4096)
    for _ in range(2):
        ws = _lib.workspace(L.tf_watershed_raveled_workspace_bytes(n, structure.size, MAX_CHAIN_DEPTH, guess), "watershed")
        rc = L.tf_watershed_raveled_ex(_lib.ptr(d_img), n, _lib.ptr(d_loc), marker_locations.size, st64.ctypes.data_as(_lib._P),
                                    structure.size, _lib.ptr(d_fo), _lib.ptr(d_bo), floc.ctypes.data_as(_lib._P),
                                    bloc.ctypes.data_as(_lib._P), _lib.ptr(d_mask), strides.ctypes.data_as(_lib._P), strides.size,
                                    float(compactness), _lib.ptr(d_out), int(bool(wsl)), MAX_CHAIN_DEPTH, 2 if reference_order else 0,
                                    _lib.ptr(ws), ws.numel(),
                                    stats.ctypes.data_as(_lib._P), _lib.stream_ptr())
        if rc == -2 and stats[6] > guess:
            guess = int(stats[6])
            continue
        break
    if rc == TF_EDEPTH:
        raise WatershedDepthError(f"watershed_raveled: {int(stats[11])} pixel(s) still tie at chain depth {int(stats[8])}")
    if rc == TF_WS_AMBIGUOUS:
        warnings.warn(f"watershed_raveled: the labels of {int(stats[9])} pixel(s) depend on the order in which the reference's "
                      "binary heap pops equal-valued markers; resolved by marker_locations order", WatershedAmbiguityWarning,
                      stacklevel=2)
    elif rc:
        _lib.check(rc, "tf_watershed_raveled")
    output[...] = d_out.cpu().numpy()
    return None


__all__ = ("watershed_raveled",)
