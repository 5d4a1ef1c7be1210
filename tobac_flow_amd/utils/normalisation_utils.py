"""Normalisation helpers (host glue; mirrors /root/reference/tobac_flow/utils/normalisation_utils.py).

`to_8bit` and `linear_norm` keep the reference's numpy semantics for arbitrary arrays; inside
`calculate_flow` the frame-pair composition to_8bit(linear_norm(pair), 0, 1) runs on the GPU
(`to_8bit_pair_dev`, tf_to8bit_pair).
"""
from typing import Callable

import numpy as np
import scipy.ndimage as ndi

from tobac_flow_amd import _lib


def to_8bit(array, vmin=None, vmax=None, fill_value=127):
    """Scale to 0..255 and truncate to uint8 (reference: normalisation_utils.py:10-33).
    Non-finite values become `fill_value`, then are patched from the other frame of a pair."""
    array = np.asarray(array)
    if vmin is None:
        vmin = np.nanmin(array)
    if vmax is None:
        vmax = np.nanmax(array)
    factor = 0 if vmin == vmax else 255 / (vmax - vmin)
    scaled = (array - vmin) * factor
    finite = np.isfinite(scaled)
    scaled[~finite] = fill_value
    if scaled.ndim >= 2:
        # (the reference indexes frames 0/1 unconditionally and fails on 1-D input; its own tests
        #  tests/test_flow.py:53-91 pass 1-D arrays, so 1-D input simply skips the pair patch here)
        scaled[0][~finite[0]] = scaled[1][~finite[0]]
        scaled[1][~finite[1]] = scaled[0][~finite[1]]
    with np.errstate(invalid="ignore"):
        return scaled.astype("uint8")


def to_8bit_pair_dev(frame0, frame1, tag="to8bit", out=None):
    """GPU: to_8bit(linear_norm(stack([frame0, frame1])), 0, 1) -> two uint8 torch tensors."""
    t = _lib.torch()
    L = _lib.lib()
    H, W = frame0.shape
    o0, o1 = out if out is not None else (_lib.empty((H, W), t.uint8), _lib.empty((H, W), t.uint8))
    ws = _lib.workspace(max(L.tf_to8bit_workspace_bytes(H, W), 256), tag)
    rc = L.tf_to8bit_pair(_lib.ptr(frame0), _lib.ptr(frame1), H, W, _lib.ptr(o0), _lib.ptr(o1),
                          _lib.ptr(ws), ws.numel(), _lib.stream_ptr())
    _lib.check(rc, "tf_to8bit_pair")
    return o0, o1


def linearise_field(field, lower_threshold, upper_threshold):
    """Clip-and-scale to [0, 1]; reversed thresholds flip the ramp (reference: :36-56)."""
    if lower_threshold == upper_threshold:
        raise ValueError("lower and upper thresholds must have different values")
    if lower_threshold > upper_threshold:
        upper_threshold, lower_threshold = lower_threshold, upper_threshold
        return 1 - np.maximum(np.minimum((field - lower_threshold) / (upper_threshold - lower_threshold), 1), 0)
    return np.maximum(np.minimum((field - lower_threshold) / (upper_threshold - lower_threshold), 1), 0)


# ---- joint normalisation of a frame pair before the 8-bit quantisation ----------------------------------------------
# Six methods, selected by name (calculate_flow's `normalisation_method`).  They are kept value-for-value equal to the
# reference's formulas (tobac_flow/utils/normalisation_utils.py:59-160), including the way log_norm / inverse_log_norm
# reuse the data minimum / maximum as a bound of the LOG values (tests/test_host_logic.py spells out the consequence);
# what is shared between them is factored out: the NaN-ignoring bounds, the unit clip, the log compression.
def _clip_unit(values):
    return np.maximum(np.minimum(values, 1), 0)


def _bounds(array, vmin, vmax):
    lower = np.nanmin(array) if vmin is None else vmin
    upper = np.nanmax(array) if vmax is None else vmax
    return lower, upper


def _log_compress(distance):
    return np.log(distance + 1)


def linear_norm(array, vmin=None, vmax=None):
    """Position of each value between vmin and vmax (defaults: the NaN-ignoring range), clipped to [0, 1];
    an empty or inverted range maps everything to 0."""
    lower, upper = _bounds(array, vmin, vmax)
    scale = 1 / (upper - lower) if upper > lower else 0
    return _clip_unit((array - lower) * scale)


def log_norm(array, vmin=None, vmax=None):
    """linear_norm of log(distance above the data minimum + 1); the data minimum doubles as the lower bound."""
    floor = np.nanmin(array)
    return linear_norm(_log_compress(array - floor), vmin=floor, vmax=vmax)


def inverse_log_norm(array, vmin=None, vmax=None):
    """linear_norm of log(distance below the data maximum + 1); the data maximum doubles as the upper bound."""
    ceiling = np.nanmax(array)
    return linear_norm(_log_compress(ceiling - array), vmin=vmin, vmax=ceiling)


def z_norm(array, max_std=3):
    """Standard score, with +-max_std standard deviations spanning [0, 1]."""
    score = (array - np.nanmean(array)) / np.nanstd(array)
    return linear_norm(score, vmin=-max_std, vmax=max_std)


def uniform_norm(array, quantiles=256):
    """Histogram equalisation: the index of the quantile bin a value falls into, rescaled to [0, 1]."""
    edges = np.quantile(array, np.linspace(0, 1, quantiles + 1))
    edges[-1] = edges[-1] + 1                       # the maximum belongs to the last bin, not to one beyond it
    return linear_norm(np.digitize(array, edges))


def local_linear_norm(data, size=100):
    """Position of each value between the minimum and maximum of its size^n neighbourhood (0 where that
    neighbourhood is flat); NaNs are first replaced by the mean, on a copy."""
    if not np.all(np.isfinite(data)):
        data = np.copy(data)
        data[np.isnan(data)] = np.nanmean(data)
    lowest, highest = ndi.minimum_filter(data, size), ndi.maximum_filter(data, size)
    span = highest - lowest
    flat = span == 0
    scale = np.where(flat, 0, 1 / np.where(flat, 1, span))
    return (data - lowest) * scale


NORMALISATION_METHODS = {
    "linear": linear_norm,
    "log": log_norm,
    "inverse_log": inverse_log_norm,
    "z_score": z_norm,
    "uniform": uniform_norm,
    "local_linear": local_linear_norm,
}


def select_normalisation_method(method: str) -> Callable:
    try:
        return NORMALISATION_METHODS[method]
    except KeyError:
        raise ValueError(f"{method} not an acceptable normalisation method, method must be one of "
                         f"{list(NORMALISATION_METHODS)}") from None


__all__ = ("to_8bit", "linearise_field", "linear_norm", "log_norm", "inverse_log_norm", "z_norm",
           "uniform_norm", "local_linear_norm", "select_normalisation_method")
