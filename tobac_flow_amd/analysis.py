"""Label filters used by the detection recipes (host glue; mirrors the first three groups of
/root/reference/tobac_flow/analysis.py:15-201) and the per-label statistics of analysis.py:204-245 / 293-376 as
segmented reductions on the GPU (tf_label_stats).  The coverage / unique-count maps of analysis.py:248-290 and the
xarray packaging (names, long_name, units attributes) are out of scope: xarray is not in this image."""
import numpy as np
from scipy import ndimage as ndi


def find_object_lengths(labels, axis: int = 0):
    """Extent of every label along `axis` (reference: analysis.py:15-35)."""
    return np.array([o[axis].stop - o[axis].start for o in ndi.find_objects(labels)])


def mask_labels(labels, mask):
    """Boolean per label (1..max): does it overlap `mask`? (reference: analysis.py:38-63)"""
    assert labels.shape == mask.shape, "Labels and mask parameters must have the same shape"
    hit = np.unique(labels[mask])
    out = np.zeros(labels.max() + 1, dtype=bool)
    out[hit] = True
    return out[1:]


def _keep(labels, wh):
    lut = np.zeros([np.nanmax(labels) + 1], labels.dtype)
    lut[1:] = np.cumsum(wh) * wh
    return lut[labels]


def _lengths_ok(labels, min_length):
    return np.array([o[0].stop - o[0].start for o in ndi.find_objects(labels)]) >= min_length


def _any_in(labels, mask, dtype=None, default=None):
    return ndi.labeled_comprehension(mask, labels, range(1, np.nanmax(labels) + 1), np.any, dtype, default)


def filter_labels_by_length(labels, min_length):
    return _keep(labels, _lengths_ok(labels, min_length))


def filter_labels_by_mask(labels, mask):
    return _keep(labels, _any_in(labels, mask))


def filter_labels_by_length_and_mask(labels, mask, min_length):
    return _keep(labels, np.logical_and(_lengths_ok(labels, min_length), _any_in(labels, mask)))


def filter_labels_by_multimask(labels, masks):
    if type(masks) is not list:
        raise ValueError("masks input must be a list of masks to process")
    return _keep(labels, np.logical_and.reduce([_any_in(labels, m, np.bool_, 0) for m in masks]))


def filter_labels_by_length_and_multimask(labels, masks, min_length):
    if type(masks) is not list:
        raise ValueError("masks input must be a list of masks to process")
    return _keep(labels, np.logical_and(_lengths_ok(labels, min_length),
                                        np.logical_and.reduce([_any_in(labels, m, np.bool_, 0) for m in masks])))


def _legacy_filter(labels, keep_fn):
    """in-place renumbering in ascending label order (analysis.py:142-201)"""
    flat = labels.ravel()
    edges = np.cumsum(np.bincount(flat))
    order = np.argsort(flat)
    lengths = np.array([o[0].stop - o[0].start for o in ndi.find_objects(labels)])
    nxt = 1
    for i in range(edges.size - 1):
        if edges[i + 1] > edges[i]:
            where = order[edges[i]:edges[i + 1]]
            if keep_fn(lengths[i], where):
                flat[where] = nxt
                nxt += 1
            else:
                flat[where] = 0
    return labels


def filter_labels_by_length_legacy(labels, min_length):
    return _legacy_filter(labels, lambda n, where: n >= min_length)


def filter_labels_by_length_and_mask_legacy(labels, mask, min_length):
    return _legacy_filter(labels, lambda n, where: n >= min_length and np.any(mask.ravel()[where]))


def filter_labels_by_length_and_multimask_legacy(labels, masks, min_length):
    if type(masks) is not list:
        raise ValueError("masks input must be a list of masks to process")
    return _legacy_filter(labels, lambda n, where: n >= min_length and np.all([np.any(m.ravel()[where]) for m in masks]))


def _label_stats(labels, field, weights, dtype):
    """(mean, std, max, min) per label 1 .. labels.max() through tf_label_stats; NaN where a label carries no weight."""
    import ctypes
    from tobac_flow_amd import _lib
    t = _lib.torch()
    L = _lib.lib()
    if tuple(np.shape(labels)) != tuple(np.shape(field)) or (weights is not None and tuple(np.shape(weights)) != tuple(np.shape(field))):
        raise ValueError("Input labels and field do not have the same shape")          # legacy_utils.py:44-45
    lab = _lib.to_dev(labels, t.int32).contiguous()
    n_labels = int(lab.max()) if lab.numel() else 0
    if dtype is None:
        dtype = np.float32 if _lib.is_tensor(field) else np.asarray(field).dtype
    if n_labels <= 0:
        return tuple(np.zeros(0, dtype) for _ in range(4))
    x = _lib.to_dev(field, t.float32).contiguous()
    w = None if weights is None else _lib.to_dev(weights, t.float32).contiguous()
    out = _lib.empty((n_labels, 6), t.float64)
    ws = _lib.workspace(L.tf_label_stats_workspace_bytes(n_labels), "label_stats")
    _lib.check(L.tf_label_stats(_lib.ptr(lab), _lib.ptr(x), _lib.ptr(w) if w is not None else ctypes.c_void_p(0), lab.numel(),
                                n_labels, _lib.ptr(out), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "tf_label_stats")
    o = out.cpu().numpy()
    carries_weight = o[:, 0] > 0                                  # `if np.nansum(w) > 0 else [nan] * 4`
    return tuple(np.where(carries_weight, o[:, k], np.nan).astype(dtype) for k in (2, 3, 4, 5))


def get_stats_for_labels(labels, da, dim=None, dtype=None):
    """np.nanmean / nanstd / nanmax / nanmin of `da` over every label 1 .. max: four arrays of length labels.max()
    (reference: analysis.py:204-245, which wraps them as DataArrays named f"{dim}_{da.name}_mean" etc.).  Accumulated in
    double on the GPU; the reference sums in the data's own precision."""
    return _label_stats(labels, da, None, dtype)


def weighted_statistics_on_labels(labels, da, weights, name=None, dim=None, dtype=None):
    """Weighted mean, weighted standard deviation, and the max / min over the positively weighted values of `da` for every
    label 1 .. max, NaN values ignored, NaN for labels without weight (reference: analysis.py:293-376)."""
    return _label_stats(labels, da, weights, dtype)
