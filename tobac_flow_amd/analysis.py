"""Label filters used by the detection recipes (host glue; mirrors the first three groups of
/root/reference/tobac_flow/analysis.py:15-201; the dataset statistics below them are out of scope)."""
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
