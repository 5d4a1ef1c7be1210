"""Label bookkeeping helpers (host glue in numpy; mirrors
/root/reference/tobac_flow/utils/label_utils.py -- same names, arguments and results)."""
from typing import Callable, Optional

import numpy as np
import scipy.ndimage as ndi


def _groups(flat):
    """argsort + cumulative bin edges: pixels of label i are order[edges[i-1]:edges[i]]."""
    return np.argsort(flat), np.cumsum(np.bincount(flat))


def labeled_comprehension(field, labels, func, index=None, dtype=None, default=None, pass_positions=False):
    """ndi.labeled_comprehension with defaults filled in (reference: label_utils.py:8-55)."""
    if not dtype:
        dtype = field.dtype
    if index is None:
        index = np.unique(labels[labels != 0])
    return ndi.labeled_comprehension(field, labels, index, func, dtype, default, pass_positions)


def apply_func_to_labels(labels, *fields, func: Callable = np.mean, index=None, default=None):
    """Apply `func` to the values of each labelled region (reference: label_utils.py:58-140)."""
    arrays = np.broadcast_arrays(labels, *fields)
    blabels, bfields = arrays[0], arrays[1:]
    if index is None:
        low = np.minimum(np.min(labels), 0)
        n_bins = np.max(labels) - low + 1
        index = range(1, n_bins)
    else:
        low = np.minimum.reduce([np.min(index) - 1, np.min(labels), 0])
        n_bins = np.maximum(np.max(index), np.max(labels)) - low + 1
    edges = np.cumsum(np.bincount(blabels.ravel() - low, minlength=n_bins))
    order = np.argsort(blabels.ravel())

    def region(i):
        return [f.ravel()[order[edges[i - low - 1]:edges[i - low]]] for f in bfields]

    # shape the default like func's return value (scalar or tuple)
    try:
        iter(default)
        assert not isinstance(default, str)
    except (TypeError, AssertionError):
        first = np.where(np.diff(edges))[0][0] + 1
        probe = func(*[f.ravel()[order[edges[first - 1]:edges[first]]] for f in bfields])
        try:
            assert not isinstance(probe, str)
            default_vals = [default] * len(probe)
        except (AssertionError, TypeError):
            default_vals = default
    else:
        default_vals = default[0] if (len(default) == 1 and not isinstance(default, str)) else default
    return np.stack([func(*region(i)) if edges[i - low] > edges[i - low - 1] else default_vals for i in index],
                    -1).squeeze()


def flat_label(mask, structure=ndi.generate_binary_structure(3, 1), dtype=np.int32):
    """Connected components that never connect across the leading (time) axis
    (reference: label_utils.py:143-180)."""
    s = structure.copy()
    s[0] = 0
    s[-1] = 0
    return ndi.label(mask, structure=s, output=dtype)[0]


def make_step_labels(labels):
    """Split labels into per-time-step, per-original-label pieces (reference: label_utils.py:183-200)."""
    if hasattr(labels, "values"):
        labels = labels.values
    step = flat_label(labels)
    order, edges = _groups(step.ravel())
    nxt = 1
    for i in range(edges.size - 1):
        if edges[i + 1] > edges[i]:
            where = order[edges[i]:edges[i + 1]]
            inv = np.unique(labels.ravel()[where], return_inverse=True)[1]
            step.ravel()[where] = inv + nxt
            nxt += np.max(inv) + 1
    return step


def get_step_labels_for_label(labels, step_labels):
    """For each label, the step labels it is made of (reference: label_utils.py:202-235)."""
    order, edges = _groups(labels.ravel())
    return [np.unique(step_labels.ravel()[order[edges[i]:edges[i + 1]]]) if edges[i + 1] > edges[i] else None
            for i in range(edges.size - 1)]


def relabel_objects(labels, inplace=False):
    """Renumber labels to contiguous integers (reference: label_utils.py:238-262)."""
    order, edges = _groups(labels.ravel())
    if not inplace:
        labels = np.zeros_like(labels)
    nxt = 1
    for i in range(edges.size - 1):
        if edges[i + 1] > edges[i]:
            labels.ravel()[order[edges[i]:edges[i + 1]]] = nxt
            nxt += 1
    return labels


def remap_labels(labels, locations: Optional[np.ndarray] = None, new_labels: Optional[np.ndarray] = None):
    """Keep the labels selected by `locations`, renumbered contiguously or to `new_labels`
    (reference: label_utils.py:265-309)."""
    top = np.nanmax(labels)
    if new_labels is not None:
        top = np.maximum(top, new_labels.size)
    lut = np.zeros(top + 1, labels.dtype)
    if new_labels is None:
        new_labels = np.arange(1, np.sum(locations) + 1)
    if locations is not None:
        if locations.dtype == bool:
            lut[1:][locations] = new_labels
        else:
            lut[locations] = new_labels
    else:
        lut[1:] = new_labels
    return lut[labels]


def slice_labels(labels):
    """Give every (label, time step) pair its own id, contiguous (reference: label_utils.py:312-349)."""
    per_step_max = np.cumsum(np.max(labels, axis=tuple(range(1, labels.ndim))), dtype=np.int32)
    per_step_max[1:] = per_step_max[:-1]
    per_step_max[0] = 0
    per_step_max = per_step_max.reshape([-1] + [1] * (labels.ndim - 1))
    step = labels + per_step_max
    step[labels == 0] = 0
    present = np.where(np.bincount(step.ravel()))[0]
    lut = np.zeros(present[-1] + 1, dtype=int)
    lut[present] = np.arange(present.size)
    return lut[step]


def find_overlapping_labels(labels, locs, bins, overlap: float = 0, absolute_overlap: int = 0):
    """Labels of `labels` present at `locs` with count > absolute_overlap and
    count >= overlap * min(len(locs), size of that label) (reference: label_utils.py:352-376)."""
    n_locs = len(locs)
    if n_locs == 0:
        return []
    hit = labels.ravel()[locs]
    counts = np.bincount(np.maximum(hit, 0))
    return [lab for lab in np.unique(hit)
            if lab != 0 and counts[lab] > absolute_overlap
            and counts[lab] >= overlap * np.minimum(n_locs, bins[lab] - bins[lab - 1])]


__all__ = ("labeled_comprehension", "apply_func_to_labels", "flat_label", "make_step_labels",
           "get_step_labels_for_label", "relabel_objects", "slice_labels", "find_overlapping_labels",
           "remap_labels")
