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


def _is_sequence(value):
    """iterable but not a string (the reference's test for a multi-valued default / result)"""
    if isinstance(value, str):
        return False
    try:
        iter(value)
    except TypeError:
        return False
    return True


def apply_func_to_labels(labels, *fields, func: Callable = np.mean, index=None, default=None):
    """func(*field values of the region) for every label in `index` (default: 1 .. max label), stacked along the last
    axis and squeezed; `fields` are broadcast against `labels`.  Labels without pixels give `default`, shaped like what
    func returns (a scalar default is repeated for a multi-valued func; a one-element sequence stands for its element).
    Behaviour of the reference's function of the same name (label_utils.py:58-140); the grouping below is a stable sort
    of the label image with searchsorted boundaries, so the values of a region reach func in C order."""
    arrays = np.broadcast_arrays(labels, *fields)
    flat_labels = arrays[0].ravel()
    flat_fields = [f.ravel() for f in arrays[1:]]
    wanted = np.arange(1, max(int(np.max(labels)), 0) + 1) if index is None else np.asarray(list(index))
    by_label = np.argsort(flat_labels, kind="stable")
    sorted_labels = flat_labels[by_label]
    starts = np.searchsorted(sorted_labels, wanted, side="left")
    stops = np.searchsorted(sorted_labels, wanted, side="right")

    def evaluate(k):
        where = by_label[starts[k]:stops[k]]
        return func(*[f[where] for f in flat_fields])

    occupied = np.flatnonzero(stops > starts)
    if _is_sequence(default):
        empty_value = default[0] if len(default) == 1 else default
    else:
        # a scalar default takes the arity of func's result, probed -- as in the reference -- on the lowest label above
        # the background that has any pixel (whether or not it is in `index`); a volume without such a label is an
        # IndexError there as well
        present = np.unique(flat_labels)
        above_background = present[present > min(int(present[0]), 0)]
        if above_background.size == 0:
            raise IndexError("apply_func_to_labels: no labelled region to infer the shape of func's result from")
        lo, hi = np.searchsorted(sorted_labels, above_background[0], "left"), np.searchsorted(sorted_labels, above_background[0], "right")
        probe = func(*[f[by_label[lo:hi]] for f in flat_fields])
        empty_value = [default] * len(probe) if _is_sequence(probe) else default
    filled = set(occupied.tolist())
    return np.stack([evaluate(k) if k in filled else empty_value for k in range(len(wanted))], -1).squeeze()


def flat_label(mask, structure=ndi.generate_binary_structure(3, 1), dtype=np.int32):
    """Connected components that never connect across the leading (time) axis
    (reference: label_utils.py:143-180)."""
    s = structure.copy()
    s[0] = 0
    s[-1] = 0
    return ndi.label(mask, structure=s, output=dtype)[0]


def _dense_lut(values):
    """lookup table value -> rank (1-based) among the sorted distinct positive `values`; 0 stays 0"""
    present = np.unique(values)
    present = present[present > 0]
    lut = np.zeros((int(present[-1]) if present.size else 0) + 1, dtype=np.int64)
    lut[present] = np.arange(1, present.size + 1)
    return lut


def make_step_labels(labels):
    """One id per (time step, spatially connected piece, original label): the non-zero mask is split into the pieces
    that are connected within a time step (flat_label), every piece into the original labels it contains; ids run
    contiguously from 1, ordered by piece and then by original label (the numbering of the reference's function of the
    same name, label_utils.py:183-200).  Vectorised: the ids are the ranks of the distinct (piece, label) pairs."""
    if hasattr(labels, "values"):
        labels = labels.values
    labels = np.asarray(labels)
    pieces = flat_label(labels)
    inside = pieces > 0
    pair_key = pieces[inside].astype(np.int64) * (int(labels.max()) + 1) + labels[inside]
    out = np.zeros_like(pieces)
    out[inside] = np.unique(pair_key, return_inverse=True)[1] + 1
    return out


def get_step_labels_for_label(labels, step_labels):
    """For every label value 1 .. max: the sorted distinct `step_labels` found on its pixels, None for a value that does
    not occur (reference: label_utils.py:202-235)."""
    flat, steps = np.asarray(labels).ravel(), np.asarray(step_labels).ravel()
    top = int(flat.max()) if flat.size else 0
    keep = flat > 0
    base = int(steps.max()) + 1 if steps.size else 1
    pairs = np.unique(flat[keep].astype(np.int64) * base + steps[keep])
    owner, step = pairs // base, pairs % base
    cuts = np.searchsorted(owner, np.arange(1, top + 2))
    return [step[cuts[k]:cuts[k + 1]] if cuts[k + 1] > cuts[k] else None for k in range(top)]


def relabel_objects(labels, inplace=False):
    """Renumber the positive labels 1 .. k in ascending order of their old value; background 0 stays
    (reference: label_utils.py:238-262).  `inplace` rewrites and returns the given array."""
    renumbered = _dense_lut(labels)[labels].astype(labels.dtype, copy=False)
    if inplace:
        labels[...] = renumbered
        return labels
    return renumbered


def remap_labels(labels, locations: Optional[np.ndarray] = None, new_labels: Optional[np.ndarray] = None):
    """labels -> table[labels], where the table sends the labels picked by `locations` (a boolean array over the label
    values 1 .. max, or an array of label values; None = all) to `new_labels` (default 1 .. number picked) and every
    other label to 0 (reference: label_utils.py:265-309)."""
    top = np.nanmax(labels)
    if new_labels is not None:
        top = np.maximum(top, new_labels.size)
    table = np.zeros(top + 1, labels.dtype)
    targets = np.arange(1, np.sum(locations) + 1) if new_labels is None else new_labels
    if locations is None:
        table[1:] = targets
    elif locations.dtype == bool:
        table[1:][locations] = targets
    else:
        table[locations] = targets
    return table[labels]


def slice_labels(labels):
    """One id per (original label, time step): unlike make_step_labels the pieces of a label within a step stay
    together.  Ids are contiguous from 1, ordered by time step and then by original label
    (reference: label_utils.py:312-349)."""
    labels = np.asarray(labels)
    per_step_top = labels.reshape(labels.shape[0], -1).max(axis=1)
    offset = np.concatenate([[0], np.cumsum(per_step_top, dtype=np.int32)[:-1]]).astype(np.int32)
    shifted = labels + offset.reshape((-1,) + (1,) * (labels.ndim - 1))        # distinct id ranges per step
    shifted[labels == 0] = 0
    return _dense_lut(shifted)[shifted].astype(int)


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
