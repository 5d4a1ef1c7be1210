"""Flow-aware labelling (mirrors /root/reference/tobac_flow/label.py:84-321).

The two full-volume nearest-neighbour label warps run on the GPU (Flow.convolve with int32
data); the overlap graph and its BFS closure are host numpy, structured like the reference so that
label numbering is identical (ascending by the smallest member per-frame label).
`subsegment_labels` (label.py:13-80, needs scikit-image) is outside the hot path: production
passes subsegment_shrink=0.
"""
import warnings

import numpy as np
from scipy import ndimage as ndi

from tobac_flow_amd.utils.label_utils import find_overlapping_labels, flat_label


def _link(flow, flat_labels, structure, dtype, overlap, absolute_overlap, present_mask):
    label_struct = structure * np.array([1, 0, 1])[:, np.newaxis, np.newaxis]
    back_labels, forward_labels = flow.convolve(flat_labels, method="nearest", dtype=dtype,
                                                structure=label_struct, fill_value=0)
    flat = flat_labels.ravel()
    bins = np.cumsum(np.bincount(flat))
    args = np.argsort(flat)
    processed = np.zeros(bins.size, dtype=bool)
    groups = {}
    for label in range(1, bins.size):
        if processed[label]:
            continue
        stack = groups[label] = [label]
        processed[label] = True
        i = 0
        while i < len(stack):
            find_neighbour_labels(stack[i], stack, bins, args, processed, forward_labels, back_labels,
                                  overlap=overlap, absolute_overlap=absolute_overlap)
            i += 1
    new_labels = np.zeros(flat_labels.shape, dtype=dtype)
    out = new_labels.ravel()
    for new_id, key in enumerate(groups):
        for member in groups[key]:
            if bins[member] > bins[member - 1]:
                out[args[bins[member - 1]:bins[member]]] = new_id + 1
    if not np.all((new_labels != 0) == present_mask):
        warnings.warn("Not all regions present in labeled array", RuntimeWarning)
    return new_labels


def flow_label(flow, mask, structure=ndi.generate_binary_structure(3, 1), dtype=np.int32, overlap: float = 0.0,
               absolute_overlap: int = 0, subsegment_shrink: float = 0.0, peak_min_distance: int = 10):
    """Label 3-D connected objects in a semi-Lagrangian frame (reference: label.py:84-175)."""
    mask = np.asarray(mask)
    if subsegment_shrink != 0:
        raise NotImplementedError("subsegment_shrink != 0 (label.py:13-80, scikit-image watershed) is outside the "
                                  "MI355X hot path; production uses subsegment_shrink=0")
    flat_labels = flat_label(mask != 0, structure=structure).astype(dtype)
    return _link(flow, flat_labels, structure, dtype, overlap, absolute_overlap, mask != 0)


def find_neighbour_labels(label, label_stack, bins, args, processed_labels, forward_labels, back_labels,
                          overlap: float = 0, absolute_overlap: int = 1):
    """Append the not-yet-visited labels that overlap `label` at t+1 / t-1 (reference: label.py:178-245)."""
    if bins[label] > bins[label - 1]:
        locs = args[bins[label - 1]:bins[label]]
        for warped in (forward_labels, back_labels):
            for new_label in find_overlapping_labels(warped, locs, bins, overlap=overlap,
                                                     absolute_overlap=absolute_overlap):
                if not processed_labels[new_label]:
                    label_stack.append(new_label)
                    processed_labels[new_label] = True


def flow_link_overlap(flow, flat_labels, structure=ndi.generate_binary_structure(3, 1), dtype=np.int32,
                      overlap: float = 0.0, absolute_overlap: int = 0):
    """Link existing per-step labels into contiguous objects (reference: label.py:249-321)."""
    flat_labels = np.asarray(flat_labels)
    return _link(flow, flat_labels, structure, dtype, overlap, absolute_overlap, flat_labels.astype(bool))


__all__ = ("flow_label", "find_neighbour_labels", "flow_link_overlap")
