"""Flow-aware labelling (mirrors /root/reference/tobac_flow/label.py:84-321).

What the reference does per label in a Python loop (a bincount / unique over the label's pixels in the
two nearest-neighbour-warped label volumes, label.py:139-170 + utils/label_utils.py:352-376) is done
once for the whole volume:
  * the two label warps run on the GPU (Flow.convolve, int32 / nearest -> tf_convolve),
  * the overlap counts of every (label, warped label) pair come from one sort/unique of packed keys,
  * the grouping keeps the reference's semantics exactly: labels are visited in ascending order, each
    unvisited label starts a group and absorbs, breadth first, every not-yet-visited label it overlaps
    (the relation is DIRECTED -- the relative criterion uses min(size of the current label, size of
    the other) but the count is taken over the current label's pixels -- so first come, first served).
`subsegment_labels` (label.py:13-80, needs scikit-image) is outside the hot path: production passes
subsegment_shrink=0.
"""
import warnings

import numpy as np
from scipy import ndimage as ndi

from tobac_flow_amd import _lib
from tobac_flow_amd.utils.label_utils import find_overlapping_labels, flat_label


def _overlap_edges(flat, warped, sizes, overlap, absolute_overlap):
    """Directed edges a -> b (b seen in `warped` under the pixels of a) that satisfy the reference's
    criterion; returned as (a, b) arrays sorted by (a, b)."""
    t = _lib.torch()
    a = flat.reshape(-1).to(t.int64)
    b = warped.reshape(-1).to(t.int64)
    both = (a > 0) & (b != 0)
    a, b = a[both], b[both]
    neg = b < 0                      # np.bincount(np.maximum(hit, 0)): negative warped labels count as 0 and are dropped
    a, b = a[~neg], b[~neg]
    nlab = int(sizes.numel())
    key, cnt = t.unique(a * nlab + b, return_counts=True)
    ea, eb = key // nlab, key % nlab
    ok = (cnt > absolute_overlap) & (cnt.to(t.float64) >= overlap * t.minimum(sizes[ea], sizes[eb]).to(t.float64))
    return ea[ok].cpu().numpy(), eb[ok].cpu().numpy()


def _link(flow, flat_labels, structure, dtype, overlap, absolute_overlap, present_mask, on_device=False):
    t = _lib.torch()
    label_struct = structure * np.array([1, 0, 1])[:, np.newaxis, np.newaxis]
    flat_dev = _lib.to_dev(flat_labels, t.int32)
    back_labels, forward_labels = flow.convolve(flat_dev, method="nearest", dtype=dtype, structure=label_struct,
                                                fill_value=0)
    n = int(flat_dev.max().item()) + 1
    sizes = t.bincount(flat_dev.reshape(-1).to(t.int64), minlength=n)
    edges = [_overlap_edges(flat_dev, w, sizes, overlap, absolute_overlap) for w in (forward_labels, back_labels)]
    # adjacency in the reference's visiting order: forward-warp neighbours (ascending), then backward-warp ones
    order = []
    for ea, eb in edges:
        idx = np.searchsorted(ea, np.arange(n + 1))
        order.append((idx, eb))
    sizes_np = sizes.cpu().numpy()
    processed = np.zeros(n, dtype=bool)
    group_of = np.zeros(n, dtype=np.int64)
    n_groups = 0
    for label in range(1, n):
        if processed[label]:
            continue
        n_groups += 1
        stack = [label]
        processed[label] = True
        i = 0
        while i < len(stack):
            cur = stack[i]
            if sizes_np[cur] > 0:
                for idx, eb in order:
                    for new in eb[idx[cur]:idx[cur + 1]]:
                        if not processed[new]:
                            processed[new] = True
                            stack.append(new)
            i += 1
        group_of[stack] = n_groups
    group_of[sizes_np == 0] = 0          # labels without pixels are never written (label.py:166-170)
    lut = t.from_numpy(group_of.astype(np.int64)).to(flat_dev.device)
    new_dev = lut[flat_dev.to(t.int64)]
    if on_device:
        if not bool(((new_dev != 0) == present_mask).all()):
            warnings.warn("Not all regions present in labeled array", RuntimeWarning)
        return new_dev.to(t.int32)
    new_labels = new_dev.cpu().numpy().astype(dtype)
    if not np.all((new_labels != 0) == present_mask):
        warnings.warn("Not all regions present in labeled array", RuntimeWarning)
    return new_labels


def flow_label(flow, mask, structure=ndi.generate_binary_structure(3, 1), dtype=np.int32, overlap: float = 0.0,
               absolute_overlap: int = 0, subsegment_shrink: float = 0.0, peak_min_distance: int = 10):
    """Label 3-D connected objects in a semi-Lagrangian frame (reference: label.py:84-175)."""
    if subsegment_shrink != 0:
        raise NotImplementedError("subsegment_shrink != 0 (label.py:13-80, scikit-image watershed) is outside the "
                                  "MI355X hot path; production uses subsegment_shrink=0")
    t = _lib.torch()
    from tobac_flow_amd import ndimage_dev as nd
    on_device = isinstance(mask, t.Tensor)
    m = _lib.to_dev(mask) != 0
    # per-frame connected components on the GPU (tf_label: SciPy's numbering, tests/test_gpu_detection.py)
    flat_labels = nd.flat_label(m, structure)
    if on_device:
        return _link(flow, flat_labels, structure, dtype, overlap, absolute_overlap, m, on_device=True)
    return _link(flow, flat_labels, structure, dtype, overlap, absolute_overlap, np.asarray(mask) != 0)


def find_neighbour_labels(label, label_stack, bins, args, processed_labels, forward_labels, back_labels,
                          overlap: float = 0, absolute_overlap: int = 1):
    """Append the not-yet-visited labels that overlap `label` at t+1 / t-1 (reference: label.py:178-245;
    kept for API parity -- flow_label / flow_link_overlap use the vectorised form above)."""
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
