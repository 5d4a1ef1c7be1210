"""ORACLE (test infrastructure, never shipped or measured as the product).

Loop-for-loop numpy restatement of the reference's flow-aware labelling
  flow_label            /root/reference/tobac_flow/label.py:84-175
  find_neighbour_labels /root/reference/tobac_flow/label.py:178-245
  flow_link_overlap     /root/reference/tobac_flow/label.py:249-321
  flat_label            /root/reference/tobac_flow/utils/label_utils.py:143-180
  find_overlapping_labels /root/reference/tobac_flow/utils/label_utils.py:352-376
with the nearest-neighbour label warps taken from oracle/np_ops.convolve.  The reference has no test
for these functions; the restatement keeps its control flow (per-label bincount / unique, BFS with a
`processed` array) so that label numbering and the first-come grouping of asymmetric overlaps are the
reference's.
"""
import numpy as np
import scipy.ndimage as ndi

from . import np_ops


def flat_label(mask, structure=ndi.generate_binary_structure(3, 1), dtype=np.int32):
    s = structure.copy()
    s[0] = 0
    s[-1] = 0
    return ndi.label(mask, structure=s, output=dtype)[0]


def find_overlapping_labels(labels, locs, bins, overlap=0, absolute_overlap=0):
    n_locs = len(locs)
    if n_locs == 0:
        return []
    hit = labels.ravel()[locs]
    counts = np.bincount(np.maximum(hit, 0))
    return [new for new in np.unique(hit)
            if new != 0 and counts[new] > absolute_overlap
            and counts[new] >= overlap * np.minimum(n_locs, bins[new] - bins[new - 1])]


def _link(flat_labels, fwd, bwd, structure, dtype, overlap, absolute_overlap):
    label_struct = structure * np.array([1, 0, 1])[:, np.newaxis, np.newaxis]
    back_labels, forward_labels = np_ops.convolve(flat_labels, fwd, bwd, label_struct, "nearest", dtype, 0)
    bins = np.cumsum(np.bincount(flat_labels.ravel()))
    args = np.argsort(flat_labels.ravel())
    processed = np.zeros(bins.size, dtype=bool)
    label_map = {}
    for label in range(1, bins.size):
        if not processed[label]:
            label_map[label] = [label]
            processed[label] = True
            i = 0
            while i < len(label_map[label]):
                cur = label_map[label][i]
                if bins[cur] > bins[cur - 1]:
                    locs = args[bins[cur - 1]:bins[cur]]
                    for warped in (forward_labels, back_labels):
                        for new in find_overlapping_labels(warped, locs, bins, overlap, absolute_overlap):
                            if not processed[new]:
                                label_map[label].append(new)
                                processed[new] = True
                i += 1
    new_labels = np.zeros(flat_labels.shape, dtype=dtype)
    for ik, k in enumerate(label_map):
        for i in label_map[k]:
            if bins[i] > bins[i - 1]:
                new_labels.ravel()[args[bins[i - 1]:bins[i]]] = ik + 1
    return new_labels


def flow_label(fwd, bwd, mask, structure=ndi.generate_binary_structure(3, 1), dtype=np.int32, overlap=0.0,
               absolute_overlap=0):
    return _link(flat_label(mask != 0, structure).astype(dtype), fwd, bwd, structure, dtype, overlap, absolute_overlap)


def flow_link_overlap(fwd, bwd, flat_labels, structure=ndi.generate_binary_structure(3, 1), dtype=np.int32, overlap=0.0,
                      absolute_overlap=0):
    return _link(np.asarray(flat_labels), fwd, bwd, structure, dtype, overlap, absolute_overlap)


# ---- cross-window linking: /root/reference/tobac_flow/linking.py:33-47 (find_overlaps), :49-93
# (find_overlap_between_cores; :96-140 is the same for anvils), :153-161 (find_new_labels) ----------------------
def find_overlaps(x, atol, rtol, max_label, label_counts):
    overlap_counts = np.bincount(x, minlength=max_label + 1)
    wh_overlap = overlap_counts >= atol if atol > 0 else overlap_counts > 0
    if rtol > 0:
        wh_overlap = np.logical_and(
            wh_overlap, np.maximum(overlap_counts / x.size, overlap_counts / label_counts) >= rtol)
    wh_overlap[0] = False
    return np.where(wh_overlap)[0]


def link_overlap_pairs(current_overlap, next_overlap, atol=5, rtol=0.5):
    """current_overlap / next_overlap: the labels two consecutive windows give to their common frames, the first and the
    last common frame already dropped (linking.py:55-56 `t_overlap[1:-1]`).  Returns (x, y): left ids repeated per
    linked right id, as linking.py:78-91 builds them."""
    from functools import partial
    cur = np.asarray(current_overlap)
    nxt = np.asarray(next_overlap)
    max_label = int(nxt.max()) if nxt.size else 0
    index = np.unique(cur[cur > 0])                                   # current_ds.core.values: the labels of the window
    label_counts = np.maximum(np.bincount(nxt.ravel(), minlength=max_label + 1), 1)
    comp = partial(find_overlaps, atol=atol, rtol=rtol, max_label=max_label, label_counts=label_counts)
    overlap_labels = ndi.labeled_comprehension(nxt.ravel(), cur.ravel(), index, comp, list, [[]])
    x = np.repeat(index, [len(n) for n in overlap_labels])
    y = np.concatenate([np.asarray(n, np.int64) for n in overlap_labels]) if len(overlap_labels) else np.zeros(0, np.int64)
    return x.astype(np.int64), y.astype(np.int64)


def find_new_labels(x, y, size):
    import scipy.sparse
    import scipy.sparse.csgraph
    graph = scipy.sparse.coo_array((np.ones(x.size), (x, y)), shape=(size, size))
    return scipy.sparse.csgraph.connected_components(graph, directed=False)[1]
