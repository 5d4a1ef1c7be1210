"""Flow-aware labelling (mirrors /root/reference/tobac_flow/label.py:84-321).

The whole of it runs behind two C-ABI entry points of the HIP library (include/tobac_flow_hip.h):
tf_flow_label (label.py:84-175: per-step connected components, then the linking) and tf_flow_link_overlap
(label.py:249-321).  What the reference does per label in a Python loop -- a bincount / unique over the label's
pixels in the two nearest-neighbour-warped label volumes (label.py:139-170, utils/label_utils.py:352-376) -- is one
run-length / sort / reduce-by-key pass over the volume on the GPU; the small label graph is then walked in the
reference's own order inside the library (labels ascending, each unvisited label opens a group and absorbs, breadth
first, every not-yet-visited label it overlaps, forward neighbours before backward ones; the relation is DIRECTED, so
first come, first served).
`subsegment_labels` (label.py:13-80, needs scikit-image) is outside the hot path: production passes
subsegment_shrink=0.
"""
import ctypes
import warnings

import numpy as np
from scipy import ndimage as ndi

from tobac_flow_amd import _lib
from tobac_flow_amd.utils.label_utils import find_overlapping_labels


def _structure_bytes(structure):
    s = np.asarray(structure)
    if s.shape != (3, 3, 3):
        raise ValueError("structure must be a (3, 3, 3) array")
    s = np.ascontiguousarray(s != 0, np.uint8)
    if int(s[0].sum()) != 1 or int(s[2].sum()) != 1:
        # label.py:129-131 unpacks Flow.convolve's stack into exactly two arrays
        raise ValueError("structure must have exactly one element in each of its first and last planes "
                         f"(got {int(s[0].sum())} and {int(s[2].sum())}): the reference unpacks two warped label stacks")
    return s


def _call_with_run_retry(fn, size_fn, shape, tag):
    """The library sizes its pair-count scratch for a number of label runs; on TF_ENOMEM it reports the number it
    needs (in the object-count slot) and the call is repeated once with that."""
    T, H, W = shape
    n_obj = ctypes.c_int(0)
    guess = max(T * H * W // 16, 65536)
    for attempt in range(2):
        ws = _lib.workspace(size_fn(T, H, W, guess), tag)
        rc = fn(ws, n_obj)
        if rc == -2 and attempt == 0 and n_obj.value > guess:
            guess = int(n_obj.value) + 1024
            continue
        break
    _lib.check(rc, tag)
    return n_obj.value


def link_overlap_dev(flow, flat_dev, structure, overlap, absolute_overlap):
    """tf_flow_link_overlap on a device int32 tensor of per-step labels; returns the device int32 object labels."""
    t = _lib.torch()
    L = _lib.lib()
    st = _structure_bytes(structure)
    fw, bw = flow._dev_flows()
    T, H, W = flat_dev.shape
    out = _lib.empty((T, H, W), t.int32)
    _call_with_run_retry(
        lambda ws, n_obj: L.tf_flow_link_overlap(_lib.ptr(flat_dev), _lib.ptr(fw), _lib.ptr(bw), T, H, W,
                                                 st.ctypes.data_as(_lib._P), float(overlap), int(absolute_overlap),
                                                 _lib.ptr(out), ctypes.byref(n_obj), _lib.ptr(ws), ws.numel(),
                                                 _lib.stream_ptr()),
        L.tf_flow_link_workspace_bytes, (T, H, W), "tf_flow_link_overlap")
    return out


def flow_label_dev(flow, mask_dev, structure, overlap, absolute_overlap):
    """tf_flow_label on a device uint8 mask; returns the device int32 object labels."""
    t = _lib.torch()
    L = _lib.lib()
    st = _structure_bytes(structure)
    fw, bw = flow._dev_flows()
    T, H, W = mask_dev.shape
    out = _lib.empty((T, H, W), t.int32)
    _call_with_run_retry(
        lambda ws, n_obj: L.tf_flow_label(_lib.ptr(mask_dev), _lib.ptr(fw), _lib.ptr(bw), T, H, W,
                                          st.ctypes.data_as(_lib._P), float(overlap), int(absolute_overlap),
                                          _lib.ptr(out), ctypes.byref(n_obj), _lib.ptr(ws), ws.numel(),
                                          _lib.stream_ptr()),
        L.tf_flow_label_workspace_bytes, (T, H, W), "tf_flow_label")
    return out


def _finish(new_dev, present_dev, dtype, on_device):
    if not bool(((new_dev != 0) == present_dev).all()):
        warnings.warn("Not all regions present in labeled array", RuntimeWarning)      # label.py:172-174
    return new_dev if on_device else _lib.to_host(new_dev).astype(dtype, copy=False)


def flow_label(flow, mask, structure=ndi.generate_binary_structure(3, 1), dtype=np.int32, overlap: float = 0.0,
               absolute_overlap: int = 0, subsegment_shrink: float = 0.0, peak_min_distance: int = 10):
    """Label 3-D connected objects in a semi-Lagrangian frame (reference: label.py:84-175)."""
    if subsegment_shrink != 0:
        raise NotImplementedError("subsegment_shrink != 0 (label.py:13-80, scikit-image watershed) is outside the "
                                  "MI355X hot path; production uses subsegment_shrink=0")
    t = _lib.torch()
    on_device = isinstance(mask, t.Tensor)
    m = (_lib.to_dev(mask) != 0)
    if tuple(m.shape) != tuple(flow.shape):
        raise AssertionError("Data input must have the same shape as the Flow object")
    new_dev = flow_label_dev(flow, m.to(t.uint8).contiguous(), structure, overlap, absolute_overlap)
    return _finish(new_dev, m, dtype, on_device)


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
    t = _lib.torch()
    on_device = isinstance(flat_labels, t.Tensor)
    flat_dev = _lib.to_dev(flat_labels, t.int32)
    if tuple(flat_dev.shape) != tuple(flow.shape):
        raise AssertionError("Data input must have the same shape as the Flow object")
    new_dev = link_overlap_dev(flow, flat_dev, structure, overlap, absolute_overlap)
    return _finish(new_dev, flat_dev != 0, dtype, on_device)


def pair_counts(a, b, include_b_zero=False, _keep_on_device=False):
    """Every distinct pair (a[i], b[i]) with a[i] > 0 and b[i] > 0 (b[i] >= 0 with include_b_zero) of two int32 label
    volumes and how often it occurs, sorted by (a, b): host int64 arrays (ids_a, ids_b, counts).  tf_pair_counts --
    the per-label np.bincount / np.unique of the reference (label_utils.py:352-376, linking.py:33-47, dataset.py:292-297)
    as one run-length / sort / reduce-by-key pass on the GPU."""
    t = _lib.torch()
    L = _lib.lib()
    a_dev, b_dev = _lib.to_dev(a, t.int32).contiguous(), _lib.to_dev(b, t.int32).contiguous()
    if a_dev.shape != b_dev.shape:
        raise ValueError("label volumes must have the same shape")
    n = a_dev.numel()
    if n == 0:
        z = np.zeros(0, np.int64)
        return z, z.copy(), z.copy()
    runs = cap = max(n // 16, 65536)
    n_out = ctypes.c_int64(0)
    for _ in range(3):
        ws = _lib.workspace(L.tf_pair_counts_workspace_bytes(n, runs), "pair_counts")
        oa, ob, oc = _lib.empty((cap,), t.int32), _lib.empty((cap,), t.int32), _lib.empty((cap,), t.int64)
        rc = L.tf_pair_counts(_lib.ptr(a_dev), _lib.ptr(b_dev), n, 1 if include_b_zero else 0, _lib.ptr(oa), _lib.ptr(ob),
                              _lib.ptr(oc), cap, ctypes.byref(n_out), _lib.ptr(ws), ws.numel(), _lib.stream_ptr())
        if rc == -2 and n_out.value > 0:                      # needs room for that many runs / pairs
            runs = cap = int(n_out.value) + 1024
            continue
        break
    _lib.check(rc, "tf_pair_counts")
    k = int(n_out.value)
    if _keep_on_device:
        return oa[:k], ob[:k], oc[:k]
    return (oa[:k].cpu().numpy().astype(np.int64), ob[:k].cpu().numpy().astype(np.int64), oc[:k].cpu().numpy())


def make_step_labels_dev(labels):
    """utils.label_utils.make_step_labels (reference: label_utils.py:183-200) on a device int32 volume: the pieces of the
    non-zero mask connected within a time step (flat_label = tf_label, t planes of the structure zeroed), every piece split
    into the original labels it contains, ids contiguous from 1 ordered by piece, then by label = the rank of the voxel's
    (piece, label) pair among the distinct pairs (tf_pair_counts returns them sorted; tf_pair_rank looks the rank up).
    Labels must be >= 0 (label volumes of the detection recipes are)."""
    from tobac_flow_amd import ndimage_dev as nd
    t = _lib.torch()
    lab = _lib.to_dev(labels, t.int32).contiguous()
    out = t.zeros_like(lab)
    if lab.numel() == 0:
        return out
    if int(lab.min().item()) < 0:
        raise ValueError("make_step_labels_dev: negative labels (the device form ranks (piece, label) pairs of positive labels)")
    plane = ndi.generate_binary_structure(3, 1)
    plane[0] = 0
    plane[2] = 0
    pieces, n_pieces = nd.label(lab != 0, plane)
    if n_pieces == 0:
        return out
    pa, pb, _ = pair_counts(pieces, lab, _keep_on_device=True)
    _lib.check(_lib.lib().tf_pair_rank(_lib.ptr(pieces), _lib.ptr(lab), lab.numel(), _lib.ptr(pa), _lib.ptr(pb), pa.numel(),
                                       _lib.ptr(out), _lib.stream_ptr()), "tf_pair_rank")
    return out


def label_sizes(labels, n_labels=None):
    """np.bincount(labels.ravel(), minlength=n_labels + 1) of a non-negative int32 volume (tf_label_sizes); ids above
    n_labels are not counted."""
    t = _lib.torch()
    L = _lib.lib()
    lab = _lib.to_dev(labels, t.int32).contiguous()
    if n_labels is None:
        n_labels = int(lab.max()) if lab.numel() else 0
    out = _lib.empty((int(n_labels) + 1,), t.int64)
    if lab.numel() == 0:
        return np.zeros(int(n_labels) + 1, np.int64)
    _lib.check(L.tf_label_sizes(_lib.ptr(lab), lab.numel(), int(n_labels), _lib.ptr(out), _lib.stream_ptr()), "tf_label_sizes")
    return _lib.to_host(out)


def slice_labels_dev(labels):
    """utils.label_utils.slice_labels on the GPU (tf_slice_labels): (device int32 step labels, number of step labels)."""
    t = _lib.torch()
    L = _lib.lib()
    lab = _lib.to_dev(labels, t.int32).contiguous()
    T = lab.shape[0]
    hw = lab.numel() // max(T, 1)
    out = _lib.empty(tuple(lab.shape), t.int32)
    if lab.numel() == 0:
        return out, 0
    n_ids = ctypes.c_int64(0)
    cap = 1 << 22
    for _ in range(2):
        ws = _lib.workspace(L.tf_slice_labels_workspace_bytes(T, cap), "slice_labels")
        rc = L.tf_slice_labels(_lib.ptr(lab), T, hw, _lib.ptr(out), ctypes.byref(n_ids), _lib.ptr(ws), ws.numel(), _lib.stream_ptr())
        if rc == -2 and n_ids.value > cap:
            cap = int(n_ids.value)
            continue
        break
    _lib.check(rc, "tf_slice_labels")
    return out, int(n_ids.value)


__all__ = ("flow_label", "find_neighbour_labels", "flow_link_overlap", "flow_label_dev", "link_overlap_dev",
           "pair_counts", "label_sizes", "slice_labels_dev", "make_step_labels_dev")
