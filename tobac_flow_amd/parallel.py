"""Multi-GPU scheme: time windows are independent units (one window per GPU, no data-path
collective); label IDs are stitched at the end with one all-gather.

This is the MI355X form of the reference's own scale-out: independent windows with overlap frames
(scripts/dcc_detect_goes.py:153 `n_pad_files`), then overlap-based linking of label IDs
(/root/reference/tobac_flow/linking.py:49-161): on the frames two consecutive windows share -- minus the first and
the last of them (linking.py:55-56) -- a label of the left window and a label of the right window are the same
object when they coincide in >= 5 pixels AND in >= 0.5 of the pixels of either one (linking.py:33-47, atol / rtol);
the stitched objects are the connected components of those pairs (linking.py:153-161).  The collective is a
`torch.distributed.all_gather` (RCCL on GPUs, gloo in the CPU tests) of each rank's label count and pair list, after
which every rank runs the same union-find and rewrites its own labels (tf_apply_lut).  Pair counting on the GPU is
the library's tf_window_overlap_pairs; CPU tensors (the gloo rehearsals) take the numpy statement of the same rule.
"""
import numpy as np

LINK_ATOL, LINK_RTOL = 5, 0.5          # linking.py:70-76


def window_bounds(T, world, overlap=1):
    """Split T frames into `world` contiguous windows sharing `overlap` frames: [(start, stop), ...]."""
    if world < 1 or T < world + overlap * (world - 1):
        raise ValueError("not enough frames for the requested number of windows")
    body = T - overlap
    edges = [round(i * body / world) for i in range(world + 1)]
    return [(edges[i], edges[i + 1] + overlap) for i in range(world)]


def _find(parent, i):
    while parent[i] != i:
        parent[i] = parent[parent[i]]
        i = parent[i]
    return i


def stitch_lut(counts, pairs_per_boundary):
    """Global relabelling tables from per-rank label counts and per-boundary (id_left, id_right) pairs.

    counts[r] = number of labels (max id) of rank r; pairs_per_boundary[r] = (k, 2) array of label
    pairs that coincide in the frame shared by rank r and rank r+1.  Returns one LUT per rank
    (index = local id, value = global id, contiguous from 1 in order of first appearance over
    (rank, local id)), identical on every rank."""
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    n = int(offs[-1])
    parent = np.arange(n + 1)
    for r, pairs in enumerate(pairs_per_boundary):
        for a, b in np.asarray(pairs, np.int64).reshape(-1, 2):
            ra, rb = _find(parent, offs[r] + a), _find(parent, offs[r + 1] + b)
            if ra != rb:
                parent[max(ra, rb)] = min(ra, rb)
    root = np.array([_find(parent, i) for i in range(n + 1)])
    new = np.zeros(n + 1, np.int64)
    nxt = 0
    for i in range(1, n + 1):           # canonical numbering: by smallest member
        if root[i] == i:
            nxt += 1
            new[i] = nxt
    new = new[root]
    return [np.concatenate([[0], new[offs[r] + 1: offs[r + 1] + 1]]) for r in range(len(counts))]


def compare_frames(overlap):
    """Which of the `overlap` common frames the linking looks at: all but the first and the last (linking.py:55-56).
    The reference links nothing when two windows share fewer than three frames; here one or two shared frames are
    all used (an extension: the rule is otherwise the same)."""
    return slice(1, overlap - 1) if overlap > 2 else slice(0, overlap)


def _overlap_pairs_host(left, right, atol, rtol):
    """linking.py:33-47 + :58-93 in numpy (CPU tensors of the gloo rehearsals; also the statement the GPU path is
    tested against).  left / right: integer arrays of the same frames; returns (k, 2) int64 sorted by (left, right)."""
    a, b = np.asarray(left).reshape(-1).astype(np.int64), np.asarray(right).reshape(-1).astype(np.int64)
    keep = (a > 0) & (b >= 0)
    a, b = a[keep], b[keep]
    if a.size == 0:
        return np.zeros((0, 2), np.int64)
    rb = np.asarray(right).reshape(-1).astype(np.int64)
    right_size = np.maximum(np.bincount(rb[rb > 0], minlength=int(max(b.max(), 1)) + 1), 1)
    base = int(b.max()) + 1
    key, cnt = np.unique(a * base + b, return_counts=True)
    ka, kb = key // base, key % base
    n_left = np.bincount(ka, weights=cnt).astype(np.int64)
    ok = (cnt >= atol) if atol > 0 else (cnt > 0)
    if rtol > 0:
        ok &= np.maximum(cnt / n_left[ka], cnt / right_size[kb]) >= rtol
    ok &= kb != 0
    return np.stack([ka[ok], kb[ok]], 1).astype(np.int64)


def overlap_pairs(left, right, atol=LINK_ATOL, rtol=LINK_RTOL):
    """(id_left, id_right) pairs that the reference's linking rule joins (module docstring).  `left` / `right`: the
    labels two consecutive windows give to the SAME frames (already reduced to compare_frames); torch tensors, on the
    GPU (library path, tf_window_overlap_pairs) or on the CPU (numpy path).  Returns a (k, 2) int64 numpy array."""
    import ctypes
    import torch
    if tuple(left.shape) != tuple(right.shape):
        raise ValueError("the two windows must hold the same frames")
    if left.numel() == 0:
        return np.zeros((0, 2), np.int64)
    if not left.is_cuda:
        return _overlap_pairs_host(left.numpy(), right.numpy(), atol, rtol)
    from tobac_flow_amd import _lib
    L = _lib.lib()
    a, b = left.to(torch.int32).contiguous(), right.to(torch.int32).contiguous()
    n = a.numel()
    runs, cap = max(n // 16, 65536), 1 << 16
    n_out = ctypes.c_int64(0)
    for _ in range(4):
        ws = _lib.workspace(L.tf_pair_counts_workspace_bytes(n, runs), "pair_counts")
        out = np.zeros((cap, 2), np.int32)
        rc = L.tf_window_overlap_pairs(_lib.ptr(a), _lib.ptr(b), n, int(atol), float(rtol), out.ctypes.data_as(_lib._P), cap,
                                       ctypes.byref(n_out), _lib.ptr(ws), ws.numel(), _lib.stream_ptr())
        if rc == -2 and n_out.value > runs and b"runs" in L.tf_last_error():
            runs = int(n_out.value) + 1024
            continue
        if rc == -2 and n_out.value > cap:
            cap = int(n_out.value)
            continue
        break
    _lib.check(rc, "tf_window_overlap_pairs")
    return out[:n_out.value].astype(np.int64)


def boundary_pairs(left_last, right_first, base=None, min_overlap=1):
    """Pairs of positive labels that coincide in >= `min_overlap` pixels of one shared frame (the absolute criterion
    alone: overlap_pairs with rtol = 0).  `base` is ignored (kept for callers of the round-1 signature)."""
    import torch
    return torch.from_numpy(overlap_pairs(left_last, right_first, atol=max(int(min_overlap), 1), rtol=0.0))


def stitch_window_list(windows, min_overlap=None, overlap=1, atol=LINK_ATOL, rtol=LINK_RTOL):
    """Single-process form of stitch_labels: `windows` is a list of (T_w, H, W) int32 label tensors (CPU or GPU), the
    first `overlap` frames of window w + 1 being the last `overlap` frames of window w (e.g. a 144-frame stack processed
    as twelve windows on one GPU, window_bounds).  Returns the relabelled windows: positive ids made globally consistent
    (contiguous from 1 in order of first appearance), zero and negative ids kept.  Same rule and LUT logic as the
    distributed version.  `min_overlap` (round-1 signature) = absolute criterion only: atol = min_overlap, rtol = 0."""
    import torch
    if len(windows) == 0:
        return []
    if min_overlap is not None:
        atol, rtol = max(int(min_overlap), 1), 0.0
    counts = [int(torch.clamp(w.max(), min=0).item()) if w.numel() else 0 for w in windows]
    sel = compare_frames(overlap)
    pairs = []
    for r in range(len(windows) - 1):
        if windows[r].shape[1:] != windows[r + 1].shape[1:]:
            raise ValueError("windows must share their spatial shape")
        left = windows[r][windows[r].shape[0] - overlap:][sel]
        right = windows[r + 1][:overlap][sel]
        pairs.append(overlap_pairs(left, right, atol, rtol))
    luts = stitch_lut(counts, pairs)
    return [apply_global_lut(w, lut) for w, lut in zip(windows, luts)]


def stitch_labels(labels, group=None, min_overlap=None, overlap=1, atol=LINK_ATOL, rtol=LINK_RTOL):
    """Make the positive label IDs of per-rank windows globally consistent.

    labels: (T_w, H, W) int32 torch tensor of this rank (negative and zero labels are kept).  The last `overlap`
    frames of rank r and the first `overlap` frames of rank r+1 are the same time steps.  Communication:
      1. all_gather of the per-rank label counts (one int64 each),
      2. neighbour exchange: rank r+1 sends the label frames the linking compares (compare_frames) to rank r -- one
         point-to-point message per boundary over a single xGMI link,
      3. all_gather of the (id_left, id_right) pair lists found on each boundary, padded to the longest
         list -- a few KB..MB; every rank then runs the same union-find and rewrites its own labels.
    `min_overlap` (round-1 signature) = absolute criterion only: atol = min_overlap, rtol = 0."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return labels
    if min_overlap is not None:
        atol, rtol = max(int(min_overlap), 1), 0.0
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = labels.device
    if dist.get_backend(group) == "gloo" and labels.is_cuda:
        # gloo has no GPU collectives for these ops: stage the exchanged frames through the host
        return stitch_labels(labels.cpu(), group, None, overlap, atol, rtol).to(dev)
    if labels.shape[0] < overlap:
        raise ValueError("window shorter than the overlap")
    count = torch.clamp(labels.max(), min=0).to(torch.int64).reshape(1)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, count, group=group)
    counts = [int(c.item()) for c in counts]
    # neighbour exchange of the compared frames
    sel = compare_frames(overlap)
    first = labels[:overlap][sel].contiguous()
    right_first = torch.empty_like(first)
    ops = []
    if rank > 0:
        ops.append(dist.P2POp(dist.isend, first, rank - 1, group))
    if rank < world - 1:
        ops.append(dist.P2POp(dist.irecv, right_first, rank + 1, group))
    for req in dist.batch_isend_irecv(ops) if ops else []:
        req.wait()
    # pairs on my right boundary
    if rank < world - 1:
        mine = torch.from_numpy(overlap_pairs(labels[labels.shape[0] - overlap:][sel], right_first, atol, rtol)).to(dev)
    else:
        mine = torch.zeros((0, 2), dtype=torch.int64, device=dev)
    n_mine = torch.tensor([mine.shape[0]], dtype=torch.int64, device=dev)
    n_all = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(n_all, n_mine, group=group)
    n_all = [int(v.item()) for v in n_all]
    width = max(max(n_all), 1)
    padded = torch.zeros((width, 2), dtype=torch.int64, device=dev)
    padded[:mine.shape[0]] = mine
    gathered = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(gathered, padded, group=group)
    pairs = [gathered[r][:n_all[r]].cpu().numpy() for r in range(world - 1)]
    lut = stitch_lut(counts, pairs)[rank]
    return apply_global_lut(labels, lut)


def apply_global_lut(labels, lut):
    """labels -> lut[labels] for the positive ids, zero and negative ids kept.  GPU tensors go through the library's
    one-pass gather (tf_apply_lut: 4 B read + 4 B written per voxel; a boolean-mask update in torch would compact and
    scatter several GB of int64 temporaries for a 12 x 5424^2 window); CPU tensors (gloo rehearsals) use torch."""
    import torch
    lut = np.asarray(lut)
    if not labels.is_cuda:
        lut_t = torch.from_numpy(lut.astype(np.int32))
        pos = labels > 0
        out = labels.clone()
        out[pos] = lut_t[labels[pos].to(torch.int64)]
        return out
    from tobac_flow_amd import _lib
    lab = labels.to(torch.int32).contiguous()
    lut_t = torch.from_numpy(np.ascontiguousarray(lut, np.int32)).to(lab.device)
    out = torch.empty_like(lab)
    _lib.check(_lib.lib().tf_apply_lut(_lib.ptr(lab), lab.numel(), _lib.ptr(lut_t), lut_t.numel(), _lib.ptr(out),
                                       _lib.stream_ptr()), "tf_apply_lut")
    if bool((lab.min() < 0).item()):                 # tf_apply_lut maps ids outside the table to 0: put negatives back
        out = torch.where(lab < 0, lab, out)
    return out
