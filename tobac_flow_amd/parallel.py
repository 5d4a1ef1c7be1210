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


def stitch_lut(counts, pairs_per_boundary):
    """Global relabelling tables from per-rank label counts and per-boundary (id_left, id_right) pairs.

    counts[r] = number of labels (max id) of rank r; pairs_per_boundary[r] = (k, 2) array of label
    pairs that coincide in the frame shared by rank r and rank r+1.  Returns one LUT per rank
    (index = local id, value = global id, contiguous from 1 in order of first appearance over
    (rank, local id)), identical on every rank.

    The stitched objects are the connected components of the pair graph (linking.py:153-161); a component is numbered by
    its smallest member (rank-major, then local id).  Vectorised: every rank runs this after every step, and with eight
    ranks of twelve windows (4 x 10^5 labels, 2 x 10^5 pairs) a Python-loop union-find took 0.6 s of a 4.3 s step."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    n = int(offs[-1])
    us, vs = [], []
    for r, pairs in enumerate(pairs_per_boundary):
        p = np.asarray(pairs, np.int64).reshape(-1, 2)
        if p.size:
            if p.min() < 1 or p[:, 0].max() > counts[r] or p[:, 1].max() > counts[r + 1]:
                raise ValueError("stitch_lut: a pair names a label outside 1 .. count of its window")
            us.append(offs[r] + p[:, 0])
            vs.append(offs[r + 1] + p[:, 1])
    node = np.arange(n + 1, dtype=np.int64)
    if us:
        u, v = np.concatenate(us), np.concatenate(vs)
        n_comp, comp = connected_components(coo_matrix((np.ones(u.size, np.int8), (u, v)), shape=(n + 1, n + 1)), directed=False)
        smallest = np.full(n_comp, n + 1, np.int64)
        np.minimum.at(smallest, comp, node)                  # smallest member of every component
        root = smallest[comp]
    else:
        root = node
    is_root = root == node
    is_root[0] = False                                       # index 0 is "no label"
    new = np.zeros(n + 1, np.int64)
    new[is_root] = np.arange(1, int(is_root.sum()) + 1)      # canonical numbering: by smallest member
    new = new[root]
    return [np.concatenate([[0], new[offs[r] + 1: offs[r + 1] + 1]]) for r in range(len(counts))]


def compare_frames(overlap, short_overlap_ok=False):
    """Which of the `overlap` common frames the linking looks at: all but the first and the last (linking.py:55-56).
    With fewer than three common frames the reference links NOTHING (its `[1:-1]` slice is empty): the same here
    (an empty slice) unless the caller opts in with `short_overlap_ok=True` (extension: one or two shared frames are
    then all used, the rule is otherwise the same)."""
    if overlap > 2:
        return slice(1, overlap - 1)
    return slice(0, overlap) if short_overlap_ok else slice(0, 0)


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


DEFAULT_OVERLAP = 4                    # frames consecutive windows share (scripts/dcc_detect_goes.py:153 n_pad_files; bench.py)


def _rule(min_overlap, overlap, atol, rtol, short_overlap_ok):
    """round-1 signature: `min_overlap` = absolute criterion only (atol = min_overlap, rtol = 0), which is also an
    explicit opt-in to linking on one or two shared frames"""
    if min_overlap is not None:
        atol, rtol, short_overlap_ok = max(int(min_overlap), 1), 0.0, True
    return atol, rtol, compare_frames(overlap, short_overlap_ok)


def _local_pairs(windows, overlap, sel, atol, rtol):
    pairs = []
    for r in range(len(windows) - 1):
        if windows[r].shape[1:] != windows[r + 1].shape[1:]:
            raise ValueError("windows must share their spatial shape")
        if windows[r].shape[0] < overlap or windows[r + 1].shape[0] < overlap:
            raise ValueError("window shorter than the overlap")
        left = windows[r][windows[r].shape[0] - overlap:][sel]
        right = windows[r + 1][:overlap][sel]
        pairs.append(overlap_pairs(left, right, atol, rtol))
    return pairs


def _count(w):
    import torch
    return int(torch.clamp(w.max(), min=0).item()) if w.numel() else 0


def stitch_window_list(windows, min_overlap=None, overlap=DEFAULT_OVERLAP, atol=LINK_ATOL, rtol=LINK_RTOL,
                       short_overlap_ok=False, inplace=False):
    """Single-process form of stitch_labels: `windows` is a list of (T_w, H, W) int32 label tensors (CPU or GPU), the
    first `overlap` frames of window w + 1 being the last `overlap` frames of window w (e.g. a 144-frame stack processed
    as twelve windows on one GPU, window_bounds).  Returns the relabelled windows: positive ids made globally consistent
    (contiguous from 1 in order of first appearance), zero and negative ids kept.  Same rule and LUT logic as the
    distributed version.  `min_overlap` (round-1 signature) = absolute criterion only: atol = min_overlap, rtol = 0."""
    if len(windows) == 0:
        return []
    atol, rtol, sel = _rule(min_overlap, overlap, atol, rtol, short_overlap_ok)
    luts = stitch_lut([_count(w) for w in windows], _local_pairs(windows, overlap, sel, atol, rtol))
    return [apply_global_lut(w, lut, inplace) for w, lut in zip(windows, luts)]


def stitch_rank_windows(windows, group=None, min_overlap=None, overlap=DEFAULT_OVERLAP, atol=LINK_ATOL, rtol=LINK_RTOL,
                        short_overlap_ok=False, _force_collectives=False, inplace=False):
    """Make the positive label IDs of ALL windows of ALL ranks globally consistent.

    windows: this rank's list of (T_w, H, W) int32 label tensors, consecutive windows sharing `overlap` frames; the
    last window of rank r and the first window of rank r + 1 share `overlap` frames too (one long sequence cut into
    world x len(windows) windows, rank-major).  Every rank may hold a different number of windows (>= 1).
    Communication:
      1. all_gather of the per-rank window count, then of the per-window label counts (a few int64),
      2. neighbour exchange: rank r + 1 sends the label frames the linking compares (compare_frames) of its FIRST
         window to rank r -- one point-to-point message per rank boundary over a single xGMI link,
      3. all_gather of the (boundary, id_left, id_right) triples found on each rank's boundaries (its internal ones and
         the one to its right neighbour), padded to the longest list -- a few KB..MB; every rank then runs the same
         union-find and rewrites its own windows (tf_apply_lut).
    Without an initialised process group (or world size 1) this is stitch_window_list (`_force_collectives`: tests run
    the collective path in a one-rank RCCL group, the only kind a one-GPU box can form)."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size(group) == 1 and not _force_collectives):
        return stitch_window_list(windows, min_overlap, overlap, atol, rtol, short_overlap_ok, inplace)
    if len(windows) and dist.get_backend(group) == "gloo" and windows[0].is_cuda:
        # gloo has no GPU collectives for these ops: stage the exchanged frames through the host
        dev = windows[0].device
        out = stitch_rank_windows([w.cpu() for w in windows], group, min_overlap, overlap, atol, rtol, short_overlap_ok, _force_collectives)
        return [w.to(dev) for w in out]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if len(windows):
        dev = windows[0].device
    else:
        dev = torch.device("cpu") if dist.get_backend(group) == "gloo" else torch.device("cuda", torch.cuda.current_device())
    # A rank with a bad input must not raise on its own: the others would wait for it in the first collective for ever
    # (ADVICE r3).  The first exchange is therefore a validity code, and every rank raises the same error together.
    bad = 1 if len(windows) == 0 else (2 if any(w.shape[0] < overlap for w in windows) else
                                       (3 if any(w.shape[1:] != windows[0].shape[1:] for w in windows) else 0))
    codes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(codes, torch.tensor([bad], dtype=torch.int64, device=dev), group=group)
    codes = [int(c.item()) for c in codes]
    if any(codes):
        r_bad = next(r for r, c in enumerate(codes) if c)
        raise ValueError("stitch_rank_windows: rank %d: %s" % (r_bad, {1: "every rank must hold at least one window", 2: "window shorter than the overlap",
                                                                       3: "windows must share their spatial shape"}[codes[r_bad]]))
    atol, rtol, sel = _rule(min_overlap, overlap, atol, rtol, short_overlap_ok)

    def gather_i64(vec):
        """all_gather of variable-length int64 vectors -> list of 1-D CPU tensors (length exchange, then padded payload)"""
        vec = vec.to(dev).to(torch.int64).reshape(-1)
        n = torch.tensor([vec.numel()], dtype=torch.int64, device=dev)
        ns = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(ns, n, group=group)
        ns = [int(v.item()) for v in ns]
        padded = torch.zeros(max(max(ns), 1), dtype=torch.int64, device=dev)
        padded[:vec.numel()] = vec
        got = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(got, padded, group=group)
        return [g[:k].cpu() for g, k in zip(got, ns)]

    counts = gather_i64(torch.tensor([_count(w) for w in windows], dtype=torch.int64))
    n_win = [int(c.numel()) for c in counts]
    first_window = np.concatenate([[0], np.cumsum(n_win)])             # global index of each rank's first window
    # neighbour exchange of the compared frames of my first window
    first = windows[0][:overlap][sel].contiguous()
    right_first = torch.empty_like(first)
    ops = []
    if rank > 0:
        ops.append(dist.P2POp(dist.isend, first, rank - 1, group))
    if rank < world - 1:
        ops.append(dist.P2POp(dist.irecv, right_first, rank + 1, group))
    for req in dist.batch_isend_irecv(ops) if ops else []:
        req.wait()
    mine = _local_pairs(windows, overlap, sel, atol, rtol)             # my internal boundaries ...
    if rank < world - 1:                                               # ... and the one to my right neighbour
        last = windows[-1]
        mine.append(overlap_pairs(last[last.shape[0] - overlap:][sel], right_first, atol, rtol))
    triples = [np.concatenate([np.full((len(p), 1), first_window[rank] + k, np.int64), np.asarray(p, np.int64).reshape(-1, 2)], 1)
               for k, p in enumerate(mine)]
    flat = np.concatenate(triples, 0).reshape(-1) if triples else np.zeros(0, np.int64)
    gathered = gather_i64(torch.from_numpy(flat))
    n_total = int(first_window[-1])
    triples_all = np.concatenate([g.numpy().reshape(-1, 3) for g in gathered], 0) if gathered else np.zeros((0, 3), np.int64)
    triples_all = triples_all[np.argsort(triples_all[:, 0], kind="stable")]
    cuts = np.searchsorted(triples_all[:, 0], np.arange(1, n_total - 1))
    pairs = [p.reshape(-1, 2) for p in np.split(triples_all[:, 1:], cuts)] if n_total > 1 else []
    all_counts = [int(v) for c in counts for v in c.tolist()]
    luts = stitch_lut(all_counts, pairs)
    return [apply_global_lut(w, luts[first_window[rank] + k], inplace) for k, w in enumerate(windows)]


def stitch_labels(labels, group=None, min_overlap=None, overlap=DEFAULT_OVERLAP, atol=LINK_ATOL, rtol=LINK_RTOL,
                  short_overlap_ok=False):
    """One window per rank (stitch_rank_windows with a one-element list).  labels: (T_w, H, W) int32 torch tensor of
    this rank (negative and zero labels are kept); the last `overlap` frames of rank r and the first `overlap` frames
    of rank r + 1 are the same time steps.  `min_overlap` (round-1 signature) = absolute criterion only: atol =
    min_overlap, rtol = 0."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return labels
    return stitch_rank_windows([labels], group, min_overlap, overlap, atol, rtol, short_overlap_ok)[0]


def apply_global_lut(labels, lut, inplace=False):
    """labels -> lut[labels] for the positive ids, zero and negative ids kept.  GPU tensors go through the library's
    one-pass gather (tf_apply_lut: 4 B read + 4 B written per voxel; a boolean-mask update in torch would compact and
    scatter several GB of int64 temporaries for a 12 x 5424^2 window); CPU tensors (gloo rehearsals) use torch.
    inplace: rewrite `labels` itself (an int32 contiguous GPU tensor) instead of returning a new volume -- the twelve windows
    of a 144 x 5424^2 stack are 23 GB, which a caller that is done with the window-local ids need not hold twice."""
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
    out = lab if (inplace and lab is labels) else torch.empty_like(lab)        # (elementwise: output may alias input)
    # ids <= 0 (background seeds -1, unlabelled 0) pass through in the same pass
    _lib.check(_lib.lib().tf_apply_lut_keep_nonpositive(_lib.ptr(lab), lab.numel(), _lib.ptr(lut_t), lut_t.numel(), _lib.ptr(out),
                                                        _lib.stream_ptr()), "tf_apply_lut_keep_nonpositive")
    return out
