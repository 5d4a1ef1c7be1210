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
import os

import numpy as np

LINK_ATOL, LINK_RTOL = 5, 0.5          # linking.py:70-76


def window_bounds(T, world, overlap=1):
    """Split T frames into `world` contiguous windows sharing `overlap` frames: [(start, stop), ...]."""
    if world < 1 or T < world + overlap * (world - 1):
        raise ValueError("not enough frames for the requested number of windows")
    body = T - overlap
    edges = [round(i * body / world) for i in range(world + 1)]
    return [(edges[i], edges[i + 1] + overlap) for i in range(world)]


def rank_windows(bounds, rank, world):
    """STRONG sharding of ONE stack over the ranks of a node (BASELINE configs 4 - 5: "288 frames, frame-sharded across 8 x
    MI355X"; the reference's analogue is one window job per process, scripts/dcc_detect_seviri_nat.py:152 +
    scripts/linking_parallel.py:26-27): rank r takes the windows [r W / N, (r + 1) W / N) of `bounds` and with them the frames
    from its first window's start to its last window's stop -- `overlap` of which it shares with each neighbour, exactly
    the frames two consecutive windows share.  Returns (frame_start, frame_stop, bounds relative to frame_start): what
    detect_stack_windows(stack[frame_start:frame_stop], local_bounds, ..., group=...) is called with on that rank; the
    stitch over the ranks (stitch_rank_windows) then equals the stitch of the one-process run over all windows."""
    n = len(bounds)
    if not (0 <= rank < world) or n < world:
        raise ValueError("rank_windows: need 0 <= rank < world <= number of windows")
    edges = [round(i * n / world) for i in range(world + 1)]
    mine = [(int(lo), int(hi)) for lo, hi in bounds[edges[rank]:edges[rank + 1]]]
    start, stop = mine[0][0], mine[-1][1]
    return start, stop, [(lo - start, hi - start) for lo, hi in mine]


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
    first = windows[0][:overlap][sel].contiguous()
    last = windows[-1]
    luts = _stitch_tables([_count(w) for w in windows], _local_pairs(windows, overlap, sel, atol, rtol), first,
                          last[last.shape[0] - overlap:][sel], group, atol, rtol, dev)
    return [apply_global_lut(w, luts[k], inplace) for k, w in enumerate(windows)]


def _stitch_tables(counts_local, internal_pairs, first_head, last_tail, group, atol, rtol, dev):
    """The collective part of stitch_rank_windows on what it needs of this rank's windows -- their label counts, the pairs of its
    INTERNAL boundaries, the compared frames of its first window (`first_head`) and of its last one (`last_tail`) -- so that
    a caller that hands its windows out one by one (detect_stack_windows(on_window=...)) never holds them together.  Returns
    the relabelling table of every local window (index = local id -> global id)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)

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

    counts = gather_i64(torch.tensor(list(counts_local), dtype=torch.int64))
    n_win = [int(c.numel()) for c in counts]
    first_window = np.concatenate([[0], np.cumsum(n_win)])             # global index of each rank's first window
    # neighbour exchange of the compared frames of my first window
    first = first_head.to(dev).contiguous()
    right_first = torch.empty_like(first)
    ops = []
    if rank > 0:
        ops.append(dist.P2POp(dist.isend, first, rank - 1, group))
    if rank < world - 1:
        ops.append(dist.P2POp(dist.irecv, right_first, rank + 1, group))
    for req in dist.batch_isend_irecv(ops) if ops else []:
        req.wait()
    mine = list(internal_pairs)                                        # my internal boundaries ...
    if rank < world - 1:                                               # ... and the one to my right neighbour
        mine.append(overlap_pairs(last_tail.to(dev), right_first, atol, rtol))
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
    return [luts[first_window[rank] + k] for k in range(len(counts_local))]


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


# ---- a stack processed as overlapping time windows on ONE device (round 5: the scheduler bench.py used to carry) --------
# The reference processes a long sequence as independent windows of files with `n_pad_files` shared frames
# (scripts/dcc_detect_goes.py:153), one process per window (scripts/linking_parallel.py:26-27 fans the linking out the same
# way), and links the label ids afterwards (linking.py:49-161).  On one MI355X the whole stack is resident, so:
#   * the flow of the stack's T - 1 frame pairs is computed ONCE (Flow.window_view gives every window the Flow
#     create_flow(window) would return, bit for bit);
#   * a window is BEGUN -- seeds, edge field, the device part of its flood (watershed_begin) -- as soon as the Farneback batch
#     with its last frame pair is enqueued (create_flow(on_frames_ready=...)), and its host replay of the reference heap's
#     order (WatershedJob.replay: sequential host work, 0.2 - 1 s per 16 x 5424^2 window) runs on a worker thread beside the
#     device's next batches;
#   * floods are FINISHED out of order, whichever replay ends first, on a second stream (the main one is busy with the flow);
#   * the label ids of all windows (of all ranks, with a process group) are made consistent by stitch_rank_windows.
# Every window's labels equal Flow.watershed on create_flow(window) -- tests/test_gpu_windows.py compares voxel for voxel.
_REPLAY_POOL = None
_SIDE_STREAMS = {}
_BUDGET_MEMO = {}


def _replay_pool(n):
    global _REPLAY_POOL
    import os
    from concurrent.futures import ThreadPoolExecutor
    if _REPLAY_POOL is None:
        cpus = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 4)
        _REPLAY_POOL = ThreadPoolExecutor(max_workers=max(1, min(12, cpus)), thread_name_prefix="tf-ws-replay")
    return _REPLAY_POOL


def _side_stream():
    """the second stream of (device, calling stream) -- the floods of ready windows are finished on it while the calling
    stream is busy with the flow.  (Round 5 could confine it to k CUs of every XCD or give it the lowest priority: both
    measured slower than a plain stream, profiles/round5_scheduling_experiments.txt, and gone with round 6 --
    tools/experiments/stream_experiments.hip.)"""
    import torch
    key = (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream()
    return _SIDE_STREAMS[key]


class _WindowFloods:
    """The floods of one channel of one stack: begun window by window (each with its seeds and edge field), their host
    replays on worker threads, finished -- out of order -- as soon as their replay has ended."""

    def __init__(self, owner, bt, channel, pieces, n_fly):
        from collections import deque
        self.o, self.bt, self.c, self.pieces, self.n_fly = owner, bt, channel, pieces, n_fly
        self.pending = deque()                               # floods in flight: (job, future, stats, scratch, window index)
        self.wins = [None] * len(owner.bounds)
        self.next = 0                                        # next window to begin
        self.family = None                                   # detect_stack_sequence: the stacks in flight share the flood slots
        self.own_stream = False                              # True: driven from a flood thread on a stream of its own (_ready)
        # hand-out mode (detect_stack_windows(on_window=...)): every window leaves as soon as it is finished, with window-local
        # ids; what stays are its label count and the pairs it forms with its neighbours (computed as soon as both exist)
        self.on_window = None
        n = len(owner.bounds)
        self.handed, self.counts, self.pairs = [None] * n, [0] * n, [None] * max(n - 1, 0)
        self._heads, self._tails, self.first_head, self.last_tail = {}, {}, None, None

    def _hand_out(self, k, lab):
        """the finished window k leaves: its count and the label pairs it forms with windows k - 1 / k + 1 on the frames the
        linking compares (linking.py:55-56; computed now if that neighbour is done, else the compared frames -- two of the
        four shared ones -- are kept until it is) stay; on_window(channel, k, labels) gets the window-local labels"""
        o = self.o
        n, ov, sel = len(o.bounds), o.overlap, o.sel
        self.counts[k] = _count(lab)
        head, tail = lab[:ov][sel], lab[lab.shape[0] - ov:][sel]
        if k > 0:
            if (k - 1) in self._tails:
                self.pairs[k - 1] = overlap_pairs(self._tails.pop(k - 1), head, o.atol, o.rtol)
            else:
                self._heads[k] = head.clone()
        elif o.multi:
            self.first_head = head.clone()
        if k < n - 1:
            if (k + 1) in self._heads:
                self.pairs[k] = overlap_pairs(tail, self._heads.pop(k + 1), o.atol, o.rtol)
            else:
                self._tails[k] = tail.clone()
        elif o.multi:
            self.last_tail = tail.clone()
        self.handed[k] = self.on_window(self.c, k, lab)
        self.wins[k] = None
        o.mark("window %d of channel %d handed out" % (k, self.c))

    def tables(self):
        """hand-out mode, every window finished: the relabelling table of every window (index = window-local id -> id that is
        consistent over all windows, and over all ranks of the group: the collective of stitch_rank_windows on counts, pairs
        and the two outer windows' compared frames)"""
        import torch
        import torch.distributed as dist
        o = self.o
        if o.multi:
            gloo = dist.get_backend(o.group) == "gloo"
            dev = torch.device("cpu") if gloo else torch.device("cuda", torch.cuda.current_device())
            return _stitch_tables(self.counts, self.pairs, self.first_head, self.last_tail, o.group, o.atol, o.rtol, dev)
        return stitch_lut(self.counts, self.pairs)

    def _ready(self, n_frames):
        """may the next window be begun now that the flow of the stack's first n_frames frames is enqueued?  Its last frame
        must be among them -- and, when the windows are driven from a flood thread on a stream of its own, must not be the
        hand-over's last frame unless the stack ends there: create_flow hands over n = pairs_done + 1 frames, of which
        forward[n - 1] is written by the NEXT part of the flow on the calling stream, and window_view saves, patches and
        restores exactly that frame of a window that ends at n -- on the calling stream (between two parts) that is ordered, on
        the flood stream the restore of the stale copy would race with the flow's real write (the frame would then
        hold garbage for the next overlapping window).  Such a window waits for the next hand-over."""
        bounds = self.o.bounds
        if self.next >= len(bounds):
            return False
        hi = bounds[self.next][1]
        if hi > n_frames:
            return False
        return (not self.own_stream) or hi < n_frames or n_frames >= len(self.bt)

    def _finish_any(self, block=True):
        """finish one ready flood -- of this stack, or (detect_stack_sequence) of any stack still in flight"""
        return self.family.finish_any(block) if self.family is not None else self.finish_one(block)

    def _begin(self, flow, w, scratch):
        """seeds -> edge field -> device part of the watershed of this channel over the window `w` of the stack; the host
        replay of the reference's heap order (if this window needs one) starts on a worker thread"""
        from tobac_flow_amd.detection import get_combined_edge_field
        from tobac_flow_amd.watershed import watershed_begin
        o = self.o
        field, seeds = o.seeds_fn(w, self.c)
        # Flow.sobel(uphill, cubic) in float64 + detection.py:638-642, rounded to float32 as watershed.py:64-65 does
        e = get_combined_edge_field(flow, field, dtype=np.float32)
        fw, bw = flow._dev_flows()
        st = {}
        o.mark("begin: seeds + edge field enqueued")
        # (defer_sweeps: the call returns after the set-up and the export -- the parts that read the window's flow fields; phase
        # A and the chain levels are begin_up_to's second pass, when the replays of ALL windows of this hand-over are under way)
        job = watershed_begin(fw, bw, e, seeds, None, o.nbr, o.chain_depth, stats=st, on_ambiguous=o.on_ambiguous, workspace=scratch,
                              defer_sweeps=os.environ.get("TF_WINDOWS_DEFER_SWEEPS", "1") == "1")     # (development switch: 0 = one pass)
        o.mark("begin: set up (replay %s)" % ("submitted" if job.needs_replay else "none"))
        fut = o.pool.submit(job.replay) if job.needs_replay else None
        return job, fut, st, scratch

    def _finish(self, job, fut, st, scratch):
        """root phase (with the pop ranks), labels written: one label volume -- or None if the library had to export for a host
        replay after the root phase (no guessed tie value, or one that was too low): the caller queues the job again"""
        o = self.o
        if fut is not None:
            fut.result()
        o.mark("finish: enter")
        done, lab = job.step(stream=o.side)                  # (WatershedJob.step records the caller's stream on the labels)
        o.mark("finish: %s" % ("done" if done else "exported again, replay pending"))
        if not done:
            return None
        if st.get("reference_order", {}).get("microseconds", 0) > 0:
            d = st["reference_order_detail"]
            o.info["reference_order"].append((st["reference_order"]["microseconds"], d["replay_form"], d["replay_us"], d["export_us"], d["guessed"],
                                              d["guess_covered_the_tie"], st["root_phases"]))
        o.info["floods"].append(st["sweeps"] + [st["chain_depth"], st["ambiguous_pixels"], st["marker_tie_origins"], st["depth_origins"]])
        return lab

    def finish_one(self, block=True):
        """finish a flood whose host replay has ended (the oldest such one); if none has, wait for the first that does: a
        window whose replay takes long -- the dense form, ~1 s -- does not hold up the others.  block=False: only if one is
        ready now (called after every begin: a flood whose guessed tie value turns out too low gets its second export -- and
        with it the start of its long replay -- as early as possible)"""
        from concurrent.futures import FIRST_COMPLETED, wait
        pending = self.pending
        ready = [p for p in pending if p[1] is None or p[1].done()]
        if not ready and not block:
            return False
        if not ready:
            wait([p[1] for p in pending], return_when=FIRST_COMPLETED)
            ready = [p for p in pending if p[1] is None or p[1].done()]
        done = ready[0]
        pending.remove(done)
        lab = self._finish(*done[:4])
        if lab is None:                                      # exported after its root phase: the replay goes to a worker, the job comes back
            pending.append((done[0], self.o.pool.submit(done[0].replay)) + done[2:])
            return True
        self.wins[done[4]] = lab
        self.pieces.append(done[3])
        if self.on_window is not None:
            self._hand_out(done[4], lab)
        return True

    def setup_up_to(self, flow, n_frames, wait_for=None):
        """first pass: set up every window that ends within the first n_frames frames of the stack (their flow is final) --
        seeds, edge field, the flood's set-up and export: the only part that reads the flow fields; the host replays start.
        Returns the jobs for sweep().
        wait_for: an event on the main stream behind the flow these windows need: until it has passed, floods whose replay
        has ended are finished (on the second stream) instead of blocking in the first synchronisation of a begin"""
        import time
        bounds = self.o.bounds
        while wait_for is not None and self._ready(n_frames) and not wait_for.query():
            if not self._finish_any(block=False):
                time.sleep(0.0005)
        begun = []
        while self._ready(n_frames):
            lo, hi = bounds[self.next]
            while not self.pieces:                           # every flood slot is taken (by this stack's floods, or an earlier stack's)
                self._finish_any()                           # (a flood begun in this call is completed there: its sweeps first)
            # the Flow create_flow(bt[lo:hi]) would return, bit for bit: the flow of a frame pair does not depend on the window
            # it is in, only the two end frames of a window are mirrored (flow.py:425-426); window_view patches those two frames
            # in the stack's arrays for the duration of the block instead of copying the window's 7.5 GB of flow vectors.  Only
            # the set-up of the flood reads the flows (its neighbour table has the displacements applied): sweeps and root
            # phase run outside the block.
            with flow.window_view(lo, hi) as flow_w:
                item = self._begin(flow_w, self.bt[lo:hi], self.pieces.pop()) + (self.next,)
            self.pending.append(item)
            begun.append(item[0])
            self.next += 1
        return begun

    def sweep(self, begun):
        """second pass: phase A and the chain levels of the windows set up by setup_up_to -- all their host replays are running
        by now (at the end of a stack, where nothing else is left to do, the last window's replay so starts one window's
        sweeps earlier)"""
        for job in begun:
            job.sweeps()                                     # (a no-op for a job finish_one has completed meanwhile)
            self.o.mark("begin: swept")
        while self._finish_any(block=False):
            pass

    def begin_up_to(self, flow, n_frames, wait_for=None):
        """begin every window that ends within the first n_frames frames of the stack: setup_up_to, then sweep"""
        self.sweep(self.setup_up_to(flow, n_frames, wait_for))

    def finish_all(self):
        while self.pending:
            self.finish_one()
        return self.wins

    def abandon_all(self):
        """failure path: give up every flood still in flight -- wait for its host replay (a worker thread that still reads the
        job), free the job, return its slot and scratch (they used to stay held until garbage collection)"""
        while self.pending:
            job, fut, _, scratch = self.pending.popleft()[:4]
            if fut is not None:
                try:
                    fut.result()
                except BaseException:                        # noqa: BLE001 -- the first failure is the one that is reported
                    pass
            try:
                job.abandon()
            except BaseException:                            # noqa: BLE001
                pass
            self.pieces.append(scratch)


class _StackRun:
    pass


class _ChannelGroup:
    """The window floods of several channels of ONE stack driven together (BASELINE config F3: channels are independent
    detections that share the Flow, detection.py:203-254): a hand-over of frames sets up that window for every channel -- all
    on the one thread and stream that drives the floods, sharing the flood slots -- then sweeps them; whoever waits for a slot
    finishes whichever channel's flood is ready."""

    def __init__(self, floods):
        self.floods = list(floods)
        if len(self.floods) > 1:
            fam = _SequenceFloods(lambda wf: None)           # (nothing is delivered from inside: the caller collects per channel)
            fam.active = list(self.floods)
            for wf in self.floods:
                wf.family = fam

    def begin_up_to(self, flow, n_frames, wait_for=None):
        begun = [(wf, wf.setup_up_to(flow, n_frames, wait_for)) for wf in self.floods]
        for wf, jobs in begun:
            wf.sweep(jobs)

    def finish_all(self):
        for wf in self.floods:
            wf.finish_all()

    def abandon_all(self):
        for wf in self.floods:
            wf.abandon_all()

    def set_own_stream(self, flag):
        for wf in self.floods:
            wf.own_stream = flag


def detect_stack_windows(bt, bounds, seeds_fn, channels=1, consume=None, overlap=DEFAULT_OVERLAP, stitch=True, group=None,
                         model="Farneback", vr_steps=1, smoothing_passes=1, interp_method="cubic", connectivity=1,
                         chain_depth=3, on_ambiguous="reference", max_in_flight=12, stream_windows=True,
                         flow_workspace_gb=None, flood_thread=None, info=None, mark=None, on_window=None):
    """Flow -> edge field -> marker-controlled watershed over a stack processed as overlapping time windows, on this device.

    bt: (T, H, W) float32 device tensor (the stack, resident).  bounds: [(start, stop), ...] consecutive windows sharing
    `overlap` frames (window_bounds).  seeds_fn(window_bt, channel) -> (field, seeds): the caller's recipe for a window --
    `field` the (linearised) field whose uphill Sobel edges are flooded (detection.get_combined_edge_field), `seeds` int32
    markers (-1 = background seed), both device tensors of the window's shape; e.g. the detect_anvils recipe
    (detection.py:547-561).  channels > 1: several detections share the one Flow (BASELINE config F3), processed one after
    the other so that one channel's labels are resident at a time: `consume(channel, windows)` is called with each
    channel's stitched windows and its return value collected (default: the windows themselves).

    Returns (results, info): results[c] = the list of int32 label windows of channel c (ids consistent over all windows, and
    over all ranks of `group` -- stitch_rank_windows -- unless stitch=False), or what `consume` made of them;
    info["floods"] = per flood the library's sweep counts and tie statistics, info["reference_order"] = per flood that
    needed it the host replay's figures, info["floods_in_flight"], info["flow_batches"].

    on_window(channel, k, labels) (round 6) -- HAND-OUT mode, the reference's own product structure (one file of window-local
    labels per window job, scripts/dcc_detect_goes.py:316-330, made consistent afterwards by linking.py:49-161): every window is
    handed out as soon as its flood is finished -- int32 labels with WINDOW-LOCAL ids, on the thread and stream that finished
    it; the callee stores, reduces or downloads them and must not keep the tensor unless it clones it -- and nothing of it
    stays resident but its label count and the pairs it forms with its neighbours.  results[c] is then {"windows": [what
    on_window returned per window], "luts": [per window the table local id -> consistent id (numpy int64; index 0 = 0)]}
    (`tf_apply_lut` / apply_global_lut(labels, lut) gives the stitched window).  `consume` is not called.  With the labels of
    no channel resident, the windows of ALL channels are begun during the flow (config F3 on one device: 3 x 45 GB of labels
    beside 136 GB of flow vectors would not fit).

    Scheduling (no effect on results): stream_windows -- begin a window as soon as its flow is enqueued -- for every channel
    whose labels fit beside the flow (round 6; the others after the stack's flow, their scratch borrowed from the then idle
    Farneback workspace);
    max_in_flight -- floods in flight at most; flood_thread -- drive the windows from a thread of their own on the second stream
    (default since the end of round 5: the calling thread then only enqueues the flow, whose batches no longer wait for the begins
    between them) or from create_flow's callback on the calling thread (False); flow_workspace_gb -- scratch budget of the Farneback batches while floods run
    beside them (default: what the device has left after the flow vectors, the labels and the floods in flight, memoised per
    stack shape so that every call of a sweep batches alike).  Each window's labels are those of
    Flow.watershed(get_combined_edge_field(create_flow(window), field), seeds) bit for bit."""
    import os
    import time
    import torch
    import tobac_flow_amd.flow as tf
    from tobac_flow_amd import _lib
    from tobac_flow_amd.watershed import neighbour_offsets
    if not (isinstance(bt, torch.Tensor) and bt.is_cuda and bt.dim() == 3):
        raise ValueError("detect_stack_windows: bt must be a (T, H, W) tensor on the GPU")
    T, H, W = bt.shape
    bounds = [(int(lo), int(hi)) for lo, hi in bounds]
    if not bounds or any(not (0 <= lo < hi <= T) for lo, hi in bounds) or any(b[1] < a[1] or b[0] < a[0] for a, b in zip(bounds[:-1], bounds[1:])):
        raise ValueError("detect_stack_windows: bounds must be ascending (start, stop) windows inside the stack")
    C = int(channels)
    o = _StackRun()
    o.bounds, o.seeds_fn, o.nbr, o.chain_depth, o.on_ambiguous = bounds, seeds_fn, neighbour_offsets(connectivity), chain_depth, on_ambiguous
    o.info = info if info is not None else {}
    o.info.setdefault("floods", [])
    o.info.setdefault("reference_order", [])
    t_start = time.perf_counter()
    o.mark = (lambda what: mark(what, (time.perf_counter() - t_start) * 1e3)) if mark is not None else (lambda what: None)
    o.pool = _replay_pool(max_in_flight)
    o.side = None
    n_windows = len(bounds)
    longest = max(hi - lo for lo, hi in bounds)
    per_job = 18 * longest * H * W                           # scratch of a flood in flight (~17 B per window voxel)
    import torch.distributed as dist_
    o.overlap, o.group = int(overlap), group
    o.atol, o.rtol, o.sel = LINK_ATOL, LINK_RTOL, compare_frames(int(overlap))
    o.multi = dist_.is_available() and dist_.is_initialized() and dist_.get_world_size(group) > 1
    flow_kw = dict(model=model, vr_steps=vr_steps, smoothing_passes=smoothing_passes, interp_method=interp_method)
    total = torch.cuda.mem_get_info()[1]
    first = None
    sum_win = sum(hi - lo for lo, hi in bounds)
    label_bytes = 4 * sum_win * H * W                        # the windows of ONE channel
    # How many channels have their windows begun DURING the flow: with on_window no labels stay resident -- all of them; else
    # one channel always (round 4/5), and more only if their labels fit beside the flow vectors, the floods in flight and a
    # Farneback scratch worth having (40 GB).  Same labels either way.
    n_fly = int(max(1, min(max_in_flight, n_windows * C, 5)))

    def room_for(cs, slots=None):
        free = torch.cuda.mem_get_info()[0] + (torch.cuda.memory_reserved() - torch.cuda.memory_allocated())
        held = sum(int(v.numel()) for k, v in list(_lib._WS.items()) if v is not None and k[1] == torch.cuda.current_device())
        resident = 0 if on_window is not None else cs * label_bytes
        # several channels: what the single-channel estimate leaves out is no longer small beside what is left -- the raw flow
        # vectors and 8-bit frames of a batch, the refinement's and the labelling's scratch, the flood stream's cached transients
        # (measured on config F3: 25 GB; the device ran out, the allocator flushed its cache -- a 4.4 s hole in the step)
        other = int(0.08 * total) if C > 1 else 0
        return free + held - (2 * T * H * W * 8 + resident + ((n_fly if slots is None else slots) + 1) * (3 * per_job // 2) + other)
    if not (bool(stream_windows) and n_windows > 1):
        Cs = 0
    elif on_window is not None or C == 1:
        Cs = C
    else:
        Cs = max([cs for cs in range(1, C + 1) if room_for(cs) >= 40e9] or [0])
    if C > 1 and Cs > 0:
        # fewer flood slots before a starved flow: the floods run one after the other on the one flood stream anyway, the
        # slots beyond the third only keep more host replays in flight
        n_fly = next((nf for nf in range(n_fly, 2, -1) if room_for(Cs, nf) >= 50e9), min(n_fly, 3))
    stream = Cs > 0
    o.info["channels_begun_during_the_flow"] = Cs
    if not stream and T * H * W * (1 + 4 + C) * 4 > 0.6 * total:
        # a stack that takes most of the device: the flood slots of the previous call go back to the allocator's cache before
        # the flow is sized (held, they cost the Farneback batches a third of their pairs); the floods take them again afterwards
        for k in range(64):
            _lib.release_workspaces("watershed_job%d" % k)
    retries_before = torch.cuda.memory_stats().get("num_alloc_retries", 0)
    budget_key = None
    if stream:
        if flow_workspace_gb is None:
            # floods in flight beside the flow need scratch of their own (the Farneback workspace is busy): the Farneback batches
            # get what is left after the flow vectors, the labels of all windows, the floods (scratch + field + seeds ~ 1.5 x
            # the scratch each, and one window's transients) -- with 30 % of it kept back: free memory that sits in the caching
            # allocator as fragments cannot serve the one block the Farneback scratch is, and a budget the device can only just
            # hold costs allocator retries (measured: 98.8 GB -> batches of 64 pairs, 268 GB at the peak, steps of 4.7 / 8.4 /
            # 6.4 s instead of 4.6).  Memoised per stack shape: the batches of every call of a sweep are then the same, whatever
            # an earlier call left cached.  Config F (144 x 5424^2, 12 windows) on 288 GB: 86 GB -- any budget from 65 to 98 GB
            # gives the library's batch hint 21 full-resolution pairs, i.e. 42-pair batches finished in two parts of 21.
            key = (torch.cuda.current_device(), T, H, W, tuple(bounds), n_fly, Cs, on_window is not None)
            if key not in _BUDGET_MEMO:
                _BUDGET_MEMO[key] = max(4.0, 0.7 * room_for(Cs) / 1e9)
            flow_workspace_gb = _BUDGET_MEMO[key]
            budget_key = key
        slots = [None] * n_fly                               # the flood slots, shared by the channels begun during the flow
        streamed = [_WindowFloods(o, bt, c, slots, n_fly) for c in range(Cs)]
        for wf_ in streamed:
            wf_.on_window = on_window
        first = _ChannelGroup(streamed)
        o.info["floods_in_flight"] = n_fly
        o.side = _side_stream()

        # (measured, config F.  Middle of round 5, six steps each: flood thread 4.48 - 5.11 s per step, mean 4.72; calling thread
        # 4.66 - 4.75, mean 4.71; flood thread on a low-priority stream 4.72 - 4.87.  End of round 5, with the build that no longer
        # pairs float operations (csrc/Makefile), two runs of eight steps each, alternating: flood thread 4.53 / 4.59 s per step
        # (4.44 - 4.70), calling thread 4.62 / 4.65 (4.49 - 4.86) -- the flow's seven parts are enqueued back to back (4.04 s
        # instead of 4.36 s until the last part has passed), the begins run beside them at 70 - 290 ms each instead of 50 ms
        # between them: the flood thread is the default, TF_WINDOWS_THREAD=0 / flood_thread=False the calling thread)
        threaded = flood_thread if flood_thread is not None else os.environ.get("TF_WINDOWS_THREAD", "1") == "1"
        o.info["flood_thread"] = bool(threaded)
        if not threaded:
            # the calling thread begins the windows itself, inside create_flow's callback -- on the calling stream, i.e. BETWEEN the
            # flow's batches (a window's set-up, phase A and chain phases: small, latency-bound launches with a host
            # synchronisation every 32 sweeps); finished floods' root phases run on the second stream
            def frames_ready(fl, n):
                o.mark("flow enqueued for %d frames" % n)
                ev = torch.cuda.Event()
                ev.record()
                first.begin_up_to(fl, n, wait_for=ev)
            worker = None
        else:
            # A FLOOD THREAD with the second stream as its current stream drives every window -- seeds, edge field, begin,
            # finish -- while the calling thread does nothing but enqueue the flow's batches: the floods' latency-bound sweeps
            # run BESIDE the next batches' kernels instead of between them (the library is re-entrant per stream; scratch and
            # memos of the Python layer are keyed by (device, stream)).  The callback hands over (flow, n, event behind the
            # batch); the thread polls the event on the host, finishing ready floods meanwhile, then begins the windows that
            # end within the first n frames.  What it reads of the stack's flow arrays is final by then (the flow of a frame
            # pair is written once, before its event); window_view's two patched end frames lie inside windows of this same
            # thread only, which it handles one after the other.
            import queue
            import threading
            handover = queue.Queue()
            failure = []
            flood_stream, dev_index = o.side, torch.cuda.current_device()
            first.set_own_stream(True)
            o.side = None                                      # (the thread's CURRENT stream is the second stream: floods are finished on it)

            def flood_loop():
                try:
                    torch.cuda.set_device(dev_index)
                    with torch.cuda.stream(flood_stream):
                        while True:
                            item = handover.get()
                            if item is None:
                                break
                            first.begin_up_to(item[0], item[1], wait_for=item[2])
                        first.finish_all()
                        flood_stream.synchronize()
                except BaseException as exc:                    # noqa: BLE001 -- re-raised on the calling thread
                    failure.append(exc)
                    try:
                        flood_stream.synchronize()
                    except BaseException:                       # noqa: BLE001
                        pass
                    first.abandon_all()                        # (slots and scratch of the floods in flight go back now)
                    while True:                                # drain: the calling thread must never block on a full hand-over
                        try:
                            if handover.get_nowait() is None:
                                break
                        except queue.Empty:
                            break

            worker = threading.Thread(target=flood_loop, name="tf-window-floods", daemon=True)
            worker.start()

            def frames_ready(fl, n):
                if failure:                                    # the flood thread has failed: stop the flow here, not after the
                    raise failure[0]                           # rest of the stack has been enqueued
                o.mark("flow enqueued for %d frames" % n)
                if os.environ.get("TF_WINDOWS_MEMDEBUG"):      # development aid: who holds the device's memory at this point
                    ws = sorted(((int(v.numel()), k[0], k[2]) for k, v in list(_lib._WS.items()) if v is not None), reverse=True)
                    print("  mem: allocated %.1f GB, reserved %.1f GB; workspaces %s" % (
                        torch.cuda.memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9,
                        ", ".join("%s@%x %.1f" % (t_, s_ & 0xffff, b_ / 1e9) for b_, t_, s_ in ws[:14])), flush=True)
                ev = torch.cuda.Event()
                ev.record()
                handover.put((fl, n, ev))
        # a batch has twice the pairs at the pyramid levels >= 2 (with half, their launches are one half-empty round of
        # workgroups) and is finished -- finest levels, refinement, smoothing, hand-over of its frames -- in two parts
        try:
            flow_all = tf.create_flow(bt, on_frames_ready=frames_ready, workspace_gb=flow_workspace_gb, split_parts=2, **flow_kw)
        except BaseException:
            if worker is None:                                 # (the calling thread drives the floods: theirs to give up)
                first.abandon_all()
            raise
        finally:
            if worker is not None:
                handover.put(None)
                worker.join()
        if worker is not None and failure:
            raise failure[0]
    else:
        flow_all = tf.create_flow(bt, deferred_check=True, **flow_kw)     # (flow_all.check() below, behind the last flood)
    o.info["flow_workspace_gb"] = None if flow_workspace_gb is None else round(float(flow_workspace_gb), 1)
    o.mark("create_flow returned (device still working)")
    flow_released = False
    if T * H * W * (1 + 4 + C) * 4 > 0.6 * total:
        # a stack whose frames + flow vectors + one channel's labels take most of the device (F3: 34 + 136 + 44 GB):
        # the Farneback scratch goes back to the allocator (create_flow would otherwise keep it for the next call)
        _lib.release_workspaces("farneback")
        flow_released = True
    results = []
    if first is not None:
        first.begin_up_to(flow_all, T)                       # (every window has been begun by the last hand-over: a no-op then)
        first.finish_all()
    for c in range(C):                                       # channels one after the other: their labels leave in that order
        wq = None
        if first is not None and c < Cs:
            wq = first.floods[c]
        else:
            # Every flood in flight owns ~17 B of scratch per window voxel until it is finished.  The Farneback scratch of
            # create_flow (up to 115 GB) is idle from here to the next create_flow: the floods take their scratch from it,
            # piece by piece, instead of allocating another 50 - 100 GB beside it (which the device does not have).
            fb = None if flow_released else _lib.borrow_workspace("farneback")
            if fb is not None and fb.numel() >= per_job:
                n_fly = int(max(1, min(max_in_flight, n_windows, fb.numel() // per_job)))
                piece = fb.numel() // n_fly // 256 * 256
                pieces = [fb[k * piece:(k + 1) * piece] for k in range(n_fly)]
            else:
                # no scratch to borrow (released above): every flood in flight allocates ~8 GB + its field and seeds; the
                # labels of the channel (4 B per window voxel, rewritten in place by the stitch) still have to fit beside them
                free = torch.cuda.mem_get_info()[0] + torch.cuda.memory_reserved() - torch.cuda.memory_allocated()
                held = sum(int(v.numel()) for k, v in list(_lib._WS.items()) if v is not None and k[0].startswith("watershed_job"))
                # (a flood in flight: scratch + field + seeds ~ 1.5 x the scratch; one window's transients: ~1.5 x more)
                room = 0.75 * (free + held - 4 * sum(hi - lo for lo, hi in bounds) * H * W - 3 * per_job // 2)
                n_fly = int(max(1, min(max_in_flight, n_windows, room // (3 * per_job // 2))))
                o.mark("floods in flight: %d (free %.1f GB, flood slots held %.1f GB)" % (n_fly, free / 1e9, held / 1e9))
                pieces = [None] * n_fly
            o.info["floods_in_flight"] = n_fly
            wq = _WindowFloods(o, bt, c, pieces, n_fly)
            wq.on_window = on_window
        wq.begin_up_to(flow_all, T)
        wins = wq.finish_all()
        if on_window is not None:
            # hand-out mode: the windows have left one by one; what is returned are the callee's values and the tables
            results.append({"windows": wq.handed, "luts": wq.tables() if stitch else None})
            o.mark("channel %d: tables" % c)
            del wins, wq
            continue
        if o.info.get("flood_thread"):
            # the flood thread allocated the labels under ITS stream; from here on they are the caller's, used on the caller's
            # stream: tell the caching allocator, or a later allocation of the flood stream could take a block the caller has
            # dropped while kernels of the caller's stream still read it (the same rule as WatershedJob.step(stream=...))
            for w_ in wins:
                if w_ is not None and w_.is_cuda:
                    w_.record_stream(torch.cuda.current_stream())
        o.mark("all windows finished")
        # label ids of all windows (of all ranks) made consistent: pair counting on the GPU, one union-find, one LUT pass
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        if stitch and (len(wins) > 1 or multi):
            wins = stitch_rank_windows(wins, group=group, overlap=overlap, inplace=True)
            o.mark("stitched")
        results.append(wins if consume is None else consume(c, wins))
        del wins, wq
    # the stack's flow was computed asynchronously: by now every window has been finished (host synchronisations behind the
    # flow's last launch) -- a starved chain of the iteration kernel (NaN rows, TF_ESTARVED) is reported here at the latest
    flow_all.check()
    del flow_all
    # The budget above is an estimate.  If the device ran out during this call -- the caching allocator had to hand its cache
    # back to the driver and retry, which synchronises everything: seconds, measured on config F3 (three channels beside 136 GB
    # of flow vectors: a 4.4 s hole in a 14.6 s step) -- the next call of this shape gives the Farneback batches a quarter less.
    retries = torch.cuda.memory_stats().get("num_alloc_retries", 0) - retries_before
    o.info["allocator_retries"] = int(retries)
    if retries > 0 and budget_key is not None:
        _BUDGET_MEMO[budget_key] = max(4.0, 0.75 * _BUDGET_MEMO[budget_key])
    return results, o.info


class _SequenceFloods:
    """The stacks of a detect_stack_sequence call that still have floods in flight, oldest first: they share the flood slots,
    and whoever waits -- for a slot, for a hand-over's event -- finishes whichever flood of whichever stack is ready."""

    def __init__(self, deliver):
        self.active, self.deliver = [], deliver

    def finish_any(self, block=True):
        from concurrent.futures import FIRST_COMPLETED, wait
        while True:
            for wf in list(self.active):
                if wf.pending and wf.finish_one(block=False):
                    self.settle(wf)
                    return True
            if not block:
                return False
            futs = [p[1] for wf in self.active for p in wf.pending if p[1] is not None]
            if not futs:
                return False
            wait(futs, return_when=FIRST_COMPLETED)

    def settle(self, wf):
        """a stack whose every window has its labels: stitch, hand over, forget"""
        if wf in self.active and wf.next == len(wf.o.bounds) and not wf.pending and getattr(wf, "flow_enqueued", False):
            self.active.remove(wf)
            self.deliver(wf)


def detect_stack_sequence(stacks, bounds, seeds_fn, consume=None, overlap=DEFAULT_OVERLAP, stitch=True, group=None,
                          model="Farneback", vr_steps=1, smoothing_passes=1, interp_method="cubic", connectivity=1,
                          chain_depth=3, on_ambiguous="reference", max_in_flight=12, flow_workspace_gb=None, info=None, mark=None):
    """detect_stack_windows(stack, bounds, seeds_fn, ...) for every stack of `stacks` (an iterable of (T, H, W) float32 device
    tensors of ONE shape: the days of a sweep), one channel, windows begun during the flow -- with the END of a stack run beside
    the flow of the next one: a stack's last windows are set up (the only part of a flood that reads the flow fields) as soon
    as its last frames' flow has passed, the calling thread then releases that stack's flow vectors and enqueues the next
    stack's flow at once, and the flood thread finishes the old stack's sweeps, host replays, root phases and its stitch beside
    it.  Alone, the end of a stack is 0.25 - 0.6 s in which the device waits for a sequential host replay.
    Every window's labels are those of the per-stack call, bit for bit (tests/test_gpu_windows.py).

    consume(k, windows): called ON THE FLOOD THREAD (its stream current) when stack k is complete -- stitched, unless
    stitch=False -- and its return value collected (default: the windows themselves; a sweep that keeps every stack's label
    windows resident runs out of memory: reduce or store them there).  Returns (results, info) like detect_stack_windows.
    With a process group every rank must pass the same number of stacks (the stitch of stack k is a collective)."""
    import os
    import queue
    import threading
    import time
    import torch
    import tobac_flow_amd.flow as tf
    from tobac_flow_amd.watershed import neighbour_offsets
    bounds = [(int(lo), int(hi)) for lo, hi in bounds]
    if len(bounds) < 2 or any(b[1] < a[1] or b[0] < a[0] for a, b in zip(bounds[:-1], bounds[1:])):
        raise ValueError("detect_stack_sequence: bounds must be two or more ascending (start, stop) windows")
    info = info if info is not None else {}
    info.setdefault("floods", [])
    info.setdefault("reference_order", [])
    t_start = time.perf_counter()
    marker = (lambda what: mark(what, (time.perf_counter() - t_start) * 1e3)) if mark is not None else (lambda what: None)
    n_windows = len(bounds)
    n_fly = int(max(1, min(max_in_flight, n_windows, int(os.environ.get("TF_WINDOWS_SLOTS", "5")))))      # (development switch: flood slots)
    pieces = [None] * n_fly                                  # the flood slots, shared by the stacks in flight
    pool = _replay_pool(max_in_flight)
    nbr = neighbour_offsets(connectivity)
    flow_kw = dict(model=model, vr_steps=vr_steps, smoothing_passes=smoothing_passes, interp_method=interp_method)
    results, failure = {}, []
    handover = queue.Queue()
    dev_index = torch.cuda.current_device()
    main_stream = torch.cuda.current_stream()
    flood_stream = _side_stream()
    info["floods_in_flight"], info["flood_thread"], info["stacks_pipelined"] = n_fly, True, True
    info["channels_begun_during_the_flow"] = 1

    def deliver(wf):
        wins = wf.wins
        wf.o.mark("stack %d: all windows finished" % wf.index)
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        if stitch and (len(wins) > 1 or multi):
            wins = stitch_rank_windows(wins, group=group, overlap=overlap, inplace=True)
            wf.o.mark("stack %d: stitched" % wf.index)
        for w_ in wins:                                      # (allocated under the flood stream; the caller's stream may use them afterwards)
            if w_ is not None and w_.is_cuda:
                w_.record_stream(main_stream)
        results[wf.index] = wins if consume is None else consume(wf.index, wins)
        wf.wins = None
        wf.delivered.set()

    fam = _SequenceFloods(deliver)

    def flood_loop():
        try:
            torch.cuda.set_device(dev_index)
            with torch.cuda.stream(flood_stream):
                while True:
                    try:
                        item = handover.get(timeout=0.0005) if fam.active else handover.get()
                    except queue.Empty:
                        fam.finish_any(block=False)
                        continue
                    if item is None:
                        break
                    what, wf = item[0], item[1]
                    if what == "new":
                        fam.active.append(wf)
                    elif what == "frames":
                        begun = wf.setup_up_to(item[2], item[3], wait_for=item[4])
                        item = None                              # (the hand-over holds the stack's Flow: 68 GB at config F)
                        if wf.next == n_windows and not wf.setups_done.is_set():
                            # the stack's last window is set up: nothing reads its flow vectors any more (the sweeps that follow
                            # work on the compact graph) -- the calling thread releases them and enqueues the next stack's flow
                            flood_stream.synchronize()           # (window_view's restored frames have landed)
                            wf.setups_done.set()
                        wf.sweep(begun)
                        begun = None
                    elif what == "end":
                        wf.flow_enqueued = True
                        fam.settle(wf)
                    item = wf = None                             # (a hand-over holds the stack's Flow: 68 GB at config F that must be free for the next stack)
                while fam.active:                                # the last stack's end: nothing left to run beside it
                    if not fam.finish_any(block=True):
                        for wf in list(fam.active):
                            fam.settle(wf)
                        if fam.active:
                            raise RuntimeError("detect_stack_sequence: a stack is left with windows that were never begun")
                flood_stream.synchronize()
        except BaseException as exc:                             # noqa: BLE001 -- re-raised on the calling thread
            failure.append(exc)
            try:
                flood_stream.synchronize()
            except BaseException:                                # noqa: BLE001
                pass
            for wf_ in list(fam.active):                         # (slots and scratch of the floods in flight go back now)
                wf_.abandon_all()
            while True:                                          # drain: the calling thread must never block on the hand-over
                try:
                    if handover.get(timeout=0.05) is None:
                        break
                except queue.Empty:
                    if stop.is_set():
                        break

    stop = threading.Event()
    worker = threading.Thread(target=flood_loop, name="tf-window-floods", daemon=True)
    worker.start()
    n_stacks = 0
    try:
        for k, bt in enumerate(stacks):
            if not (isinstance(bt, torch.Tensor) and bt.is_cuda and bt.dim() == 3):
                raise ValueError("detect_stack_sequence: every stack must be a (T, H, W) tensor on the GPU")
            T, H, W = bt.shape
            if any(not (0 <= lo < hi <= T) for lo, hi in bounds) or bounds[-1][1] != T:
                raise ValueError("detect_stack_sequence: bounds must lie inside every stack and end with it")
            o = _StackRun()
            o.bounds, o.seeds_fn, o.nbr, o.chain_depth, o.on_ambiguous = bounds, seeds_fn, nbr, chain_depth, on_ambiguous
            o.info, o.pool, o.side, o.mark = info, pool, None, (lambda what, k_=k: marker("[stack %d] %s" % (k_, what)))
            if flow_workspace_gb is None:
                longest = max(hi - lo for lo, hi in bounds)
                per_job = 18 * longest * H * W
                key = (torch.cuda.current_device(), T, H, W, tuple(bounds), n_fly)
                if key not in _BUDGET_MEMO:
                    free = torch.cuda.mem_get_info()[0] + (torch.cuda.memory_reserved() - torch.cuda.memory_allocated())
                    from tobac_flow_amd import _lib
                    held = sum(int(v.numel()) for kk, v in list(_lib._WS.items()) if v is not None and kk[1] == torch.cuda.current_device())
                    need = 2 * T * H * W * 8 + 4 * sum(hi - lo for lo, hi in bounds) * H * W + (n_fly + 1) * (3 * per_job // 2)
                    _BUDGET_MEMO[key] = max(4.0, 0.7 * (free + held - need) / 1e9)
                flow_workspace_gb = _BUDGET_MEMO[key]
                info["flow_workspace_gb"] = round(float(flow_workspace_gb), 1)
            wf = _WindowFloods(o, bt, 0, pieces, n_fly)
            wf.family, wf.index, wf.flow_enqueued, wf.own_stream = fam, k, False, True
            wf.setups_done, wf.delivered = threading.Event(), threading.Event()
            handover.put(("new", wf))

            def frames_ready(fl, n, wf_=wf, o_=o):
                if failure:                                      # (stop this stack's flow where the flood thread failed)
                    raise failure[0]
                o_.mark("flow enqueued for %d frames" % n)
                ev = torch.cuda.Event()
                ev.record()
                handover.put(("frames", wf_, fl, n, ev))
            flow = tf.create_flow(bt, on_frames_ready=frames_ready, workspace_gb=flow_workspace_gb, split_parts=2, **flow_kw)
            handover.put(("end", wf))
            n_stacks += 1
            # this stack's flow vectors are read until its last window has been set up: then they go, and the next stack's flow
            # is enqueued while the flood thread is still sweeping, replaying and stitching this one
            while not wf.setups_done.wait(0.002):
                if failure:
                    raise failure[0]
            flow.check()                                         # (a starved chain of the iteration kernel is reported per stack)
            del flow
    finally:
        stop.set()
        handover.put(None)
        worker.join()
    if failure:
        raise failure[0]
    return [results[k] for k in range(n_stacks)], info
