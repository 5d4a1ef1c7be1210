"""Multi-GPU scheme: time windows are independent units (one window per GPU, no data-path
collective); label IDs are stitched at the end with one all-gather.

This is the MI355X form of the reference's own scale-out: independent windows with overlap frames,
then overlap-based linking of label IDs (/root/reference/tobac_flow/linking.py:49-161 -- overlap
counts in the shared frames, then connected components).  Here consecutive windows share ONE frame
(the last frame of rank r is the first frame of rank r+1); the collective is a
`torch.distributed.all_gather` (RCCL on GPUs, gloo in the CPU tests) of each rank's boundary label
frame and label count, after which every rank runs the same union-find and rewrites its own labels.
"""
import numpy as np


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


def boundary_pairs(left_last, right_first, base, min_overlap=1):
    """(id_left, id_right) pairs of positive labels that coincide in the frame two consecutive windows share, seen in at
    least `min_overlap` pixels; `base` > the largest id of the right window.  Tensors (CPU or GPU), (k, 2) int64 out."""
    import torch
    a, b = left_last.reshape(-1).to(torch.int64), right_first.reshape(-1).to(torch.int64)
    both = (a > 0) & (b > 0)
    key, cnt = torch.unique(a[both] * base + b[both], return_counts=True)
    key = key[cnt >= min_overlap]
    return torch.stack([key // base, key % base], 1)


def stitch_window_list(windows, min_overlap=1):
    """Single-process form of stitch_labels: `windows` is a list of (T_w, H, W) int32 label tensors (CPU or GPU), window
    w + 1 starting with the frame window w ends with (e.g. a 144-frame stack processed as twelve windows on one GPU,
    window_bounds).  Returns the relabelled windows: positive ids made globally consistent (contiguous from 1 in order
    of first appearance), zero and negative ids kept.  Same LUT logic as the distributed version."""
    import torch
    if len(windows) == 0:
        return []
    counts = [int(torch.clamp(w.max(), min=0).item()) if w.numel() else 0 for w in windows]
    pairs = []
    for r in range(len(windows) - 1):
        if windows[r].shape[1:] != windows[r + 1].shape[1:]:
            raise ValueError("windows must share their spatial shape")
        pairs.append(boundary_pairs(windows[r][-1], windows[r + 1][0], counts[r + 1] + 1, min_overlap).cpu().numpy())
    luts = stitch_lut(counts, pairs)
    return [apply_global_lut(w, lut) for w, lut in zip(windows, luts)]


def stitch_labels(labels, group=None, min_overlap=1):
    """Make the positive label IDs of per-rank windows globally consistent.

    labels: (T_w, H, W) int32 torch tensor of this rank (negative and zero labels are kept).
    Rank r's last frame and rank r+1's first frame are the same time step.  Communication:
      1. all_gather of the per-rank label counts (one int64 each),
      2. neighbour exchange: rank r+1 sends its first label frame to rank r (one point-to-point
         message per boundary over a single xGMI link),
      3. all_gather of the (id_left, id_right) pair lists found on each boundary, padded to the longest
         list -- a few KB..MB; every rank then runs the same union-find and rewrites its own labels."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return labels
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = labels.device
    if dist.get_backend(group) == "gloo" and labels.is_cuda:
        # gloo has no GPU collectives for these ops: stage the exchanged frames through the host
        return stitch_labels(labels.cpu(), group, min_overlap).to(dev)
    count = torch.clamp(labels.max(), min=0).to(torch.int64).reshape(1)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, count, group=group)
    counts = [int(c.item()) for c in counts]
    # neighbour exchange of the shared frame
    first = labels[0].contiguous()
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
        mine = boundary_pairs(labels[-1], right_first, counts[rank + 1] + 1, min_overlap)
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
