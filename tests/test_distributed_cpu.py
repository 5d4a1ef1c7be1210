"""N > 1 path on CPU: two gloo ranks process adjacent time windows of one label volume and stitch
their label IDs with the all-gather of tobac_flow_amd/parallel.py; the result must equal a global
labelling of the whole volume after ID canonicalisation."""
import os
import socket

import numpy as np
import pytest
import scipy.ndimage as ndi


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, volume, bounds, out_dir):
    import torch
    import torch.distributed as dist
    from tobac_flow_amd.parallel import stitch_labels
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a, b = bounds[rank]
    local = ndi.label(volume[a:b] > 0)[0].astype(np.int32)      # window-local IDs
    local[volume[a:b] < 0] = -1                                 # background seeds survive untouched
    got = stitch_labels(torch.from_numpy(local), min_overlap=1, overlap=1)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), got.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _canonical(lab):
    """relabel positive IDs by first occurrence in raster (t, y, x) order"""
    flat = lab.ravel()
    pos = flat > 0
    _, first = np.unique(flat[pos], return_index=True)
    order = flat[pos][np.sort(first)]
    lut = np.zeros(flat.max() + 1, np.int64)
    lut[order] = np.arange(1, order.size + 1)
    out = flat.copy()
    out[pos] = lut[flat[pos]]
    return out.reshape(lab.shape)


@pytest.mark.parametrize("world", [2, 3])
def test_two_rank_label_stitch_equals_global_labelling(tmp_path, world):
    import torch.multiprocessing as mp
    from tobac_flow_amd.parallel import window_bounds
    rng = np.random.default_rng(5)
    T, H, W = 9, 24, 30
    vol = (ndi.gaussian_filter(rng.normal(size=(T, H, W)), (1.5, 2, 2)) > 0.02).astype(np.int32)
    vol[:, :2, :2] = -1
    bounds = window_bounds(T, world)
    mp.spawn(_worker, args=(world, _free_port(), vol, bounds, str(tmp_path)), nprocs=world, join=True)
    merged = np.zeros((T, H, W), np.int64)
    for r, (a, b) in enumerate(bounds):
        part = np.load(tmp_path / f"r{r}.npy")
        overlap = merged[a:b] != 0
        assert np.array_equal(merged[a:b][overlap], part[overlap])      # shared frames agree after stitching
        merged[a:b] = part
    want = ndi.label(vol > 0)[0]
    want[vol < 0] = -1
    assert np.array_equal(_canonical(merged), _canonical(want))


def _ws_worker(rank, world, port, field, markers, bounds, overlap, out_dir):
    """One rank = one time window: its own flood (here the CPU oracle stands in for the HIP flood, which needs a GPU) with
    WINDOW-LOCAL marker ids, then the distributed stitch with the reference's linking rule."""
    import torch
    import torch.distributed as dist
    from oracle import ws_oracle
    from tobac_flow_amd.parallel import stitch_labels
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a, b = bounds[rank]
    local_markers = ndi.label(markers[a:b] > 0)[0].astype(np.int32)
    local_markers[markers[a:b] < 0] = -1
    zero = np.zeros(field[a:b].shape + (2,), np.float32)
    lab = ws_oracle.watershed(zero, zero, field[a:b], local_markers, None, 1)
    got = stitch_labels(torch.from_numpy(lab), overlap=overlap)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), got.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_stitched_watershed_windows_equal_the_whole_volume_flood_away_from_the_seams(tmp_path, world):
    """The production scheme end to end: overlapping time windows flooded independently, label ids
    stitched over gloo by the reference's overlap rule (linking.py:49-161: >= 5 px and >= 0.5, the outer common frames
    dropped).  Every window is then compared with the flood of the WHOLE volume on the frames that are at least two
    steps away from a cut: same partition, and one global id per object across all windows."""
    import torch.multiprocessing as mp
    from oracle import ws_oracle
    from tobac_flow_amd.parallel import window_bounds
    rng = np.random.default_rng(11)
    T, H, W, overlap = 6 * world + 4, 48, 64, 4
    yy, xx = np.mgrid[0:H, 0:W]
    field = np.zeros((T, H, W), np.float32)
    markers = np.zeros((T, H, W), np.int32)
    for k in range(8):                                   # drifting blobs on a 2 x 4 grid of cells (their seeds never meet),
        cy = 12 + 24 * (k // 4) + rng.uniform(-2, 2)     # each seeded in every frame it lives in
        cx = 8 + 16 * (k % 4) + rng.uniform(-2, 2)
        vy, vx = rng.uniform(-0.2, 0.2, 2)
        t0, t1 = (0, T) if k < 3 else sorted(rng.integers(0, T, 2))
        for t in range(t0, max(t1, t0 + 3)):
            if t >= T:
                break
            y, x = cy + vy * t, cx + vx * t
            field[t] -= np.exp(-((yy - y) ** 2 + (xx - x) ** 2) / 24.0).astype(np.float32)
            if 1 <= y < H - 2 and 1 <= x < W - 2:
                markers[t, int(y):int(y) + 2, int(x):int(x) + 2] = 1
    field += ndi.gaussian_filter(rng.normal(size=field.shape), (0, 1, 1)).astype(np.float32) * 0.02
    markers[field > -0.01] = -1                          # background seed
    bounds = window_bounds(T, world, overlap)
    mp.spawn(_ws_worker, args=(world, _free_port(), field, markers, bounds, overlap, str(tmp_path)), nprocs=world, join=True)
    glob_markers = ndi.label(markers > 0)[0].astype(np.int32)
    glob_markers[markers < 0] = -1
    zero = np.zeros(field.shape + (2,), np.float32)
    whole = ws_oracle.watershed(zero, zero, field, glob_markers, None, 1)
    mapping = {}
    for r, (a, b) in enumerate(bounds):
        part = np.load(tmp_path / f"r{r}.npy")
        lo = 0 if r == 0 else 2                          # frames at least two steps from a cut
        hi = (b - a) if r == world - 1 else (b - a) - 2
        got, want = part[lo:hi], whole[a + lo:a + hi]
        assert np.array_equal(got <= 0, want <= 0) and np.array_equal(got[got < 0], want[got < 0])
        pairs = np.unique(np.stack([want[want > 0], got[want > 0]], 1), axis=0)
        for w_id, g_id in pairs:                         # one stitched id per object of the whole-volume flood, everywhere
            assert mapping.setdefault(int(w_id), int(g_id)) == int(g_id), (r, w_id, g_id)
    assert len(set(mapping.values())) == len(mapping)    # ... and different objects keep different ids


def _multi_window_worker(rank, world, port, windows_per_rank, overlap, out_dir):
    """every rank holds SEVERAL consecutive windows of one sequence (bench.py config F under --gpus N)"""
    import torch
    import torch.distributed as dist
    from tobac_flow_amd.parallel import stitch_rank_windows
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    got = stitch_rank_windows([torch.from_numpy(w) for w in windows_per_rank[rank]], overlap=overlap)
    for k, g in enumerate(got):
        np.save(os.path.join(out_dir, f"r{rank}_w{k}.npy"), g.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("split", [(2, 2), (3, 1), (1, 2, 2)])
def test_rank_window_lists_stitch_like_one_process(tmp_path, split):
    """parallel.stitch_rank_windows over gloo (ranks holding 2+2, 3+1 and 1+2+2 windows of ONE sequence) returns exactly
    what stitch_window_list returns for the concatenated window list in one process: same rule (linking.py:49-161), same
    global numbering."""
    import torch
    import torch.multiprocessing as mp
    from tobac_flow_amd.parallel import stitch_window_list, window_bounds
    rng = np.random.default_rng(3)
    overlap, n_windows = 4, sum(split)
    T, H, W = 6 * n_windows + overlap, 32, 40
    truth = ndi.label(ndi.gaussian_filter(rng.normal(size=(T, H, W)), (2.0, 1.5, 1.5)) > 0.03)[0].astype(np.int32)
    truth[:, :2, :2] = -1
    wins = []
    for a, b in window_bounds(T, n_windows, overlap):
        w = truth[a:b].copy()
        ids = np.unique(w[w > 0])
        perm = np.zeros(max(int(w.max()), 0) + 1, np.int32)
        perm[ids] = rng.permutation(len(ids)) + 1               # window-local numbering
        w[w > 0] = perm[w[w > 0]]
        wins.append(w)
    want = [x.numpy() for x in stitch_window_list([torch.from_numpy(w) for w in wins], overlap=overlap)]
    per_rank, k = [], 0
    for n in split:
        per_rank.append(wins[k:k + n])
        k += n
    mp.spawn(_multi_window_worker, args=(len(split), _free_port(), per_rank, overlap, str(tmp_path)), nprocs=len(split), join=True)
    k = 0
    for r, n in enumerate(split):
        for j in range(n):
            assert np.array_equal(np.load(tmp_path / f"r{r}_w{j}.npy"), want[k]), (r, j)
            k += 1
    joined = sum(int(np.unique(w[w > 0]).size) for w in wins) - int(max(w.max() for w in want))
    assert joined > 0                                           # the rule did link objects across boundaries


def _bad_input_worker(rank, world, port, out_dir):
    """rank 1 hands in a window shorter than the overlap: BOTH ranks must raise (a rank that raised on its own
    left the others waiting in the first collective)"""
    import torch
    import torch.distributed as dist
    from tobac_flow_amd.parallel import stitch_rank_windows
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=60))
    wins = [torch.ones((6 if rank == 0 else 2, 8, 8), dtype=torch.int32)]
    try:
        stitch_rank_windows(wins, overlap=4)
        msg = "no error"
    except ValueError as e:
        msg = str(e)
    with open(os.path.join(out_dir, f"r{rank}.txt"), "w") as fh:
        fh.write(msg)
    dist.barrier()
    dist.destroy_process_group()


def test_a_bad_window_on_one_rank_raises_on_every_rank(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_bad_input_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    msgs = [open(tmp_path / f"r{r}.txt").read() for r in range(2)]
    assert msgs[0] == msgs[1] and "rank 1" in msgs[0] and "shorter than the overlap" in msgs[0]
