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
    got = stitch_labels(torch.from_numpy(local))
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
