"""tobac_flow_amd.parallel.detect_stack_windows -- a stack processed as overlapping time windows on one device: the flow of
the stack once, windows begun from create_flow's callback while the later frames' flow is computed, floods in parts with
their host replays on worker threads, finished out of order on a second stream, label ids stitched in place (round 5: the
scheduler bench.py used to carry; reference: scripts/dcc_detect_goes.py:153, scripts/linking_parallel.py:26-27,
linking.py:49-161).  Compared VOXEL FOR VOXEL with the same windows processed by the plain calls one after the other
(create_flow(window) -> seeds -> get_combined_edge_field -> Flow.watershed -> stitch_window_list): streamed and not, in one
process and as two gloo ranks on one device (the count comparison accepted a label volume with the right
count and wrong voxels)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

T, H, W, N_WIN, OVERLAP = 44, 1500, 2500, 4, 4


def _seeds(w, c):
    from tools.synth import anvil_seeds
    return anvil_seeds(w)


def _serial_windows(bt, bounds, overlap, per_window_flow=False):
    """the plain calls, one window after the other; per_window_flow: create_flow on the window itself (what the reference's
    scripts do per file group) instead of Flow.window of the stack's flow -- the same vectors bit for bit"""
    import tobac_flow_amd.flow as tf
    from tobac_flow_amd.detection import get_combined_edge_field
    from tobac_flow_amd.parallel import stitch_window_list
    kw = dict(model="Farneback", vr_steps=1, smoothing_passes=1, interp_method="cubic")
    flow = None if per_window_flow else tf.create_flow(bt, **kw)
    labs = []
    for a, b in bounds:
        fl = tf.create_flow(bt[a:b], **kw) if per_window_flow else flow.window(a, b)
        lin, seeds = _seeds(bt[a:b], 0)
        e = get_combined_edge_field(fl, lin, dtype=np.float32)
        labs.append(fl.watershed(e, seeds, connectivity=1))
    return stitch_window_list(labs, overlap=overlap)


def test_detect_stack_windows_equals_the_serial_plain_calls_voxel_for_voxel():
    import torch
    from tobac_flow_amd.parallel import detect_stack_windows, window_bounds
    from tools.synth import blob_stack
    bt = blob_stack(T, H, W, seed=20240601, t0=0)
    bounds = window_bounds(T, N_WIN, OVERLAP)
    want = _serial_windows(bt, bounds, OVERLAP)
    assert int(max(int(w.max()) for w in want)) > 10
    # the first window through create_flow on the window itself: the reference's own per-window call
    first = _serial_windows(bt, bounds[:1], OVERLAP, per_window_flow=True)[0]
    assert torch.equal(first > 0, want[0] > 0)
    # streamed with the windows driven by a thread of their own on the second stream (the default), streamed from create_flow's
    # callback on the calling thread, and all windows after the stack's flow
    for stream, thread in ((True, True), (True, False), (False, None)):
        info = {}
        (got,), info = detect_stack_windows(bt, bounds, _seeds, overlap=OVERLAP, stream_windows=stream, flood_thread=thread, info=info)
        assert len(got) == N_WIN and len(info["floods"]) == N_WIN
        for k, (g, w) in enumerate(zip(got, want)):
            assert g.dtype == torch.int32 and g.shape == w.shape
            assert torch.equal(g, w), (stream, thread, k, int((g != w).sum()))
        assert (info.get("flow_workspace_gb") is not None) == stream and info.get("flood_thread", None) == (thread if stream else None)
        del got
    # an exception on the flood thread (here: the caller's seeds_fn) reaches the caller
    def bad_seeds(w, c):
        raise RuntimeError("seeds_fn failed")
    with pytest.raises(RuntimeError, match="seeds_fn failed"):
        detect_stack_windows(bt, bounds, bad_seeds, overlap=OVERLAP)
    # unstitched: every window's labels are those of the plain call on that window (window-local ids)
    (raw,), _ = detect_stack_windows(bt, bounds, _seeds, overlap=OVERLAP, stitch=False)
    import tobac_flow_amd.flow as tf
    from tobac_flow_amd.detection import get_combined_edge_field
    fl = tf.create_flow(bt[bounds[2][0]:bounds[2][1]], model="Farneback", vr_steps=1, smoothing_passes=1, interp_method="cubic")
    lin, seeds = _seeds(bt[bounds[2][0]:bounds[2][1]], 0)
    assert torch.equal(raw[2], fl.watershed(get_combined_edge_field(fl, lin, dtype=np.float32), seeds, connectivity=1))


def test_detect_stack_windows_validates_its_input():
    import torch
    from tobac_flow_amd.parallel import detect_stack_windows
    bt = torch.zeros((6, 32, 32), device="cuda")
    with pytest.raises(ValueError, match="bounds"):
        detect_stack_windows(bt, [(0, 4), (2, 9)], _seeds)
    with pytest.raises(ValueError, match="bounds"):
        detect_stack_windows(bt, [], _seeds)
    with pytest.raises(ValueError, match="GPU"):
        detect_stack_windows(bt.cpu(), [(0, 6)], _seeds)


def _rank_worker(rank, world, port, out_dir, stream):
    import torch
    import torch.distributed as dist
    from tobac_flow_amd.parallel import detect_stack_windows, window_bounds
    from tools.synth import blob_stack
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t_rank = 24                                                     # rank r holds frames 20 r .. 20 r + 23: two windows of 14
    bt = blob_stack(t_rank, H, W, seed=20240601, t0=rank * (t_rank - OVERLAP))
    (got,), _ = detect_stack_windows(bt, window_bounds(t_rank, 2, OVERLAP), _seeds, overlap=OVERLAP, stream_windows=stream)
    for j, g in enumerate(got):
        np.save(os.path.join(out_dir, f"r{rank}_w{j}.npy"), g.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("stream", [True, False])
def test_two_gloo_ranks_on_one_device_equal_the_one_process_run_voxel_for_voxel(tmp_path, stream):
    """rank r holds frames 20 r .. 20 r + 23 of one sequence as two 14-frame windows (consecutive ranks share four frames);
    one process holding frames 0 .. 43 as the same four windows: every window's labels -- ids consistent over both ranks after
    stitch_rank_windows -- equal, voxel for voxel."""
    import socket
    import torch
    import torch.multiprocessing as mp
    from tobac_flow_amd.parallel import window_bounds
    from tools.synth import blob_stack
    bt = blob_stack(T, H, W, seed=20240601, t0=0)
    bounds = window_bounds(T, N_WIN, OVERLAP)
    assert bounds == [(0, 14), (10, 24), (20, 34), (30, 44)]
    want = [w.cpu().numpy() for w in _serial_windows(bt, bounds, OVERLAP)]
    del bt
    torch.cuda.empty_cache()
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    mp.spawn(_rank_worker, args=(2, port, str(tmp_path), stream), nprocs=2, join=True)
    for r in range(2):
        for j in range(2):
            got = np.load(tmp_path / f"r{r}_w{j}.npy")
            assert np.array_equal(got, want[2 * r + j]), (r, j, int((got != want[2 * r + j]).sum()))


def test_several_channels_share_the_flow_and_equal_the_plain_calls():
    """BASELINE config F3's shape of the problem at a small size: several detections (channels: the same stack at offsets 0 / -2
    K) share ONE Flow.  Round 6: the windows of EVERY channel whose labels fit are begun during the flow, on the one flood
    thread and stream (or from the calling thread), sharing the flood slots; `consume` is called per channel, in order.  In
    hand-out mode (on_window) every window leaves as soon as it is finished, with window-local ids, and the call returns the
    relabelling tables: tables applied to the handed-out windows == the stitched windows.  Every channel's stitched windows
    equal the serial plain calls on that channel, voxel for voxel, in every form."""
    import torch
    import tobac_flow_amd.flow as tf
    from tobac_flow_amd.detection import get_combined_edge_field
    from tobac_flow_amd.parallel import apply_global_lut, detect_stack_windows, stitch_window_list, window_bounds
    from tools.synth import anvil_seeds, blob_stack
    t_, h_, w_ = 38, 400, 600
    bt = blob_stack(t_, h_, w_, seed=20240601, t0=5)
    bounds = window_bounds(t_, 3, 4)
    offsets = (0.0, -2.0)

    def seeds_fn(w, c):
        return anvil_seeds(w + offsets[c] if c else w)
    flow = tf.create_flow(bt, model="Farneback", vr_steps=1, smoothing_passes=1, interp_method="cubic")
    want = []
    for c in range(2):
        labs = []
        for a, b in bounds:
            fl = flow.window(a, b)
            lin, seeds = seeds_fn(bt[a:b], c)
            labs.append(fl.watershed(get_combined_edge_field(fl, lin, dtype=np.float32), seeds, connectivity=1))
        want.append(stitch_window_list(labs, overlap=4))
    del flow
    assert not torch.equal(want[0][0], want[1][0])                      # the channels do differ
    for stream, thread in ((True, True), (True, False), (False, None)):
        seen = []

        def consume(c, wins):
            seen.append(c)
            return [w.clone() for w in wins]
        got, info = detect_stack_windows(bt, bounds, seeds_fn, channels=2, consume=consume, overlap=4, stream_windows=stream, flood_thread=thread)
        assert seen == [0, 1] and len(got) == 2 and info["floods_in_flight"] >= 1
        assert info["channels_begun_during_the_flow"] == (2 if stream else 0), info          # (a small stack: both channels fit)
        for c in range(2):
            for k, (g, w) in enumerate(zip(got[c], want[c])):
                assert torch.equal(g, w), (stream, thread, c, k, int((g != w).sum()))
        # hand-out mode: window-local labels leave one by one, the tables make them consistent
        handed = {}

        def on_window(c, k, lab):
            assert lab.dtype == torch.int32 and (c, k) not in handed
            handed[(c, k)] = lab.clone()
            return int(lab.max())
        res, info = detect_stack_windows(bt, bounds, seeds_fn, channels=2, on_window=on_window, overlap=4, stream_windows=stream, flood_thread=thread)
        assert len(handed) == 6 and info["channels_begun_during_the_flow"] == (2 if stream else 0)
        for c in range(2):
            assert set(res[c]) == {"windows", "luts"} and len(res[c]["luts"]) == 3
            assert res[c]["windows"] == [int(handed[(c, k)].max()) for k in range(3)]
            for k in range(3):
                g = apply_global_lut(handed[(c, k)], res[c]["luts"][k])
                assert torch.equal(g, want[c][k]), ("hand-out", stream, thread, c, k, int((g != want[c][k]).sum()))


def test_a_sequence_of_stacks_equals_the_per_stack_calls_voxel_for_voxel():
    """detect_stack_sequence (round 5): three stacks of one sequence processed back to back, the end of a stack -- its last
    windows' sweeps, host replays, root phases, the stitch -- beside the next stack's flow.  Every stack's stitched windows equal
    those of detect_stack_windows on that stack alone, voxel for voxel; `consume` is called once per stack, in order; an
    exception of the caller's seeds_fn on the flood thread reaches the caller."""
    import torch
    from tobac_flow_amd.parallel import detect_stack_sequence, detect_stack_windows, window_bounds
    from tools.synth import blob_stack
    t_, h_, w_, n_win = 34, 700, 900, 3
    all_bt = blob_stack(t_ + 2 * 5, h_, w_, seed=20240601, t0=3)
    stacks = [all_bt[5 * k:5 * k + t_] for k in range(3)]
    bounds = window_bounds(t_, n_win, OVERLAP)
    want = []
    for bt in stacks:
        (w,), _ = detect_stack_windows(bt, bounds, _seeds, overlap=OVERLAP)
        want.append([x.clone() for x in w])
        del w
    assert int(max(int(x.max()) for x in want[0])) > 3
    seen = []

    def consume(k, wins):
        seen.append(k)
        return [x.clone() for x in wins]
    info = {}
    got, info = detect_stack_sequence(iter(stacks), bounds, _seeds, consume=consume, overlap=OVERLAP, info=info)
    assert seen == [0, 1, 2] and len(got) == 3 and info["stacks_pipelined"] and len(info["floods"]) == 3 * n_win
    for k in range(3):
        for j, (g, w) in enumerate(zip(got[k], want[k])):
            assert g.dtype == torch.int32 and torch.equal(g, w), (k, j, int((g != w).sum()))
    assert not torch.equal(got[0][0], got[1][0])                        # (the stacks do differ)
    # default consume: the windows themselves
    got2, _ = detect_stack_sequence(stacks[:2], bounds, _seeds, overlap=OVERLAP)
    assert all(torch.equal(g, w) for k in range(2) for g, w in zip(got2[k], want[k]))

    def bad_seeds(w, c):
        raise RuntimeError("seeds_fn failed")
    with pytest.raises(RuntimeError, match="seeds_fn failed"):
        detect_stack_sequence(stacks, bounds, bad_seeds, overlap=OVERLAP)
    with pytest.raises(ValueError, match="bounds"):
        detect_stack_sequence(stacks, [(0, t_)], _seeds)

    def bad_consume(k, wins):
        if k == 1:
            raise KeyError("consume failed")
        return None
    with pytest.raises(KeyError, match="consume failed"):                # (raised on the flood thread while stack 2's flow is enqueued)
        detect_stack_sequence(stacks, bounds, _seeds, consume=bad_consume, overlap=OVERLAP)
    (again,), _ = detect_stack_windows(stacks[0], bounds, _seeds, overlap=OVERLAP)      # the library is usable afterwards
    assert all(torch.equal(g, w) for g, w in zip(again, want[0]))


def test_a_window_that_ends_at_a_hand_over_waits_for_the_next_one_on_the_flood_thread(monkeypatch):
    """ADVICE r5 (high).  create_flow hands over n = pairs_done + 1 frames; forward[n - 1] is written by the flow's next part.
    A window that ends exactly at n has that frame saved / patched / restored by Flow.window_view: ordered on the calling
    stream, a race on the flood thread's stream (the stale copy restored over the flow's real write -> the next overlapping
    window reads garbage as an interior frame).  T = 30 in one forced-split batch hands over 16 then 30 frames; bounds
    [(0, 16), (12, 30)] put a window end on the first hand-over; a slow seeds_fn lets the flow run ahead of the flood thread.
    flood thread == calling thread == the serial plain calls, voxel for voxel."""
    import time
    import torch
    from tobac_flow_amd.parallel import detect_stack_windows
    from tools.synth import blob_stack
    monkeypatch.setenv("TF_FLOW_SPLIT_FORCE", "1")
    t_, h_, w_ = 30, 500, 700
    bt = blob_stack(t_, h_, w_, seed=20240601, t0=2)
    bounds = [(0, 16), (12, 30)]
    want = _serial_windows(bt, bounds, OVERLAP)

    def slow_seeds(w, c):
        time.sleep(0.4)
        return _seeds(w, c)
    marks = []
    (on_caller,), _ = detect_stack_windows(bt, bounds, slow_seeds, overlap=OVERLAP, flood_thread=False, mark=lambda what, ms: marks.append(what))
    assert "flow enqueued for 16 frames" in marks, marks        # the window end does coincide with a hand-over
    for k in range(2):
        assert torch.equal(on_caller[k], want[k]), ("calling thread", k, int((on_caller[k] != want[k]).sum()))
    for attempt in range(3):
        marks = []
        (on_thread,), info = detect_stack_windows(bt, bounds, slow_seeds, overlap=OVERLAP, flood_thread=True, mark=lambda what, ms: marks.append(what))
        assert info["flood_thread"] is True
        for k in range(2):
            assert torch.equal(on_thread[k], want[k]), ("flood thread", attempt, k, int((on_thread[k] != want[k]).sum()))
        # window 0 was begun only after the whole stack's flow had been handed over
        first_setup = next(i for i, m in enumerate(marks) if m.startswith("begin:"))
        assert marks.index("flow enqueued for 30 frames") < first_setup, marks
        del on_thread


def _strong_worker(rank, world, port, out_dir, hand_out):
    import torch
    import torch.distributed as dist
    from tobac_flow_amd.parallel import detect_stack_windows, rank_windows, window_bounds
    from tools.synth import anvil_seeds, blob_stack
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t_all, h_, w_ = 54, 700, 900
    lo, hi, local = rank_windows(window_bounds(t_all, 5, OVERLAP), rank, world)
    bt = blob_stack(hi - lo, h_, w_, seed=20240601, t0=lo)              # this rank's frames of the ONE stack

    def seeds_fn(w, c):
        return anvil_seeds(w - 2.0 * c if c else w)
    if hand_out:
        kept = {}

        def on_window(c, k, lab):
            kept[(c, k)] = lab.cpu().numpy()
        res, info = detect_stack_windows(bt, local, seeds_fn, channels=2, overlap=OVERLAP, on_window=on_window)
        assert info["channels_begun_during_the_flow"] == 2
        for c in range(2):
            for k in range(len(local)):
                lut = np.asarray(res[c]["luts"][k])
                lab = kept[(c, k)]
                np.save(os.path.join(out_dir, f"r{rank}_c{c}_w{k}.npy"), np.where(lab > 0, lut[np.maximum(lab, 0)], lab).astype(np.int32))
    else:
        got, _ = detect_stack_windows(bt, local, seeds_fn, channels=2, overlap=OVERLAP)
        for c in range(2):
            for k, g in enumerate(got[c]):
                np.save(os.path.join(out_dir, f"r{rank}_c{c}_w{k}.npy"), g.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("hand_out", [False, True])
def test_strong_sharding_two_ranks_on_one_device_equal_the_one_process_run_voxel_for_voxel(tmp_path, hand_out):
    """ONE 54-frame stack of five windows, two channels; rank 0 takes windows 0 - 1 (round(5 / 2) = 2), rank 1
    windows 2 - 4, each the frames its windows cover (parallel.rank_windows); stitched over both ranks (stitched windows, or
    hand-out mode's tables applied to the window-local labels) == the one-process run over all five windows, voxel for
    voxel, both channels."""
    import socket
    import torch
    import torch.multiprocessing as mp
    from tobac_flow_amd.parallel import detect_stack_windows, rank_windows, window_bounds
    from tools.synth import anvil_seeds, blob_stack
    t_all, h_, w_ = 54, 700, 900
    bounds = window_bounds(t_all, 5, OVERLAP)
    shares = [rank_windows(bounds, r, 2) for r in range(2)]
    assert [len(s[2]) for s in shares] == [2, 3] and shares[0][0] == 0 and shares[1][1] == t_all
    assert shares[0][1] - shares[1][0] == OVERLAP                       # consecutive ranks share the frames two windows share
    bt = blob_stack(t_all, h_, w_, seed=20240601, t0=0)

    def seeds_fn(w, c):
        return anvil_seeds(w - 2.0 * c if c else w)
    got, _ = detect_stack_windows(bt, bounds, seeds_fn, channels=2, overlap=OVERLAP)
    want = [[w.cpu().numpy() for w in got[c]] for c in range(2)]
    assert max(int(w.max()) for w in want[0]) > 5
    del got, bt
    torch.cuda.empty_cache()
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    mp.spawn(_strong_worker, args=(2, port, str(tmp_path), hand_out), nprocs=2, join=True)
    for c in range(2):
        k_glob = 0
        for r in range(2):
            for k in range(len(shares[r][2])):
                have = np.load(tmp_path / f"r{r}_c{c}_w{k}.npy")
                assert np.array_equal(have, want[c][k_glob]), (hand_out, c, r, k, int((have != want[c][k_glob]).sum()))
                k_glob += 1


def test_hand_out_mode_with_one_window_and_without_stitching():
    """on_window with a single window (no pair to form: the table is the identity on 1 .. count) and with stitch=False (no tables)"""
    import torch
    import tobac_flow_amd.flow as tf
    from tobac_flow_amd.detection import get_combined_edge_field
    from tobac_flow_amd.parallel import apply_global_lut, detect_stack_windows, window_bounds
    from tools.synth import anvil_seeds, blob_stack
    bt = blob_stack(14, 400, 600, seed=20240601, t0=5)
    fl = tf.create_flow(bt, model="Farneback", vr_steps=1, smoothing_passes=1, interp_method="cubic")
    lin, seeds = anvil_seeds(bt)
    want = fl.watershed(get_combined_edge_field(fl, lin, dtype=np.float32), seeds, connectivity=1)
    got = {}
    res, _ = detect_stack_windows(bt, [(0, 14)], lambda w, c: anvil_seeds(w), on_window=lambda c, k, lab: got.setdefault(k, lab.clone()))
    assert list(got) == [0] and torch.equal(got[0], want)
    lut = np.asarray(res[0]["luts"][0])
    assert lut[0] == 0 and np.array_equal(lut[1:], np.arange(1, int(want.max()) + 1))
    assert torch.equal(apply_global_lut(got[0], lut), want)
    res, _ = detect_stack_windows(bt, window_bounds(14, 2, 4), lambda w, c: anvil_seeds(w), stitch=False, on_window=lambda c, k, lab: int(lab.max()))
    assert res[0]["luts"] is None and len(res[0]["windows"]) == 2 and all(isinstance(v, int) for v in res[0]["windows"])
