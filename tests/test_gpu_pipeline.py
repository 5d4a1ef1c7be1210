"""The drop-in script's sequence end to end on a synthetic scene (scripts/dcc_detect_goes.py:162-330): create_flow with
refinement -> detect_cores -> get_anvil_markers -> detect_anvils (thick, thin) -> the output-file label contract and the
per-label statistics.  Every stage has its own parity test; this one checks that the stages fit together (dtypes, label
ranges, containers) and that the contract's variables are consistent with each other and with the oracle's restatement."""
import os
import warnings

import numpy as np
import pytest
import scipy.ndimage as ndi

from helpers import blob_sequence
from oracle import np_dataset
from test_gpu_detection import FakeDataArray

pytestmark = pytest.mark.gpu


def test_detection_script_sequence_end_to_end():
    import tobac_flow_amd.flow as tf
    from tobac_flow_amd import dataset as D
    from tobac_flow_amd.analysis import weighted_statistics_on_labels
    from tobac_flow_amd.detection import detect_anvils, detect_cores, get_anvil_markers
    T, minutes = 8, 2
    rng = np.random.default_rng(42)
    bt0 = blob_sequence(rng, T, 96, 120, n_blobs=5, vmax=2.0, noise=0.5)
    flow = tf.create_flow(bt0, model="Farneback", vr_steps=1, smoothing_passes=1, interp_method="cubic")
    ramp = np.linspace(0.5, 1.4, T, dtype=np.float32)[:, None, None]
    cold = np.clip(250.0 - bt0, 0, None)
    bt = FakeDataArray((290.0 - ramp * cold).astype(np.float32), minutes=minutes)
    # WVD - SWD crosses the thick-anvil thresholds (-5 / -12.5) and WVD + SWD the thin ones (0 / -7.5) around the cold blobs
    wvd = FakeDataArray((0.8 * cold - 11).astype(np.float32), minutes=minutes)
    swd = FakeDataArray(np.full(cold.shape, 3.0, np.float32), minutes=minutes)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        core = np.asarray(detect_cores(flow, bt, wvd, swd, use_wvd=False, min_length=3)).astype(np.int32)
        wd, ws = np.asarray(wvd) - np.asarray(swd), np.asarray(wvd) + np.asarray(swd)
        markers = np.asarray(get_anvil_markers(flow, wd, threshold=-5, overlap=0.5, absolute_overlap=4, min_length=3))
        thick = np.asarray(detect_anvils(flow, wd, markers=markers, upper_threshold=-5, lower_threshold=-12.5,
                                         erode_distance=2, min_length=3)).astype(np.int32)
        thin = np.asarray(detect_anvils(flow, ws, markers=thick, upper_threshold=0, lower_threshold=-7.5,
                                        erode_distance=2, min_length=3)).astype(np.int32)
    assert core.max() >= 1 and thick.max() >= 1 and thin.max() >= 1
    assert np.all(thin[thick > 0] == thick[thick > 0])               # the thin anvil grows from the thick one's labels

    ds = D.LabelDataset(coords={"t": np.asarray(bt.t.values if hasattr(bt.t, "values") else bt.t)})
    ref = {"coords": {"t": ds.coords["t"]}}
    for name, v in (("core_label", core), ("thick_anvil_label", thick), ("thin_anvil_label", thin)):
        ds.add(name, v.copy(), ("t", "y", "x")); ref[name] = v.copy()
    for fn in ("add_label_coords", "link_cores_and_anvils", "add_step_labels", "add_label_coords", "link_step_labels",
               "flag_edge_labels"):
        getattr(D, fn)(ds); getattr(np_dataset, fn)(ref)
    nan_field = np.asarray(wvd).copy(); nan_field[3, 40:43, 50:52] = np.nan
    D.flag_nan_adjacent_labels(ds, nan_field); np_dataset.flag_nan_adjacent_labels(ref, nan_field)
    for k, want in ref.items():
        if k == "coords":
            for c, v in want.items():
                assert np.array_equal(ds.coords[c], v), c
        else:
            assert np.array_equal(np.asarray(ds[k]), want) and np.asarray(ds[k]).dtype == want.dtype, k
    # the variables agree with each other
    cores, anvils = ds.coords["core"], ds.coords["anvil"]
    assert set(ds["core_anvil_index"]) <= set(anvils) | {0}
    assert ds["anvil_core_count"].sum() == np.count_nonzero(ds["core_anvil_index"])
    linked = ds["core_anvil_index"] > 0
    for c, a in zip(cores[linked], ds["core_anvil_index"][linked]):
        assert np.all(ds["thick_anvil_label"][ds["core_label"] == c] == a)      # cores were written into their anvil
    step = ds["core_step_label"]
    for k, parent in zip(ds.coords["core_step"], ds["core_step_core_index"]):
        where = step == k
        assert np.all(ds["core_label"][where] == parent) and len(np.unique(np.nonzero(where)[0])) == 1   # one step each
    # per-label statistics of the brightness temperature over the cores
    mean, std, mx, mn = weighted_statistics_on_labels(ds["core_label"], np.asarray(bt), np.ones(core.shape, np.float32))
    assert mean.shape == (core.max(),)
    for c in cores:
        vals = np.asarray(bt)[ds["core_label"] == c].astype(np.float64)
        assert abs(mean[c - 1] - vals.mean()) < 1e-3 and mx[c - 1] == vals.max().astype(np.float32) and mn[c - 1] == vals.min().astype(np.float32)
        assert abs(std[c - 1] - vals.std()) < 1e-3


def test_flow_window_equals_create_flow_on_the_slice():
    """Flow.window(a, b) is the Flow create_flow(data[a:b]) returns, bit for bit, on the device and for numpy stacks:
    the flow of a frame pair does not depend on the stack it is cut from; only the two end frames of a stack are
    mirrored (flow.py:425-426).  bench.py computes the flow of a long stack once and floods overlapping windows of it."""
    import torch
    import tobac_flow_amd.flow as tf
    from tools.synth import blob_stack
    bt = blob_stack(9, 150, 210, seed=3)
    kw = dict(model="Farneback", vr_steps=1, smoothing_passes=1, interp_method="cubic")
    whole = tf.create_flow(bt, **kw)
    for a, b in ((0, 9), (0, 4), (3, 8), (5, 9), (4, 6), (7, 8)):
        want = tf.create_flow(bt[a:b], **kw)
        got = whole.window(a, b)
        assert got.shape == want.shape
        for g, w in ((got.forward_flow, want.forward_flow), (got.backward_flow, want.backward_flow)):
            assert torch.equal(torch.nan_to_num(g, nan=-777.0), torch.nan_to_num(w, nan=-777.0)), (a, b)
    whole_np = tf.create_flow(bt.cpu().numpy(), **kw)
    got = whole_np.window(3, 8)
    want = tf.create_flow(bt[3:8].cpu().numpy(), **kw)
    assert isinstance(got.forward_flow, np.ndarray)
    assert np.array_equal(got.forward_flow, want.forward_flow, equal_nan=True) and np.array_equal(got.backward_flow, want.backward_flow, equal_nan=True)
    assert not np.shares_memory(got.forward_flow, whole_np.forward_flow)         # a window never aliases the stack it was cut from


def test_flow_window_view_is_the_window_without_the_copy_and_restores_the_stack():
    """`with flow.window_view(a, b) as w` (what bench.py floods its windows through): w equals window(a, b) bit for bit,
    aliases the stack's arrays, and the stack is bit-identical to what it was once the block is left -- also when the
    block raises."""
    import torch
    import tobac_flow_amd.flow as tf
    from tools.synth import blob_stack
    bt = blob_stack(8, 96, 130, seed=5)
    whole = tf.create_flow(bt, model="Farneback", vr_steps=0, smoothing_passes=1, interp_method="linear")
    f0, b0 = whole.forward_flow.clone(), whole.backward_flow.clone()
    same = lambda x, y: torch.equal(torch.nan_to_num(x, nan=-777.0), torch.nan_to_num(y, nan=-777.0))
    for a, b in ((0, 8), (0, 3), (2, 7), (5, 8), (3, 4), (0, 1), (7, 8)):
        want = whole.window(a, b)
        with whole.window_view(a, b) as w:
            assert w.shape == want.shape
            assert same(w.forward_flow, want.forward_flow) and same(w.backward_flow, want.backward_flow), (a, b)
            assert w.forward_flow.data_ptr() == whole.forward_flow[a].data_ptr()
        assert same(whole.forward_flow, f0) and same(whole.backward_flow, b0), (a, b)
    with pytest.raises(RuntimeError, match="inside"):
        with whole.window_view(2, 6):
            raise RuntimeError("inside")
    assert same(whole.forward_flow, f0) and same(whole.backward_flow, b0)
    with pytest.raises(ValueError):
        with whole.window_view(4, 4):
            pass
    whole_np = tf.Flow(f0.cpu().numpy(), b0.cpu().numpy())
    want = whole_np.window(2, 6)
    with whole_np.window_view(2, 6) as w:
        assert np.array_equal(w.forward_flow, want.forward_flow, equal_nan=True) and np.array_equal(w.backward_flow, want.backward_flow, equal_nan=True)
        assert np.shares_memory(w.forward_flow, whole_np.forward_flow)
    assert np.array_equal(whole_np.forward_flow, f0.cpu().numpy(), equal_nan=True) and np.array_equal(whole_np.backward_flow, b0.cpu().numpy(), equal_nan=True)


@pytest.mark.parametrize("smoothing_passes", [0, 1, 2])
def test_create_flow_clip_is_the_elementwise_clip_of_the_unclipped_flow(smoothing_passes):
    """create_flow clips both flow arrays to +-max_value (flow.py:60-63) and mirrors the end frames (flow.py:425-426).  With
    smoothing the clip rides on the last smoothing pass's store (tf_smooth_flow_step_clip) and only the end frames are
    written afterwards (tf_flow_finalize_ends); without, one pass does both (tf_flow_finalize).  Either way the result is
    np.clip of the unclipped flow, NaN kept, and a single-frame stack is all NaN."""
    import torch
    import tobac_flow_amd.flow as tf
    from tools.synth import blob_stack
    bt = blob_stack(5, 120, 150, seed=9)
    kw = dict(model="Farneback", vr_steps=0, smoothing_passes=smoothing_passes, interp_method="linear")
    free = tf.create_flow(bt, max_value=1e9, **kw)
    assert float(free.forward_flow.abs().max()) > 0.5                     # the clip below does cut something
    for mv in (0.4, 20):
        got = tf.create_flow(bt, max_value=mv, **kw)
        for g, f in ((got.forward_flow, free.forward_flow), (got.backward_flow, free.backward_flow)):
            assert torch.equal(g, torch.clamp(f, -mv, mv))
        assert torch.equal(got.forward_flow[-1], -got.backward_flow[-1]) and torch.equal(got.backward_flow[0], -got.forward_flow[0])
    one = tf.create_flow(bt[:1], max_value=0.4, **kw)
    assert bool(torch.isnan(one.forward_flow).all()) and bool(torch.isnan(one.backward_flow).all())


def test_flow_batches_and_stream_overlap_do_not_change_the_flow(monkeypatch):
    """The flow of a stack does not depend on how its pairs are batched, nor on whether the refinement / smoothing of a
    batch runs on the second stream while the next batch's Farneback is under way (flow.py: _calculate_flow_impl)."""
    import torch
    import tobac_flow_amd.flow as tf
    from tools.synth import blob_stack
    bt = blob_stack(8, 180, 236, seed=9)
    kw = dict(model="Farneback", vr_steps=1, smoothing_passes=1, interp_method="cubic")
    monkeypatch.setenv("TF_FLOW_OVERLAP", "0")
    monkeypatch.setenv("TF_FLOW_BATCH", "64")
    want = tf.create_flow(bt, **kw)                                    # one batch, one stream
    for batch, overlap in (("2", "1"), ("3", "1"), ("2", "0"), ("1", "1")):
        monkeypatch.setenv("TF_FLOW_BATCH", batch)
        monkeypatch.setenv("TF_FLOW_OVERLAP", overlap)
        for _ in range(2):                                             # twice: the second run reuses the side stream and its buffers
            got = tf.create_flow(bt, **kw)
            torch.cuda.synchronize()
            for g, w in ((got.forward_flow, want.forward_flow), (got.backward_flow, want.backward_flow)):
                assert torch.equal(torch.nan_to_num(g, nan=-777.0), torch.nan_to_num(w, nan=-777.0)), (batch, overlap)


def test_a_batch_that_does_not_fit_is_halved_and_the_flow_is_the_same(monkeypatch):
    """_calculate_flow_impl sizes its Farneback batches from an ESTIMATE of the free memory; a batch whose scratch cannot
    be allocated after all (fragmented allocator) is halved and tried again, together with the batches after it -- the
    flow does not change.  Simulated here with a model that refuses every batch of more than two pairs."""
    import torch
    import tobac_flow_amd.flow as tf
    from tobac_flow_amd.utils.flow_utils import FarnebackFlow
    from tools.synth import blob_stack
    bt = blob_stack(10, 120, 168, seed=4)
    want = tf.create_flow(bt, vr_steps=1, smoothing_passes=1, interp_method="cubic")
    refused = []
    real = FarnebackFlow.calc_batch_dev

    def picky(self, prev, nxt, fwd_out, bwd_out, tag="farneback", parts=1):
        if prev.shape[0] > 2:
            refused.append(int(prev.shape[0]))
            raise torch.OutOfMemoryError("simulated: scratch for %d pairs does not fit" % prev.shape[0])
        return real(self, prev, nxt, fwd_out, bwd_out, tag, parts)

    monkeypatch.setattr(FarnebackFlow, "calc_batch_dev", picky)
    got = tf.create_flow(bt, vr_steps=1, smoothing_passes=1, interp_method="cubic")
    assert refused and max(refused) == 9                               # 9 pairs -> 4, 4, 1 -> 2, 2, 2, 2, 1
    for g, w in ((got.forward_flow, want.forward_flow), (got.backward_flow, want.backward_flow)):
        assert torch.equal(torch.nan_to_num(g, nan=-777.0), torch.nan_to_num(w, nan=-777.0))


def _rccl_one_rank_worker(rank, port, wins, overlap, out_dir):
    import torch
    import torch.distributed as dist
    from tobac_flow_amd.parallel import stitch_rank_windows
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    got = stitch_rank_windows([torch.from_numpy(w).cuda() for w in wins], overlap=overlap, _force_collectives=True)
    for j, g in enumerate(got):
        assert g.is_cuda
        np.save(os.path.join(out_dir, f"w{j}.npy"), g.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_stitch_collectives_over_rccl_in_a_one_rank_group(tmp_path):
    """The collective path of parallel.stitch_rank_windows (all_gather of counts and pair lists on DEVICE tensors over RCCL,
    pair counting and LUT passes on the GPU) in the only process group a one-GPU box can form: one rank.  Same labels as
    stitch_window_list.  (The neighbour message between ranks is covered by the gloo tests, tests/test_distributed_cpu.py.)"""
    import socket
    import torch
    import torch.multiprocessing as mp
    from tobac_flow_amd.parallel import stitch_window_list, window_bounds
    rng = np.random.default_rng(11)
    overlap, n_windows = 4, 3
    T, H, W = 6 * n_windows + overlap, 48, 64
    truth = ndi.label(ndi.gaussian_filter(rng.normal(size=(T, H, W)), (2.0, 1.5, 1.5)) > 0.03)[0].astype(np.int32)
    truth[:, :2, :2] = -1
    wins = []
    for a, b in window_bounds(T, n_windows, overlap):
        w = truth[a:b].copy()
        ids = np.unique(w[w > 0])
        perm = np.zeros(max(int(w.max()), 0) + 1, np.int32)
        perm[ids] = rng.permutation(len(ids)) + 1
        w[w > 0] = perm[w[w > 0]]
        wins.append(w)
    want = [x.cpu().numpy() for x in stitch_window_list([torch.from_numpy(w).cuda() for w in wins], overlap=overlap)]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_rccl_one_rank_worker, args=(port, wins, overlap, str(tmp_path)), nprocs=1, join=True)
    for j in range(n_windows):
        assert np.array_equal(np.load(tmp_path / f"w{j}.npy"), want[j]), j


def test_create_flow_on_frames_ready_hands_out_final_windows():
    """create_flow(on_frames_ready=f) (round 4): f(flow, n) is called after every batch of frame pairs, and every window
    that ends within the first n frames is then already the window of the finished Flow, bit for bit -- which is what lets
    bench.py begin a window's Sobel / seeds / flood (and its host replay) while the device computes the flow of the later
    frames.  Checked: the windows copied out inside the callbacks equal create_flow(window) of a fresh call."""
    import os
    import torch
    import tobac_flow_amd.flow as tf
    from tools.synth import blob_stack
    bt = blob_stack(11, 96, 128, seed=5, t0=3)
    bounds = [(0, 4), (2, 7), (5, 9), (7, 11)]
    got, seen = {}, []

    def ready(flow, n):
        seen.append(n)
        for a, b in bounds:
            if b <= n and (a, b) not in got:
                with flow.window_view(a, b) as w:
                    got[(a, b)] = (w.forward_flow.clone(), w.backward_flow.clone())
    old = os.environ.get("TF_FLOW_BATCH")
    os.environ["TF_FLOW_BATCH"] = "3"                                  # four batches of 3 / 3 / 2 / 2 pairs
    try:
        full = tf.create_flow(bt, vr_steps=1, smoothing_passes=1, interp_method="cubic", on_frames_ready=ready)
    finally:
        if old is None:
            del os.environ["TF_FLOW_BATCH"]
        else:
            os.environ["TF_FLOW_BATCH"] = old
    assert seen == sorted(seen) and seen[-1] == 11 and len(seen) >= 3
    assert set(got) == set(bounds)
    plain = tf.create_flow(bt, vr_steps=1, smoothing_passes=1, interp_method="cubic")
    assert torch.equal(full.forward_flow, plain.forward_flow) and torch.equal(full.backward_flow, plain.backward_flow)
    for (a, b), (fw, bw) in got.items():
        want = tf.create_flow(bt[a:b], vr_steps=1, smoothing_passes=1, interp_method="cubic")
        assert torch.equal(fw, want.forward_flow) and torch.equal(bw, want.backward_flow), (a, b)


def test_watershed_job_in_parts_on_a_second_stream_equals_the_one_call(golden_ws):
    """tf_watershed_begin / _replay / _finish (WatershedJob): begin, the host replay on a worker thread, step() on a second
    stream -- with and without a guessed tie value, and with a guess that is too low (re-entrant finish) -- give the labels
    of the one-call form (which are the reference's, tests/test_gpu_reference_order.py)."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from tobac_flow_amd import _lib
    from tobac_flow_amd import watershed as W
    c = golden_ws["E_const_plateau_c1"]
    args = (_lib.to_dev(c["fwd"], torch.float32), _lib.to_dev(c["bwd"], torch.float32), _lib.to_dev(c["field"], torch.float32),
            _lib.to_dev(c["markers"], torch.int32), None, W.neighbour_offsets(int(c["conn"])))
    side = torch.cuda.Stream()
    pool = ThreadPoolExecutor(2)
    key = next(iter(()), None)
    for guess in ("none", "memo", "too_low"):
        with W._MEMO_LOCK:
            k = (*c["field"].shape, len(args[5]), W.DEFAULT_CHAIN_DEPTH, torch.cuda.current_stream().cuda_stream)
            if guess == "none":
                W._tie_memo.pop(k, None)
            elif guess == "too_low":
                W._tie_memo[k] = [0]                              # the smallest ordered key: below every marker value
        st = {}
        job = W.watershed_begin(*args, stats=st, expect_conflict=True)
        assert job.needs_replay == (guess != "none")
        rounds = 0
        while True:
            if job.needs_replay:
                pool.submit(job.replay).result()
            done, lab = job.step(stream=side)
            rounds += 1
            if done:
                break
        assert rounds == (1 if guess == "memo" else 2), (guess, rounds)
        assert np.array_equal(lab.cpu().numpy(), c["labels"]), guess
        assert st["reference_order_detail"]["guess_covered_the_tie"] == (guess == "memo")
        assert st["root_phases"] == (1 if guess == "memo" else 2)
    # TF_WS_DEFER_SWEEPS (round 5): begin returns after the set-up and the export; phase A and the chain levels run in
    # sweeps() -- beside the replay -- or, if nobody calls it, at the start of the first step(): the same labels and statistics
    for explicit in (True, False):
        with W._MEMO_LOCK:
            W._tie_memo.pop(k, None)
            W._conflict_memo.pop(k, None)
        ref = {}
        want = W.watershed_begin(*args, stats=ref, expect_conflict=None).finish()
        with W._MEMO_LOCK:
            W._tie_memo.pop(k, None)
            W._conflict_memo.pop(k, None)
        st = {}
        job = W.watershed_begin(*args, stats=st, expect_conflict=None, defer_sweeps=True)
        assert job._sweeps_pending and k not in W._conflict_memo          # (the scheduling probe belongs to the sweeps)
        fut = pool.submit(job.replay) if job.needs_replay else None
        if explicit:
            job.sweeps()
            assert not job._sweeps_pending and k in W._conflict_memo
            job.sweeps()                                                     # (a second call does nothing)
        if fut is not None:
            fut.result()
        lab = job.finish()
        assert not job._sweeps_pending and k in W._conflict_memo
        assert torch.equal(lab, want) and np.array_equal(lab.cpu().numpy(), c["labels"]), explicit
        # (sweeps[7], the entries processed, depends on the order the chaotic relaxation happened to take)
        assert st["sweeps"][:7] == ref["sweeps"][:7] and st["root_phases"] == ref["root_phases"] and st["chain_depth"] == ref["chain_depth"], (st, ref)


def test_create_flow_with_split_batches_is_bit_identical_and_hands_out_parts(monkeypatch):
    """TF_FLOW_SPLIT=2: a batch runs its coarse pyramid levels for all pairs at once and its two finest levels, refinement and
    smoothing part by part (tf_farneback_batch_phase); the flows are those of the unsplit schedule bit for bit, and
    on_frames_ready fires after every PART."""
    import torch
    import tobac_flow_amd.flow as tf
    from tools.synth import blob_stack
    bt = blob_stack(11, 203, 331, seed=9, t0=1)
    want = tf.create_flow(bt, vr_steps=1, smoothing_passes=1, interp_method="cubic")
    monkeypatch.setenv("TF_FLOW_SPLIT", "2")
    seen = []
    got = tf.create_flow(bt, vr_steps=1, smoothing_passes=1, interp_method="cubic", on_frames_ready=lambda fl, n: seen.append(n))
    assert seen == [11]                                               # parts of five pairs of 203 x 331 would not fill the GPU: not split
    monkeypatch.setenv("TF_FLOW_SPLIT_FORCE", "1")                    # (split whatever the size)
    seen = []
    got = tf.create_flow(bt, vr_steps=1, smoothing_passes=1, interp_method="cubic", on_frames_ready=lambda fl, n: seen.append(n))
    assert seen == [6, 11]                                            # ten pairs: two parts of five
    assert torch.equal(torch.nan_to_num(got.forward_flow, nan=-7.0), torch.nan_to_num(want.forward_flow, nan=-7.0))
    assert torch.equal(torch.nan_to_num(got.backward_flow, nan=-7.0), torch.nan_to_num(want.backward_flow, nan=-7.0))
    raw = tf.calculate_flow(bt, "Farneback")                          # no refinement, no smoothing: written straight into the arrays
    monkeypatch.delenv("TF_FLOW_SPLIT")
    monkeypatch.delenv("TF_FLOW_SPLIT_FORCE")
    raw0 = tf.calculate_flow(bt, "Farneback")
    assert torch.equal(torch.nan_to_num(raw[0], nan=-7.0), torch.nan_to_num(raw0[0], nan=-7.0))


def test_out_of_memory_in_the_callback_is_the_callers_and_handed_out_frames_are_never_recomputed(monkeypatch):
    """(i) an OutOfMemoryError raised by the caller's on_frames_ready callback propagates -- it used to be
    taken for "this batch's scratch did not fit": workspace released, the batch recomputed at half size, the callback
    re-entered with a SMALLER n.  (ii) When the library's own work of a split batch runs out of memory in part k, only the
    pairs from that part on are queued again: the parts already handed out are not rewritten, the n the callback sees never
    decreases, and the flow is the plain call's bit for bit."""
    import pytest
    import torch
    import tobac_flow_amd.flow as tf
    from tobac_flow_amd.utils.flow_utils import FarnebackFlow
    from tools.synth import blob_stack
    bt = blob_stack(11, 203, 331, seed=9, t0=1)
    kw = dict(vr_steps=1, smoothing_passes=1, interp_method="cubic")
    want = tf.create_flow(bt, **kw)
    # (i)
    calls = []

    def greedy(flow, n):
        calls.append(n)
        raise torch.OutOfMemoryError("simulated: the caller's own allocation failed")
    monkeypatch.setenv("TF_FLOW_BATCH", "4")
    with pytest.raises(torch.OutOfMemoryError, match="caller's own"):
        tf.create_flow(bt, on_frames_ready=greedy, **kw)
    assert calls == [4]                                               # (batches of 3 / 4 / 3 pairs) entered once, not again with a smaller batch
    monkeypatch.delenv("TF_FLOW_BATCH")
    # (ii)
    monkeypatch.setenv("TF_FLOW_SPLIT", "2")
    monkeypatch.setenv("TF_FLOW_SPLIT_FORCE", "1")
    real = FarnebackFlow.calc_phase_dev
    state = {"phase2_calls": 0, "refused": 0}

    def picky(self, prev, nxt, fwd_out, bwd_out, phase, ws_pairs, tag="farneback"):
        if phase == 2:
            state["phase2_calls"] += 1
            if state["phase2_calls"] == 2:                            # the SECOND part of the first (only) batch of ten pairs
                state["refused"] += 1
                raise torch.OutOfMemoryError("simulated: scratch of the second part does not fit")
        return real(self, prev, nxt, fwd_out, bwd_out, phase, ws_pairs, tag)
    monkeypatch.setattr(FarnebackFlow, "calc_phase_dev", picky)
    seen, first_part = [], {}

    def ready(flow, n):
        seen.append(n)
        if n == 6 and not first_part:
            with flow.window_view(0, 6) as w:
                first_part["f"], first_part["b"] = w.forward_flow.clone(), w.backward_flow.clone()
    got = tf.create_flow(bt, on_frames_ready=ready, **kw)
    assert state["refused"] == 1
    assert seen == sorted(seen) and seen[0] == 6 and seen[-1] == 11 and seen.count(6) == 1, seen
    for g, w in ((got.forward_flow, want.forward_flow), (got.backward_flow, want.backward_flow)):
        assert torch.equal(torch.nan_to_num(g, nan=-7.0), torch.nan_to_num(w, nan=-7.0))
    w6 = tf.create_flow(bt[:6], **kw)
    assert torch.equal(first_part["f"], w6.forward_flow) and torch.equal(first_part["b"], w6.backward_flow)


def test_create_flow_from_host_input_keeps_the_vectors_in_hbm_until_somebody_reads_them():
    """Round 5: the drop-in scripts hand create_flow a host array and only pass the Flow object on (scripts/dcc_detect_goes.py:
    164-303).  The vectors therefore stay on the device, where every Flow method works, and the numpy arrays
    `forward_flow` / `backward_flow` of the reference's object are downloaded when first read -- same values as
    calculate_flow's numpy result, and the object behaves like the eager one (slicing, assignment, window)."""
    import tobac_flow_amd.flow as tf
    from tools.synth import blob_stack
    bt = blob_stack(5, 96, 128, seed=6).cpu().numpy()
    kw = dict(vr_steps=1, smoothing_passes=1, interp_method="cubic")
    flow = tf.create_flow(bt, **kw)
    assert flow._fw is None and flow.shape == (5, 96, 128)
    edges = flow.sobel(bt, direction="uphill", method="cubic")
    lab = flow.label(bt < 270)
    assert isinstance(edges, np.ndarray) and isinstance(lab, np.ndarray) and flow._fw is None      # nothing was downloaded
    fw, bw = tf.calculate_flow(bt, **kw)
    fw, bw = np.clip(fw, -20, 20), np.clip(bw, -20, 20)
    assert isinstance(flow.forward_flow, np.ndarray) and flow._fw is not None
    assert np.array_equal(flow.forward_flow, fw, equal_nan=True) and np.array_equal(flow.backward_flow, bw, equal_nan=True)
    eager = tf.Flow(fw, bw)
    assert np.array_equal(eager.sobel(bt, direction="uphill", method="cubic"), edges, equal_nan=True)
    part = flow[1:4]
    assert part.shape == (3, 96, 128) and np.array_equal(part.forward_flow, fw[1:4], equal_nan=True)
    w = flow.window(1, 4)
    assert np.array_equal(w.forward_flow[-1], -bw[3], equal_nan=True)
    flow.forward_flow = bw                                            # assignment drops the device copy
    assert flow._dev is None and np.array_equal(flow.forward_flow, bw, equal_nan=True)
