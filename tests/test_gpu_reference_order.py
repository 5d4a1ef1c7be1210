"""on_ambiguous="reference" / TF_WS_REFERENCE_ORDER: where labels hang on the order of EQUAL-VALUED MARKERS -- the one
thing the reference decides by the array mechanics of its binary heap (_watershed.pyx:67-152, 278-284) -- the library
replays those mechanics on the host for the pop ranks and the device flood uses them: the labels are then the
reference's bit for bit ("What's missing" 1).  Checked against the reference's own golden outputs
(tests/golden/watershed_ref.npz, produced by the reference's watershed.py + compiled _watershed.pyx) and against the C
twin of that kernel (oracle/c/ws_heap.c, tie_mode 0 = the reference's semantics) on tie-heavy random volumes."""
import os
import sys
import warnings

import numpy as np
import pytest
import scipy.ndimage as ndi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu

ALL_GOLDENS = ["A_cont_c1", "B_cont_mask_c2", "B_cont_mask_c3", "C_quant4_c1", "C_quant32_c1", "D_anvil_like_c1",
               "E_const_plateau_c1", "F_zero_flow_c1", "G_big_flow_c1"]


@pytest.fixture(scope="module")
def tf():
    import tobac_flow_amd.flow as tf
    return tf


@pytest.mark.parametrize("name", ALL_GOLDENS)
def test_every_reference_golden_is_reproduced_bit_for_bit(tf, golden_ws, name):
    """All nine outputs of the reference itself -- the tie-heavy C_quant4 (21 px) and E_const_plateau (37 px) included,
    which the default mode resolves by raster order and reports."""
    c = golden_ws[name]
    with warnings.catch_warnings():
        warnings.simplefilter("error")                           # reference order applied: nothing left to warn about
        got = tf.watershed(c["fwd"], c["bwd"], c["field"], c["markers"], mask=c.get("mask"), connectivity=int(c["conn"]),
                           on_ambiguous="reference")
    assert got.dtype == np.int32
    assert np.array_equal(got, c["labels"]), f"{int((got != c['labels']).sum())} px differ from the reference"


@pytest.mark.parametrize("name", ["C_quant4_c1", "E_const_plateau_c1"])
def test_the_raveled_twin_in_reference_order(tf, golden_ws, name):
    """the same through the reference's native seam (tf_watershed_raveled_ex, padded flat arrays as watershed.py:59-149
    prepares them)"""
    from oracle import ws_oracle
    from tobac_flow_amd._watershed import watershed_raveled
    c = golden_ws[name]
    conn = int(c["conn"])
    p = ws_oracle.prepare(c["fwd"], c["bwd"], c["field"], c["markers"], c.get("mask"), conn)
    out = p["out"].ravel().copy()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        watershed_raveled(np.ascontiguousarray(p["field"].ravel()), p["markers"].astype(np.intp), p["nbr"].astype(np.intp),
                          p["fwd_off"], p["bwd_off"], p["fwd_loc"], p["bwd_loc"], p["mask"], p["strides"].astype(np.int32), 0.0,
                          out, False, reference_order=True)
    pd = p["pad"]
    o = out.reshape(p["out"].shape)
    assert np.array_equal(o[pd[0]:o.shape[0] - pd[0], pd[1]:o.shape[1] - pd[1], pd[2]:o.shape[2] - pd[2]], c["labels"])


def _tie_heavy_case(seed):
    """quantised fields (few distinct values, large exact plateaus), many seeds of several labels sharing those values,
    non-zero flows, masks, all three connectivities"""
    rng = np.random.default_rng(7000 + seed)
    shape = (int(rng.integers(1, 6)), int(rng.integers(24, 64)), int(rng.integers(24, 72)))
    levels = [1, 2, 3, 4, 8, 32][seed % 6]
    smooth = ndi.gaussian_filter(rng.normal(size=shape), (0.5, 3, 3))
    smooth = (smooth - smooth.min()) / (smooth.max() - smooth.min() + 1e-9)
    field = (np.floor(smooth * levels) / max(levels, 1)).astype(np.float32)
    if seed % 4 == 3:
        field[rng.random(shape) < 0.01] = np.inf
    markers = np.zeros(shape, np.int32)
    n_seeds = int(rng.integers(6, 40))
    for k in range(n_seeds):
        t, y, x = (int(rng.integers(0, s)) for s in shape)
        h, w = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        markers[t, y:y + h, x:x + w] = (k % 9) + 1 if seed % 2 else k + 1
    if seed % 3 == 0:
        markers[:, :2, :] = -1
    mask = None if seed % 2 == 0 else ndi.gaussian_filter(rng.normal(size=shape), (0, 2, 2)) > -0.05
    amp = [0.0, 1.5, 3.0][seed % 3]
    fwd = (rng.normal(size=shape + (2,)) * amp).astype(np.float32)
    bwd = (rng.normal(size=shape + (2,)) * amp).astype(np.float32)
    return fwd, bwd, field, markers, mask, [1, 2, 3][(seed // 2) % 3]


@pytest.mark.parametrize("seed", range(24))
def test_tie_heavy_random_volumes_equal_the_reference_kernel(tf, seed):
    """against the C twin of the reference's kernel in the reference's own semantics (tie_mode 0, pinned by the goldens
    and by the live compiled .pyx, tests/test_oracle_golden.py)"""
    from oracle import ws_oracle
    fwd, bwd, field, markers, mask, conn = _tie_heavy_case(seed)
    want = ws_oracle.watershed(fwd, bwd, field, markers, mask, conn, tie_mode=0)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        got = tf.watershed(fwd, bwd, field, markers, mask=mask, connectivity=conn, on_ambiguous="reference")
    assert np.array_equal(got, want), f"{int((got != want).sum())} px differ from the reference kernel"


def test_reference_order_costs_nothing_when_no_tie_matters(tf, golden_ws):
    """a clean input: no replay (stats), same labels"""
    import torch
    from tobac_flow_amd import _lib
    from tobac_flow_amd.watershed import neighbour_offsets, watershed_dev
    from tobac_flow_amd import watershed as W
    c = golden_ws["A_cont_c1"]
    st = {}
    with W._MEMO_LOCK:
        W._tie_memo.clear()       # no guessed tie value left by an earlier flood of this shape (a guess starts the export, and a replay, regardless)
    lab = watershed_dev(_lib.to_dev(c["fwd"], torch.float32), _lib.to_dev(c["bwd"], torch.float32), _lib.to_dev(c["field"], torch.float32),
                        _lib.to_dev(c["markers"], torch.int32), None, neighbour_offsets(int(c["conn"])), stats=st, on_ambiguous="reference")
    assert st["reference_order"] == {"replayed_pops": 0, "seeds": 0, "microseconds": 0}
    assert np.array_equal(lab.cpu().numpy(), c["labels"])
    c = golden_ws["E_const_plateau_c1"]
    lab = watershed_dev(_lib.to_dev(c["fwd"], torch.float32), _lib.to_dev(c["bwd"], torch.float32), _lib.to_dev(c["field"], torch.float32),
                        _lib.to_dev(c["markers"], torch.int32), None, neighbour_offsets(int(c["conn"])), stats=st, on_ambiguous="reference")
    # the replay is handed the seeds at or below the largest tie value only (all of them on a constant plateau)
    assert st["reference_order"]["replayed_pops"] > 0 and 0 < st["reference_order"]["seeds"] <= int((c["markers"] != 0).sum())
    assert np.array_equal(lab.cpu().numpy(), c["labels"])


@pytest.mark.parametrize("seed", range(0, 24, 2))
def test_sparse_dense_and_plain_replay_agree(tf, seed, monkeypatch):
    """The three host replays of csrc/ws_replay.h behind one flood: sparse (the default: only the heap items at or below
    the largest tie value are followed, the others are anonymous occupants of their positions), dense (every seed in place
    as an 8-byte entry, runs of equal seeds popped in one scan, the tree of equal seeds walked with a saved path;
    TF_WS_REFERENCE_DENSE=1) and plain (every item as the reference keeps it; =2): same labels, same number of replayed
    pops, the sparse form never handed more seeds.  (tools/replay_check compares them on 200 000 random instances on the CPU.)"""
    import torch
    from tobac_flow_amd import _lib
    from tobac_flow_amd.watershed import neighbour_offsets, watershed_dev
    fwd, bwd, field, markers, mask, conn = _tie_heavy_case(seed)
    args = (_lib.to_dev(fwd, torch.float32), _lib.to_dev(bwd, torch.float32), _lib.to_dev(field, torch.float32),
            _lib.to_dev(markers, torch.int32), None if mask is None else _lib.to_dev(mask.astype(np.int8), torch.int8), neighbour_offsets(conn))
    st_sparse, st_dense, st_plain = {}, {}, {}
    lab_sparse = watershed_dev(*args, stats=st_sparse, on_ambiguous="reference").cpu().numpy()
    monkeypatch.setenv("TF_WS_REFERENCE_DENSE", "1")
    lab_dense = watershed_dev(*args, stats=st_dense, on_ambiguous="reference").cpu().numpy()
    monkeypatch.setenv("TF_WS_REFERENCE_NO_CODES", "1")                # the dense form's seeds as 8-byte entries instead of 2 bits + exceptions
    lab_dense8 = watershed_dev(*args, on_ambiguous="reference").cpu().numpy()
    monkeypatch.delenv("TF_WS_REFERENCE_NO_CODES")
    monkeypatch.setenv("TF_WS_REFERENCE_DENSE", "2")
    lab_plain = watershed_dev(*args, stats=st_plain, on_ambiguous="reference").cpu().numpy()
    assert np.array_equal(lab_sparse, lab_dense) and np.array_equal(lab_plain, lab_dense) and np.array_equal(lab_dense8, lab_dense)
    a, b, c = st_sparse["reference_order"], st_dense["reference_order"], st_plain["reference_order"]
    assert a["replayed_pops"] == b["replayed_pops"] == c["replayed_pops"] and a["seeds"] <= b["seeds"] == c["seeds"]
    if a["replayed_pops"]:
        assert st_sparse["reference_order_detail"]["replay_form"] == "sparse" and st_dense["reference_order_detail"]["replay_form"] == "dense"


def _one_value_class_case(seed):
    """The detect_anvils shape of a tie: a plateau at the SMALLEST value of the field carrying seeds of many different
    labels (striped, with floodable plateau pixels between them), every other seed larger and (float noise) distinct, and
    the plateau early in raster order -- so every heap item at or below the tie value is a seed of that one value and none
    of them sits among the last S heap positions: the case the device evaluates in closed form (k_ws_tie_*)."""
    rng = np.random.default_rng(9100 + seed)
    T, H, W = int(rng.integers(2, 5)), int(rng.integers(40, 90)), int(rng.integers(48, 120))
    field = (0.1 + ndi.gaussian_filter(rng.random((T, H, W)), (0.5, 2, 2)) + 1e-3 * rng.random((T, H, W))).astype(np.float32)
    markers = np.zeros((T, H, W), np.int32)
    y0, y1 = 3, 3 + int(rng.integers(8, 18))
    field[0, y0:y1, 2:W - 2] = -1.0                                     # the plateau
    step = int(rng.integers(2, 5))
    lab = 1
    for y in range(y0, y1, 2 if seed % 2 else 1):
        for x in range(2 + int(rng.integers(0, step)), W - 2, step):
            if rng.random() < 0.8:
                markers[0, y, x] = lab if seed % 3 else (lab % 7) + 1
                lab += 1
    if seed % 4 == 1:                                                  # a second plateau frame, still early
        field[1, y0:y1, 2:W // 2] = -1.0
        markers[1, y0:y1:2, 3:W // 2:step] = np.arange(lab, lab + len(range(3, W // 2, step)), dtype=np.int32)[None, :]
    # the larger seeds: the last frame's border band as background (-1) + scattered ones, all at values > -1
    markers[T - 1, H - 12:, :] = -1
    for k in range(int(rng.integers(5, 20))):
        t, y, x = int(rng.integers(1, T)), int(rng.integers(y1 + 2, H - 14)), int(rng.integers(0, W - 3))
        markers[t, y:y + 2, x:x + 3] = 1000 + k
    amp = [0.0, 1.0, 2.5][seed % 3]
    fwd = (rng.normal(size=(T, H, W, 2)) * amp).astype(np.float32)
    bwd = (rng.normal(size=(T, H, W, 2)) * amp).astype(np.float32)
    mask = None if seed % 2 == 0 else rng.random((T, H, W)) > 0.03
    return fwd, bwd, field, markers, mask, [1, 2, 3][(seed // 2) % 3]


@pytest.mark.parametrize("seed", range(12))
def test_pop_ranks_in_closed_form_on_the_device_equal_the_host_replay_and_the_reference_kernel(tf, seed, monkeypatch):
    """Round 4: when every heap item at or below the tie value is a seed of that one value, the pop ranks are computed on
    the device (no export, no host pass).  Same labels as the host replay (TF_WS_REFERENCE_HOST=1) and as the C twin of the
    reference's kernel (tie_mode 0); run twice, so that both the export after the root phase (first flood of a shape: no
    guess) and the export on a guessed tie value (second) take the device form."""
    import torch
    from oracle import ws_oracle
    from tobac_flow_amd import _lib
    from tobac_flow_amd import watershed as Wm
    from tobac_flow_amd.watershed import neighbour_offsets, watershed_dev
    fwd, bwd, field, markers, mask, conn = _one_value_class_case(seed)
    want = ws_oracle.watershed(fwd, bwd, field, markers, mask, conn, tie_mode=0)
    args = (_lib.to_dev(fwd, torch.float32), _lib.to_dev(bwd, torch.float32), _lib.to_dev(field, torch.float32),
            _lib.to_dev(markers, torch.int32), None if mask is None else _lib.to_dev(mask.astype(np.int8), torch.int8), neighbour_offsets(conn))
    with Wm._MEMO_LOCK:
        Wm._tie_memo.clear()
    st1, st2, sth, st0 = {}, {}, {}, {}
    raster = watershed_dev(*args, stats=st0, on_ambiguous="ignore").cpu().numpy()
    with Wm._MEMO_LOCK:
        Wm._tie_memo.clear()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        lab1 = watershed_dev(*args, stats=st1, on_ambiguous="reference").cpu().numpy()
        lab2 = watershed_dev(*args, stats=st2, on_ambiguous="reference").cpu().numpy()
        monkeypatch.setenv("TF_WS_REFERENCE_HOST", "1")
        labh = watershed_dev(*args, stats=sth, on_ambiguous="reference").cpu().numpy()
    assert st0["ambiguous_pixels"] > 0, "the case has no label that hangs on the order of equal-valued markers"
    d1, d2, dh = st1["reference_order_detail"], st2["reference_order_detail"], sth["reference_order_detail"]
    assert d1["replay_form"] == "device" and not d1["guessed"], d1
    assert d2["replay_form"] == "device" and d2["guessed"] and d2["guess_covered_the_tie"], d2
    assert dh["replay_form"] == "sparse", dh
    assert st2["root_phases"] < st1["root_phases"]                      # the guess saves the second root phase
    assert np.array_equal(lab1, labh) and np.array_equal(lab2, labh)
    assert np.array_equal(lab1, want), f"{int((lab1 != want).sum())} px differ from the reference kernel ({int((raster != want).sum())} in raster order)"


def test_dense_replay_on_full_disk_frames_equals_the_reference_kernel(monkeypatch):
    """The dense form at the benchmark's frame size (2 x 5424^2: 56 M seeds, most of them the background's at exactly 0, so
    the runs and the saved path are really exercised) against the C twin of the reference's kernel, every voxel."""
    import torch
    import tobac_flow_amd.flow as tf
    from oracle import ws_oracle
    from tobac_flow_amd.detection import get_combined_edge_field
    from tobac_flow_amd.watershed import neighbour_offsets, watershed_dev
    from tools.synth import anvil_seeds, blob_stack
    bt = blob_stack(2, 5424, 5424, seed=20240601, t0=30)
    fl = tf.create_flow(bt, vr_steps=1, smoothing_passes=1, interp_method="cubic")
    lin, seeds = anvil_seeds(bt)
    e = get_combined_edge_field(fl, lin, dtype=np.float32)
    fw, bw = fl._dev_flows()
    st = {}
    monkeypatch.setenv("TF_WS_REFERENCE_DENSE", "1")
    lab = watershed_dev(fw, bw, e, seeds, None, neighbour_offsets(1), stats=st, on_ambiguous="reference")
    want = ws_oracle.watershed(fw.cpu().numpy(), bw.cpu().numpy(), e.cpu().numpy(), seeds.cpu().numpy(), None, 1, tie_mode=0)
    print("2 x 5424^2, dense replay:", st["reference_order"], st["reference_order_detail"])
    assert st["reference_order"]["replayed_pops"] > 0
    assert np.array_equal(lab.cpu().numpy(), want)


def test_config_C_window_with_component_seeds_in_reference_order(tf):
    """the 14 x 1500 x 2500 window of config C with component-labelled seeds -- 327 reported
    voxels in the default mode, 21 of which the reference's heap orders the other way -- equals the reference kernel
    (C twin, tie_mode 0) in every voxel; the cost of the replay is printed."""
    import torch
    from oracle import ws_oracle
    from tobac_flow_amd.detection import get_combined_edge_field
    from tobac_flow_amd.watershed import neighbour_offsets, watershed_dev
    from tools.synth import anvil_inputs, blob_stack
    bt = blob_stack(14, 1500, 2500, seed=11, t0=0)
    fl = tf.create_flow(bt, vr_steps=1, smoothing_passes=1, interp_method="cubic")
    lin, markers = anvil_inputs(bt)
    comp = ndi.label(markers.cpu().numpy() > 0)[0].astype(np.int32)
    comp[markers.cpu().numpy() < 0] = -1
    e = get_combined_edge_field(fl, lin, dtype=np.float32)
    fw, bw = fl._dev_flows()
    st, st0 = {}, {}
    seeds_dev = torch.from_numpy(comp).cuda()
    base = watershed_dev(fw, bw, e, seeds_dev, None, neighbour_offsets(1), stats=st0, on_ambiguous="ignore")
    lab = watershed_dev(fw, bw, e, seeds_dev, None, neighbour_offsets(1), stats=st, on_ambiguous="reference")
    want = ws_oracle.watershed(fw.cpu().numpy(), bw.cpu().numpy(), e.cpu().numpy(), comp, None, 1, tie_mode=0)
    got = lab.cpu().numpy()
    print("config C window: default mode differs from the reference in", int((base.cpu().numpy() != want).sum()), "of",
          st0["ambiguous_pixels"], "reported voxels; reference order:", st["reference_order"])
    assert st0["ambiguous_pixels"] > 0                                # the case is present in this window
    assert np.array_equal(got, want), f"{int((got != want).sum())} px differ from the reference kernel"


def test_raveled_flood_along_a_path_of_several_thousand_pixels(tf):
    """the raveled form knows no (T, H, W), its sweep limit has to come from the number of pixels: one seed at
    the end of a snake-shaped mask whose flood path is ~7000 pixels long (used to stop with TF_ENOCONV at 4096 sweeps)."""
    from tobac_flow_amd._watershed import watershed_raveled
    H, W = 123, 121
    mask2 = np.zeros((H, W), np.int8)
    for r in range(1, H - 1, 2):                                       # corridors joined alternately at the right / left end
        mask2[r, 1:W - 1] = 1
        if r + 2 < H - 1:
            mask2[r + 1, W - 2 if (r // 2) % 2 == 0 else 1] = 1
    n = H * W
    rng = np.random.default_rng(0)
    img = rng.random(n).astype(np.float32)
    out = np.zeros(n, np.int32)
    start = 1 * W + 1
    out[start] = 5
    z = np.zeros(n, np.int32)
    watershed_raveled(img, np.array([start], np.intp), np.array([-W, -1, 1, W], np.intp), z, z, np.zeros(4, np.int32),
                      np.zeros(4, np.int32), mask2.ravel(), np.array([W, 1], np.int32), 0.0, out, False)
    path = int(mask2.sum())
    assert path > 5000
    assert np.array_equal(out.reshape(H, W) == 5, mask2 == 1)


def test_full_disk_frames_with_component_seeds_in_reference_order():
    """3 x 5424 x 5424 with SURVEY 8(d)'s component-labelled seeds: reference order against the C twin of the reference's
    kernel in its own semantics (tie_mode 0), every voxel; the twin needs ~15 s."""
    import torch
    import tobac_flow_amd.flow as tf
    from oracle import ws_oracle
    from tobac_flow_amd.detection import get_combined_edge_field
    from tobac_flow_amd.watershed import neighbour_offsets, watershed_dev
    from tools.synth import anvil_seeds, blob_stack
    bt = blob_stack(3, 5424, 5424, seed=20240601, t0=12)
    fl = tf.create_flow(bt, vr_steps=1, smoothing_passes=1, interp_method="cubic")
    lin, seeds = anvil_seeds(bt)
    e = get_combined_edge_field(fl, lin, dtype=np.float32)
    fw, bw = fl._dev_flows()
    st = {}
    lab = watershed_dev(fw, bw, e, seeds, None, neighbour_offsets(1), stats=st, on_ambiguous="reference")
    want = ws_oracle.watershed(fw.cpu().numpy(), bw.cpu().numpy(), e.cpu().numpy(), seeds.cpu().numpy(), None, 1, tie_mode=0)
    got = lab.cpu().numpy()
    print("3 x 5424^2, component seeds: ambiguous voxels %d, reference order %s, %s" % (st["ambiguous_pixels"], st["reference_order"], st["reference_order_detail"]))
    assert np.array_equal(got, want), f"{int((got != want).sum())} px differ from the reference kernel"


def test_reference_order_cost_on_a_full_disk_window():
    """the cost of the reference order on a 12 x 5424 x 5424 window (353 M voxels; no oracle at
    this size).  Properties: outside the voxels the default mode REPORTS as depending on the order of equal-valued
    markers, both modes give the same label; inside them the label is one of the labels of the tying seeds (here: it
    stays a valid seed label); the call returns clean (no warning)."""
    import time
    import torch
    import tobac_flow_amd.flow as tf
    from tobac_flow_amd.detection import get_combined_edge_field
    from tobac_flow_amd.watershed import neighbour_offsets, watershed_dev
    from tools.synth import anvil_seeds, blob_stack
    bt = blob_stack(12, 5424, 5424, seed=20240601, t0=0)
    fl = tf.create_flow(bt, vr_steps=1, smoothing_passes=1, interp_method="cubic")
    lin, seeds = anvil_seeds(bt)
    e = get_combined_edge_field(fl, lin, dtype=np.float32)
    fw, bw = fl._dev_flows()
    nbr = neighbour_offsets(1)
    st0, st1 = {}, {}
    base, rep = watershed_dev(fw, bw, e, seeds, None, nbr, stats=st0, on_ambiguous="ignore", return_ambiguous=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    base2 = watershed_dev(fw, bw, e, seeds, None, nbr, on_ambiguous="ignore")
    torch.cuda.synchronize()
    t_default = time.perf_counter() - t0
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        t0 = time.perf_counter()
        ref = watershed_dev(fw, bw, e, seeds, None, nbr, stats=st1, on_ambiguous="reference")
        torch.cuda.synchronize()
        t_ref = time.perf_counter() - t0
    print("12 x 5424^2 window: default flood %.3f s, reference-order flood %.3f s; %d voxels depend on the order of equal-valued "
          "markers, %d of them change; replay %s" % (t_default, t_ref, st0["ambiguous_pixels"], int((ref != base).sum()), st1["reference_order"]))
    assert torch.equal(base, base2)
    assert st0["ambiguous_pixels"] > 0 and st1["reference_order"]["replayed_pops"] > 0
    differs = ref != base
    assert not bool((differs & ((rep & 1) == 0)).any())                # only reported voxels can change
    assert bool((ref[seeds != 0] == seeds[seeds != 0]).all()) and int((ref == 0).sum()) == 0
