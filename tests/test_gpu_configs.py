"""The BASELINE.json configurations that round 1 never exercised as such ("Configs not exercised"):

  S   16 x 512 x 512 translating-blob sequence: the WHOLE pipeline (create_flow with the drop-in scripts' settings ->
      combined edge field -> watershed -> flow_label) against the oracle
  C   24 x 1500 x 2500: the full 24-frame stack as two overlapping windows + stitch, window floods against the oracle
  F   144 x 5424 x 5424 as twelve 12-frame windows through stitch_window_list on one GPU; the same procedure on a stack the
      oracle can afford is compared with the oracle window by window
  F3  three channels (offsets 0 / -2 / -4 K) sharing ONE Flow: growth detection per channel, device-resident recipe against
      the SciPy-glue recipe, and detect_growth_markers_multichannel against the recipe assembled from oracle pieces
  V   288 x 3712 x 3712 as twenty-four windows on one GPU (property checks + stitch, like F)
  F3 at its stated size: 288 x 5424 x 5424, three channels sharing one Flow, twenty-four windows per channel
"""
import os
import sys
import warnings
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest
import scipy.ndimage as ndi

from helpers import blob_sequence

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


def _oracle_flow(bt, vr_steps, smoothing_passes, interp, max_value=20):
    """calculate_flow + the clip of create_flow (flow.py:23-65, 362-428, 499-527) from oracle pieces; frame pairs on a
    thread pool (the C restatements release the GIL)."""
    from oracle import np_ops
    from test_gpu_parity import _oracle_farneback
    T, H, W = bt.shape
    fw = np.full((T, H, W, 2), np.nan, np.float32)
    bw = np.full((T, H, W, 2), np.nan, np.float32)

    def pair(i):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            p8 = np_ops.to_8bit(np_ops.linear_norm(bt[i:i + 2].copy()), 0, 1)
            f, b = _oracle_farneback(p8[0], p8[1]), _oracle_farneback(p8[1], p8[0])
            if vr_steps > 0:
                f, b = np_ops.variational_refinement(p8[0], p8[1], f), np_ops.variational_refinement(p8[1], p8[0], b)
            for _ in range(smoothing_passes):
                f, b = np_ops.smooth_flow_step(f, b, interp)
        return f, b
    with ThreadPoolExecutor(8) as pool:
        for i, (f, b) in enumerate(pool.map(pair, range(T - 1))):
            fw[i], bw[i + 1] = f, b
    fw[-1], bw[0] = -bw[-1], -fw[0]
    return np.clip(fw, -max_value, max_value), np.clip(bw, -max_value, max_value)


def _anvil_markers(field):
    """detect_anvils' seeds (reference detection.py:545-561)"""
    s = ndi.generate_binary_structure(3, 1) * np.array([0, 1, 0])[:, None, None].astype(bool)
    markers = ndi.label(field >= 1)[0].astype(np.int32)
    markers = markers * ndi.binary_erosion(markers != 0, structure=s)
    nan = np.isnan(field)
    bg = ndi.binary_erosion(np.logical_or(field <= 0, nan), structure=np.ones([3, 3, 3]), border_value=1)
    bg[nan] = True
    markers[bg] = -1
    return markers.astype(np.int32)


def test_config_S_whole_pipeline_matches_the_oracle():
    """Config S: 16 x 512 x 512 translating blobs, the settings of scripts/dcc_detect_goes.py:164-166
    (vr_steps=1, smoothing_passes=1, cubic).
    Flow (round 4): BIT FOR BIT the oracle's, raw and composed.  Rounds 1 - 3 accepted a maximum of 0.05 px with up to 2e-5
    of the components beyond 1e-4 for the composed flow: the raw vectors were ~1e-7 away (window sums in another order
    than OpenCV's running sum along the row, a reciprocal with a Newton step in the solve, inv(G) in closed form), and
    cv2's remap quantises sampling coordinates to 1/32 px, so a vector on a bin edge moved the refined one by ~1e-2.
    With the sequential row sums (k_fb_iter), OpenCV's solve and the oracle's inverse every stage is bit-identical, and so
    is the composition.
    Everything downstream is integer / bit-exact work and is compared bit for bit on the GPU's own flows."""
    import tobac_flow_amd.flow as tf
    from oracle import np_label, np_ops, ws_oracle
    from tobac_flow_amd.detection import get_combined_edge_field
    from tobac_flow_amd.utils import linearise_field
    rng = np.random.default_rng(20240601)
    bt = blob_sequence(rng, 16, 512, 512, n_blobs=8)
    bt[5, 100:150, 200:260] = np.nan                         # one frame with a NaN patch (to_8bit patching, edge field inf)
    flow = tf.create_flow(bt, model="Farneback", vr_steps=1, smoothing_passes=1, interp_method="cubic")
    want_f, want_b = _oracle_flow(bt, 1, 1, "cubic")
    for got, want in ((flow.forward_flow, want_f), (flow.backward_flow, want_b)):
        assert got.shape == (16, 512, 512, 2) and got.dtype == np.float32
        assert np.array_equal(np.isnan(got), np.isnan(want))
        d = np.abs(np.nan_to_num(got) - np.nan_to_num(want))
        print("config S composed flow vs oracle: max %.3g, %d of %d components differ" % (d.max(), int((d != 0).sum()), d.size))
        assert np.array_equal(got, want, equal_nan=True), (d.max(), int((d != 0).sum()))
    raw_f, raw_b = tf.calculate_flow(bt, "Farneback")           # no refinement, no smoothing
    want_rf, want_rb = _oracle_flow(bt, 0, 0, "linear", max_value=np.inf)
    assert np.array_equal(raw_f, want_rf, equal_nan=True) and np.array_equal(raw_b, want_rb, equal_nan=True)
    for i in (0, 5, 14):                                        # oracle stages on the GPU's raw vectors: bit-exact
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            p8 = np_ops.to_8bit(np_ops.linear_norm(bt[i:i + 2].copy()), 0, 1)
            f = np_ops.variational_refinement(p8[0], p8[1], raw_f[i])
            b = np_ops.variational_refinement(p8[1], p8[0], raw_b[i + 1])
            f, b = np_ops.smooth_flow_step(f, b, "cubic")
        assert np.array_equal(np.clip(f, -20, 20), flow.forward_flow[i], equal_nan=True)
        assert np.array_equal(np.clip(b, -20, 20), flow.backward_flow[i + 1], equal_nan=True)
    fwd, bwd = flow.forward_flow, flow.backward_flow
    field = linearise_field(bt, 270, 250).astype(np.float32)
    edges = get_combined_edge_field(flow, field)
    want_e = np_ops.sobel(field, fwd, bwd, "cubic", None, np.nan, "uphill")
    want_e[want_e > 0] += 1
    want_e = want_e - field
    want_e[np.isnan(field)] = np.inf
    assert edges.dtype == np.float64 and np.array_equal(edges, want_e)
    markers = _anvil_markers(field)
    assert markers.max() >= 3 and (markers == -1).any()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        labels = flow.watershed(edges, markers, connectivity=ndi.generate_binary_structure(3, 1), on_ambiguous="warn")
        labels2, report = tf.watershed(fwd, bwd, edges, markers, connectivity=1, return_ambiguous=True, on_ambiguous="warn")
    with warnings.catch_warnings():
        warnings.simplefilter("error")                               # the DEFAULT call: the reference's labels, nothing to report
        ref_mode = flow.watershed(edges, markers, connectivity=ndi.generate_binary_structure(3, 1))
    assert np.array_equal(labels, labels2)
    ideal = ws_oracle.watershed(fwd, bwd, edges, markers, None, 1, tie_mode=1)
    ref = ws_oracle.watershed(fwd, bwd, edges, markers, None, 1)     # tie_mode 0: the reference kernel's own semantics
    assert np.array_equal(ref_mode, ref)                             # default = reference order: the reference's labels, every voxel
    assert np.array_equal(labels, ideal)                             # opt-out mode: raster order of equal-valued markers ...
    assert not ((labels != ref) & ((report & 1) == 0)).any()         # ... and whatever differs from the reference is reported
    mask = field >= 0.5
    got_l = flow.label(mask, overlap=0.5, absolute_overlap=5)
    assert np.array_equal(got_l, np_label.flow_label(fwd, bwd, mask, overlap=0.5, absolute_overlap=5))


def test_config_C_24_frames_as_two_windows():
    """Config C: the 24-frame 1500 x 2500 stack, processed the production way: two 14-frame windows sharing four frames,
    each flooded on the GPU in reference order (against the reference kernel's C twin on the same window, bit for bit), stitched by the reference's
    overlap rule; every object of the overlap keeps one id in both windows."""
    import torch
    import tobac_flow_amd.flow as tf
    from oracle import ws_oracle
    from tobac_flow_amd.parallel import stitch_window_list, window_bounds
    from tobac_flow_amd.detection import get_combined_edge_field
    from tools.synth import anvil_inputs, blob_stack
    T, H, W, overlap = 24, 1500, 2500, 4
    bounds = window_bounds(T, 2, overlap)
    assert bounds == [(0, 14), (10, 24)]
    out = []
    for a, b in bounds:
        bt = blob_stack(b - a, H, W, seed=11, t0=a)
        fl = tf.create_flow(bt, vr_steps=1, smoothing_passes=1, interp_method="cubic")
        lin, markers = anvil_inputs(bt)
        # window-local object ids: connected components of the seeds (the bench's seeds are all 1)
        comp = ndi.label(markers.cpu().numpy() > 0)[0].astype(np.int32)
        comp[markers.cpu().numpy() < 0] = -1
        e = get_combined_edge_field(fl, lin, dtype=np.float32)       # detection.py:620-642 (NaN -> +inf)
        lab = fl.watershed(e, torch.from_numpy(comp).cuda(), connectivity=1, on_ambiguous="reference")
        ref = ws_oracle.watershed(fl.forward_flow.cpu().numpy(), fl.backward_flow.cpu().numpy(), e.cpu().numpy(), comp,
                                  None, 1, tie_mode=0)                # the reference kernel's own semantics
        assert np.array_equal(lab.cpu().numpy(), ref)
        out.append(lab)
    left, right = stitch_window_list(out, overlap=overlap)
    l, r = left[-overlap:][1:-1].cpu().numpy(), right[:overlap][1:-1].cpu().numpy()      # the compared frames
    both = (l > 0) & (r > 0)
    agree = (l[both] == r[both]).mean()
    # (windows see different temporal context and the rule joins only pairs above its thresholds: not 100 %)
    assert both.sum() > 10000 and agree > 0.9, agree
    # ids are global: contiguous from 1 over both windows
    ids = np.unique(np.concatenate([left.cpu().numpy()[left.cpu().numpy() > 0], right.cpu().numpy()[right.cpu().numpy() > 0]]))
    assert ids[0] == 1 and ids[-1] == len(ids)


def _windowed_detection(frames_of, T, n_windows, overlap, oracle):
    """The production procedure: per window create_flow -> edge field -> seeds -> watershed; returns the window labels."""
    import torch
    import tobac_flow_amd.flow as tf
    from tobac_flow_amd import ndimage_dev as nd
    from tobac_flow_amd.detection import get_combined_edge_field
    from tobac_flow_amd.parallel import window_bounds
    from tools.synth import anvil_inputs
    labs = []
    for a, b in window_bounds(T, n_windows, overlap):
        bt = frames_of(a, b)
        fl = tf.create_flow(bt, vr_steps=1, smoothing_passes=1, interp_method="cubic")
        lin, markers = anvil_inputs(bt)
        e = get_combined_edge_field(fl, lin, dtype=np.float32)       # detection.py:620-642 (NaN -> +inf)
        # window-local object ids: connected components of the seeds on the device (tf_label = scipy.ndimage.label)
        comp = torch.where(markers < 0, markers, nd.label(markers > 0)[0])
        with warnings.catch_warnings():
            warnings.simplefilter("error")                           # default call = the reference's own order: clean
            lab = fl.watershed(e, comp, connectivity=1)
        if oracle:
            from oracle import ws_oracle
            ref = ws_oracle.watershed(fl.forward_flow.cpu().numpy(), fl.backward_flow.cpu().numpy(), e.cpu().numpy(),
                                      comp.cpu().numpy(), None, 1, tie_mode=0)    # the reference kernel's own semantics
            assert np.array_equal(lab.cpu().numpy(), ref)
        labs.append(lab)
        del fl, e, lin, markers, bt
    return labs


def _check_stitched(windows, overlap):
    """after the stitch: ids contiguous from 1, background kept, and in the frames the linking compared every pixel
    that carries an object in both windows carries the SAME id in most cases (windows see different temporal context,
    and the rule joins only pairs above its thresholds)"""
    import torch
    top = 0
    for w in windows:
        assert bool((w != 0).all())                                           # mask=None: everything floods
        top = max(top, int(w.max()))
    seen = torch.zeros(top + 1, dtype=torch.bool, device=windows[0].device)
    for w in windows:
        seen[w[w > 0].long()] = True
    assert bool(seen[1:].all())
    for left, right in zip(windows[:-1], windows[1:]):
        l, r = left[-overlap:][1:-1], right[:overlap][1:-1]
        both = (l > 0) & (r > 0)
        assert int(both.sum()) > 0 and float((l[both] == r[both]).float().mean()) > 0.9
        assert bool(((l < 0) == (r < 0)).float().mean() > 0.95)


def test_config_F_procedure_on_a_stack_the_oracle_can_afford():
    """Config F's procedure (twelve 12-frame windows of a 144-frame stack, four shared frames, stitch) on 40 frames of
    384 x 512 as four windows: every window flood equals the sequential oracle bit for bit, then the stitch checks."""
    from tobac_flow_amd.parallel import stitch_window_list
    from tools.synth import blob_stack
    T, H, W, overlap = 40, 384, 512, 4
    labs = _windowed_detection(lambda a, b: blob_stack(b - a, H, W, seed=5, t0=a), T, 4, overlap, oracle=True)
    _check_stitched(stitch_window_list(labs, overlap=overlap), overlap)


def _stack_detection_full_size(T, H, W, n_windows, overlap, offsets=(0.0,), seed=20240601):
    """A BASELINE configuration at its stated size on ONE GPU, the way bench.py runs it: the flow of the T - 1 frame pairs
    once for the stack (create_flow in library-sized batches), then per channel and per window Flow.window -> SURVEY 8(d)
    seeds (component-labelled on the device) -> combined edge field -> watershed, and the stitch of the label ids over
    all windows of the channel.  The channels (F3: offsets 0 / -2 / -4 K) share the ONE Flow and are processed one after
    the other, so that only one channel's label volumes are resident.  Checked: _check_stitched per channel."""
    import torch
    import tobac_flow_amd.flow as tf
    from tobac_flow_amd import _lib
    from tobac_flow_amd.detection import get_combined_edge_field
    from tobac_flow_amd.parallel import stitch_window_list, window_bounds
    from tools.synth import anvil_seeds, blob_stack
    bounds = window_bounds(T, n_windows, overlap)
    assert len(bounds) == n_windows and bounds[0][0] == 0 and bounds[-1][1] == T
    bt = torch.empty((T, H, W), dtype=torch.float32, device="cuda")
    for f0 in range(0, T, 12):
        f1 = min(f0 + 12, T)
        bt[f0:f1] = blob_stack(f1 - f0, H, W, seed=seed, t0=f0)
    flow_all = tf.create_flow(bt, vr_steps=1, smoothing_passes=1, interp_method="cubic")
    _lib.release_workspaces()                                          # the flow scratch (tens of GB) is not needed any more
    assert flow_all.shape == (T, H, W)
    n_objects = []
    for off in offsets:
        labs = []
        for a, b in bounds:
            fl = flow_all.window(a, b)
            lin, seeds = anvil_seeds(bt[a:b] + off if off else bt[a:b])
            e = get_combined_edge_field(fl, lin, dtype=np.float32)   # detection.py:620-642 (NaN -> +inf)
            with warnings.catch_warnings():
                warnings.simplefilter("error")                       # the default call: reference order, nothing to warn about
                labs.append(fl.watershed(e, seeds, connectivity=1))
            del fl, lin, seeds, e
        out = stitch_window_list(labs, overlap=overlap)
        del labs
        _check_stitched(out, overlap)
        n_objects.append(max(int(w.max()) for w in out))
        del out
    return n_objects


def test_config_F_144_full_disk_frames_as_twelve_windows():
    """Config F itself on ONE GPU: 144 frames of 5424 x 5424 cut from one sequence, twelve overlapping windows through
    the whole hot path, label ids stitched over all of them (stitch_window_list: tf_window_overlap_pairs + union-find +
    tf_apply_lut).  ~12 x 16 x 29.4 M voxels of labels stay resident (22 GB), the stack's flow 68 GB."""
    n = _stack_detection_full_size(144, 5424, 5424, 12, 4)
    assert n[0] > 100


def test_config_V_288_frames_as_24_windows():
    """Config V at its stated size on ONE GPU: SEVIRI full-disk, 288 frames of 3712 x 3712 as twenty-four windows
    (BASELINE.json shards them over 8 GPUs; the windows are the same, here one GPU visits them in turn)."""
    n = _stack_detection_full_size(288, 3712, 3712, 24, 4)
    assert n[0] > 100


def test_config_F3_288_frames_three_channels():
    """Config F3 at its stated size on ONE GPU: 288 full-disk frames, three channel stacks (offsets 0 / -2 / -4 K,
    SURVEY.md 8d) as three independent detections sharing ONE Flow (136 GB of flow vectors resident), twenty-four
    windows each.  A colder channel (more negative offset) has more seeded area, so its object count differs."""
    n = _stack_detection_full_size(288, 5424, 5424, 24, 4, offsets=(0.0, -2.0, -4.0))
    assert len(n) == 3 and all(v > 100 for v in n) and len(set(n)) > 1


# ---- F3 -------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def three_channels():
    import tobac_flow_amd.flow as tf
    from test_gpu_detection import FakeDataArray
    rng = np.random.default_rng(77)
    bt = blob_sequence(rng, 8, 160, 200, n_blobs=6, vmax=1.5, noise=0.4)
    # growing cold tops: deepen the blobs with time so that the growth metrics exceed their thresholds
    grow = np.linspace(0.6, 1.4, 8, dtype=np.float32)[:, None, None]
    bt = (290.0 - (290.0 - bt) * grow).astype(np.float32)
    flow = tf.create_flow(bt, vr_steps=1, smoothing_passes=1, interp_method="cubic")
    chans = [FakeDataArray(((250.0 - (bt + off)) / 2.0 - 10.0).astype(np.float32), minutes=5) for off in (0.0, -2.0, -4.0)]
    return dict(tf=tf, bt=FakeDataArray(bt, minutes=5), flow=flow, chans=chans)


def test_config_F3_three_channels_share_one_flow(three_channels):
    """F3: three channel stacks, ONE Flow, three independent growth detections.  The device-resident recipe
    (detect_growth_markers) against the same recipe with the reference's SciPy glue between the device operators."""
    from tobac_flow_amd.detection import _detect_growth_markers_host, detect_growth_markers
    flow = three_channels["flow"]
    n_marked = 0
    for wvd in three_channels["chans"]:
        try:
            got_s, got_m = detect_growth_markers(flow, wvd)
        except ValueError:
            with pytest.raises(ValueError):                    # no survivor: both paths do what the reference's call does
                _detect_growth_markers_host(flow, wvd)
            continue
        want_s, want_m = _detect_growth_markers_host(flow, wvd)
        assert np.array_equal(np.isnan(got_s), np.isnan(want_s)) and np.array_equal(np.nan_to_num(got_s), np.nan_to_num(want_s))
        assert np.array_equal(got_m, want_m)
        n_marked += int(got_m.max() > 0)
    assert n_marked >= 1


def test_config_F3_multichannel_growth_markers_match_the_oracle_recipe(three_channels):
    """detect_growth_markers_multichannel (reference detection.py:203-254) against the recipe assembled from ORACLE
    pieces: Flow.diff / filtered_tdiff from oracle/np_ops (numpy restatement of convolve.py), flow labelling from the
    loop-form oracle of label.py, SciPy for the ndimage steps, the legacy length / multi-mask filter restated inline."""
    from oracle import np_label, np_ops
    from tobac_flow_amd.detection import detect_growth_markers_multichannel, get_curvature_filter
    from tobac_flow_amd.utils import get_time_diff_from_coord
    flow, bt, wvd = three_channels["flow"], three_channels["bt"], three_channels["chans"][0]
    fwd, bwd = flow.forward_flow, flow.backward_flow
    got_w, got_b, got_m = detect_growth_markers_multichannel(flow, wvd, bt, min_length=2, lower_threshold=0.05,
                                                             upper_threshold=0.1)
    t_struct = np.zeros([3, 3, 3])
    t_struct[:, 1, 1] = 1
    nanmean0 = lambda x: np.nanmean(x, 0)

    def smoothed(field):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            raw = np_ops.diff(np.asarray(field), fwd, bwd) / get_time_diff_from_coord(field.t)[:, np.newaxis, np.newaxis]
            return np_ops.convolve(raw, fwd, bwd, t_struct, "linear", np.float32, np.nan, nanmean0)
    want_w, want_b = smoothed(wvd), smoothed(bt)
    for g, w in ((got_w, want_w), (got_b, want_b)):
        assert np.array_equal(np.isnan(g), np.isnan(w)) and np.array_equal(np.nan_to_num(g), np.nan_to_num(w))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        markers = np.logical_or((want_w * get_curvature_filter(np.asarray(wvd))) >= 0.05,
                                (want_b * get_curvature_filter(np.asarray(bt), direction="positive")) <= -0.05)
    markers = np_label.flow_label(fwd, bwd, ndi.binary_opening(markers, structure=ndi.generate_binary_structure(2, 1)[np.newaxis, ...]),
                                  overlap=0.5)
    assert markers.max() > 0
    masks = [want_w >= 0.1, want_b <= -0.1, np.asarray(wvd) > -5]
    want_m = np.zeros_like(markers)                           # analysis.py:182-201 restated
    counter = 1
    objs = ndi.find_objects(markers)
    for i, sl in enumerate(objs):
        if sl is None:
            continue
        sel = markers == i + 1
        if sl[0].stop - sl[0].start >= 2 and all(m[sel].any() for m in masks):
            want_m[sel] = counter
            counter += 1
    assert np.array_equal(np.asarray(got_m), want_m)
