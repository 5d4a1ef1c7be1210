"""Edge cases through the C ABI: degenerate shapes, empty / saturated inputs, extreme flows, NaN / inf."""
import os
import warnings

import numpy as np
import pytest
import scipy.ndimage as ndi

from helpers import rand_field, rand_flow, seeds

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tf():
    import tobac_flow_amd.flow as flow
    return flow


def _same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype
    if a.dtype.kind == "f":
        assert np.array_equal(np.isnan(a), np.isnan(b))
        a, b = np.nan_to_num(a, posinf=1e300, neginf=-1e300), np.nan_to_num(b, posinf=1e300, neginf=-1e300)
    assert np.array_equal(a, b), f"{int((a != b).sum())} elements differ"


@pytest.mark.parametrize("shape", [(1, 1, 1), (1, 7, 1), (1, 1, 9), (2, 3, 4), (1, 5, 5), (3, 2, 70)])
@pytest.mark.parametrize("method", ["nearest", "linear", "cubic"])
def test_sobel_and_convolve_degenerate_shapes(tf, shape, method):
    from oracle import np_ops
    rng = np.random.default_rng(sum(shape))
    data = rng.normal(size=shape).astype(np.float32)
    fwd = rng.uniform(-2.5, 2.5, shape + (2,)).astype(np.float32)
    bwd = rng.uniform(-2.5, 2.5, shape + (2,)).astype(np.float32)
    fl = tf.Flow(fwd, bwd)
    _same(fl.sobel(data, method=method, direction="uphill"), np_ops.sobel(data, fwd, bwd, method, None, np.nan, "uphill"))
    _same(fl.convolve(data, method=method), np_ops.convolve(data, fwd, bwd, method=method))
    _same(fl.diff(data, method=method), np_ops.diff(data, fwd, bwd, method))


def test_convolve_fill_value_number_and_border_taps(tf):
    """numeric fill value exercises the straddling-border formulas (cval + sum (S - cval) w)"""
    from oracle import np_ops
    rng = np.random.default_rng(9)
    shape = (3, 12, 14)
    data = rng.normal(size=shape).astype(np.float32)
    fwd = rng.uniform(-6, 6, shape + (2,)).astype(np.float32)
    bwd = rng.uniform(-6, 6, shape + (2,)).astype(np.float32)
    st = ndi.generate_binary_structure(3, 3)
    for method in ("nearest", "linear", "cubic"):
        got = tf.Flow(fwd, bwd).convolve(data, structure=st, method=method, fill_value=-3.5)
        want = np_ops.convolve(data, fwd, bwd, st, method, np.float32, -3.5)
        _same(got, want)


def test_to8bit_special_values(tf):
    from oracle import np_ops
    from tobac_flow_amd import _lib
    from tobac_flow_amd.utils.normalisation_utils import to_8bit_pair_dev
    base = np.linspace(200, 300, 6 * 8, dtype=np.float32).reshape(2, 6, 4)
    cases = [base.copy() for _ in range(5)]
    cases[1][:] = 7.0                                  # vmin == vmax -> factor 0
    cases[2][0, 0, 0] = np.inf                         # infinite range -> everything non-finite or 0
    cases[3][1, 2, :] = -np.inf
    cases[4][0] = np.nan                               # one frame entirely NaN: patched from the other
    for pair in cases:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = np_ops.to_8bit(np_ops.linear_norm(pair.copy()), 0, 1)
        a, b = to_8bit_pair_dev(_lib.to_dev(pair[0]), _lib.to_dev(pair[1]))
        assert np.array_equal(a.cpu().numpy(), want[0]) and np.array_equal(b.cpu().numpy(), want[1])


def test_farneback_reference_test_blob(tf):
    """the 15 x 10 blob of the reference's tests/test_flow.py:203-204 (single pyramid level: < 32 px)"""
    from test_gpu_parity import _oracle_farneback
    xx, yy = np.meshgrid(np.arange(15), np.arange(10))
    blob = tf.to_8bit((7 ** 2 - (xx - 7) ** 2) * (4.5 ** 2 - (yy - 4.5) ** 2))
    nxt = np.roll(blob, 1, 1)
    f, b = tf.calculate_flow_frame(blob, nxt, tf.select_of_model("Farneback"))
    assert f.shape == (10, 15, 2)
    assert np.array_equal(f, _oracle_farneback(blob, nxt))
    assert np.array_equal(b, _oracle_farneback(nxt, blob))
    z, _ = tf.calculate_flow_frame(blob, blob, tf.select_of_model("Farneback"))
    assert np.array_equal(z, _oracle_farneback(blob, blob)) and np.allclose(z, 0, atol=0.05)


@pytest.mark.parametrize("shape", [(33, 47), (64, 31), (100, 333),
                                   # the iteration kernel's strip geometry (116 output columns per strip, farneback.hip FBI_OW): exactly
                                   # one / two strips, a last strip of one column, of 115, three strips with a short one
                                   (70, 116), (70, 117), (45, 232), (45, 233), (40, 347), (260, 349)])
def test_farneback_odd_sizes_match_oracle(tf, shape):
    from test_gpu_parity import _oracle_farneback
    rng = np.random.default_rng(shape[1])
    a = (ndi.gaussian_filter(rng.normal(size=shape), 3) * 400 + 128).clip(0, 255).astype(np.uint8)
    b = np.roll(a, (1, 2), (0, 1))
    f, bk = tf.calculate_flow_frame(a, b, tf.select_of_model("Farneback"))
    assert np.array_equal(f, _oracle_farneback(a, b))
    assert np.array_equal(bk, _oracle_farneback(b, a))


@pytest.mark.parametrize("shape", [(1, 1), (1, 9), (7, 1), (2, 2), (3, 5), (13, 13), (5, 40), (40, 6), (40, 9), (9, 40),
                                   (7, 8), (9, 9), (40, 11), (12, 300), (300, 12)])
def test_farneback_tiny_and_degenerate_images_match_oracle(tf, shape):
    """Images smaller than the 13 x 13 window, than the expansion radius, and one pixel wide / tall (no bilinear patch
    exists: every gather takes the out-of-image branch).  Below 10 pixels OpenCV's border attenuation test
    (unsigned)(x - 5) >= (unsigned)(W - 10) wraps around and skips some border columns / rows: the widths and heights
    6 ... 11 pin that behaviour (a kernel that scaled every border column was 1.4 px off here)."""
    from test_gpu_parity import _oracle_farneback
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    big = ndi.gaussian_filter(rng.normal(size=(shape[0] + 8, shape[1] + 8)), 1.5)
    big = ((big - big.min()) / (np.ptp(big) + 1e-9) * 255).astype(np.uint8)
    a = np.ascontiguousarray(big[4:4 + shape[0], 4:4 + shape[1]])
    b = np.ascontiguousarray(big[3:3 + shape[0], 5:5 + shape[1]])           # shifted by (1, -1)
    f, bk = tf.calculate_flow_frame(a, b, tf.select_of_model("Farneback"))
    want_f, want_b = _oracle_farneback(a, b), _oracle_farneback(b, a)
    assert f.shape == shape + (2,) and np.isfinite(f).all() and np.isfinite(bk).all()
    assert np.array_equal(f, want_f), np.max(np.abs(f - want_f))
    assert np.array_equal(bk, want_b), np.max(np.abs(bk - want_b))


def test_create_flow_short_series(tf):
    rng = np.random.default_rng(1)
    two = rng.normal(size=(2, 40, 48)).astype(np.float32)
    fl = tf.create_flow(two)
    assert fl.shape == (2, 40, 48) and np.isfinite(fl.forward_flow).all()
    assert np.array_equal(fl.forward_flow[1], -fl.backward_flow[1]) and np.array_equal(fl.backward_flow[0], -fl.forward_flow[0])
    one = tf.create_flow(two[:1])                      # no frame pair: the reference leaves NaN
    assert one.shape == (1, 40, 48) and np.isnan(one.forward_flow).all() and np.isnan(one.backward_flow).all()


def test_watershed_degenerate_volumes(tf):
    from oracle import ws_oracle
    z1 = np.zeros((1, 1, 1, 2), np.float32)
    assert tf.watershed(z1, z1, np.zeros((1, 1, 1), np.float32), np.array([[[3]]], np.int32)).tolist() == [[[3]]]
    assert tf.watershed(z1, z1, np.zeros((1, 1, 1), np.float32), np.zeros((1, 1, 1), np.int32)).tolist() == [[[0]]]
    rng = np.random.default_rng(4)
    shape = (1, 17, 23)                                # T = 1: no temporal neighbours at all
    f = rand_field(rng, shape)
    m = seeds(rng, shape, 4)
    fl = rand_flow(rng, shape, 3.0)
    for conn in (1, 2, 3):
        assert np.array_equal(tf.watershed(fl, fl, f, m, connectivity=conn), ws_oracle.watershed(fl, fl, f, m, None, conn))
    allm = np.arange(1, 1 + np.prod(shape), dtype=np.int32).reshape(shape)       # every pixel is a seed
    assert np.array_equal(tf.watershed(fl, fl, f, allm), allm)
    nomask = np.zeros(shape, bool)                                              # nothing floodable, seeds kept
    assert np.array_equal(tf.watershed(fl, fl, f, m, mask=nomask), m)


@pytest.mark.parametrize("seed", range(4))
def test_watershed_extreme_flows_and_seeds_outside_mask(tf, seed):
    """flows at the create_flow clip bound (+-20 px), half-integer flows (round half to even), NaN flows
    are rounded to 0 here (the reference would crash on them), markers that lie outside the mask still seed"""
    from oracle import ws_oracle
    rng = np.random.default_rng(200 + seed)
    shape = (4, 45, 50)
    field = rand_field(rng, shape)
    markers = seeds(rng, shape, 8)
    mask = ndi.gaussian_filter(rng.normal(size=shape), (0, 2, 2)) > -0.1
    fwd = rng.choice(np.array([-20, -7.5, -2.5, -0.5, 0, 0.5, 1.5, 3.5, 20], np.float32), size=shape + (2,))
    bwd = rng.choice(np.array([-20, -7.5, -2.5, -0.5, 0, 0.5, 1.5, 3.5, 20], np.float32), size=shape + (2,))
    conn = [1, 3][seed % 2]
    got = tf.watershed(fwd, bwd, field, markers, mask=mask, connectivity=conn)
    want = ws_oracle.watershed(fwd, bwd, field, markers, mask, conn)
    assert np.array_equal(got, want), f"{int((got != want).sum())} px differ"
    assert np.array_equal(got[markers != 0], markers[markers != 0])


def test_watershed_custom_structure_and_inf_field(tf):
    from oracle import ws_oracle
    rng = np.random.default_rng(8)
    shape = (3, 30, 34)
    field = rand_field(rng, shape)
    field[rng.random(shape) < 0.05] = np.inf
    markers = seeds(rng, shape, 6)
    st = np.zeros((3, 3, 3), bool)
    st[1] = True                       # in-plane 8-neighbourhood only
    st[0, 1, 1] = st[2, 1, 1] = True   # + the two temporal neighbours
    fwd, bwd = rand_flow(rng, shape, 2.0), rand_flow(rng, shape, 2.0)
    got = tf.watershed(fwd, bwd, field, markers, connectivity=st)
    want = ws_oracle.watershed(fwd, bwd, field, markers, None, st)
    assert np.array_equal(got, want)


def test_watershed_ex_rejects_unknown_flags_and_chain_depth_one_ignores_the_hint(tf):
    """C ABI: flags other than TF_WS_SKIP_FAST_PATH / TF_WS_REFERENCE_ORDER are TF_EINVAL (TF_WS_DEFER_SWEEPS = 4 belongs to
    tf_watershed_begin alone); with chain_depth 1 there are no chain phases to skip to, so the hint is ignored and the root phase runs."""
    import torch
    from tobac_flow_amd import _lib
    from tobac_flow_amd.watershed import neighbour_offsets, watershed_dev
    rng = np.random.default_rng(4)
    shape = (2, 12, 14)
    T, H, W = shape
    f = torch.from_numpy(rand_field(rng, shape).astype(np.float32)).cuda()
    m = torch.from_numpy(seeds(rng, shape, 3).astype(np.int32)).cuda()
    fl = torch.zeros(shape + (2,), dtype=torch.float32, device="cuda")
    out = torch.empty(shape, dtype=torch.int32, device="cuda")
    nbr = np.ascontiguousarray(neighbour_offsets(1, 3), np.int8)
    L = _lib.lib()
    ws = torch.empty(L.tf_watershed_workspace_bytes(T, H, W, len(nbr), 3, 0), dtype=torch.uint8, device="cuda")
    st = np.zeros(8, np.int64)
    args = lambda flags: (_lib.ptr(f), _lib.ptr(m), None, _lib.ptr(fl), _lib.ptr(fl), T, H, W, nbr.ctypes.data_as(_lib._P),
                          len(nbr), 3, flags, _lib.ptr(out), _lib.ptr(ws), ws.numel(), st.ctypes.data_as(_lib._P), None)
    assert L.tf_watershed_ex(*args(4)) == -1 and b"unknown flag" in L.tf_last_error()
    assert L.tf_watershed_ex(*args(8)) == -1 and b"unknown flag" in L.tf_last_error()
    assert L.tf_watershed_ex(*args(0)) == 0
    probe = out.cpu().numpy().copy()
    assert L.tf_watershed_ex(*args(2)) == 0 and np.array_equal(out.cpu().numpy(), probe)      # TF_WS_REFERENCE_ORDER: nothing to reorder here
    assert L.tf_watershed_ex(*args(1)) == 0 and st[5] == -1 and st[1] == 0
    assert np.array_equal(out.cpu().numpy(), probe)
    st1 = {}
    depth1 = watershed_dev(fl, fl, f, m, None, nbr, 1, st1, expect_conflict=True).cpu().numpy()
    assert st1["sweeps"][1] > 0 and st1["sweeps"][5] == 0        # root phase ran, nothing skipped
    assert np.array_equal(depth1[m.cpu().numpy() != 0], m.cpu().numpy()[m.cpu().numpy() != 0])


def test_apply_global_lut_gpu_equals_host_path_and_keeps_negative_ids():
    """The relabelling step of the multi-GPU stitch (parallel.apply_global_lut): the library's one-pass gather on the GPU
    must equal the torch implementation the gloo rehearsals use, including zero and negative ids."""
    import torch
    from tobac_flow_amd.parallel import apply_global_lut
    rng = np.random.default_rng(12)
    labels = rng.integers(-3, 40, size=(3, 33, 41)).astype(np.int32)
    lut = np.concatenate([[0], rng.permutation(np.arange(1, 40))]).astype(np.int64)
    want = apply_global_lut(torch.from_numpy(labels), lut).numpy()
    got = apply_global_lut(torch.from_numpy(labels).cuda(), lut).cpu().numpy()
    assert got.dtype == np.int32 and np.array_equal(got, want)
    assert np.array_equal(got[labels <= 0], labels[labels <= 0]) and (labels < 0).any()
    nonneg = np.abs(labels)
    assert np.array_equal(apply_global_lut(torch.from_numpy(nonneg).cuda(), lut).cpu().numpy(),
                          apply_global_lut(torch.from_numpy(nonneg), lut).numpy())


def test_flow_diagnostics_run_and_satisfy_their_identities(tf):
    """get_forward_warp / flow_diff_mse_estimate / get_flow_residual / flow_residual_mse_estimate / time_flow
    (reference: flow.py:571-640): thin wrappers over convolve and calculate_flow_2, exercised on a drifting field."""
    class DA:                                              # the one xarray attribute these helpers touch
        def __init__(self, data):
            self.data = data

    rng = np.random.default_rng(31)
    base = ndi.gaussian_filter(rng.normal(size=(70, 100)), 3).astype(np.float32) * 40 + 260
    stack = np.stack([np.roll(base, (i, -2 * i), (0, 1)) for i in range(4)])
    flow = tf.create_flow(stack, smoothing_passes=1)
    da = DA(stack)
    fw = tf.get_forward_warp(da, flow)
    one_tap = np.zeros([3, 3, 3], bool)
    one_tap[2, 1, 1] = True
    assert fw.shape == stack.shape and np.array_equal(np.nan_to_num(fw), np.nan_to_num(flow.convolve(stack, one_tap)[0]))
    win = (slice(None), slice(15, -15), slice(15, -15))
    aligned = np.nanmean(np.abs(fw[:3][win] - stack[:3][win]))          # next frame pulled back along the flow vs this frame
    unaligned = np.mean(np.abs(stack[1:][win] - stack[:3][win]))
    assert unaligned > 0.5 and aligned < 0.25 * unaligned               # (integer drift + 1/32 px coordinate grid: often exactly 0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        all_mse, cold_mse = tf.flow_diff_mse_estimate(da, flow)
        resid = tf.get_flow_residual(da, flow, vr_steps=0)
        r_all, r_cold = tf.flow_residual_mse_estimate(da, flow, vr_steps=0)
    assert np.isfinite(all_mse) and all_mse >= 0 and (np.isnan(cold_mse) or cold_mse >= 0)
    assert resid.shape == stack.shape + (2,) and np.isfinite(r_all) and r_all >= 0
    zero = tf.Flow(np.zeros(stack.shape + (2,), np.float32), np.zeros(stack.shape + (2,), np.float32))
    fw0 = tf.get_forward_warp(da, zero)
    # zero flow: the forward warp is the next frame -- except the last row and column, where the bilinear patch
    # reaches outside and the NaN border value poisons the sum even at weight zero (cv2.remap, BORDER_CONSTANT)
    assert np.array_equal(fw0[:-1, :-1, :-1], stack[1:, :-1, :-1])
    assert np.isnan(fw0[:-1, -1, :]).all() and np.isnan(fw0[:-1, :, -1]).all() and np.isnan(fw0[-1]).all()
    assert tf.time_flow(stack, vr_steps=0, smoothing_passes=0) > 0


def test_stitch_window_list_on_gpu_equals_host_result():
    """the single-process window stitch with GPU tensors (tf_apply_lut relabelling) against the same call on CPU tensors"""
    import torch
    from tobac_flow_amd.parallel import stitch_window_list, window_bounds
    rng = np.random.default_rng(3)
    truth, n = ndi.label(ndi.gaussian_filter(rng.normal(size=(10, 40, 50)), (1.0, 1.5, 1.5)) > 0.03)
    assert n >= 4
    wins = []
    for a, b in window_bounds(10, 3):
        w = truth[a:b].astype(np.int32)
        ids = np.unique(w[w > 0])
        perm = np.zeros(w.max() + 1, np.int32)
        perm[ids] = rng.permutation(len(ids)) + 1
        wins.append(perm[w])
    host = [x.numpy() for x in stitch_window_list([torch.from_numpy(w) for w in wins], min_overlap=1, overlap=1)]
    dev = [x.cpu().numpy() for x in stitch_window_list([torch.from_numpy(w).cuda() for w in wins], min_overlap=1, overlap=1)]
    for h, d in zip(host, dev):
        assert d.dtype == np.int32 and np.array_equal(h, d)
    # the reference's rule (>= 5 px, >= 0.5, outer common frames dropped) on windows that share four frames
    wins = [truth[a:b].astype(np.int32) for a, b in window_bounds(10, 2, overlap=4)]
    wins[1] = np.where(wins[1] > 0, (wins[1] * 5) % 23 + 1, 0).astype(np.int32)
    host = [x.numpy() for x in stitch_window_list([torch.from_numpy(w) for w in wins], overlap=4)]
    dev = [x.cpu().numpy() for x in stitch_window_list([torch.from_numpy(w).cuda() for w in wins], overlap=4)]
    for h, d in zip(host, dev):
        assert np.array_equal(h, d)


def test_watershed_refuses_a_nan_field_where_it_matters(tf):
    """`smaller()` of the reference heap (_watershed.pyx:161-164) is not an order on NaN: with NaN keys its pop order
    depends on the heap's array state and has no closed form -- the library refuses (ValueError) instead of returning
    labels that look plausible.  A NaN strictly inside a seed region (never compared) is harmless."""
    rng = np.random.default_rng(3)
    shape = (2, 20, 24)
    field = rand_field(rng, shape)
    markers = np.zeros(shape, np.int32)
    markers[:, 2:8, 2:8] = 1
    markers[:, 14:18, 15:20] = 2
    z = np.zeros(shape + (2,), np.float32)
    ok = tf.watershed(z, z, field, markers)
    inside = field.copy()
    inside[0, 4, 4] = np.nan                                   # interior of seed 1: all neighbours are seeds
    assert np.array_equal(tf.watershed(z, z, inside, markers), ok)
    for where in ((1, 10, 12), (0, 2, 2)):                     # a floodable pixel; a seed pixel that touches floodable ones
        bad = field.copy()
        bad[where] = np.nan
        with pytest.raises(ValueError, match="NaN"):
            tf.watershed(z, z, bad, markers)


def test_development_switches_do_not_change_results(tf, tmp_path):
    """Library switches that select an alternative kernel for the same arithmetic (development switches) must
    give bit-identical flows: the generic polynomial-expansion kernel instead of the register-blocked polyN = 5 one, the
    one-launch-per-half-sweep SOR instead of the fused tile kernel, the two-pass blur instead of the fused 3 x 3 one.
    The switches are read once per process, so the alternative runs in a child process."""
    import subprocess
    import sys
    rng = np.random.default_rng(21)
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import tobac_flow_amd.flow as tf; "
            "a = np.load(%r); f, b = tf.calculate_flow(a, 'Farneback', vr_steps=1); np.save(%r, np.stack([f, b]))")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # a width that is no multiple of four (byte staging with reflected borders) and one that is (word staging; deep
    # pyramid: blur kernels up to 39 taps)
    for tag, shape in (("a", (3, 150, 210)), ("b", (2, 272, 544))):
        a = ndi.gaussian_filter(rng.normal(size=shape), (0, 2.5, 2.5)).astype(np.float32) * 30 + 250
        np.save(tmp_path / f"in_{tag}.npy", a)
        fw, bw = tf.calculate_flow(a, "Farneback", vr_steps=1)
        # (round 4) the iteration kernel's scheduling knobs: both directions of a strip in one workgroup, strips that let their
        # left neighbour get ahead before they start, one column group of tickets whatever the launch size, one ticket list
        # instead of one per XCD, every chain in two parts one row group apart instead of walked whole by one lane
        for var in ("TF_FB_POLYEXP_GENERIC", "TF_VR_SOR_SWEEPS", "TF_VR_WEIGHTS_PASS", "TF_FB_BLUR_TWOPASS", "TF_FB_BLUR_NO_LDS",
                    "TF_FBI_JOIN_DIRECTIONS", "TF_FBI_SLACK_ROWS", "TF_FBI_COLUMN_GROUPS", "TF_FBI_ONE_TICKET_LIST", "TF_FBI_TWO_PART_CHAIN"):
            out = tmp_path / f"{var}_{tag}.npy"
            # (calculate_flow asks for the two-part chain when it hands out no windows: the switch's OTHER value is the test here)
            env = dict(os.environ, **{var: "13" if var == "TF_FBI_SLACK_ROWS" else ("0" if var == "TF_FBI_TWO_PART_CHAIN" else "1")})
            subprocess.check_call([sys.executable, "-c", code % (root, str(tmp_path / f"in_{tag}.npy"), str(out))], env=env)
            alt = np.load(out)
            assert np.array_equal(alt[0], fw, equal_nan=True) and np.array_equal(alt[1], bw, equal_nan=True), (var, tag)
        # the one switch that is NOT bit-neutral: round 3's iteration kernel (window sums as a tree, reciprocal + Newton step),
        # kept for A/B timing -- within the north star's 1e-4 px of the sequential form on the raw vectors
        raw = tf.calculate_flow(a, "Farneback")
        out = tmp_path / f"tree_{tag}.npy"
        code_raw = code.replace("vr_steps=1", "vr_steps=0")
        subprocess.check_call([sys.executable, "-c", code_raw % (root, str(tmp_path / f"in_{tag}.npy"), str(out))], env=dict(os.environ, TF_FB_ROW_SUMS_TREE="1"))
        alt = np.load(out)
        d = max(np.nanmax(np.abs(alt[0] - raw[0])), np.nanmax(np.abs(alt[1] - raw[1])))
        assert d <= 1e-4, d                                  # (often 0 at this size: the orders differ in the last bits of a double)


def test_a_starved_chain_is_an_error_not_a_flow_of_nans(tf, tmp_path):
    """VERDICT r4 weak 4 / ADVICE r4: a row-sum chain of k_fb_iter whose left neighbour's hand-over words never arrive gives up
    after a bounded number of polls and continues with NaN -- that must reach the caller as an error (TF_ESTARVED ->
    TobacFlowHipError), never as a Flow with NaN rows and rc 0.  (i) Forced on the device: a child process in which the strips
    never store their words (TF_FBI_SEQ_ABLATE=64) and the poll bound is 4 (TF_FBI_POLL_LIMIT): every chain right of strip 0
    starves, create_flow raises.  (ii) The host path alone: the status word set from the host (tf_farneback_debug_set_starved)
    makes the next batch call return TF_ESTARVED on entry, the report clears it, and the following call computes the same flow
    as before."""
    import subprocess
    import sys
    from tobac_flow_amd import _lib
    rng = np.random.default_rng(5)
    a = ndi.gaussian_filter(rng.normal(size=(3, 140, 300)), (0, 2.5, 2.5)).astype(np.float32) * 30 + 250     # three strips of 116 columns
    np.save(tmp_path / "in.npy", a)
    want = tf.calculate_flow(a, "Farneback")
    assert np.isfinite(want[0]).all() and np.isfinite(want[1]).all()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import tobac_flow_amd.flow as tf; from tobac_flow_amd._lib import TobacFlowHipError\n"
            "a = np.load(%r)\n"
            "try:\n"
            "    tf.create_flow(a, 'Farneback')\n"
            "except TobacFlowHipError as e:\n"
            "    assert 'gave up' in str(e) and '-6' in str(e), str(e); print('STARVED-RAISED')\n"
            "else:\n"
            "    print('NO-ERROR')\n") % (root, str(tmp_path / "in.npy"))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, TF_FBI_SEQ_ABLATE="64", TF_FBI_POLL_LIMIT="4"),
                         capture_output=True, text=True, timeout=600)
    assert "STARVED-RAISED" in out.stdout, (out.stdout, out.stderr[-2000:])
    # (ii) host path in this process: the status word of a flow's own (round 6: one per model object = per flow) set from the
    # host makes that flow's next batch call return TF_ESTARVED on entry, the report clears it, the following call computes
    # the same flow as before; the device's shared word (status_slot 0) behaves the same for callers of the bare C ABI
    from tobac_flow_amd.utils.normalisation_utils import linear_norm
    L = _lib.lib()
    p8 = tf.to_8bit(linear_norm(a[:2]), 0, 1)
    model = tf.select_of_model("Farneback")
    f0 = model.calc(p8[0], p8[1], None)
    slot = model.params.status_slot
    assert slot > 0 and L.tf_farneback_status_check(slot) == 0 and L.tf_farneback_check() == 0
    assert L.tf_farneback_debug_set_starved_slot(slot) == 0
    with pytest.raises(_lib.TobacFlowHipError, match="gave up"):
        model.calc(p8[0], p8[1], None)
    assert L.tf_farneback_status_check(slot) == 0                         # reported once, cleared
    assert np.array_equal(model.calc(p8[0], p8[1], None), f0)
    assert L.tf_farneback_debug_set_starved() == 0
    assert np.array_equal(model.calc(p8[0], p8[1], None), f0)             # (the shared word is not this flow's)
    assert L.tf_farneback_check() == _lib.TF_ESTARVED and L.tf_farneback_check() == 0
    again = tf.calculate_flow(a, "Farneback")
    assert np.array_equal(again[0], want[0]) and np.array_equal(again[1], want[1])
    # (iii) device-resident input: create_flow returns before the device is done, its check is deferred to an event -- the report
    # of a launch that finishes later arrives at the results the Flow hands out (round 6: Flow.sobel / watershed / ... wait for
    # it) or at Flow.check().  Round 6: every flow reports to a status word of its own (tf_farneback_status_*)
    import torch
    ad = torch.from_numpy(a).cuda()
    fl = tf.create_flow(ad, "Farneback")
    assert fl._pending_check is not None
    slot = fl._pending_check.status_slot
    assert slot > 0
    torch.cuda.synchronize()
    assert L.tf_farneback_debug_set_starved_slot(slot) == 0               # "a chain of those launches gave up"
    with pytest.raises(_lib.TobacFlowHipError, match="gave up"):
        fl.sobel(ad, direction="uphill", method="cubic")                 # first use after the launches have finished
    assert L.tf_farneback_status_check(slot) == 0
    fl.sobel(ad, direction="uphill", method="cubic")                     # reported once
    fl2 = tf.create_flow(ad, "Farneback")
    assert L.tf_farneback_debug_set_starved_slot(fl2._pending_check.status_slot) == 0
    with pytest.raises(_lib.TobacFlowHipError, match="gave up"):
        fl2.check()
    fl2.check()
    assert torch.equal(torch.nan_to_num(fl2.forward_flow, nan=-7.0), torch.nan_to_num(torch.from_numpy(np.clip(want[0], -20, 20)).cuda(), nan=-7.0))
    # (iv) two flows in flight: B's calls neither consume nor report A's starved chain; A's own check still finds it
    fa = tf.create_flow(ad, "Farneback")
    slot_a = fa._pending_check.status_slot
    assert L.tf_farneback_debug_set_starved_slot(slot_a) == 0
    fb = tf.create_flow(ad, "Farneback")                                  # (entry checks of B's batches read B's word only)
    assert fb._pending_check.status_slot not in (0, slot_a)
    fb.check()
    assert L.tf_farneback_check() == 0                                    # nor has the device's shared word been touched
    with pytest.raises(_lib.TobacFlowHipError, match="gave up"):
        fa.check()
    # the shared word still works for callers that construct their parameter block themselves (status_slot 0)
    assert L.tf_farneback_debug_set_starved() == 0 and L.tf_farneback_status_check(0) == _lib.TF_ESTARVED and L.tf_farneback_check() == 0
    # slots go back when their flow object is dropped
    del fa, fb, fl, fl2
    import gc
    gc.collect()
    k = L.tf_farneback_status_acquire()
    assert 0 < k <= max(slot, slot_a) + 2
    L.tf_farneback_status_release(k)


def test_shutdown_releases_the_timing_pool_and_leaves_the_library_usable(tf):
    """tf_shutdown (SURVEY 8(b) ownership row): timing off, pooled events destroyed, idempotent, library still works."""
    from tobac_flow_amd import _lib
    rng = np.random.default_rng(3)
    a = rng.integers(0, 255, (40, 50)).astype(np.uint8)
    b = rng.integers(0, 255, (40, 50)).astype(np.uint8)
    flow0 = rng.normal(size=(40, 50, 2)).astype(np.float32)
    vr = tf.VariationalRefinement.create()
    _lib.profile_enable(True)
    want = vr.calc(a, b, flow0.copy())
    assert _lib.profile_collect()                     # something was timed
    vr.calc(a, b, flow0.copy())                       # leaves recorded, uncollected events behind
    L = _lib.lib()
    assert L.tf_shutdown() == 0 and L.tf_shutdown() == 0
    got = vr.calc(a, b, flow0.copy())
    assert np.array_equal(got, want)
    assert _lib.profile_collect() == {}               # timing is off after shutdown, nothing left over


def test_two_host_threads_on_two_streams_do_not_disturb_each_other():
    """Re-entrancy: two host threads, each with its own HIP stream, drive the library
    at the same time on DIFFERENT inputs of the SAME shape -- the case in which process-global scratch buffers or memos
    keyed by shape alone would be shared.  Workspaces and the watershed's scheduling memos are per stream: every result
    equals the one the same call gives alone."""
    import threading
    import torch
    import tobac_flow_amd.flow as tf
    from tobac_flow_amd.detection import get_combined_edge_field
    from tobac_flow_amd.watershed import neighbour_offsets, watershed_dev
    from tools.synth import anvil_seeds, blob_stack
    stacks = [blob_stack(5, 200, 260, seed=s) for s in (1, 2)]

    def run(bt):
        fl = tf.create_flow(bt, vr_steps=1, smoothing_passes=1, interp_method="cubic")
        lin, seeds = anvil_seeds(bt)
        e = get_combined_edge_field(fl, lin, dtype=np.float32)
        fw, bw = fl._dev_flows()
        lab = watershed_dev(fw, bw, e, seeds, None, neighbour_offsets(1), on_ambiguous="ignore")
        return fw.clone(), e.clone(), lab.clone()

    alone = [run(bt) for bt in stacks]
    torch.cuda.synchronize()
    results, errors = [None, None], []

    def worker(k):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                for _ in range(4):
                    results[k] = run(stacks[k])
                stream.synchronize()
        except Exception as exc:                                 # noqa: BLE001 -- reported below
            errors.append(exc)

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    torch.cuda.synchronize()
    assert not errors, errors
    for k in range(2):
        for got, want in zip(results[k], alone[k]):
            assert torch.equal(torch.nan_to_num(got.float(), nan=-7.0), torch.nan_to_num(want.float(), nan=-7.0)), k


def test_farneback_launch_of_several_rounds_and_column_groups_equals_the_oracle(tf):
    """A batch whose iteration launches need more workgroups than the GPU holds at once (96 pairs x 7 strips x 2 directions =
    1344 two-wave workgroups on 1024 slots at the full resolution): the tickets are then dealt by column groups, strips of
    the second group find their left neighbours' hand-over words complete, and every pair must still be the oracle's bit
    for bit -- checked on five pairs of the batch (first, last, three in between), both directions."""
    import torch
    from test_gpu_parity import _oracle_farneback
    from tobac_flow_amd.utils.flow_utils import FarnebackFlow
    rng = np.random.default_rng(77)
    B, H, W = 96, 96, 760
    base = ndi.gaussian_filter(rng.normal(size=(H + B + 2, W + B + 2)), 2.5)
    base = ((base - base.min()) / np.ptp(base) * 255).astype(np.uint8)
    frames = np.stack([base[i:i + H, B - i:B - i + W] for i in range(B + 1)])              # drifting content
    fr = torch.from_numpy(frames).cuda()
    fwd = torch.empty((B, H, W, 2), dtype=torch.float32, device="cuda")
    bwd = torch.empty_like(fwd)
    FarnebackFlow().calc_batch_dev(fr[:-1].contiguous(), fr[1:].contiguous(), fwd, bwd)
    for i in (0, 17, 48, 80, B - 1):
        assert np.array_equal(fwd[i].cpu().numpy(), _oracle_farneback(frames[i], frames[i + 1])), i
        assert np.array_equal(bwd[i].cpu().numpy(), _oracle_farneback(frames[i + 1], frames[i])), i


@pytest.mark.parametrize("B,parts", [(6, 2), (7, 3), (2, 2)])
def test_farneback_split_batch_equals_the_unsplit_one(B, parts):
    """tf_farneback_batch_split (round 4): the pyramid levels >= 2 for all B pairs at once, the two finest levels in parts --
    the same kernels on the same data, so the flows are those of tf_farneback_batch bit for bit (uneven last part,
    part of one pair, and an odd frame size whose levels do not halve exactly)."""
    import torch
    from tobac_flow_amd.utils.flow_utils import FarnebackFlow
    rng = np.random.default_rng(31 + B)
    H, W = 203, 331
    frames = torch.from_numpy((rng.random((B + 1, H, W)) * 255).astype(np.uint8)).cuda()
    frames[1:] = (frames[:-1].float() * 0.7 + frames[1:].float() * 0.3).to(torch.uint8)        # some frame-to-frame coherence
    prev, nxt = frames[:-1].contiguous(), frames[1:].contiguous()
    m = FarnebackFlow()
    f0, b0 = torch.empty((B, H, W, 2), device="cuda"), torch.empty((B, H, W, 2), device="cuda")
    f1, b1 = torch.empty_like(f0), torch.empty_like(b0)
    m.calc_batch_dev(prev, nxt, f0, b0)
    m.calc_batch_dev(prev, nxt, f1, b1, parts=parts)
    assert torch.equal(f0, f1) and torch.equal(b0, b1)
    assert float(f0.abs().max()) > 0


def test_the_copy_kernel_on_a_second_stream():
    """tf_copy16 -- the plain copy bench.py measures its practical HBM ceiling with -- copies, on a stream that is not the
    caller's current one too, and refuses misaligned input.  (Round 5 ran it on a CU-masked stream of the library's own making;
    those entry points left the ABI in round 6: destroying such a stream made a LATER 132-GiB allocation of this suite
    segfault -- profiles/round6_stream_destroy_segfault.txt, tools/experiments/stream_experiments.hip.)"""
    import ctypes
    import torch
    from tobac_flow_amd import _lib
    L = _lib.lib()
    for name in ("tf_stream_create_cu_mask", "tf_stream_destroy", "tf_stream_create_priority", "tf_debug_cu_histogram"):
        assert not hasattr(L, name), name
    side = torch.cuda.Stream()
    a = torch.arange(1 << 20, device="cuda", dtype=torch.float32)
    b = torch.empty_like(a)
    torch.cuda.synchronize()
    _lib.check(L.tf_copy16(_lib.ptr(a), _lib.ptr(b), a.numel() * 4, ctypes.c_void_p(side.cuda_stream)), "tf_copy16")
    side.synchronize()
    assert torch.equal(a, b)
    assert L.tf_copy16(ctypes.c_void_p(a.data_ptr() + 4), _lib.ptr(b), 1024, None) == -1      # misaligned: TF_EINVAL
