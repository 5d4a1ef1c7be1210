"""CPU-only tests: the C ABI library loads and exports every symbol include/*.h declares, the host
layer mirrors the reference's interface (names, defaults, exception types) and the numpy glue gives
the reference's known answers (ports of /root/reference/tests/test_flow.py:8-49,364-412,
tests/test_analysis.py, tests/test_label_utils.py, tests/test_detection.py:7-33)."""
import os
import re

import numpy as np
import pytest
import scipy.ndimage as ndi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import ctypes
    from tobac_flow_amd import _lib
    header = open(os.path.join(ROOT, "include", "tobac_flow_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(tf_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 18
    L = ctypes.CDLL(_lib.lib_path())
    missing = [n for n in sorted(declared) if not hasattr(L, n)]
    assert not missing, missing
    assert set(_lib.EXPORTS) <= declared
    assert _lib.lib().tf_version() >= 100


def test_a_starved_chain_report_becomes_an_exception(monkeypatch):
    """Host side of TF_ESTARVED (no device needed): with no status word allocated tf_farneback_check reports nothing; a
    non-zero report (-6) from the library makes FarnebackFlow.check_launches -- what create_flow / calculate_flow /
    calculate_flow_frame / FarnebackFlow.calc call after their launches -- raise TobacFlowHipError (a RuntimeError)."""
    import pytest
    import torch
    from tobac_flow_amd import _lib
    from tobac_flow_amd.utils.flow_utils import FarnebackFlow
    L = _lib.lib()
    assert L.tf_farneback_check() == 0
    model = FarnebackFlow()
    assert model.params.chain_form == _lib.FB_CHAIN_DEFAULT

    class _Stream:
        def synchronize(self):
            pass
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: _Stream())
    model.check_launches()                                                  # nothing reported: no exception
    monkeypatch.setattr(L, "tf_farneback_check", lambda: _lib.TF_ESTARVED, raising=False)
    with pytest.raises(_lib.TobacFlowHipError, match="code -6"):
        model.check_launches("create_flow")
    with pytest.raises(RuntimeError):
        model.check_launches()
    # status slots (round 6): without a device there is no pinned ring -- acquire hands out 0 (= share the device's word), the
    # checks report nothing, a slot number outside the ring is an argument error
    assert L.tf_farneback_status_acquire() == 0 and L.tf_farneback_status_check(0) == 0 and L.tf_farneback_status_check(7) == 0
    assert L.tf_farneback_status_check(4096) == -1
    L.tf_farneback_status_release(0)
    L.tf_farneback_status_release(4096)
    assert model.params.status_slot == 0


def test_device_field_is_a_tensor_with_a_time_coordinate():
    """detection.DeviceField: what a device-resident pipeline hands the recipes in place of an xr.DataArray (a torch tensor
    has no coordinates and Tensor.t is the transpose): `.data`, `.t` with `.data` / `.values`, the arithmetic the drop-in
    scripts use on their fields (wvd - swd, wvd + swd, -bt), and the checks of its constructor.  (CPU tensors do for the
    container; the recipes themselves need the GPU.)"""
    import numpy as np
    import pytest
    import torch
    from tobac_flow_amd.detection import DeviceField, _is_device
    from tobac_flow_amd.flow import _unwrap_device_field
    times = np.datetime64("2020-06-01T00:00") + np.arange(3) * np.timedelta64(10, "m")
    a = DeviceField(torch.arange(24, dtype=torch.float32).reshape(3, 2, 4), times)
    b = DeviceField(torch.ones(3, 2, 4), times)
    assert a.shape == (3, 2, 4) and a.dtype == torch.float32 and len(a.t) == 3 and np.array_equal(a.t.data, times) and np.array_equal(np.asarray(a.t), times)
    for got, want in (((a - b), a.data - 1), ((a + b), a.data + 1), ((-a), -a.data), ((a - 2.0), a.data - 2)):
        assert isinstance(got, DeviceField) and torch.equal(got.data, want) and np.array_equal(got.t.values, times)
    assert _is_device(a) and _is_device(a.data) and not _is_device(np.zeros(3))
    assert _unwrap_device_field(a) is a.data and _unwrap_device_field(a.data) is a.data
    arr = np.zeros((3, 2, 4), np.float32)
    assert _unwrap_device_field(arr) is arr
    with pytest.raises(TypeError):
        DeviceField(arr, times)
    with pytest.raises(ValueError, match="time coordinate"):
        DeviceField(torch.zeros(4, 2, 2), times)


def test_farneback_launch_arithmetic_of_the_host_side():
    """Host-only arithmetic of the C ABI (no device work): workspace sizes grow with the batch, hold the strips' hand-over
    words of the iteration kernel at every size (1 x 1 included), and the workgroup count the batching decisions rest on
    is two directions x pairs x strips of 116 columns (farneback.hip FBI_OW)."""
    import ctypes
    from tobac_flow_amd import _lib
    L = _lib.lib()
    p = _lib.FarnebackParams(5, 0.5, 13, 10, 5, 1.1)
    resident = ctypes.c_int64(0)
    assert L.tf_farneback_iteration_workgroups(5424, 5424, ctypes.byref(p), 21, ctypes.byref(resident)) == 2 * 21 * 47
    assert resident.value > 0 and resident.value % 4 == 0                       # four two-wave workgroups per CU
    assert L.tf_farneback_iteration_workgroups(1500, 2500, ctypes.byref(p), 23, None) == 2 * 23 * 22
    assert L.tf_farneback_iteration_workgroups(0, 10, ctypes.byref(p), 1, None) == 0
    one = L.tf_farneback_workspace_bytes_batch(1, 5424, 5424, ctypes.byref(p))
    assert L.tf_farneback_workspace_bytes_batch(21, 5424, 5424, ctypes.byref(p)) > 20 * one
    # hand-over words of a pair at the full resolution: 2 directions x 46 strip boundaries x 5424 rows x 20 words of 8 bytes
    assert one > 2 * 46 * 5424 * 20 * 8
    tiny = L.tf_farneback_workspace_bytes(1, 1, ctypes.byref(p))
    assert tiny >= 16384                                                        # the ticket counters alone
    assert L.tf_farneback_can_split(5424, 5424, ctypes.byref(p)) == 1 and L.tf_farneback_can_split(40, 40, ctypes.byref(p)) == 0


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import tobac_flow_amd.flow as tf
    from tobac_flow_amd._lib import TobacFlowHipError
    z = np.zeros((3, 5, 2), np.float32)
    with pytest.raises(TobacFlowHipError):
        tf.smooth_flow_step(z, z)
    with pytest.raises(TobacFlowHipError):
        tf.Flow(z[None], z[None]).sobel(np.zeros((1, 3, 5), np.float32))


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "tobac_flow_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, f


def test_select_of_model_errors():
    import tobac_flow_amd.flow as tf
    assert hasattr(tf.select_of_model("Farneback"), "calc")
    with pytest.raises(NotImplementedError):
        tf.select_of_model("DenseRLOF")
    with pytest.raises(NotImplementedError):
        tf.select_of_model("DIS")
    with pytest.raises(ValueError):
        tf.select_of_model("not_an_of_model")


def _stitch_lut_loops(counts, pairs_per_boundary):
    """the stitch as plain loops (a union-find that attaches the larger root to the smaller, numbering by smallest member)"""
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    n = int(offs[-1])
    parent = np.arange(n + 1)

    def find(i):
        while parent[i] != i:
            parent[i] = parent[parent[i]]
            i = parent[i]
        return i
    for r, pairs in enumerate(pairs_per_boundary):
        for a, b in np.asarray(pairs, np.int64).reshape(-1, 2):
            ra, rb = find(offs[r] + a), find(offs[r + 1] + b)
            if ra != rb:
                parent[max(ra, rb)] = min(ra, rb)
    root = np.array([find(i) for i in range(n + 1)])
    new, nxt = np.zeros(n + 1, np.int64), 0
    for i in range(1, n + 1):
        if root[i] == i:
            nxt += 1
            new[i] = nxt
    new = new[root]
    return [np.concatenate([[0], new[offs[r] + 1: offs[r + 1] + 1]]) for r in range(len(counts))]


def test_stitch_lut_equals_the_loop_form_on_random_graphs():
    """parallel.stitch_lut (sparse connected components, every rank runs it after every step) against the loop form:
    windows without labels, boundaries without pairs, chains over many windows, repeated pairs."""
    from tobac_flow_amd.parallel import stitch_lut
    rng = np.random.default_rng(5)
    for _ in range(150):
        n_win = int(rng.integers(1, 8))
        counts = [int(c) for c in rng.integers(0, 12, size=n_win)]
        pairs = []
        for r in range(n_win - 1):
            k = int(rng.integers(0, 12)) if counts[r] and counts[r + 1] else 0
            pairs.append(np.stack([rng.integers(1, counts[r] + 1, size=k), rng.integers(1, counts[r + 1] + 1, size=k)], 1)
                         if k else np.zeros((0, 2), np.int64))
        got, want = stitch_lut(counts, pairs), _stitch_lut_loops(counts, pairs)
        assert len(got) == len(want) and all(np.array_equal(g, w) for g, w in zip(got, want)), (counts, pairs)
    with pytest.raises(ValueError):
        stitch_lut([2, 2], [np.array([[3, 1]])])


def test_flow_window_and_window_view_on_host_arrays():
    """Flow.window mirrors the end frames of the cut (flow.py:425-426); window_view is the same window on the stack's own
    memory, which is restored when the block is left."""
    import tobac_flow_amd.flow as tf
    rng = np.random.default_rng(2)
    fw = rng.normal(size=(6, 4, 5, 2)).astype(np.float32)
    bw = rng.normal(size=(6, 4, 5, 2)).astype(np.float32)
    fw[-1], bw[0] = -bw[-1], -fw[0]
    f0, b0 = fw.copy(), bw.copy()
    whole = tf.Flow(fw, bw)
    for a, b in ((0, 6), (1, 4), (0, 2), (3, 6), (2, 3)):
        w = whole.window(a, b)
        assert w.shape == (b - a, 4, 5) and not np.shares_memory(w.forward_flow, fw)
        if b - a > 1:
            assert np.array_equal(w.forward_flow[:-1], f0[a:b - 1]) and np.array_equal(w.forward_flow[-1], -b0[b - 1])
            assert np.array_equal(w.backward_flow[1:], b0[a + 1:b]) and np.array_equal(w.backward_flow[0], -f0[a])
        else:
            assert np.isnan(w.forward_flow).all() and np.isnan(w.backward_flow).all()
        with whole.window_view(a, b) as v:
            assert np.shares_memory(v.forward_flow, fw)
            assert np.array_equal(v.forward_flow, w.forward_flow, equal_nan=True)
            assert np.array_equal(v.backward_flow, w.backward_flow, equal_nan=True)
        assert np.array_equal(fw, f0) and np.array_equal(bw, b0)
    with pytest.raises(ValueError):
        whole.window(3, 3)


def test_flow_object_contract():
    import tobac_flow_amd.flow as tf
    from tobac_flow_amd.core import AbstractFlow
    z = np.zeros([3, 5, 2])
    f = tf.Flow(z, z)
    assert isinstance(f, AbstractFlow) and f.shape == (3, 5)
    with pytest.raises(ValueError):
        tf.Flow(z, np.zeros([2, 4, 2]))
    with pytest.raises(ValueError):
        tf.Flow(np.zeros([3, 5, 1]), np.zeros([3, 5, 1]))
    fw, bw = f.flow
    assert fw is f.forward_flow and bw is f.backward_flow
    assert f[:2, :4].shape == (2, 4)
    with pytest.raises(AssertionError):
        f.convolve(np.zeros((4, 4)))
    for name in ("create_flow", "calculate_flow", "calculate_flow_2", "calculate_flow_frame", "smooth_flow_step",
                 "to_8bit", "warp_flow", "select_of_model", "vr_model", "combine_flow", "flow_magnitude"):
        assert hasattr(tf, name), name


def test_to_8bit_reference_known_answers():
    import tobac_flow_amd.flow as tf
    assert np.all(tf.to_8bit(np.zeros(5)) == 0)
    assert np.all(tf.to_8bit(np.ones(5)) == 0)
    assert np.all(tf.to_8bit(np.ones(5), vmin=0, vmax=1) == 255)
    arr = np.arange(256)
    assert np.all(tf.to_8bit(arr) == arr)
    assert np.all(tf.to_8bit(arr + 10, vmin=10, vmax=10 + 255) == arr)
    pair = np.array([[np.nan, 1.0, 0.0], [0.5, np.nan, np.nan]])
    assert np.array_equal(tf.to_8bit(pair, 0, 1), np.array([[127, 255, 0], [127, 255, 0]], np.uint8))


def test_convolve_argument_validation():
    from tobac_flow_amd.convolve import convolve
    z = np.zeros((2, 4, 4, 2), np.float32)
    d = np.zeros((2, 4, 4), np.float32)
    with pytest.raises(AssertionError):
        convolve(d, z, z, structure=np.ones((3, 3)))
    with pytest.raises(ValueError):
        convolve(d, z, z, method="spline")


def test_analysis_known_answers():
    from tobac_flow_amd import analysis
    assert analysis.find_object_lengths(np.zeros([3]).astype(int)).size == 0
    one = np.array([0, 1, 0]).astype(int)
    assert analysis.find_object_lengths(one)[0] == 1
    l3 = np.array([[1, 1, 1]]).astype(int)
    assert analysis.find_object_lengths(l3)[0] == 1 and analysis.find_object_lengths(l3, axis=1)[0] == 3
    assert np.all(analysis.find_object_lengths(np.arange(10).astype(int)) == np.ones([9]))
    empty = np.zeros([3]).astype(int)
    assert analysis.mask_labels(empty, empty).size == 0
    assert analysis.mask_labels(one, empty)[0] == False  # noqa: E712
    assert analysis.mask_labels(one, one)[0] == True     # noqa: E712


def test_label_utils_known_answers():
    from tobac_flow_amd.utils.label_utils import apply_func_to_labels, make_step_labels, slice_labels
    lab = np.zeros([5, 10, 15], dtype=np.int32)
    lab[:, 3:6, 4:8] = 1
    assert np.all(np.unique(slice_labels(lab)) == np.arange(6))
    lab[:, 5:8, 10:13] = 2
    s = slice_labels(lab)
    assert np.all(np.unique(s) == np.arange(11))
    for i in range(5):
        assert np.all(np.unique(s[i]) == np.array([0, 2 * i + 1, 2 * i + 2]))
    t = np.array([[[0, 0, 0, 1], [0, 2, 1, 0], [0, 2, 0, 3]], [[0, 0, 0, 0], [0, 2, 2, 0], [0, 2, 0, 4]]])
    want = np.array([[[0, 0, 0, 1], [0, 3, 2, 0], [0, 3, 0, 4]], [[0, 0, 0, 0], [0, 5, 5, 0], [0, 5, 0, 6]]])
    assert np.all(make_step_labels(t) == want)
    tl = np.zeros([4, 6])
    tl[1:3, 1:3] = 1
    tl[2:3, 3:6] = 3
    tl = tl.astype(int)
    d1 = np.arange(24).reshape([4, 6])
    r = apply_func_to_labels(tl, d1, func=np.mean)
    assert np.allclose(r[0], d1[tl == 1].mean()) and np.allclose(r[2], d1[tl == 3].mean())
    r = apply_func_to_labels(tl, d1, np.array([1, 2, 3, 3, 2, 1]), func=lambda a, w: (np.average(a, weights=w), np.std(a)),
                             default=np.nan)
    assert r.shape == (2, 3) and np.isnan(r[0, 1])


def test_get_watershed_mask_known_answers():
    from tobac_flow_amd.detection import get_watershed_mask
    f = np.zeros([1, 5, 5], dtype=np.float32)
    f[:, 3:] = 1
    r = get_watershed_mask(f)
    assert np.all(r[:, :2]) and not np.any(r[:, 2:])
    r = get_watershed_mask(f, erode_distance=2)
    assert np.all(r[:, :1]) and not np.any(r[:, 1:])
    assert not np.any(get_watershed_mask(f, erode_distance=3))
    f[:, 2] = np.nan
    r = get_watershed_mask(f, erode_distance=1)
    assert np.all(r[:, :3]) and not np.any(r[:, 3:])


def test_peak_local_max_documented_examples():
    from tobac_flow_amd.utils.peak_utils import peak_local_max
    img = np.zeros((7, 7))
    img[3, 4] = 1
    img[3, 2] = 1.5
    assert np.array_equal(peak_local_max(img, min_distance=1), [[3, 2], [3, 4]])
    assert np.array_equal(peak_local_max(img, min_distance=2), [[3, 2]])


def test_window_bounds_and_stitch_lut():
    from tobac_flow_amd.parallel import stitch_lut, window_bounds
    assert window_bounds(10, 1) == [(0, 10)]
    b = window_bounds(144, 8)
    assert b[0][0] == 0 and b[-1][1] == 144 and all(b[i][1] - 1 == b[i + 1][0] for i in range(7))
    with pytest.raises(ValueError):
        window_bounds(3, 4)
    # rank 0 has labels 1..3, rank 1 has 1..2, rank 2 has 1..2; 0:2 == 1:1, 1:2 == 2:1
    luts = stitch_lut([3, 2, 2], [np.array([[2, 1]]), np.array([[2, 1]])])
    assert [l.tolist() for l in luts] == [[0, 1, 2, 3], [0, 2, 4], [0, 4, 5]]


# ----------------------------------------------------------------------------- peak_local_max pinned by scikit-image
def _peak_golden():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "peak_local_max_skimage.npz"))


def test_peak_local_max_equals_scikit_image_on_tie_free_fields():
    """utils.peak_utils.peak_local_max restates scikit-image 0.18 for the call the reference makes
    (detection.py:154, 161).  tests/golden/peak_local_max_skimage.npz was produced by scikit-image 0.18.3 itself
    (tests/golden/make_peak_golden.py): coordinates AND row order must match for min_distance 1, 3, 10."""
    from tobac_flow_amd.utils.peak_utils import peak_local_max
    z = _peak_golden()
    assert str(z["skimage_version"]).startswith("0.18")
    checked = 0
    for name in z["names"]:
        if str(name) == "quantised":
            continue
        img = z[f"{name}/image"]
        for d in (1, 3, 10):
            want = z[f"{name}/peaks_d{d}"]
            got = np.asarray(peak_local_max(img, min_distance=d)).reshape(-1, 2)
            assert got.shape == want.shape and np.array_equal(got, want), f"{name} min_distance={d}"
            checked += len(want)
    assert checked > 4000


def test_peak_local_max_with_tied_intensities_is_a_valid_greedy_selection():
    """On a quantised field many candidates share an intensity; scikit-image orders them with numpy's unstable
    argsort, so its own answer changes with the numpy version (the restatement reproduces the golden exactly under the
    golden's numpy 1.26 and differs under numpy 2).  What holds under any tie order: peaks are mask pixels, sorted by
    non-increasing intensity, pairwise at least min_distance apart, and every rejected candidate lies within
    min_distance of a kept peak that is at least as high."""
    import scipy.ndimage as ndi
    from tobac_flow_amd.utils.peak_utils import peak_local_max
    z = _peak_golden()
    img = z["quantised/image"]
    for d in (1, 3, 10):
        got = np.asarray(peak_local_max(img, min_distance=d)).reshape(-1, 2)
        want = z[f"quantised/peaks_d{d}"]
        size = 2 * d + 1
        mask = (img == ndi.maximum_filter(img, footprint=np.ones((size, size), bool), mode="constant")) & (img > img.min())
        mask[:d] = mask[-d:] = False
        mask[:, :d] = mask[:, -d:] = False
        vals = img[tuple(got.T)]
        assert mask[tuple(got.T)].all() and np.all(np.diff(vals) <= 0)
        cheb = np.abs(got[:, None, :] - got[None, :, :]).max(-1)
        assert (cheb + np.eye(len(got), dtype=int) * 10 ** 6).min() >= d
        cand = np.transpose(np.nonzero(mask))
        dist = np.abs(cand[:, None, :] - got[None, :, :]).max(-1)                      # (candidates, kept)
        covered = ((dist < d) & (vals[None, :] >= img[tuple(cand.T)][:, None])).any(1)
        kept = (dist == 0).any(1)
        assert np.all(covered | kept)
        assert abs(len(got) - len(want)) <= max(2, len(want) // 20)                    # same problem, close to the golden


def test_core_cooling_filter_keeps_only_fast_cooling_cores():
    """Last stage of detect_cores (reference: detection.py:434-482) on a hand-made labelling: core 1 cools by
    1 K / min, core 2 by 0.1 K / min, core 3 lasts only three steps (no 3-step interval): only core 1 survives."""
    from tobac_flow_amd.detection import _core_cooling_filter

    class Coord:
        def __init__(self, v):
            self.values = self.data = v

    class Field(np.ndarray):
        pass

    T, H, W, minutes = 8, 12, 30, 5
    labels = np.zeros((T, H, W), np.int32)
    labels[:, 2:6, 2:8] = 1
    labels[:, 2:6, 12:18] = 2
    labels[2:5, 7:10, 22:27] = 3
    bt = np.full((T, H, W), 280.0, np.float32).view(Field)
    steps = np.arange(T, dtype=np.float32)[:, None, None]
    bt[:, 2:6, 2:8] = 280.0 - 1.0 * minutes * steps[:, :, :1]
    bt[:, 2:6, 12:18] = 280.0 - 0.1 * minutes * steps[:, :, :1]
    bt[2:5, 7:10, 22:27] = 250.0 - 3.0 * minutes * steps[2:5, :, :1]
    bt.t = Coord(np.datetime64("2020-06-01T00:00") + np.arange(T) * np.timedelta64(minutes, "m"))
    out = _core_cooling_filter(labels.copy(), bt, min_length=3)
    assert out.dtype == labels.dtype
    assert np.array_equal(out == 1, labels == 1) and out.max() == 1
    # exactly at the threshold counts as cooling (>= 0.5)
    bt[:, 2:6, 12:18] = 280.0 - 0.5 * minutes * steps[:, :, :1]
    out = _core_cooling_filter(labels.copy(), bt, min_length=3)
    assert out.max() == 2 and np.array_equal(out == 2, labels == 2)


# ----------------------------------------------------------------------------- label filters of analysis.py
def _random_labels(seed, shape=(7, 24, 30), n=14):
    import scipy.ndimage as ndi
    rng = np.random.default_rng(seed)
    blobs = ndi.gaussian_filter(rng.normal(size=shape), (0.6, 1.0, 1.0)) > 0.09     # 13 - 22 components for seeds 0 - 2
    labels, count = ndi.label(blobs)
    assert count >= 5
    return labels.astype(np.int32), rng


def _brute_filter(labels, keep):
    out = np.zeros_like(labels)
    nxt = 0
    for k in range(1, labels.max() + 1):
        if keep(k):
            nxt += 1
            out[labels == k] = nxt
    return out


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_label_filters_match_brute_force_and_their_legacy_forms(seed):
    """analysis.filter_labels_by_* (reference: analysis.py:66-201): keep the labels that last >= min_length steps along
    axis 0 and / or touch the mask(s), renumbered densely in ascending order; the *_legacy forms do the same in place."""
    from tobac_flow_amd import analysis as an
    labels, rng = _random_labels(seed)
    mask = rng.random(labels.shape) < 0.01
    mask2 = rng.random(labels.shape) < 0.02

    def length(k):
        tt = np.nonzero((labels == k).any((1, 2)))[0]
        return tt[-1] - tt[0] + 1

    def hits(k, m):
        return bool(m[labels == k].any())

    min_length = 3
    want_len = _brute_filter(labels, lambda k: length(k) >= min_length)
    want_mask = _brute_filter(labels, lambda k: hits(k, mask))
    want_both = _brute_filter(labels, lambda k: length(k) >= min_length and hits(k, mask))
    want_multi = _brute_filter(labels, lambda k: hits(k, mask) and hits(k, mask2))
    want_len_multi = _brute_filter(labels, lambda k: length(k) >= min_length and hits(k, mask) and hits(k, mask2))
    assert 0 < want_len.max() < labels.max() and 0 < want_mask.max() < labels.max()      # the filters do remove something
    assert np.array_equal(an.filter_labels_by_length(labels, min_length), want_len)
    assert np.array_equal(an.filter_labels_by_mask(labels, mask), want_mask)
    assert np.array_equal(an.filter_labels_by_length_and_mask(labels, mask, min_length), want_both)
    assert np.array_equal(an.filter_labels_by_multimask(labels, [mask, mask2]), want_multi)
    assert np.array_equal(an.filter_labels_by_length_and_multimask(labels, [mask, mask2], min_length), want_len_multi)
    for legacy, args, want in [(an.filter_labels_by_length_legacy, (min_length,), want_len),
                               (an.filter_labels_by_length_and_mask_legacy, (mask, min_length), want_both),
                               (an.filter_labels_by_length_and_multimask_legacy, ([mask, mask2], min_length), want_len_multi)]:
        work = labels.copy()
        out = legacy(work, *args)
        assert out is work and np.array_equal(out, want)                                 # in place, same answer
    for fn in (an.filter_labels_by_multimask, an.filter_labels_by_length_and_multimask_legacy):
        with pytest.raises(ValueError):
            fn(labels.copy(), mask, *(() if fn is an.filter_labels_by_multimask else (min_length,)))


def test_find_neighbour_labels_and_find_overlapping_labels():
    """label.find_neighbour_labels / label_utils.find_overlapping_labels (reference: label.py:178-245,
    label_utils.py:352-376): labels whose overlap with the given label's pixels exceeds `absolute_overlap` pixels AND
    reaches `overlap` x the smaller of the two areas are pushed once onto the stack."""
    from tobac_flow_amd.label import find_neighbour_labels
    from tobac_flow_amd.utils.label_utils import find_overlapping_labels
    labels = np.zeros((12, 12), np.int32)
    labels[1:5, 1:5] = 1                                   # 16 px
    labels[6:9, 6:9] = 2                                   # 9 px
    labels[10:12, 0:3] = 3                                 # 6 px
    fwd = np.zeros_like(labels)                            # "warped labels of the next step"
    fwd[1:5, 1:3] = 2                                      # covers 8 px of label 1
    fwd[1:2, 4:5] = 3                                      # covers 1 px of label 1
    bwd = np.zeros_like(labels)
    bwd[3:5, 3:5] = 3                                      # covers 4 px of label 1
    bins = np.cumsum(np.bincount(labels.ravel()))
    args = np.argsort(labels.ravel())
    locs = args[bins[0]:bins[1]]
    assert sorted(locs.tolist()) == sorted(np.flatnonzero(labels.ravel() == 1).tolist())
    assert find_overlapping_labels(fwd, locs, bins, overlap=0, absolute_overlap=0) == [2, 3]
    assert find_overlapping_labels(fwd, locs, bins, overlap=0, absolute_overlap=1) == [2]          # 1 px is not > 1
    assert find_overlapping_labels(fwd, locs, bins, overlap=0.9, absolute_overlap=0) == []         # 8 < 0.9 * min(16, 9)
    assert find_overlapping_labels(fwd, locs, bins, overlap=0.8, absolute_overlap=0) == [2]        # 8 >= 0.8 * 9
    assert find_overlapping_labels(fwd, np.array([], int), bins) == []
    stack, seen = [], np.zeros(4, bool)
    seen[1] = True
    find_neighbour_labels(1, stack, bins, args, seen, fwd, bwd, overlap=0, absolute_overlap=1)
    assert stack == [2, 3] and seen.tolist() == [False, True, True, True]                          # 2 via t+1, 3 via t-1 (4 px)
    find_neighbour_labels(1, stack, bins, args, seen, fwd, bwd, overlap=0, absolute_overlap=1)
    assert stack == [2, 3]                                                                         # nothing is pushed twice


# ----------------------------------------------------------------------------- normalisation methods and small utilities
def test_normalisation_methods_have_their_defining_properties():
    """utils.normalisation_utils (reference: normalisation_utils.py:59-160): every method maps into [0, 1]; linear / log /
    z_score / uniform are non-decreasing in the data, inverse_log non-increasing; known values."""
    from tobac_flow_amd.utils import normalisation_utils as nu
    rng = np.random.default_rng(3)
    x = (rng.gamma(2.0, 10.0, size=(2, 40, 50)) + 200).astype(np.float32)
    order = np.argsort(x.ravel())
    for name, sign in [("linear", 1), ("log", 1), ("z_score", 1), ("uniform", 1), ("inverse_log", -1)]:
        y = nu.select_normalisation_method(name)(x)
        assert y.shape == x.shape and y.min() >= 0 and y.max() <= 1, name
        assert np.all(sign * np.diff(y.ravel()[order]) >= 0), name
    assert np.array_equal(nu.linear_norm(np.array([2.0, 4.0, 6.0])), [0.0, 0.5, 1.0])
    assert np.array_equal(nu.linear_norm(np.array([5.0, 5.0])), [0.0, 0.0])                     # flat input: factor 0
    assert np.array_equal(nu.linear_norm(np.array([0.0, 10.0]), vmin=2, vmax=6), [0.0, 1.0])   # clipped
    # log_norm (:75-79) overwrites vmin with the DATA minimum and then uses it as the lower bound of the LOG values:
    # log([1, e, e^2] - 1 + 1) = [0, 1, 2] normalised from 1 to 2 -> [0, 0, 1].  Reproduced as is; with data whose minimum
    # exceeds its largest log value (any brightness temperature field) the upper bound falls below the lower one and
    # everything maps to 0.
    assert np.array_equal(nu.log_norm(np.array([1.0, np.e, np.e ** 2])), [0.0, 0.0, 1.0])
    assert not nu.log_norm(x).any()
    inv = nu.inverse_log_norm(np.array([1.0, 2.0, 3.0]))       # log(3 - x + 1) = log([3, 2, 1]) from its minimum to the DATA maximum 3
    assert np.allclose(inv, (np.log([3.0, 2.0, 1.0]) - 0.0) / 3.0)
    z = nu.z_norm(np.array([-10.0, 0.0, 10.0]), max_std=1)
    assert np.array_equal(z, [0.0, 0.5, 1.0])
    u = nu.uniform_norm(np.arange(1000.0), quantiles=4)
    assert set(np.unique(u)) == {0.0, 1 / 3, 2 / 3, 1.0} and np.allclose(np.bincount((u * 3).round().astype(int)), 250, atol=1)
    ll = nu.local_linear_norm(np.tile(np.arange(50.0), (50, 1)), size=5)
    assert ll.min() == 0 and ll.max() == 1 and ll[10, 10] == 0.5                                # centre of a 5-wide ramp
    nan_in = np.tile(np.arange(20.0), (20, 1))
    nan_in[3, 3] = np.nan
    assert np.isfinite(nu.local_linear_norm(nan_in, size=3)).all() and np.isnan(nan_in[3, 3])   # NaNs filled on a copy
    with pytest.raises(ValueError):
        nu.select_normalisation_method("quadratic")


def test_small_utilities_known_answers():
    from datetime import datetime, timedelta
    from tobac_flow_amd.utils import mse
    from tobac_flow_amd.utils.datetime_utils import get_datetime_from_coord, get_time_diff_from_coord, time_diff
    from tobac_flow_amd.utils.flow_utils import select_border_mode, select_interp_mode
    from tobac_flow_amd.utils.label_utils import get_step_labels_for_label, make_step_labels, relabel_objects
    assert mse(np.array([1.0, 2.0, 4.0]), np.array([1.0, 0.0, 1.0])) == pytest.approx((0 + 4 + 9) / 3)
    t0 = datetime(2020, 6, 1)
    times = [t0, t0 + timedelta(minutes=5), t0 + timedelta(minutes=15), t0 + timedelta(minutes=20)]
    assert time_diff(times) == [5.0, 7.5, 7.5, 5.0]            # one-sided at the ends, centred (halved) inside
    coord = np.array(times, dtype="datetime64[s]")
    assert get_datetime_from_coord(coord) == times
    assert np.array_equal(get_time_diff_from_coord(coord), [5.0, 7.5, 7.5, 5.0])
    assert select_interp_mode("nearest") == 0 and select_interp_mode("linear") == 1 and select_interp_mode("cubic") == 2
    assert select_interp_mode("lanczos") == 3
    for bad, err in [("spline", ValueError)]:
        with pytest.raises(err):
            select_interp_mode(bad)
    assert select_border_mode("constant") == "constant"
    for bad, err in [("reflect", NotImplementedError), ("edge", ValueError)]:
        with pytest.raises(err):
            select_border_mode(bad)
    labels = np.array([[[0, 4, 4], [0, 0, 9]], [[4, 4, 0], [9, 0, 0]]])
    assert np.array_equal(relabel_objects(labels), np.array([[[0, 1, 1], [0, 0, 2]], [[1, 1, 0], [2, 0, 0]]]))
    steps = make_step_labels(labels)
    per_label = get_step_labels_for_label(labels, steps)     # one entry per label value 1 .. max, None where absent
    assert len(per_label) == labels.max()
    for lab in range(1, labels.max() + 1):
        if lab in (4, 9):
            assert np.array_equal(per_label[lab - 1], np.unique(steps[labels == lab]))
            assert len(per_label[lab - 1]) == 2            # both labels live in two time steps
        else:
            assert per_label[lab - 1] is None


@pytest.mark.parametrize("n_windows", [1, 2, 4])
def test_stitch_window_list_reassembles_a_labelling(n_windows):
    """parallel.stitch_window_list: cut a labelled (t, y, x) volume into windows that share one frame, renumber every
    window independently, stitch: the result is the original labelling up to renumbering in order of first appearance
    (the single-process form of the multi-GPU stitch, reference scheme: linking.py:49-161)."""
    import scipy.ndimage as ndi
    import torch
    from tobac_flow_amd.parallel import stitch_window_list, window_bounds
    rng = np.random.default_rng(8)
    blobs = ndi.gaussian_filter(rng.normal(size=(13, 30, 36)), (1.2, 1.5, 1.5)) > 0.03
    truth, n = ndi.label(blobs)
    assert n >= 4
    truth = truth.astype(np.int32)
    truth[0, 0, 0] = -7                                        # negative ids pass through untouched
    windows = []
    for a, b in window_bounds(truth.shape[0], n_windows):
        w = truth[a:b].copy()
        ids = np.unique(w[w > 0])
        perm = np.zeros(max(w.max(), 0) + 1, np.int32)
        perm[ids] = rng.permutation(len(ids)) + 1              # what an independent per-window labelling would give
        w[w > 0] = perm[w[w > 0]]
        windows.append(torch.from_numpy(w))
    out = [x.numpy() for x in stitch_window_list(windows, min_overlap=1, overlap=1)]
    bounds = window_bounds(truth.shape[0], n_windows)
    for (a, b), w in zip(bounds, out):                          # shared frames agree between neighbours
        assert w.shape == truth[a:b].shape
    for i in range(n_windows - 1):
        assert np.array_equal(out[i][-1], out[i + 1][0])
    whole = np.concatenate([out[0]] + [w[1:] for w in out[1:]])
    assert whole[0, 0, 0] == -7 and np.array_equal(whole == 0, truth == 0)
    # same partition: a bijection between the stitched ids and the true ids
    pos = truth > 0
    pairs = np.unique(np.stack([truth[pos], whole[pos]], 1), axis=0)
    assert len(pairs) == len(np.unique(truth[pos])) == len(np.unique(whole[pos]))
    assert sorted(np.unique(whole[pos]).tolist()) == list(range(1, len(pairs) + 1))
    assert stitch_window_list([]) == []


def test_apply_func_to_labels_behaviours():
    """utils.label_utils.apply_func_to_labels beyond the reference's own test (tests/test_label_utils.py:5-75): explicit
    index incl. absent and unordered labels, broadcasting, sequence defaults, multi-valued func, region values in C order."""
    from tobac_flow_amd.utils.label_utils import apply_func_to_labels
    labels = np.array([[0, 2, 2, 0], [5, 5, 0, 2], [0, 0, 0, 5]])
    data = np.arange(12.0).reshape(3, 4)
    assert np.array_equal(apply_func_to_labels(labels, data, func=np.sum, default=-1.0), [-1, 1 + 2 + 7, -1, -1, 4 + 5 + 11])
    assert np.array_equal(apply_func_to_labels(labels, data, func=np.sum, index=[5, 3, 2], default=0.0), [20.0, 0.0, 10.0])
    # values reach func in C order of the pixels
    assert apply_func_to_labels(labels, data, func=lambda v: v[0] * 100 + v[-1], index=[2])[()] == 1 * 100 + 7
    # broadcasting of a lower-dimensional field
    row = np.array([10.0, 20.0, 30.0, 40.0])
    assert np.array_equal(apply_func_to_labels(labels, row, func=np.max, index=[2, 5]), [40.0, 40.0])
    # multi-valued func: scalar default repeated, one-element sequence unwrapped, full sequence used as given
    mm = lambda v: (v.min(), v.max())
    assert np.array_equal(apply_func_to_labels(labels, data, func=mm, index=[2, 4], default=np.nan),
                          [[1.0, np.nan], [7.0, np.nan]], equal_nan=True)
    assert np.array_equal(apply_func_to_labels(labels, data, func=mm, index=[4, 5], default=(-1.0, -2.0)), [[-1.0, 4.0], [-2.0, 11.0]])
    assert np.array_equal(apply_func_to_labels(labels, data, func=np.sum, index=[4, 5], default=[9.0]), [9.0, 20.0])
    with pytest.raises(IndexError):
        apply_func_to_labels(np.zeros((2, 2), int), np.zeros((2, 2)), func=np.sum, index=[1], default=0.0)


def test_window_linking_rule_equals_the_reference_statement():
    """parallel._overlap_pairs_host (the CPU-tensor path of the stitch, and the statement the GPU kernel is tested against)
    against the loop form of linking.py:33-93 (oracle/np_label.link_overlap_pairs: bincount per left label through
    scipy.ndimage.labeled_comprehension, atol / rtol as the reference applies them)."""
    import scipy.ndimage as ndi
    from oracle import np_label
    from tobac_flow_amd.parallel import _overlap_pairs_host, compare_frames
    rng = np.random.default_rng(0)
    for trial in range(12):
        a = ndi.label(ndi.gaussian_filter(rng.normal(size=(2, 40, 50)), (0, 1.5, 1.5)) > 0.2)[0]
        b = ndi.label(ndi.gaussian_filter(rng.normal(size=(2, 40, 50)), (0, 1.5, 1.5)) > 0.2)[0]
        if trial % 2:
            b = np.roll(a, (1, 2), (1, 2))
            b[b > 0] = (b[b > 0] * 7) % 31 + 1
        for atol, rtol in ((5, 0.5), (0, 0), (1, 0), (3, 0.9), (0, 0.3)):
            x, y = np_label.link_overlap_pairs(a, b, atol, rtol)
            assert np.array_equal(_overlap_pairs_host(a, b, atol, rtol), np.stack([x, y], 1)), (trial, atol, rtol)
    assert _overlap_pairs_host(np.zeros((1, 3, 3), int), np.ones((1, 3, 3), int), 5, 0.5).shape == (0, 2)
    # linking.py:55-56: of the common frames the first and the last are not compared
    assert compare_frames(4) == slice(1, 3) and compare_frames(24) == slice(1, 23) and compare_frames(1) == slice(0, 0) and compare_frames(1, True) == slice(0, 1) and compare_frames(2, True) == slice(0, 2)


def test_stitch_with_the_reference_rule_joins_only_well_overlapping_labels():
    """Two windows sharing four frames: an object seen alike by both is joined; one that the right window sees shifted
    so that < 50 % of either footprint coincides is not, nor is one that coincides in < 5 pixels (linking.py:70-76)."""
    import torch
    from tobac_flow_amd.parallel import stitch_window_list
    left = np.zeros((6, 20, 40), np.int32)
    right = np.zeros((6, 20, 40), np.int32)
    left[:, 2:8, 2:8] = 1
    right[:, 2:8, 2:8] = 3                  # same footprint: joined
    left[:, 10:16, 2:8] = 2
    right[:, 10:16, 6:12] = 1               # 2 of 6 columns in common: 1/3 of either -> not joined
    left[:, 2:4, 20:22] = 3
    right[:, 2:4, 20:22] = 2                # 4 px per frame x 2 compared frames = 8 >= 5 and identical: joined
    left[:, 10:11, 30:32] = 4
    right[:, 10:11, 30:32] = 4              # 2 px x 2 frames = 4 < 5: not joined
    out = stitch_window_list([torch.from_numpy(left), torch.from_numpy(right)], overlap=4)
    l, r = out[0].numpy(), out[1].numpy()
    assert l[0, 2, 2] == r[0, 2, 2] and l[0, 2, 20] == r[0, 2, 20]
    assert l[0, 10, 2] != r[0, 10, 6] and l[0, 10, 30] != r[0, 10, 30]
    assert sorted(np.unique(np.concatenate([l[l > 0], r[r > 0]])).tolist()) == [1, 2, 3, 4, 5, 6]


def test_short_overlaps_link_nothing_unless_opted_in():
    """linking.py:55-56 compares the common frames [1:-1]: with one or two shared frames the reference links nothing.
    stitch_window_list does the same by default; short_overlap_ok=True (or the round-1 `min_overlap` form) is the
    explicit opt-in to linking on all shared frames."""
    import torch
    from tobac_flow_amd.parallel import stitch_window_list
    left = np.zeros((4, 12, 12), np.int32)
    right = np.zeros((4, 12, 12), np.int32)
    left[:, 2:8, 2:8] = 1
    right[:, 2:8, 2:8] = 1
    for ov in (1, 2):
        out = stitch_window_list([torch.from_numpy(left), torch.from_numpy(right)], overlap=ov)
        assert int(out[0].max()) == 1 and int(out[1].max()) == 2            # two objects: not linked
        out = stitch_window_list([torch.from_numpy(left), torch.from_numpy(right)], overlap=ov, short_overlap_ok=True)
        assert int(out[0].max()) == 1 and int(out[1].max()) == 1            # opted in: one object
    out = stitch_window_list([torch.from_numpy(left), torch.from_numpy(right)])          # default overlap = 4: [1:-1] compared
    assert int(out[1].max()) == 1


def test_anvil_inputs_blocked_path_equals_the_single_block(monkeypatch):
    """tools/synth.anvil_inputs processes volumes beyond torch's 32-bit pooling limit in blocks of frames with a
    one-frame halo; the round-3 block size left every block two frames over the guard, so the path could only raise.
    With the limit patched small the blocked result must equal the unblocked one (first / last / interior blocks)."""
    import torch
    from tools import synth
    rng = np.random.default_rng(5)
    bt = torch.from_numpy((265.0 + 12.0 * rng.standard_normal((23, 20, 24))).astype(np.float32))
    bt[3, 2:6, 3:9] = float("nan")
    bt[11, 10:14, 0:5] = float("nan")
    lin0, m0 = synth.anvil_inputs(bt)
    assert (m0 > 0).any() and (m0 < 0).any() and (m0 == 0).any()
    for frames_per_block in (1, 3, 7):
        monkeypatch.setattr(synth, "_TORCH_INDEX_LIMIT", (frames_per_block + 4) * 22 * 26)
        assert synth._padded_elements(23, 20, 24) > synth._TORCH_INDEX_LIMIT          # the blocked path is really taken
        lin, m = synth.anvil_inputs(bt)
        assert torch.equal(m, m0) and torch.equal(torch.nan_to_num(lin, nan=-7.0), torch.nan_to_num(lin0, nan=-7.0))
    monkeypatch.setattr(synth, "_TORCH_INDEX_LIMIT", 4 * 22 * 26)                   # not even one frame + halo + padding
    with pytest.raises(ValueError):
        synth.anvil_inputs(bt)


def test_the_three_host_replays_of_the_reference_heap_agree(tmp_path):
    """csrc/ws_replay.h is pure C++ (the host half of TF_WS_REFERENCE_ORDER): tools/replay_check builds it with g++ and runs
    the plain form (every item as the reference keeps it, _watershed.pyx:67-152), the sparse form and the dense form (from
    8-byte entries and from the device's 2-bit codes) on random instances with few distinct values -- equal-valued seeds,
    floods from below and inside the tie value, runs of ballast seeds, small items at the end of the array -- and compares
    pop counts and every marker's rank.  (On the GPU the same three run behind one flood: tests/test_gpu_reference_order.py.)"""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = tmp_path / "replay_check"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-w", "-o", str(exe), os.path.join(ROOT, "tools", "replay_check", "replay_check.cpp")])
    for seed in ("101", "202"):
        out = subprocess.run([str(exe), "random", "4000", seed], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "plain == sparse == dense" in out.stdout, out.stdout + out.stderr


def test_window_scheduler_shares_its_slots_between_stacks_and_delivers_them_in_order():
    """host logic of parallel._WindowFloods / _SequenceFloods (detect_stack_windows, detect_stack_sequence) with the device
    parts replaced by stand-ins: never more floods in flight than slots, whichever stack they belong to; a flood whose first
    finish ends in a late export is queued again and finished later; a stack is delivered once its flow has been enqueued,
    every window begun and every flood finished -- the stacks in order; set-ups precede sweeps within a hand-over."""
    import contextlib
    from concurrent.futures import ThreadPoolExecutor
    from tobac_flow_amd.parallel import _SequenceFloods, _StackRun, _WindowFloods
    log, in_flight, peak = [], [0], [0]
    pool = ThreadPoolExecutor(4)

    class Job:
        def __init__(self, tag, late):
            self.tag, self.late, self.swept, self.needs_replay = tag, late, False, True

        def sweeps(self):
            log.append(("sweep", self.tag))
            self.swept = True

        def replay(self):
            return self

    class FakeFlow:
        @contextlib.contextmanager
        def window_view(self, lo, hi):
            log.append(("view", lo, hi))
            yield self

    def make(stack, n_win, slots, late=()):
        o = _StackRun()
        o.bounds = [(4 * k, 4 * k + 6) for k in range(n_win)]
        o.mark = lambda what: None
        o.pool = pool
        wf = _WindowFloods(o, list(range(4 * n_win + 2)), 0, slots, len(slots))

        def begin(flow_w, w, scratch, wf=wf):
            in_flight[0] += 1
            peak[0] = max(peak[0], in_flight[0])
            job = Job((stack, wf.next), (stack, wf.next) in late)
            log.append(("setup", job.tag))
            return job, pool.submit(job.replay), {}, scratch

        def finish(job, fut, st, scratch):
            fut.result()
            assert job.swept, "finished before its sweeps"
            if job.late:
                job.late = False
                log.append(("late", job.tag))
                return None
            in_flight[0] -= 1
            log.append(("done", job.tag))
            return "labels%s" % (job.tag,)
        wf._begin, wf._finish = begin, finish
        # (WatershedJob.step runs pending sweeps itself: the stand-in's finish is only reached through sweep() or a job swept there)
        real_finish_one = wf.finish_one

        def finish_one(block=True, wf=wf):
            for p in wf.pending:
                if not p[0].swept:
                    p[0].sweeps()
            return real_finish_one(block)
        wf.finish_one = finish_one
        return wf
    # one stack, two slots, five windows, the third one with a late export
    slots = [None, None]
    wf = make(0, 5, slots, late={(0, 2)})
    wf.begin_up_to(FakeFlow(), 14)                                       # windows 0 .. 2 end within 14 frames
    assert wf.next == 3 and peak[0] <= 2
    wf.begin_up_to(FakeFlow(), 22)
    assert wf.finish_all() == ["labels(0, %d)" % k for k in range(5)] and peak[0] <= 2 and in_flight[0] == 0 and len(slots) == 2
    assert ("late", (0, 2)) in log and log.index(("late", (0, 2))) < log.index(("done", (0, 2)))
    first = [e for e in log if e[0] in ("setup", "sweep")][:4]
    assert [e[0] for e in first[:2]] == ["setup", "setup"]               # a hand-over's windows are all set up before any is swept
    # two stacks sharing three slots
    del log[:]
    peak[0] = 0
    delivered = []
    fam = _SequenceFloods(lambda w: delivered.append((w.index, list(w.wins))))
    slots = [None, None, None]
    a, b = make(0, 4, slots, late={(0, 3)}), make(1, 4, slots)
    for k, w in enumerate((a, b)):
        w.family, w.index, w.flow_enqueued = fam, k, False
    fam.active.append(a)
    a.sweep(a.setup_up_to(FakeFlow(), 18))                               # all four windows of stack 0: three slots -> one finished on the way
    a.flow_enqueued = True
    fam.settle(a)
    fam.active.append(b)
    b.sweep(b.setup_up_to(FakeFlow(), 18))                               # stack 1 begins while stack 0 still holds slots
    assert peak[0] <= 3
    b.flow_enqueued = True
    while fam.active:
        assert fam.finish_any(block=True) or not fam.active
        for w in list(fam.active):
            fam.settle(w)
    assert [d[0] for d in delivered] == [0, 1] and in_flight[0] == 0 and len(slots) == 3
    assert delivered[0][1] == ["labels(0, %d)" % k for k in range(4)] and delivered[1][1] == ["labels(1, %d)" % k for k in range(4)]


def test_flood_thread_never_begins_a_window_whose_last_frame_is_the_handovers_last():
    """create_flow hands over n = pairs_done + 1 frames, of which forward[n - 1] is written by the flow's NEXT
    part on the calling stream; Flow.window_view saves / patches / restores exactly that frame of a window that ends at n.
    Driven from the flood thread (a stream of its own) that restore races with the flow's write, so such a window waits for
    the next hand-over -- unless the stack ends there.  On the calling thread (ordered by the stream) nothing changes.
    Also: abandon_all gives every slot back (failure path)."""
    import contextlib
    from concurrent.futures import ThreadPoolExecutor
    from tobac_flow_amd.parallel import _StackRun, _WindowFloods
    pool = ThreadPoolExecutor(2)
    views, abandoned = [], []

    class Job:
        needs_replay = True

        def sweeps(self):
            pass

        def replay(self):
            return self

        def abandon(self):
            abandoned.append(self)

    class FakeFlow:
        @contextlib.contextmanager
        def window_view(self, lo, hi):
            views.append((lo, hi))
            yield self

    def make(own_stream, slots):
        o = _StackRun()
        o.bounds = [(0, 16), (12, 30)]
        o.mark = lambda what: None
        o.pool = pool
        wf = _WindowFloods(o, list(range(30)), 0, slots, len(slots))
        wf.own_stream = own_stream
        wf._begin = lambda flow_w, w, scratch: (Job(), pool.submit(lambda: None), {}, scratch)
        return wf
    slots = [None, None, None]
    wf = make(True, slots)
    assert wf.setup_up_to(FakeFlow(), 15) == [] and views == []
    assert wf.setup_up_to(FakeFlow(), 16) == [] and views == []          # window 0 ends AT the hand-over: frame 15 is not final
    assert len(wf.setup_up_to(FakeFlow(), 17)) == 1 and views == [(0, 16)]
    assert wf.setup_up_to(FakeFlow(), 29) == []
    assert len(wf.setup_up_to(FakeFlow(), 30)) == 1 and views == [(0, 16), (12, 30)]   # the stack ends here: its last frame is final
    assert len(slots) == 1
    wf.abandon_all()
    assert len(abandoned) == 2 and len(slots) == 3 and not wf.pending
    del views[:]
    wf = make(False, [None, None])                                       # calling thread: ordered by the stream, begun at once
    assert len(wf.setup_up_to(FakeFlow(), 16)) == 1 and views == [(0, 16)]


def test_rank_windows_deals_the_windows_of_one_stack_out_to_the_ranks():
    """parallel.rank_windows (strong sharding, BASELINE configs 4 - 5): every window goes to exactly one rank, in order;
    a rank's frames run from its first window's start to its last window's stop; consecutive ranks share exactly the
    frames two consecutive windows share; one rank = the whole stack."""
    import pytest
    from tobac_flow_amd.parallel import rank_windows, window_bounds
    for T, n, overlap in ((288, 24, 4), (144, 12, 4), (54, 5, 4), (40, 3, 2)):
        bounds = window_bounds(T, n, overlap)
        assert rank_windows(bounds, 0, 1) == (0, T, bounds)
        for world in (2, 3, min(n, 8)):
            shares = [rank_windows(bounds, r, world) for r in range(world)]
            glob = [(lo + s[0], hi + s[0]) for s in shares for lo, hi in s[2]]
            assert glob == bounds
            assert shares[0][0] == 0 and shares[-1][1] == T
            for a, b in zip(shares[:-1], shares[1:]):
                assert a[1] - b[0] == overlap and len(a[2]) >= 1
            assert max(len(s[2]) for s in shares) - min(len(s[2]) for s in shares) <= 1
            assert all(s[2][0][0] == 0 and s[2][-1][1] == s[1] - s[0] for s in shares)
    with pytest.raises(ValueError):
        rank_windows(window_bounds(40, 3, 2), 0, 4)
    with pytest.raises(ValueError):
        rank_windows(window_bounds(40, 3, 2), 3, 3)


def test_every_module_of_the_package_imports():
    """(a syntax error in a module the CPU suite does not otherwise touch must not wait for the GPU box)"""
    import importlib
    import pkgutil
    import tobac_flow_amd
    names = [m.name for m in pkgutil.walk_packages(tobac_flow_amd.__path__, "tobac_flow_amd.")]
    assert "tobac_flow_amd._staging" in names and "tobac_flow_amd.parallel" in names
    for name in names:
        importlib.import_module(name)


def test_select_peaks_grid_form_equals_the_all_pairs_form():
    """utils.peak_utils.select_peaks (round 6: accepted points kept in a grid of cells) against the all-pairs statement of the
    same greedy rule: 2-D and 3-D candidates, integer and fractional min_distance, ties in intensity, num_peaks"""
    from tobac_flow_amd.utils.peak_utils import _select_peaks_all_pairs, select_peaks
    rng = np.random.default_rng(4)
    for ndim, n, span in ((2, 0, 50), (2, 1, 50), (2, 400, 60), (2, 3000, 400), (3, 800, 30), (1, 200, 100)):
        coords = rng.integers(0, span, size=(n, ndim)).astype(np.int64)
        vals = np.round(rng.normal(size=n), 1)                          # (ties in intensity: the order is argsort's, in both forms)
        for d in (0, 1, 2, 3, 10, 2.5):
            for num in (np.inf, 7):
                a, b = select_peaks(coords, vals, d, num), _select_peaks_all_pairs(coords, vals, d, num)
                assert a.shape == b.shape and np.array_equal(a, b), (ndim, n, d, num)
