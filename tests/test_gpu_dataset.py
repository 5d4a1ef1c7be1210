"""The output-file label contract (tobac_flow_amd/dataset.py) against the oracle's restatement of the reference's
dataset.py on random label volumes, and tf_slice_labels / tf_pair_counts through their wrappers."""
import copy

import numpy as np
import pytest
import scipy.ndimage as ndi

from oracle import np_dataset

pytestmark = pytest.mark.gpu


def _volumes(seed, shape=(6, 40, 50), density=0.35):
    """cores inside thick anvils inside thin anvils, like a detection output (labels are 3-D connected blobs)"""
    rng = np.random.default_rng(seed)
    sm = ndi.gaussian_filter(rng.normal(size=shape), (0.7, 2.5, 2.5))
    thin = ndi.label(sm > np.quantile(sm, 1 - density))[0].astype(np.int32)
    thick = ndi.label(sm > np.quantile(sm, 1 - 0.6 * density))[0].astype(np.int32)
    core = ndi.label(sm > np.quantile(sm, 1 - 0.25 * density))[0].astype(np.int32)
    # anvil labels of the thick field carried by the thin one where they overlap, like detect_anvils(markers=thick)
    return core, thick, np.where(thick != 0, thick, thin + (thick.max() if thin.max() else 0) * (thin != 0)).astype(np.int32)


def _pair(seed, **kw):
    from tobac_flow_amd.dataset import LabelDataset
    core, thick, thin = _volumes(seed, **kw)
    t = np.datetime64("2020-06-01T00:00") + np.arange(core.shape[0]) * np.timedelta64(600, "s")
    ds = LabelDataset(coords={"t": t})
    for name, v in (("core_label", core), ("thick_anvil_label", thick), ("thin_anvil_label", thin)):
        ds.add(name, v.copy(), ("t", "y", "x"))
    ref = {"core_label": core.copy(), "thick_anvil_label": thick.copy(), "thin_anvil_label": thin.copy(), "coords": {"t": t}}
    return ds, ref


def _same(ds, ref, names):
    for n in names:
        got, want = np.asarray(ds[n]), np.asarray(ref[n])
        assert got.shape == want.shape, n
        assert got.dtype == want.dtype, (n, got.dtype, want.dtype)
        assert np.array_equal(got, want), n


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_label_contract_matches_the_reference_restatement(seed):
    from tobac_flow_amd import dataset as D
    ds, ref = _pair(seed)
    # the script's order: scripts/dcc_detect_goes.py:316-330
    D.add_label_coords(ds); np_dataset.add_label_coords(ref)
    D.link_cores_and_anvils(ds); np_dataset.link_cores_and_anvils(ref)
    D.add_step_labels(ds); np_dataset.add_step_labels(ref)
    D.add_label_coords(ds); np_dataset.add_label_coords(ref)
    D.link_step_labels(ds); np_dataset.link_step_labels(ref)
    D.flag_edge_labels(ds); np_dataset.flag_edge_labels(ref)
    for c in ("core", "anvil", "core_step", "thick_anvil_step", "thin_anvil_step"):
        assert ds.coords[c].dtype == np.int32 and np.array_equal(ds.coords[c], ref["coords"][c]), c
    _same(ds, ref, ["core_label", "thick_anvil_label", "thin_anvil_label", "core_step_label", "thick_anvil_step_label",
                    "thin_anvil_step_label", "core_anvil_index", "anvil_core_count", "core_step_core_index",
                    "thick_anvil_step_anvil_index", "thin_anvil_step_anvil_index"]
          + [k + f for k in ("core", "thick_anvil", "thin_anvil") for f in ("_edge_label_flag", "_start_label_flag", "_end_label_flag")])
    assert ds.dims["core_anvil_index"] == ("core",) and ds.dims["anvil_core_count"] == ("anvil",)
    assert ds.dims["thin_anvil_step_anvil_index"] == ("thin_anvil_step",) and ds.dims["core_step_label"] == ("t", "y", "x")
    assert ds["core_anvil_index"].max() > 0                       # the case exercises a real link


def test_atol_and_no_merge_options():
    from tobac_flow_amd import dataset as D
    for atol, merge in ((1, False), (40, True), (10 ** 6, True)):
        ds, ref = _pair(5)
        D.add_label_coords(ds); np_dataset.add_label_coords(ref)
        D.link_cores_and_anvils(ds, atol=atol, add_cores_to_anvils=merge)
        np_dataset.link_cores_and_anvils(ref, atol=atol, add_cores_to_anvils=merge)
        _same(ds, ref, ["core_anvil_index", "anvil_core_count", "thick_anvil_label", "thin_anvil_label"])


def test_edge_flags_with_dates_and_a_time_gap():
    from tobac_flow_amd import dataset as D
    ds, ref = _pair(7, shape=(8, 30, 36))
    t = ds.coords["t"].copy()
    t[5:] += np.timedelta64(3600, "s")                            # a gap after frame 4
    ds.coords["t"] = t; ref["coords"]["t"] = t
    D.add_label_coords(ds); np_dataset.add_label_coords(ref)
    start, end = t[1], t[6]
    D.flag_edge_labels(ds, start, end); np_dataset.flag_edge_labels(ref, start, end)
    names = [k + f for k in ("core", "thick_anvil", "thin_anvil") for f in ("_edge_label_flag", "_start_label_flag", "_end_label_flag")]
    _same(ds, ref, names)
    assert ds["thin_anvil_start_label_flag"].any() and not ds["thin_anvil_start_label_flag"].all()


def test_nan_adjacent_flags():
    from tobac_flow_amd import dataset as D
    ds, ref = _pair(9)
    D.add_label_coords(ds); np_dataset.add_label_coords(ref)
    da = np.zeros(ds["core_label"].shape, np.float32)
    D.flag_nan_adjacent_labels(ds, da); np_dataset.flag_nan_adjacent_labels(ref, da)
    names = ["core_nan_flag", "thick_anvil_nan_flag", "thin_anvil_nan_flag"]
    _same(ds, ref, names)
    assert not ds["thin_anvil_nan_flag"].any()
    da[2, 10:13, 20:22] = np.nan; da[0, 0, 0] = np.nan; da[5, 39, 49] = np.nan
    D.flag_nan_adjacent_labels(ds, da); np_dataset.flag_nan_adjacent_labels(ref, da)
    _same(ds, ref, names)


def test_slice_labels_device_edge_cases():
    from tobac_flow_amd.label import slice_labels_dev
    rng = np.random.default_rng(4)
    for shape, top in (((1, 1, 1), 1), ((3, 5, 7), 4), ((5, 17, 33), 300), ((4, 8, 8), 0)):
        lab = rng.integers(0, top + 1, shape).astype(np.int32)
        if top and lab.max() == 0:
            lab.flat[0] = 1
        got, n = slice_labels_dev(lab)
        if lab.max() == 0:
            assert n == 0 and not got.cpu().numpy().any()
            continue
        want = np_dataset.slice_labels(lab)
        assert np.array_equal(got.cpu().numpy(), want) and n == want.max()
    # a step without labels between two with: offsets carry over
    lab = np.zeros((3, 4, 4), np.int32); lab[0, 0, 0] = 7; lab[2, 1, 1] = 2; lab[2, 2, 2] = 7
    got, n = slice_labels_dev(lab)
    assert n == 3 and np.array_equal(got.cpu().numpy(), np_dataset.slice_labels(lab))


def test_pair_counts_wrapper_matches_numpy():
    from tobac_flow_amd.label import pair_counts, label_sizes
    rng = np.random.default_rng(6)
    a = rng.integers(0, 9, (4, 20, 30)).astype(np.int32)
    b = rng.integers(0, 5, (4, 20, 30)).astype(np.int32)
    for zero in (False, True):
        ia, ib, cnt = pair_counts(a, b, include_b_zero=zero)
        keep = (a > 0) & ((b >= 0) if zero else (b > 0))
        keys, want = np.unique(np.stack([a[keep], b[keep]], 1), axis=0, return_counts=True)
        assert np.array_equal(np.stack([ia, ib], 1), keys) and np.array_equal(cnt, want)
    assert np.array_equal(label_sizes(a), np.bincount(a.ravel()))


def test_label_statistics_match_numpy_within_float32_summation_error():
    """tf_label_stats accumulates in double; the reference sums float32 data in float32 (np.average / np.nanmean): the
    tolerance is that of the reference's own sums, 2e-5 relative (+ 1e-6 absolute near zero); max / min are exact."""
    import warnings
    from tobac_flow_amd.analysis import get_stats_for_labels, weighted_statistics_on_labels
    rng = np.random.default_rng(11)
    core, thick, thin = _volumes(11)
    labels = thin
    x = (250 + 30 * rng.normal(size=labels.shape)).astype(np.float32)
    x[rng.random(labels.shape) < 0.05] = np.nan
    w = rng.random(labels.shape).astype(np.float32)
    w[rng.random(labels.shape) < 0.3] = 0
    biggest = np.argmax(np.bincount(labels.ravel())[1:]) + 1
    w[labels == 1] = 0                                           # a label without any weight -> NaN x 4
    x[labels == 2] = np.nan                                      # a label without any value
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want_w = np_dataset.weighted_statistics_on_labels(labels, x, w)
        want_u = np_dataset.get_stats_for_labels(labels, x)
    got_w = weighted_statistics_on_labels(labels, x, w)
    got_u = get_stats_for_labels(labels, x)
    for got, want in ((got_w, want_w), (got_u, want_u)):
        for k in range(4):
            g, v = np.asarray(got[k], np.float64), np.asarray(want[k], np.float64)
            assert g.shape == v.shape and got[k].dtype == np.float32
            assert np.array_equal(np.isnan(g), np.isnan(v)), k
            ok = ~np.isnan(v)
            if k >= 2:
                assert np.array_equal(g[ok], v[ok].astype(np.float32)), k
            else:
                assert np.allclose(g[ok], v[ok], rtol=2e-5, atol=1e-6), (k, np.abs(g[ok] - v[ok]).max())
    assert np.isnan(got_w[0][0]) and np.isnan(got_w[0][1]) and not np.isnan(got_w[0][biggest - 1])
    with pytest.raises(ValueError):
        get_stats_for_labels(labels, x[1:])


def test_retry_paths_of_the_label_wrappers():
    """pair_counts sizes its scratch for n / 16 runs and slice_labels_dev for 4 M shifted ids; both retry with what the
    library reports when the data needs more."""
    from tobac_flow_amd.label import pair_counts, slice_labels_dev
    rng = np.random.default_rng(21)
    a = rng.integers(0, 50, (8, 512, 512)).astype(np.int32)          # ~2 M runs >> n / 16
    b = rng.integers(0, 7, a.shape).astype(np.int32)
    ia, ib, cnt = pair_counts(a, b)
    keep = (a > 0) & (b > 0)
    want = np.zeros((50, 7), np.int64)
    np.add.at(want, (a[keep], b[keep]), 1)
    got = np.zeros_like(want)
    got[ia, ib] = cnt
    assert np.array_equal(got, want) and np.all(cnt > 0)
    lab = np.zeros((4, 64, 64), np.int32)
    lab[0, 3, 3] = 1_500_000; lab[1, 5, 5] = 2_000_000; lab[1, 9, 9] = 7; lab[3, 1, 1] = 1_900_000      # 5.4 M shifted ids
    step, n = slice_labels_dev(lab)
    assert n == 4 and np.array_equal(step.cpu().numpy(), np_dataset.slice_labels(lab))


def test_label_contract_accepts_device_tensors():
    """label volumes that already live on the GPU stay there (no host copies of the volumes), results are the same"""
    import torch
    from tobac_flow_amd import dataset as D
    ds, ref = _pair(2)
    for k in ("core_label", "thick_anvil_label", "thin_anvil_label"):
        ds[k] = torch.as_tensor(ds[k]).cuda()
    D.add_label_coords(ds); np_dataset.add_label_coords(ref)
    D.link_cores_and_anvils(ds); np_dataset.link_cores_and_anvils(ref)
    D.add_step_labels(ds); np_dataset.add_step_labels(ref)
    D.add_label_coords(ds); np_dataset.add_label_coords(ref)
    D.link_step_labels(ds); np_dataset.link_step_labels(ref)
    D.flag_edge_labels(ds); np_dataset.flag_edge_labels(ref)
    assert ds["core_step_label"].is_cuda and ds["thick_anvil_label"].is_cuda
    for k in ("core_label", "thick_anvil_label", "thin_anvil_label", "core_step_label", "thin_anvil_step_label"):
        assert np.array_equal(ds[k].cpu().numpy(), ref[k]), k
    for k in ("core_anvil_index", "anvil_core_count", "core_step_core_index", "thin_anvil_step_anvil_index",
              "core_edge_label_flag", "thin_anvil_end_label_flag"):
        assert np.array_equal(ds[k], ref[k]), k
