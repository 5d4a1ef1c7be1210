"""ORACLE (test infrastructure, never shipped or measured as the product).

numpy / scipy restatement, on plain dicts of arrays, of the reference's output-file label contract
  slice_labels              /root/reference/tobac_flow/utils/label_utils.py:312-349
  add_step_labels           /root/reference/tobac_flow/dataset.py:189-229
  add_label_coords          /root/reference/tobac_flow/dataset.py:232-292
  find_max_overlap          /root/reference/tobac_flow/dataset.py:294-301
  link_cores_and_anvils     /root/reference/tobac_flow/dataset.py:303-366
  link_step_labels          /root/reference/tobac_flow/dataset.py:369-457
  find_edge_labels          /root/reference/tobac_flow/dataset.py:460-517
  flag_nan_adjacent_labels  /root/reference/tobac_flow/dataset.py:643-702
  find_overlap_mode         /root/reference/tobac_flow/utils/stats_utils.py:11-20
The reference works on an xarray.Dataset (absent from this image); `ds` here is a dict name -> array with the
coordinates in ds["coords"] (a dict).  The per-label work keeps the reference's own tools
(scipy.ndimage.labeled_comprehension, scipy.stats.mode, np.bincount / np.unique), so tie-breaking is theirs.
PARITY STATUS: pinned by known answers worked out by hand in tests/test_oracle_known_answers.py; the reference holds no
test or fixture for these functions.
"""
import numpy as np
import scipy.ndimage as ndi
from scipy import stats
from scipy.ndimage import labeled_comprehension


def slice_labels(labels):
    max_step_label = np.cumsum(np.max(labels, axis=tuple(range(1, len(labels.shape)))), dtype=np.int32)
    max_step_label[1:] = max_step_label[:-1]
    max_step_label[0] = 0
    max_step_label = max_step_label.reshape([-1] + [1] * (len(labels.shape) - 1))
    step_labels = labels + max_step_label
    step_labels[labels == 0] = 0
    wh_labels = np.where(np.bincount(step_labels.ravel()))[0]
    label_map = np.zeros(wh_labels[-1] + 1, dtype=int)
    label_map[wh_labels] = np.arange(wh_labels.size)
    return label_map[step_labels]


def add_step_labels(ds):
    for k in ("core", "thick_anvil", "thin_anvil"):
        ds[k + "_step_label"] = slice_labels(ds[k + "_label"]).astype(np.int32)


def _ids(*arrays):
    s = set()
    for a in arrays:
        s |= set(np.unique(a).astype(np.int32))
    return np.asarray(sorted(list(s - set([0]))), dtype=np.int32)


def add_label_coords(ds):
    c = ds.setdefault("coords", {})
    c["core"] = _ids(ds["core_label"])
    c["anvil"] = _ids(ds["thick_anvil_label"], ds["thin_anvil_label"])
    for k in ("core", "thick_anvil", "thin_anvil"):
        if k + "_step_label" in ds:
            c[k + "_step"] = _ids(ds[k + "_step_label"])
    return ds


def find_max_overlap(x, atol, max_label):
    overlap_counts = np.bincount(x, minlength=max_label + 1)
    overlap_counts[0] = 0
    wh_overlap = np.argmax(overlap_counts)
    return wh_overlap if overlap_counts[wh_overlap] >= atol else 0


def remap_labels(labels, locations, new_labels):
    # utils/label_utils.py:265-309 with integer `locations` (label ids) and explicit new labels
    max_label = np.nanmax(labels)
    max_label = np.maximum(max_label, new_labels.size)
    remapper = np.zeros(max_label + 1, labels.dtype)
    remapper[locations] = new_labels
    return remapper[labels]


def link_cores_and_anvils(ds, atol=5, add_cores_to_anvils=True):
    core = ds["coords"]["core"]
    anvil = ds["coords"]["anvil"]
    max_label = int(core.max())
    core_anvil_index = labeled_comprehension(
        ds["thick_anvil_label"].flatten(), ds["core_label"].flatten(), core,
        lambda x: find_max_overlap(x, atol, max_label), int, 0)
    ds["core_anvil_index"] = np.asarray(core_anvil_index).astype(np.int32)
    if add_cores_to_anvils:
        remapped = remap_labels(ds["core_label"], core, core_anvil_index)
        wh = remapped != 0
        ds["thick_anvil_label"][wh] = remapped[wh]
        ds["thin_anvil_label"][wh] = remapped[wh]
    ds["anvil_core_count"] = np.asarray([np.sum(core_anvil_index == i) for i in anvil]).astype(np.int32)


def find_overlap_mode(x, background=0):
    if np.any(x != background):
        return stats.mode(x[x != background], keepdims=False)[0]
    return background


def _mode_per_label(step_labels, labels, index):
    out = np.zeros(len(index), dtype=np.int64)
    flat_s, flat_l = step_labels.ravel(), labels.ravel()
    order = np.argsort(flat_s, kind="stable")
    for i, lab in enumerate(index):
        lo, hi = np.searchsorted(flat_s[order], lab), np.searchsorted(flat_s[order], lab, side="right")
        out[i] = find_overlap_mode(flat_l[order[lo:hi]]) if hi > lo else 0
    return out


def link_step_labels(ds):
    for k, parent, name in (("core", "core_label", "core_step_core_index"),
                            ("thick_anvil", "thick_anvil_label", "thick_anvil_step_anvil_index"),
                            ("thin_anvil", "thin_anvil_label", "thin_anvil_step_anvil_index")):
        ds[name] = _mode_per_label(ds[k + "_step_label"], ds[parent], ds["coords"][k + "_step"]).astype(np.int32)


def _flags(label_dim, ids):
    ids = np.asarray(ids)
    if ids.size and ids[0] == 0:
        ids = ids[1:]
    flag = np.zeros(len(label_dim), dtype=bool)
    pos = np.searchsorted(label_dim, ids)
    if np.any(pos >= len(label_dim)) or np.any(label_dim[np.minimum(pos, len(label_dim) - 1)] != ids):
        raise KeyError("label not in the label coordinate")
    flag[pos] = True
    return flag


def find_edge_labels(labels, label_dim, t=None, start_date=None, end_date=None, max_time_gap=900):
    edge_labels = np.unique(np.concatenate([np.unique(labels[:, 0]), np.unique(labels[:, -1]),
                                            np.unique(labels[:, :, 0]), np.unique(labels[:, :, -1])]))
    edge_flag = _flags(label_dim, edge_labels)
    if start_date is not None and t is not None and t[0] < start_date:
        start_labels = np.unique(labels[t <= start_date])
    else:
        start_labels = np.unique(labels[0])
    if end_date is not None and t is not None and t[-1] > end_date:
        end_labels = np.unique(labels[t >= end_date])
    else:
        end_labels = np.unique(labels[-1])
    if t is not None and len(t) > 1:
        gaps = np.where(np.diff(t).astype("timedelta64[ns]").astype(np.int64) / 1e9 > max_time_gap)[0]
        if gaps.size:
            start_labels = np.unique(np.concatenate([start_labels, np.unique(labels[gaps])]))
            end_labels = np.unique(np.concatenate([end_labels, np.unique(labels[gaps + 1])]))
    return edge_flag, _flags(label_dim, start_labels), _flags(label_dim, end_labels)


def flag_edge_labels(ds, start_date=None, end_date=None, max_time_gap=900):
    t = ds["coords"].get("t")
    for k, dim in (("core", "core"), ("thick_anvil", "anvil"), ("thin_anvil", "anvil")):
        e, s, n = find_edge_labels(ds[k + "_label"], ds["coords"][dim], t, start_date, end_date, max_time_gap)
        ds[k + "_edge_label_flag"], ds[k + "_start_label_flag"], ds[k + "_end_label_flag"] = e, s, n


def flag_nan_adjacent_labels(ds, da):
    for k, dim in (("core", "core"), ("thick_anvil", "anvil"), ("thin_anvil", "anvil")):
        ds[k + "_nan_flag"] = np.zeros(len(ds["coords"][dim]), dtype=bool)
    if np.any(np.isnan(da)):
        wh_nan = ndi.binary_dilation(np.isnan(da), structure=np.ones([3, 3, 3]))
        for k, dim in (("core", "core"), ("thick_anvil", "anvil"), ("thin_anvil", "anvil")):
            ds[k + "_nan_flag"] = _flags(ds["coords"][dim], np.unique(ds[k + "_label"][wh_nan]))


# ---- per-label statistics: /root/reference/tobac_flow/analysis.py:204-245, 293-376, utils/legacy_utils.py:32-61 ----------
def apply_weighted_func_to_labels(labels, field, weights, func, default=None):
    if labels.shape != field.shape:
        raise ValueError("Input labels and field do not have the same shape")
    bins = np.cumsum(np.bincount(labels.ravel()))
    args = np.argsort(labels.ravel())
    return np.array([(func(field.ravel()[args[bins[i]:bins[i + 1]]], weights.ravel()[args[bins[i]:bins[i + 1]]])
                      if bins[i + 1] > bins[i] else default) for i in range(bins.size - 1)])


def weighted_statistics_on_labels(labels, da, weights):
    def weighted_average(values, weights, ignore_nan=True):
        if ignore_nan:
            wh_nan = np.isnan(values)
            values = values[~wh_nan]
            weights = weights[~wh_nan]
        if np.nansum(weights) == 0:
            return np.nan
        return np.average(values, weights=weights)

    weighted_std = lambda x, w: weighted_average((x - weighted_average(x, w)) ** 2, w) ** 0.5
    weighted_stats = lambda x, w: ([weighted_average(x, w), weighted_std(x, w), np.nanmax(x[w > 0]), np.nanmin(x[w > 0])]
                                   if np.nansum(w) > 0 else [np.nan, np.nan, np.nan, np.nan])
    stats_array = apply_weighted_func_to_labels(labels, da, weights, weighted_stats, default=[np.nan] * 4)
    return tuple(stats_array[..., k] for k in range(4))


def get_stats_for_labels(labels, da):
    idx = range(1, int(labels.max()) + 1)
    return tuple(ndi.labeled_comprehension(da, labels, idx, f, np.float64, np.nan) for f in (np.nanmean, np.nanstd, np.nanmax, np.nanmin))
