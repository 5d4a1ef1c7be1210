"""The label contract of the detection output files, at array level (mirrors /root/reference/tobac_flow/dataset.py:189-702).

After detection the drop-in scripts (scripts/dcc_detect_goes.py:316-330) derive, from the three label volumes
`core_label`, `thick_anvil_label`, `thin_anvil_label`, the variables every later stage (linking.py, analysis.py) reads:
`*_step_label`, the coordinates `core` / `anvil` / `*_step`, `core_anvil_index`, `anvil_core_count`, the
`*_step_*_index` arrays and the edge / start / end / NaN flags.  The reference keeps them in an xarray.Dataset and writes
NetCDF; xarray / netCDF4 are not in this image, so the container here is `LabelDataset` -- a dict of arrays with `.dims`
and `.coords` -- and writing files is out of scope.  Names, dtypes, dimension names and values are the reference's.

What the reference does per label in Python (scipy.ndimage.labeled_comprehension / apply_func_to_labels with a
bincount or a mode per label) is ONE pair-count pass over the two volumes on the GPU (tf_pair_counts, label.hip) and a
small host reduction over the distinct pairs; the per-step relabelling is tf_slice_labels; LUT applications are
tf_apply_lut.  Ties are broken as the reference's tools do (np.argmax / scipy.stats.mode: the smallest label wins).
"""
import numpy as np

from tobac_flow_amd import _lib
from tobac_flow_amd import label as _label

_KINDS = (("core", "core"), ("thick_anvil", "anvil"), ("thin_anvil", "anvil"))


class LabelDataset(dict):
    """name -> array (numpy, or a device tensor for the label volumes); `.dims[name]` = dimension names, `.coords` =
    coordinate name -> 1-D numpy array.  Stands in for the xr.Dataset of the reference's scripts."""

    def __init__(self, *args, coords=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.coords = dict(coords or {})
        self.dims = {}

    def add(self, name, data, dims, dtype=None):
        """add_dataarray_to_ds(create_dataarray(data, dims, name, dtype=dtype), ds) (dataset.py:20-60)"""
        if dtype is not None and not _lib.is_tensor(data):
            data = np.asarray(data).astype(dtype)
        self[name] = data
        self.dims[name] = tuple(dims)
        return data


def _host(x):
    return _lib.to_host(x) if _lib.is_tensor(x) else np.asarray(x)


def _like(result_dev, template):
    """device result in the container kind of `template` (numpy in -> numpy out)"""
    return result_dev if _lib.is_tensor(template) else _lib.to_host(result_dev)


def _apply_lut(labels_dev, lut):
    """lut[labels] for a device int32 volume and a host int32 table (tf_apply_lut; ids outside the table -> 0)"""
    t = _lib.torch()
    lab = labels_dev.contiguous()
    lut_t = t.from_numpy(np.ascontiguousarray(lut, np.int32)).to(lab.device)
    out = t.empty_like(lab)
    _lib.check(_lib.lib().tf_apply_lut(_lib.ptr(lab), lab.numel(), _lib.ptr(lut_t), lut_t.numel(), _lib.ptr(out),
                                       _lib.stream_ptr()), "tf_apply_lut")
    return out


def _present_ids(*volumes):
    """sorted non-zero label values of the volumes (np.unique minus 0), int32"""
    ids = np.zeros(0, np.int64)
    for v in volumes:
        sizes = _label.label_sizes(v)
        ids = np.union1d(ids, np.nonzero(sizes)[0])
    return ids[ids != 0].astype(np.int32)


def add_step_labels(dataset):
    """`core_step_label`, `thick_anvil_step_label`, `thin_anvil_step_label` = slice_labels of the three label volumes,
    int32 on (t, y, x) (reference: dataset.py:189-229)."""
    for kind, _ in _KINDS:
        src = dataset[kind + "_label"]
        step, _n = _label.slice_labels_dev(src)
        dataset.add(kind + "_step_label", _like(step, src), ("t", "y", "x"))


def add_label_coords(dataset):
    """Coordinates `core`, `anvil` (thick and thin anvils share it) and, where the step labels exist, `core_step`,
    `thick_anvil_step`, `thin_anvil_step`: the sorted non-zero labels, int32.  Variables already indexed by one of these
    coordinates are cut down to the labels that remain (the reference's `dataset.sel`) (reference: dataset.py:232-292)."""
    new = {"core": _present_ids(dataset["core_label"]),
           "anvil": _present_ids(dataset["thick_anvil_label"], dataset["thin_anvil_label"])}
    for kind, _ in _KINDS:
        if kind + "_step_label" in dataset:
            new[kind + "_step"] = _present_ids(dataset[kind + "_step_label"])
    for name, values in new.items():
        if name in dataset.coords:
            old = np.asarray(dataset.coords[name])
            pos = np.searchsorted(old, values)
            if np.any(pos >= old.size) or np.any(old[np.minimum(pos, old.size - 1)] != values):
                raise KeyError(f"not all values found in index '{name}'")
            for var, dims in dataset.dims.items():
                if name in dims:
                    dataset[var] = np.take(_host(dataset[var]), pos, axis=dims.index(name))
    dataset.coords.update(new)
    return dataset


def _best_partner(a, b, index, atol=0):
    """for every id in `index`: the b-label with the largest pair count among the pixels where a == id (b > 0 only),
    the smallest such label on ties, 0 if there is none or its count is below atol"""
    ia, ib, cnt = _label.pair_counts(a, b)
    out = np.zeros(len(index), np.int64)
    if ia.size:
        # pairs are sorted by (a, b): a stable sort by descending count inside each a keeps the smallest b first
        order = np.lexsort((ib, -cnt, ia))
        first = np.ones(order.size, bool)
        first[1:] = ia[order][1:] != ia[order][:-1]
        top_a, top_b, top_c = ia[order][first], ib[order][first], cnt[order][first]
        pos = np.searchsorted(top_a, index)
        ok = (pos < top_a.size)
        ok[ok] = top_a[pos[ok]] == np.asarray(index)[ok]
        sel = pos[ok]
        out[ok] = np.where(top_c[sel] >= atol, top_b[sel], 0)
    return out


def find_max_overlap(x, atol, max_label):
    """(reference: dataset.py:294-301; the per-label callback, kept for API parity)"""
    overlap_counts = np.bincount(x, minlength=max_label + 1)
    overlap_counts[0] = 0
    wh_overlap = np.argmax(overlap_counts)
    return wh_overlap if overlap_counts[wh_overlap] >= atol else 0


def link_cores_and_anvils(dataset, atol: int = 5, add_cores_to_anvils: bool = True):
    """`core_anvil_index` (core,): the thick anvil a core overlaps most (>= atol pixels, else 0); with
    add_cores_to_anvils the core's pixels are written into both anvil volumes under that anvil's label;
    `anvil_core_count` (anvil,) (reference: dataset.py:303-366)."""
    t = _lib.torch()
    core, anvil = np.asarray(dataset.coords["core"]), np.asarray(dataset.coords["anvil"])
    core_dev = _lib.to_dev(dataset["core_label"], t.int32)
    index = _best_partner(core_dev, dataset["thick_anvil_label"], core, atol)
    dataset.add("core_anvil_index", index, ("core",), np.int32)
    if add_cores_to_anvils and core.size:
        top = max(int(core.max()), int(core_dev.max()), index.size)
        lut = np.zeros(top + 1, np.int32)
        lut[core] = index
        remapped = _apply_lut(core_dev, lut)
        wh = remapped != 0
        for name in ("thick_anvil_label", "thin_anvil_label"):
            src = dataset[name]
            merged = t.where(wh, remapped.to(_lib.to_dev(src).dtype), _lib.to_dev(src))
            if _lib.is_tensor(src):
                src.copy_(merged)
            else:
                src[...] = merged.cpu().numpy()
    counts = np.bincount(index[index > 0], minlength=int(anvil.max()) + 1 if anvil.size else 1)
    dataset.add("anvil_core_count", counts[anvil] if anvil.size else np.zeros(0), ("anvil",), np.int32)


def link_step_labels(dataset):
    """`core_step_core_index`, `thick_anvil_step_anvil_index`, `thin_anvil_step_anvil_index`: for every step label the
    mode of the parent labels under it, background excluded (reference: dataset.py:369-457, stats_utils.py:11-20)."""
    for kind, parent, name in (("core", "core_label", "core_step_core_index"),
                               ("thick_anvil", "thick_anvil_label", "thick_anvil_step_anvil_index"),
                               ("thin_anvil", "thin_anvil_label", "thin_anvil_step_anvil_index")):
        index = np.asarray(dataset.coords[kind + "_step"])
        dataset.add(name, _best_partner(dataset[kind + "_step_label"], dataset[parent], index), (kind + "_step",), np.int32)


def _flags(label_dim, ids):
    ids = np.asarray(ids)
    ids = ids[ids != 0] if ids.size and ids[0] == 0 else ids
    label_dim = np.asarray(label_dim)
    pos = np.searchsorted(label_dim, ids)
    if np.any(pos >= label_dim.size) or np.any(label_dim[np.minimum(pos, max(label_dim.size - 1, 0))] != ids):
        raise KeyError("label not found in the label coordinate")
    flag = np.zeros(label_dim.size, bool)
    flag[pos] = True
    return flag


def _unique_of(*pieces):
    return np.unique(np.concatenate([_host(p).ravel() for p in pieces]))


def find_edge_labels(labels, label_dim, t=None, start_date=None, end_date=None, max_time_gap=900):
    """Flags over `label_dim`: labels touching the sides of the domain; labels present at the start (first frame, or all
    frames up to `start_date` when the volume begins before it) and, with the reference's own pairing, before a time
    gap; labels present at the end (last frame / frames from `end_date` on) and after a gap.  `t`: datetime64
    coordinate of the frames (reference: dataset.py:460-517)."""
    edge = _unique_of(labels[:, 0], labels[:, -1], labels[:, :, 0], labels[:, :, -1])
    t = None if t is None else np.asarray(t)
    if start_date is not None and t is not None and t[0] < np.datetime64(start_date):
        start = _unique_of(labels[: int(np.searchsorted(t, np.datetime64(start_date), side="right"))])
    else:
        start = _unique_of(labels[0])
    if end_date is not None and t is not None and t[-1] > np.datetime64(end_date):
        end = _unique_of(labels[int(np.searchsorted(t, np.datetime64(end_date), side="left")):])
    else:
        end = _unique_of(labels[-1])
    if t is not None and t.size > 1:
        gaps = np.where(np.diff(t).astype("timedelta64[ns]").astype(np.int64) / 1e9 > max_time_gap)[0]
        if gaps.size:
            start = _unique_of(start, *[labels[int(g)] for g in gaps])
            end = _unique_of(end, *[labels[int(g) + 1] for g in gaps])
    return _flags(label_dim, edge), _flags(label_dim, start), _flags(label_dim, end)


def flag_edge_labels(dataset, start_date=None, end_date=None, max_time_gap=900):
    """`{core,thick_anvil,thin_anvil}_{edge,start,end}_label_flag` (reference: dataset.py:520-640)."""
    for kind, dim in _KINDS:
        e, s, n = find_edge_labels(dataset[kind + "_label"], dataset.coords[dim], dataset.coords.get("t"),
                                   start_date, end_date, max_time_gap)
        dataset.add(kind + "_edge_label_flag", e, (dim,), bool)
        dataset.add(kind + "_start_label_flag", s, (dim,), bool)
        dataset.add(kind + "_end_label_flag", n, (dim,), bool)


def flag_nan_adjacent_labels(dataset, da):
    """`core_nan_flag`, `thick_anvil_nan_flag`, `thin_anvil_nan_flag`: labels with a pixel within one pixel (3 x 3 x 3
    box) of a missing value of `da` (reference: dataset.py:643-702)."""
    from tobac_flow_amd import ndimage_dev
    t = _lib.torch()
    nan = t.isnan(_lib.to_dev(da, t.float32) if not _lib.is_tensor(da) else da)
    near = None
    if bool(nan.any()):
        near = ndimage_dev.binary_dilation(nan, structure=np.ones((3, 3, 3), bool)).to(t.int32)
    for kind, dim in _KINDS:
        ids = np.zeros(0, np.int64)
        if near is not None:
            ids = np.unique(_label.pair_counts(dataset[kind + "_label"], near)[0])
        dataset.add(kind + "_nan_flag", _flags(dataset.coords[dim], ids), (dim,), bool)


__all__ = ("LabelDataset", "add_step_labels", "add_label_coords", "find_max_overlap", "link_cores_and_anvils",
           "link_step_labels", "find_edge_labels", "flag_edge_labels", "flag_nan_adjacent_labels")
