"""Watershed segmentation in a semi-Lagrangian framework, on the MI355X.

Mirrors /root/reference/tobac_flow/watershed.py:17-168 (same signature, coercions and
ValueErrors).  The reference pads the volume and calls its Cython heap flood
(tobac_flow/_watershed.pyx:222-344); here the whole flood runs in HIP (tf_watershed,
include/tobac_flow_hip.h) on the unpadded volume: out-of-volume neighbours are rejected by
coordinate tests, which is what the zero-padded mask achieves in the reference.
"""
import warnings

import numpy as np
import scipy.ndimage as ndi

from tobac_flow_amd import _lib

# Neighbour orders of skimage.morphology._util._offsets_to_raveled_neighbors for
# ndi.generate_binary_structure(3, k) as (dt, dy, dx): skimage sorts by L1 distance with numpy's
# default NON-stable argsort, so the order is data, not derivable (watershed.py:114-116;
# recorded from scikit-image 0.18.3 / numpy 1.26 by tests/golden/make_watershed_golden.py).
_NEIGHBOUR_ORDER = {
    1: [(-1, 0, 0), (0, -1, 0), (0, 0, -1), (0, 0, 1), (0, 1, 0), (1, 0, 0)],
    2: [(-1, 0, 0), (0, 0, -1), (0, -1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 0), (-1, 0, -1), (0, -1, 1),
        (0, -1, -1), (-1, 1, 0), (-1, 0, 1), (0, 1, -1), (0, 1, 1), (1, -1, 0), (-1, -1, 0), (1, 0, -1),
        (1, 0, 1), (1, 1, 0)],
    3: [(-1, 0, 0), (0, -1, 0), (0, 0, -1), (0, 0, 1), (0, 1, 0), (1, 0, 0), (-1, -1, 0), (0, -1, -1),
        (0, -1, 1), (0, 1, -1), (-1, 1, 0), (0, 1, 1), (1, -1, 0), (-1, 0, 1), (-1, 0, -1), (1, 0, 1),
        (1, 0, -1), (1, 1, 0), (-1, 1, -1), (-1, 1, 1), (-1, -1, -1), (-1, -1, 1), (1, -1, -1),
        (1, -1, 1), (1, 1, -1), (1, 1, 1)],
}

DEFAULT_CHAIN_DEPTH = 3
MAX_CHAIN_DEPTH = 12              # TF_WS_MAX_DEPTH
TF_WS_AMBIGUOUS, TF_EDEPTH = 1, -5
AMB_DEPENDS, AMB_MARKER_TIE, AMB_DEPTH = 1, 2, 4


class WatershedAmbiguityWarning(UserWarning):
    """Some labels depend on the order in which the reference's heap pops equal-valued markers."""


class WatershedAmbiguityError(RuntimeError):
    """on_ambiguous="raise": some labels depend on the order of equal-valued markers."""


class WatershedDepthError(RuntimeError):
    """Ties remain at the deepest chain level: the labels may differ from the reference there."""


def neighbour_offsets(connectivity, ndim=3):
    """(n, 3) int8 neighbour list in the reference's order (skimage `_validate_connectivity` +
    `_offsets_to_raveled_neighbors`).  Non-standard structuring elements are ordered by a stable
    sort on L1 distance (documented deviation: skimage's order for those is not reproducible)."""
    if connectivity is None:
        connectivity = 1
    if np.isscalar(connectivity):
        selem = ndi.generate_binary_structure(ndim, int(connectivity))
    else:
        selem = np.array(connectivity, bool)
        if selem.ndim != ndim:
            raise ValueError("Connectivity dimension must be same as image")
        if any(s % 2 == 0 for s in selem.shape):
            raise ValueError("Connectivity array must have an unambiguous center")
    if selem.shape != (3, 3, 3):
        raise ValueError("connectivity must be an integer or a (3, 3, 3) structuring element")
    for k, order in _NEIGHBOUR_ORDER.items():
        if np.array_equal(selem, ndi.generate_binary_structure(3, k)):
            return np.array(order, np.int8)
    offs = np.stack(np.nonzero(selem), -1) - 1
    dist = np.abs(offs).sum(1)
    offs = offs[np.argsort(dist, kind="stable")]
    return np.ascontiguousarray(offs[np.abs(offs).sum(1) > 0], np.int8)


# Scheduling memo (never affects the labels): tf_watershed first tries a cheap root phase and falls back to
# the chain phases when it finds a label conflict.  Fields with exact plateaus (detect_anvils) conflict
# every time, so for a volume shape whose last probe conflicted the speculative phase is skipped
# (TF_WS_SKIP_FAST_PATH); every _REPROBE-th call probes again so that a change of data is noticed.
# Both memos are keyed by shape AND by the calling stream, and guarded by a lock: interleaved callers (two host threads,
# two streams) each see the history of their own sequence of windows (process-global state keyed by shape).
_relevant_memo = {}      # (T, H, W, neighbours, depth, stream) -> relevant pixels of the last successful call
_conflict_memo = {}
_MEMO_LOCK = __import__("threading").Lock()
_REPROBE = 8
TF_WS_SKIP_FAST_PATH, TF_WS_REFERENCE_ORDER = 1, 2


def watershed_dev(fwd, bwd, field, markers, mask, nbr, chain_depth=DEFAULT_CHAIN_DEPTH, stats=None,
                  expect_conflict=None, max_chain_depth=MAX_CHAIN_DEPTH, on_ambiguous="warn", return_ambiguous=False):
    """Device-resident core: torch tensors in (field f32, markers i32, mask i8 or None), labels out.

    expect_conflict: True / False force the scheduling hint, None (default) uses the per-shape memo.
    Exactness contract (include/tobac_flow_hip.h, tf_watershed_ex2): the library deepens the chain comparison on its
    own up to `max_chain_depth` and reports every pixel whose label still hangs on a last-resort tie-break.
      * ties between equal-valued markers (the reference resolves them by the internal state of its heap): labels
        follow the markers' raster order; `on_ambiguous` = "warn" (default) / "raise" / "ignore";
        `on_ambiguous="reference"`: the library replays the reference heap's push / pop mechanics on the host
        (TF_WS_REFERENCE_ORDER, include/tobac_flow_hip.h) to get the markers' pop ranks and floods with those: the
        labels are then the reference's bit for bit, at the cost of a sequential pass (stats["reference_order"]);
      * ties left by the depth cut-off at `max_chain_depth`: WatershedDepthError (a warning with "ignore").
    return_ambiguous: also return the (T, H, W) uint8 report (AMB_* bits)."""
    if on_ambiguous not in ("warn", "raise", "ignore", "reference"):
        raise ValueError("on_ambiguous must be 'warn', 'raise', 'ignore' or 'reference'")
    t = _lib.torch()
    L = _lib.lib()
    T, H, W = field.shape
    labels = _lib.empty((T, H, W), t.int32)
    amb = _lib.empty((T, H, W), t.uint8) if return_ambiguous else None
    nbr = np.ascontiguousarray(nbr, np.int8)
    chain_depth = int(chain_depth)
    max_chain_depth = max(chain_depth, min(int(max_chain_depth), MAX_CHAIN_DEPTH))
    # the flood keys are compact over the relevant pixels: size the workspace from a cheap count of
    # the floodable pixels and retry once with the exact number if boundary markers exceed the slack
    st = np.zeros(16, np.int64)
    key = (T, H, W, len(nbr), chain_depth, t.cuda.current_stream().cuda_stream)
    # ... unless the previous call of this shape has reported its exact count (stats[6]): consecutive windows of one
    # sequence differ little, the library checks the size anyway, and counting costs three passes and a host sync
    with _MEMO_LOCK:
        known = _relevant_memo.get(key)
        memo = list(_conflict_memo.get(key, (False, 0)))      # [last probe conflicted, calls since that probe]
    if known is not None:
        guess = min(T * H * W, int(known * 1.25) + 4096)
    else:
        floodable = (markers == 0) if mask is None else ((markers == 0) & (mask != 0))
        guess = min(T * H * W, int(floodable.sum().item() * 1.5) + 4096)
        del floodable
    skip = expect_conflict if expect_conflict is not None else (memo[0] and memo[1] < _REPROBE)
    flags = TF_WS_SKIP_FAST_PATH if (skip and chain_depth > 1) else 0
    if on_ambiguous == "reference":
        flags |= TF_WS_REFERENCE_ORDER
    # levels beyond chain_depth are rarely needed: the first call gets room for two more, a second one for all
    start, cap = chain_depth, min(max_chain_depth, chain_depth + 2)
    probed = None
    ws = None
    while True:
        nbytes = L.tf_watershed_workspace_bytes(T, H, W, len(nbr), cap, guess)
        ws = None                                # a retry must not hold the old buffer while the larger one is allocated
        ws = _lib.workspace(nbytes, "watershed")
        rc = L.tf_watershed_ex2(_lib.ptr(field), _lib.ptr(markers), _lib.ptr(mask), _lib.ptr(fwd), _lib.ptr(bwd),
                                T, H, W, nbr.ctypes.data_as(_lib._P), len(nbr), start, cap, flags, _lib.ptr(labels),
                                _lib.ptr(amb), _lib.ptr(ws), ws.numel(), st.ctypes.data_as(_lib._P), _lib.stream_ptr())
        if rc == -2 and st[6] > guess:
            guess = int(st[6])
            continue
        if probed is None and rc in (0, TF_WS_AMBIGUOUS, TF_EDEPTH):
            probed = int(st[5])
        if rc == TF_EDEPTH and cap < max_chain_depth:
            start, cap, flags = cap + 1, max_chain_depth, TF_WS_SKIP_FAST_PATH | (flags & TF_WS_REFERENCE_ORDER)
            continue
        break
    if rc not in (TF_WS_AMBIGUOUS, TF_EDEPTH):
        _lib.check(rc, "tf_watershed")
    if probed is not None and probed >= 0:
        memo[0], memo[1] = bool(probed), 0                     # this call probed
    else:
        memo[1] += 1
    with _MEMO_LOCK:
        if rc in (0, TF_WS_AMBIGUOUS, TF_EDEPTH):
            _relevant_memo[key] = int(st[6])
        _conflict_memo[key] = memo
    if stats is not None:
        stats["sweeps"] = st[:8].tolist()
        stats["chain_depth"] = int(st[8])
        stats["ambiguous_pixels"] = int(st[9])
        stats["marker_tie_origins"] = int(st[10])
        stats["depth_origins"] = int(st[11])
        stats["reference_order"] = {"replayed_pops": int(st[13]), "seeds": int(st[14]), "microseconds": int(st[15])}
    if rc == TF_EDEPTH:
        msg = (f"watershed: {int(st[11])} pixel(s) still tie at chain depth {int(st[8])} (the deepest allowed); "
               f"{int(st[9])} label(s) may differ from the reference")
        if on_ambiguous == "ignore":
            warnings.warn(msg, WatershedAmbiguityWarning, stacklevel=2)
        else:
            raise WatershedDepthError(msg)
    elif rc == TF_WS_AMBIGUOUS and on_ambiguous != "ignore":
        msg = (f"watershed: the labels of {int(st[9])} pixel(s) depend on the order in which the reference's binary heap "
               f"pops equal-valued markers ({int(st[10])} tie point(s)); resolved by the markers' raster order")
        if on_ambiguous == "raise":
            raise WatershedAmbiguityError(msg)
        warnings.warn(msg, WatershedAmbiguityWarning, stacklevel=2)
    return (labels, amb) if return_ambiguous else labels


def watershed(
    forward_flow: np.ndarray,
    backward_flow: np.ndarray,
    field: np.ndarray,
    markers: np.ndarray,
    mask: np.ndarray | None = None,
    connectivity: int | np.ndarray = 1,
    _dev_flows=None,
    chain_depth: int = DEFAULT_CHAIN_DEPTH,
    max_chain_depth: int = MAX_CHAIN_DEPTH,
    on_ambiguous: str = "warn",
    return_ambiguous: bool = False,
) -> np.ndarray:
    """Watershed segmentation of a sequence of images in a semi-Lagrangian framework
    (reference: watershed.py:17-168).  Returns int32 labels with the shape of `field`.

    The keyword arguments after `connectivity` are not in the reference: see watershed_dev for the exactness
    contract they control (`return_ambiguous=True` returns (labels, uint8 report))."""
    t = _lib.torch()
    on_device = isinstance(field, t.Tensor)
    if hasattr(field, "to_numpy") and not isinstance(field, (np.ndarray, t.Tensor)):
        field = field.to_numpy()
    if hasattr(markers, "to_numpy") and not isinstance(markers, (np.ndarray, t.Tensor)):
        markers = markers.to_numpy()
    if tuple(markers.shape) != tuple(field.shape):
        raise ValueError(f"`markers` (shape {tuple(markers.shape)}) must have same "
                         f"shape as `image` (shape {tuple(field.shape)})")
    if mask is not None and tuple(mask.shape) != tuple(field.shape):
        raise ValueError(f"`mask` (shape {tuple(mask.shape)}) must have same shape "
                         f"as `image` (shape {tuple(field.shape)})")
    if len(field.shape) != 3:
        raise ValueError("field must have three dimensions (t, y, x)")
    nbr = neighbour_offsets(connectivity, 3)
    f = _lib.to_dev(field, t.float32)                    # watershed.py:64-65
    m = _lib.to_dev(markers, t.int32)                    # watershed.py:72-73
    k = None if mask is None else _lib.to_dev(mask).to(t.int8)   # watershed.py:78-79
    if _dev_flows is not None:
        fwd, bwd = _dev_flows
    else:
        fwd, bwd = _lib.to_dev(forward_flow, t.float32), _lib.to_dev(backward_flow, t.float32)
    if tuple(fwd.shape) != tuple(field.shape) + (2,):
        raise ValueError("flow vectors must have shape field.shape + (2,)")
    out = watershed_dev(fwd, bwd, f, m, k, nbr, chain_depth, max_chain_depth=max_chain_depth,
                        on_ambiguous=on_ambiguous, return_ambiguous=return_ambiguous)
    if return_ambiguous:
        return out if on_device else (out[0].cpu().numpy(), out[1].cpu().numpy())
    return out if on_device else out.cpu().numpy()


__all__ = ("watershed", "watershed_dev", "neighbour_offsets", "WatershedAmbiguityWarning", "WatershedAmbiguityError",
           "WatershedDepthError")
