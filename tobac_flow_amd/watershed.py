"""Watershed segmentation in a semi-Lagrangian framework, on the MI355X.

Mirrors /root/reference/tobac_flow/watershed.py:17-168 (same signature, coercions and
ValueErrors).  The reference pads the volume and calls its Cython heap flood
(tobac_flow/_watershed.pyx:222-344); here the whole flood runs in HIP (tf_watershed,
include/tobac_flow_hip.h) on the unpadded volume: out-of-volume neighbours are rejected by
coordinate tests, which is what the zero-padded mask achieves in the reference.
"""
import ctypes
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
TF_WS_AMBIGUOUS, TF_WS_REPLAY_PENDING, TF_EDEPTH = 1, 2, -5
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
TF_WS_SKIP_FAST_PATH, TF_WS_REFERENCE_ORDER, TF_WS_DEFER_SWEEPS = 1, 2, 4


_SLOTS_LOCK = __import__("threading").Lock()
_slots_busy = set()


def _take_slot():
    """workspace slot of a flood in flight: every job owns one scratch buffer (tag `watershed_job<k>`) until it is finished"""
    with _SLOTS_LOCK:
        k = 0
        while k in _slots_busy:
            k += 1
        _slots_busy.add(k)
        return k


def _give_slot(k):
    with _SLOTS_LOCK:
        _slots_busy.discard(k)


class _DepthRetry(Exception):
    """ties left by the chain depth the job was given: the synchronous wrappers start again with all twelve levels"""


class WatershedJob:
    """One flood in flight (tf_watershed_begin / _replay / _finish, include/tobac_flow_hip.h): `watershed_begin` has run the
    device part up to (not including) the root phase.  If the job was begun with a guessed tie value (on_ambiguous
    ="reference" and the previous flood of this shape had labels that hang on the order of equal-valued markers),
    `needs_replay` is True and `replay()` -- pure host work, no GIL, any thread -- computes the reference heap's pop ranks
    while the device goes on; `finish()` runs the root phase with them and returns the labels (it runs the replay itself
    if nobody has, and exports / replays after the root phase when there was no guess or the guess was too low).  Several
    jobs may be in flight: the replays of earlier windows run on worker threads while the device floods the next ones
    (bench.py)."""

    def __init__(self, handle, slot, ws, keep, shape, on_ambiguous, return_ambiguous, stats, st, memo_key, can_deepen):
        self._h, self._slot, self._ws, self._keep = handle, slot, ws, keep
        self._shape, self._on_ambiguous, self._return_ambiguous, self._stats, self._st = shape, on_ambiguous, return_ambiguous, stats, st
        self._memo_key, self._can_deepen = memo_key, can_deepen
        self.needs_replay = bool(_lib.lib().tf_watershed_needs_replay(handle))
        self.info = {}
        self._out = None
        self._sweeps_pending = False
        self._probe_noted = False

    @staticmethod
    def _info_dict(a):
        return {"replay_form": ("none", "sparse", "dense", "device")[int(a[0])], "seeds": int(a[1]), "seeds_at_or_below_tie_value": int(a[2]),
                "subgraph_pixels": int(a[3]), "relevant_pixels": int(a[4]), "export_us": int(a[5]), "replay_us": int(a[6]),
                "tie_key": int(a[7]), "guessed": bool(a[8]), "guess_covered_the_tie": bool(a[9]), "exported_for_key": int(a[10])}

    def sweeps(self):
        """Phase A and the chain levels of a job begun with defer_sweeps=True (tf_watershed_sweeps; a no-op otherwise), on the
        stream the job was begun on.  Optional -- step() / finish() run them if nobody has -- but a caller that begins several
        windows in a row gets all their set-ups (which read the flow fields) and host replays under way first."""
        if self._h is not None and self._sweeps_pending:
            self._sweeps_pending = False
            _lib.check(_lib.lib().tf_watershed_sweeps(self._h, self._st.ctypes.data_as(_lib._P)), "tf_watershed_sweeps")
            self._note_probe()
        return self

    def _note_probe(self):
        """the scheduling memo of this shape: did this job probe (speculative root phase), and did the probe find a conflict?"""
        if self._probe_noted:
            return
        self._probe_noted = True
        probed = int(self._st[5])
        with _MEMO_LOCK:
            memo = list(_conflict_memo.get(self._memo_key, (False, 0)))
            if probed >= 0:
                memo[0], memo[1] = bool(probed), 0             # this call probed
            else:
                memo[1] += 1
            _conflict_memo[self._memo_key] = memo

    def replay(self):
        if self._h is not None and self.needs_replay:
            _lib.check(_lib.lib().tf_watershed_replay(self._h), "tf_watershed_replay")
        return self

    def abandon(self):
        if self._h is not None:
            _lib.lib().tf_watershed_abandon(self._h)
            self._h = None
            self._ws = self._keep = None
            _give_slot(self._slot)

    def __del__(self):
        try:
            self.abandon()
        except Exception:                                    # interpreter shutdown
            pass

    def step(self, _deepen=False, stream=None):
        """One entry of tf_watershed_finish.  Returns (True, labels [, report]) when the flood is complete, or (False, None)
        when the library has exported for a host replay AFTER the root phase (no guess, or a guess that was too low):
        `needs_replay` is True again, run `replay()` (any thread) and call step() once more.
        stream: a torch.cuda.Stream to do this on instead of the stream the job was begun on (tf_watershed_set_stream): the
        root phase then does not queue behind whatever the main stream is busy with.  The call returns with that stream idle."""
        if self._h is None:
            raise RuntimeError("WatershedJob: the job has already been finished or abandoned")
        t = _lib.torch()
        if stream is not None:
            caller = t.cuda.current_stream()
            _lib.check(_lib.lib().tf_watershed_set_stream(self._h, ctypes.c_void_p(stream.cuda_stream)), "tf_watershed_set_stream")
            with t.cuda.stream(stream):
                res = self._step(_deepen)
            # the outputs were allocated under `stream` (allocating them on the caller's stream and writing them from this one
            # would race with whatever the caller's stream still has queued on a reused block), so the caching allocator ties
            # them to it; they are handed to the caller for use on ITS stream: tell the allocator, or a later allocation on
            # `stream` could reuse the block while the caller's kernels still read it
            for x in res[1:]:
                if isinstance(x, t.Tensor) and x.is_cuda:
                    x.record_stream(caller)
            return res
        return self._step(_deepen)

    def _step(self, _deepen):
        t = _lib.torch()
        L = _lib.lib()
        if self._out is None:
            self._out = (_lib.empty(self._shape, t.int32), _lib.empty(self._shape, t.uint8) if self._return_ambiguous else None)
        labels, amb = self._out
        st = self._st
        info = np.zeros(12, np.int64)
        rc = TF_WS_REPLAY_PENDING
        try:
            rc = L.tf_watershed_finish(self._h, _lib.ptr(labels), _lib.ptr(amb), st.ctypes.data_as(_lib._P), info.ctypes.data_as(_lib._P))
        finally:
            if rc != TF_WS_REPLAY_PENDING:                   # the library has freed the job
                self._h = None
                self._ws = self._keep = self._out = None
                _give_slot(self._slot)
        if self._sweeps_pending:                             # finish has run the deferred sweeps itself
            self._sweeps_pending = False
            self._note_probe()
        if rc == TF_WS_REPLAY_PENDING:
            self.needs_replay = True
            self.info = self._info_dict(info)
            return False, None
        self.needs_replay = False
        if rc not in (TF_WS_AMBIGUOUS, TF_EDEPTH):
            _lib.check(rc, "tf_watershed")
        self.info = self._info_dict(info)
        with _MEMO_LOCK:                                     # the next floods of this shape guess from the tie values seen
            hist = _tie_memo.setdefault(self._memo_key, [])
            hist.append(int(info[7]))
            del hist[:-_TIE_HISTORY]
        stats, on_ambiguous = self._stats, self._on_ambiguous
        if stats is not None:
            stats["sweeps"] = st[:8].tolist()
            stats["chain_depth"] = int(st[8])
            stats["ambiguous_pixels"] = int(st[9])
            stats["marker_tie_origins"] = int(st[10])
            stats["depth_origins"] = int(st[11])
            stats["root_phases"] = int(st[12])
            stats["reference_order"] = {"replayed_pops": int(st[13]), "seeds": int(st[14]), "microseconds": int(st[15])}
            stats["reference_order_detail"] = self.info
        if rc == TF_EDEPTH:
            if self._can_deepen and _deepen:
                raise _DepthRetry()
            msg = (f"watershed: {int(st[11])} pixel(s) still tie at chain depth {int(st[8])} (the deepest "
                   f"{'this job was given room for: begin it with a larger chain_depth' if self._can_deepen else 'allowed'}); "
                   f"{int(st[9])} label(s) may differ from the reference")
            if on_ambiguous == "ignore":
                warnings.warn(msg, WatershedAmbiguityWarning, stacklevel=4)
            else:
                raise WatershedDepthError(msg)
        elif rc == TF_WS_AMBIGUOUS and on_ambiguous != "ignore":
            msg = (f"watershed: the labels of {int(st[9])} pixel(s) depend on the order in which the reference's binary heap "
                   f"pops equal-valued markers ({int(st[10])} tie point(s)); resolved by the markers' raster order")
            if on_ambiguous == "raise":
                raise WatershedAmbiguityError(msg)
            warnings.warn(msg, WatershedAmbiguityWarning, stacklevel=4)
        return True, ((labels, amb) if self._return_ambiguous else labels)

    def finish(self, _deepen=False):
        """Complete the flood on this thread (replays included) and return the labels."""
        while True:
            done, out = self.step(_deepen)
            if done:
                return out
            self.replay()


# (T, H, W, neighbours, depth, stream) -> ordered keys of the largest tie value of the last floods (-1: no tie).  The guess
# handed to tf_watershed_begin is the SMALLEST of them: a guess that is too low costs what no guess costs (export after the
# root phase, second root phase), one that is too high makes the replay long -- with the background's value as the guess it
# is the dense form, ~1 s instead of 0.2 s, for every flood until the next one finishes.
_tie_memo = {}
_TIE_HISTORY = 32


def watershed_begin(fwd, bwd, field, markers, mask, nbr, chain_depth=DEFAULT_CHAIN_DEPTH, stats=None,
                    expect_conflict=None, max_chain_depth=MAX_CHAIN_DEPTH, on_ambiguous="reference", return_ambiguous=False,
                    workspace=None, guess_tie_value=True, _all_levels=False, defer_sweeps=False):
    """Device-resident core, first part: torch tensors in (field f32, markers i32, mask i8 or None) -> WatershedJob.

    expect_conflict: True / False force the scheduling hint, None (default) uses the per-shape memo.
    Exactness contract (include/tobac_flow_hip.h, tf_watershed_ex2): the library deepens the chain comparison on its
    own up to `max_chain_depth` and knows every pixel whose label still hangs on a last-resort tie-break.
      * ties between equal-valued markers (the reference resolves them by the internal state of its heap):
        `on_ambiguous="reference"` (the default since round 4: the result of the reference's own call,
        watershed.py:151 + _watershed.pyx:278-284): the library replays the reference heap's push / pop mechanics on the
        host (TF_WS_REFERENCE_ORDER) to get the markers' pop ranks and floods with those: the labels are the reference's
        bit for bit, at the cost of a sequential host pass when such a tie exists (stats["reference_order"]);
        "warn" / "raise" / "ignore": labels follow the markers' raster order, and the voxels concerned are reported;
      * ties left by the depth cut-off: finish() raises WatershedDepthError (a warning with "ignore"); watershed_dev /
        watershed start again with all twelve levels first.
    return_ambiguous: finish() also returns the (T, H, W) uint8 report (AMB_* bits).
    workspace: a uint8 device tensor the flood may use as its scratch until the job is finished (e.g. a slice of another
    stage's idle scratch, _lib.borrow_workspace); too small a one is ignored and the job allocates its own.
    guess_tie_value: with on_ambiguous="reference", begin the export for the tie value of the previous flood of this shape
    and stream (scheduling only: finish verifies the guess; see tf_watershed_begin).
    defer_sweeps: return after the set-up and the export (the parts that read the flow fields); WatershedJob.sweeps() -- or
    step() / finish() -- runs phase A and the chain levels (TF_WS_DEFER_SWEEPS; scheduling only)."""
    if on_ambiguous not in ("warn", "raise", "ignore", "reference"):
        raise ValueError("on_ambiguous must be 'warn', 'raise', 'ignore' or 'reference'")
    t = _lib.torch()
    L = _lib.lib()
    T, H, W = field.shape
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
        ties = [k for k in _tie_memo.get(key, ()) if k >= 0]
        tie_key = min(ties) if ties else -1
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
    spec = tie_key if (on_ambiguous == "reference" and guess_tie_value) else -1
    if defer_sweeps:
        flags |= TF_WS_DEFER_SWEEPS
    # levels beyond chain_depth are rarely needed: the first job gets room for two more, a second one for all
    start, cap = chain_depth, min(max_chain_depth, chain_depth + 2)
    if _all_levels:
        start, cap, flags = min(max_chain_depth, chain_depth + 3), max_chain_depth, flags | TF_WS_SKIP_FAST_PATH
    slot = _take_slot()
    handle = ctypes.c_void_p()
    try:
        while True:
            nbytes = L.tf_watershed_workspace_bytes(T, H, W, len(nbr), cap, guess)
            ws = None                            # a retry must not hold the old buffer while the larger one is allocated
            if workspace is not None and workspace.numel() >= nbytes:
                ws = workspace
            else:
                ws = _lib.workspace(nbytes, "watershed_job%d" % slot)
            rc = L.tf_watershed_begin(_lib.ptr(field), _lib.ptr(markers), _lib.ptr(mask), _lib.ptr(fwd), _lib.ptr(bwd),
                                      T, H, W, nbr.ctypes.data_as(_lib._P), len(nbr), start, cap, flags, spec,
                                      _lib.ptr(ws), ws.numel(), st.ctypes.data_as(_lib._P), _lib.stream_ptr(), ctypes.byref(handle))
            if rc == -2 and st[6] > guess:
                guess = int(st[6])
                continue
            break
        _lib.check(rc, "tf_watershed")
    except BaseException:
        _give_slot(slot)
        raise
    with _MEMO_LOCK:
        _relevant_memo[key] = int(st[6])
    job = WatershedJob(handle, slot, ws, (field, markers, mask), (T, H, W), on_ambiguous, return_ambiguous, stats, st, key,
                       can_deepen=cap < max_chain_depth)
    if defer_sweeps:
        job._sweeps_pending = True                           # (the scheduling probe is the sweeps': noted when they have run)
    else:
        job._note_probe()
    return job


def watershed_dev(fwd, bwd, field, markers, mask, nbr, chain_depth=DEFAULT_CHAIN_DEPTH, stats=None,
                  expect_conflict=None, max_chain_depth=MAX_CHAIN_DEPTH, on_ambiguous="reference", return_ambiguous=False):
    """Device-resident core: torch tensors in, labels out (`watershed_begin(...).finish()`, which see).  Round 6: when the job was
    begun with a guessed tie value, its host replay of the reference heap's order (sequential host work, 20 - 35 ms per
    16 x 5424^2 window) runs on a worker thread beside the device's phase A and chain levels instead of after them --
    what the window scheduler does for many floods, done for the one flood of a plain call."""
    def run(**kw):
        job = watershed_begin(fwd, bwd, field, markers, mask, nbr, chain_depth, stats, expect_conflict, max_chain_depth,
                              on_ambiguous, return_ambiguous, defer_sweeps=True, **kw)
        fut = _replay_worker().submit(job.replay) if job.needs_replay else None
        try:
            job.sweeps()
        finally:
            if fut is not None:
                fut.result()
        return job
    try:
        return run().finish(_deepen=True)
    except _DepthRetry:
        return run(_all_levels=True).finish()


_REPLAY_WORKER = None


def _replay_worker():
    global _REPLAY_WORKER
    if _REPLAY_WORKER is None:
        from concurrent.futures import ThreadPoolExecutor
        _REPLAY_WORKER = ThreadPoolExecutor(max_workers=2, thread_name_prefix="tf-ws-replay1")
    return _REPLAY_WORKER


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
    on_ambiguous: str = "reference",
    return_ambiguous: bool = False,
) -> np.ndarray:
    """Watershed segmentation of a sequence of images in a semi-Lagrangian framework
    (reference: watershed.py:17-168).  Returns int32 labels with the shape of `field`.

    The keyword arguments after `connectivity` are not in the reference: see watershed_begin for the exactness
    contract they control (`return_ambiguous=True` returns (labels, uint8 report)).  The default, on_ambiguous=
    "reference", returns the labels of the reference's own call bit for bit, equal-valued markers included."""
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
        return out if on_device else (_lib.to_host(out[0]), _lib.to_host(out[1]))
    return out if on_device else _lib.to_host(out)


__all__ = ("watershed", "watershed_dev", "watershed_begin", "WatershedJob", "neighbour_offsets", "WatershedAmbiguityWarning", "WatershedAmbiguityError",
           "WatershedDepthError")
