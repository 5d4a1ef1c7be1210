"""ctypes binding of libtobac_flow_hip.so (include/tobac_flow_hip.h) + device-buffer plumbing.

The HIP library is the product path: if it is missing the import of any compute entry point
fails loudly -- there is no CPU fallback.  PyTorch is used for device memory and streams only.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# TF_LIB_PATH: development switch (A/B runs of two builds of the library)
_SO = os.environ.get("TF_LIB_PATH") or os.path.join(_HERE, "csrc", "libtobac_flow_hip.so")

INTERP = {"nearest": 0, "linear": 1, "cubic": 2, "lanczos": 3}
TF_F32, TF_F64, TF_I32 = 0, 1, 2
FUNC_STACK, FUNC_SOBEL, FUNC_SOBEL_UPHILL, FUNC_SOBEL_DOWNHILL, FUNC_NANMEAN, FUNC_DIFF, FUNC_ANY, FUNC_NANMAX = range(8)


class VarRefParams(ctypes.Structure):
    _fields_ = [("fixed_point_iterations", ctypes.c_int), ("sor_iterations", ctypes.c_int), ("alpha", ctypes.c_float),
                ("delta", ctypes.c_float), ("gamma", ctypes.c_float), ("omega", ctypes.c_float)]


class FarnebackParams(ctypes.Structure):
    _fields_ = [("num_levels", ctypes.c_int), ("pyr_scale", ctypes.c_double), ("win_size", ctypes.c_int),
                ("num_iters", ctypes.c_int), ("poly_n", ctypes.c_int), ("poly_sigma", ctypes.c_double),
                ("chain_form", ctypes.c_int), ("status_slot", ctypes.c_int)]


FB_CHAIN_DEFAULT, FB_CHAIN_ONE_LANE, FB_CHAIN_TWO_PART = 0, 1, 2
TF_ESTARVED = -6


_lib = None

_c = ctypes
_P = ctypes.c_void_p
_PROTOS = {
    "tf_version": (_c.c_int, []),
    "tf_last_error": (_c.c_char_p, []),
    "tf_device_count": (_c.c_int, []),
    "tf_to8bit_workspace_bytes": (_c.c_size_t, [_c.c_int64, _c.c_int64]),
    "tf_to8bit_pair": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _P, _P, _P, _c.c_size_t, _P]),
    "tf_farneback_default_params": (None, [_c.POINTER(FarnebackParams)]),
    "tf_farneback_workspace_bytes": (_c.c_size_t, [_c.c_int64, _c.c_int64, _c.POINTER(FarnebackParams)]),
    "tf_farneback_pair": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _c.POINTER(FarnebackParams), _P, _P, _P,
                                     _c.c_size_t, _P]),
    "tf_farneback_expansion": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _c.POINTER(FarnebackParams), _P, _P, _P]),
    "tf_farneback_workspace_bytes_batch": (_c.c_size_t, [_c.c_int64, _c.c_int64, _c.c_int64, _c.POINTER(FarnebackParams)]),
    "tf_farneback_workspace_bytes_split": (_c.c_size_t, [_c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _c.POINTER(FarnebackParams)]),
    "tf_farneback_batch_split": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _c.POINTER(FarnebackParams),
                                            _P, _P, _c.c_int64, _P, _c.c_size_t, _P]),
    "tf_farneback_can_split": (_c.c_int, [_c.c_int64, _c.c_int64, _c.POINTER(FarnebackParams)]),
    "tf_farneback_workspace_bytes_phase": (_c.c_size_t, [_c.c_int64, _c.c_int64, _c.c_int64, _c.POINTER(FarnebackParams), _c.c_int]),
    "tf_farneback_batch_phase": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _c.POINTER(FarnebackParams),
                                            _P, _P, _c.c_int64, _P, _c.c_size_t, _P, _c.c_int]),
    "tf_farneback_batch_hint": (_c.c_int64, [_c.c_int64, _c.c_int64, _c.POINTER(FarnebackParams), _c.c_int64, _c.c_size_t]),
    "tf_farneback_check": (_c.c_int, []),
    "tf_farneback_debug_set_starved": (_c.c_int, []),
    "tf_farneback_status_acquire": (_c.c_int, []),
    "tf_farneback_status_check": (_c.c_int, [_c.c_int]),
    "tf_farneback_status_release": (None, [_c.c_int]),
    "tf_farneback_debug_set_starved_slot": (_c.c_int, [_c.c_int]),
    "tf_farneback_iteration_workgroups": (_c.c_int64, [_c.c_int64, _c.c_int64, _c.POINTER(FarnebackParams), _c.c_int64, _P]),
    "tf_farneback_batch": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _c.POINTER(FarnebackParams),
                                      _P, _P, _c.c_int64, _P, _c.c_size_t, _P]),
    "tf_varref_default_params": (None, [_c.POINTER(VarRefParams)]),
    "tf_varref_workspace_bytes": (_c.c_size_t, [_c.c_int64, _c.c_int64]),
    "tf_varref": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _c.POINTER(VarRefParams), _P, _P, _c.c_size_t, _P]),
    "tf_varref_ex": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _c.POINTER(VarRefParams), _P, _c.c_int, _P, _c.c_size_t, _P]),
    "tf_varref_workspace_bytes_batch": (_c.c_size_t, [_c.c_int64, _c.c_int64, _c.c_int64]),
    "tf_varref_batch": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _c.POINTER(VarRefParams), _P, _c.c_int64, _c.c_int,
                                   _P, _c.c_size_t, _P]),
    "tf_smooth_flow_step": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _c.c_int, _P, _P, _P]),
    "tf_smooth_flow_step_clip": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _c.c_int, _P, _P, _c.c_float, _P]),
    "tf_warp_flow": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _c.c_int, _P, _P]),
    "tf_flow_finalize": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_float, _P]),
    "tf_flow_finalize_ends": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_float, _P]),
    "tf_convolve": (_c.c_int, [_P, _c.c_int, _c.c_int64, _c.c_int64, _c.c_int64, _P, _P, _P, _c.c_int,
                               _c.c_double, _c.c_int, _P, _c.c_int, _c.c_int64, _c.c_int64, _P]),
    "tf_edge_field": (_c.c_int, [_P, _P, _c.c_int64, _P, _c.c_int, _P]),
    "tf_sobel_edge_field": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _c.c_int64, _P, _P, _c.c_int, _P, _c.c_int, _P]),
    "tf_watershed_workspace_bytes": (_c.c_size_t, [_c.c_int64, _c.c_int64, _c.c_int64, _c.c_int, _c.c_int, _c.c_int64]),
    "tf_watershed": (_c.c_int, [_P, _P, _P, _P, _P, _c.c_int64, _c.c_int64, _c.c_int64, _P, _c.c_int, _c.c_int,
                                _P, _P, _c.c_size_t, _P, _P]),
    "tf_watershed_ex": (_c.c_int, [_P, _P, _P, _P, _P, _c.c_int64, _c.c_int64, _c.c_int64, _P, _c.c_int, _c.c_int,
                                   _c.c_int, _P, _P, _c.c_size_t, _P, _P]),
    "tf_watershed_ex2": (_c.c_int, [_P, _P, _P, _P, _P, _c.c_int64, _c.c_int64, _c.c_int64, _P, _c.c_int, _c.c_int,
                                    _c.c_int, _c.c_int, _P, _P, _P, _c.c_size_t, _P, _P]),
    "tf_watershed_begin": (_c.c_int, [_P, _P, _P, _P, _P, _c.c_int64, _c.c_int64, _c.c_int64, _P, _c.c_int, _c.c_int,
                                      _c.c_int, _c.c_int, _c.c_int64, _P, _c.c_size_t, _P, _P, _c.POINTER(_P)]),
    "tf_watershed_needs_replay": (_c.c_int, [_P]),
    "tf_watershed_sweeps": (_c.c_int, [_P, _P]),
    "tf_watershed_replay": (_c.c_int, [_P]),
    "tf_watershed_finish": (_c.c_int, [_P, _P, _P, _P, _P]),
    "tf_watershed_abandon": (None, [_P]),
    "tf_watershed_set_stream": (_c.c_int, [_P, _P]),
    "tf_watershed_job_info": (_c.c_int, [_P, _P]),
    "tf_watershed_raveled_workspace_bytes": (_c.c_size_t, [_c.c_int64, _c.c_int, _c.c_int, _c.c_int64]),
    "tf_watershed_raveled": (_c.c_int, [_P, _c.c_int64, _P, _c.c_int64, _P, _c.c_int, _P, _P, _P, _P, _P, _P, _c.c_int,
                                        _c.c_double, _P, _c.c_int, _c.c_int, _P, _c.c_size_t, _P, _P]),
    "tf_watershed_raveled_ex": (_c.c_int, [_P, _c.c_int64, _P, _c.c_int64, _P, _c.c_int, _P, _P, _P, _P, _P, _P, _c.c_int,
                                           _c.c_double, _P, _c.c_int, _c.c_int, _c.c_int, _P, _c.c_size_t, _P, _P]),
    "tf_binary_morph": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _c.c_int64, _P, _c.c_int, _c.c_int, _c.c_int, _P, _P, _P]),
    "tf_correlate1d_sym": (_c.c_int, [_P, _c.c_int, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int, _P, _c.c_int, _P, _P]),
    "tf_grey_morph": (_c.c_int, [_P, _c.c_int, _c.c_int64, _c.c_int64, _c.c_int64, _P, _c.c_int, _P, _P]),
    "tf_linearise": (_c.c_int, [_P, _c.c_int64, _c.c_double, _c.c_double, _P, _P]),
    "tf_label_extent": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int, _P, _P, _P, _P]),
    "tf_apply_lut": (_c.c_int, [_P, _c.c_int64, _P, _c.c_int, _P, _P]),
    "tf_apply_lut_keep_nonpositive": (_c.c_int, [_P, _c.c_int64, _P, _c.c_int, _P, _P]),
    "tf_field_masks": (_c.c_int, [_P, _c.c_int64, _P, _P, _P, _P]),
    "tf_merge_seeds": (_c.c_int, [_P, _P, _P, _c.c_int64, _P, _P]),
    "tf_label_workspace_bytes": (_c.c_size_t, [_c.c_int64, _c.c_int64, _c.c_int64]),
    "tf_label": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _c.c_int64, _P, _P, _P, _P, _c.c_size_t, _P]),
    "tf_pair_counts_workspace_bytes": (_c.c_size_t, [_c.c_int64, _c.c_int64]),
    "tf_slice_labels_workspace_bytes": (_c.c_size_t, [_c.c_int64, _c.c_int64]),
    "tf_label_stats_workspace_bytes": (_c.c_size_t, [_c.c_int64]),
    "tf_label_stats": (_c.c_int, [_P, _P, _P, _c.c_int64, _c.c_int64, _P, _P, _c.c_size_t, _P]),
    "tf_slice_labels": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _P, _P, _P, _c.c_size_t, _P]),
    "tf_pair_counts": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int, _P, _P, _P, _c.c_int64, _P, _P, _c.c_size_t, _P]),
    "tf_label_sizes": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _P, _P]),
    "tf_pair_rank": (_c.c_int, [_P, _P, _c.c_int64, _P, _P, _c.c_int64, _P, _P]),
    "tf_flow_link_workspace_bytes": (_c.c_size_t, [_c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64]),
    "tf_flow_link_overlap": (_c.c_int, [_P, _P, _P, _c.c_int64, _c.c_int64, _c.c_int64, _P, _c.c_double, _c.c_int64, _P, _P,
                                        _P, _c.c_size_t, _P]),
    "tf_flow_label_workspace_bytes": (_c.c_size_t, [_c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64]),
    "tf_flow_label": (_c.c_int, [_P, _P, _P, _c.c_int64, _c.c_int64, _c.c_int64, _P, _c.c_double, _c.c_int64, _P, _P,
                                 _P, _c.c_size_t, _P]),
    "tf_window_overlap_pairs": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _c.c_double, _P, _c.c_int64, _P, _P,
                                           _c.c_size_t, _P]),
    "tf_profile_enable": (_c.c_int, [_c.c_int]),
    "tf_profile_kernel_count": (_c.c_int, []),
    "tf_profile_kernel_name": (_c.c_char_p, [_c.c_int]),
    "tf_profile_collect": (_c.c_int, [_P, _P, _P]),
    "tf_host_alloc": (_c.c_int, [_c.c_size_t, _c.POINTER(_P)]),
    "tf_host_free": (_c.c_int, [_P]),
    "tf_host_is_pinned": (_c.c_int, [_P, _c.c_size_t]),
    "tf_host_pool_stats": (_c.c_int, [_P, _P]),
    "tf_host_pool_spare": (_c.c_int, [_c.c_size_t]),
    "tf_host_pool_trim": (_c.c_int, [_c.c_size_t]),
    "tf_upload": (_c.c_int, [_P, _P, _c.c_size_t, _P, _P]),
    "tf_download": (_c.c_int, [_P, _P, _c.c_size_t, _P]),
    "tf_hash_host": (_c.c_int, [_P, _c.c_size_t, _P]),
    "tf_hash_dev": (_c.c_int, [_P, _c.c_size_t, _P, _P]),
    "tf_copy16": (_c.c_int, [_P, _P, _c.c_size_t, _P]),
    "tf_copy16_variant": (_c.c_int, [_P, _P, _c.c_size_t, _P, _c.c_int]),
    "tf_shutdown": (_c.c_int, []),
    "tf_selftest_shared_divide": (_c.c_int, [_c.c_int64, _c.c_uint64, _P, _P]),
}

EXPORTS = tuple(_PROTOS)


def lib_path():
    return _SO


def lib():
    """Load the HIP library (once). Raises ImportError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise ImportError(
                f"{_SO} not found: build it with `make -C tobac_flow_amd/csrc` (or __graft_entry__.build()). "
                "tobac_flow_amd has no CPU fallback.")
        # torch first: it brings its own HIP runtime (libamdhip64), and the library must bind to THAT instance -- loaded the
        # other way round the process ends up with two runtimes and the library's sees no device (hipErrorNoDevice)
        torch()
        L = ctypes.CDLL(_SO)
        for name, (res, args) in _PROTOS.items():
            if os.environ.get("TF_LIB_PATH") and not hasattr(L, name):
                continue                         # A/B runs against an older build of the library: entry points it lacks stay unbound
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


class TobacFlowHipError(RuntimeError):
    pass


def check(rc, what):
    if rc == 0:
        return
    msg = lib().tf_last_error().decode(errors="replace")
    if rc == -2:
        raise MemoryError(f"{what}: {msg}")
    if rc == -1:
        raise ValueError(f"{what}: {msg}")
    raise TobacFlowHipError(f"{what} failed (code {rc}): {msg}")


# ---- device plumbing (torch) ---------------------------------------------------------------------
_torch = None


def torch():
    global _torch
    if _torch is None:
        import torch as _t
        _torch = _t
    return _torch


def device():
    t = torch()
    if not t.cuda.is_available():
        raise TobacFlowHipError("no HIP device visible: tobac_flow_amd computes on an MI355X only (no CPU fallback)")
    lib()
    return t.device("cuda", t.cuda.current_device())


def is_tensor(x):
    return _torch is not None and isinstance(x, _torch.Tensor) or (type(x).__module__.startswith("torch"))


def to_dev(x, dtype=None, share=False):
    """numpy array / torch tensor -> contiguous torch tensor on the current HIP device.  Host arrays go through the pinned
    staging ring (_staging.upload: host threads + pipelined DMA instead of the runtime's pageable copy).  share=True -- for
    inputs the caller only READS (the fields and label volumes handed to the entry points): an array whose content has been
    uploaded or downloaded before is served from HBM instead of crossing PCIe again (recognised by a content checksum)."""
    t = torch()
    dev = device()
    if isinstance(x, t.Tensor):
        y = x.to(dev)
    else:
        from tobac_flow_amd import _staging
        a = np.asarray(x)
        try:
            y = _staging.upload(a, fresh=not share)
        except TypeError:                       # a dtype the staging path does not take (object, big-endian, ...): torch's own
            if a.dtype == np.bool_:
                a = a.astype(np.uint8)
            y = t.from_numpy(np.ascontiguousarray(a)).to(dev)
    if dtype is not None and y.dtype != dtype:
        y = y.to(dtype)
    return y.contiguous()


def to_host(x, remember=False):
    """device tensor -> numpy array in a pooled pinned block (_staging.download: DMA at the link's rate, no second host
    copy).  remember=True -- for results handed back through the reference's interface: the device tensor stays cached as
    the array's twin, so that the array coming back as an input (`markers=`) does not cross PCIe again."""
    from tobac_flow_amd import _staging
    return _staging.download(x, remember=remember)


def ptr(tensor):
    return ctypes.c_void_p(tensor.data_ptr()) if tensor is not None else None


def stream_ptr():
    t = torch()
    return ctypes.c_void_p(t.cuda.current_stream().cuda_stream)


def empty(shape, dtype):
    t = torch()
    return t.empty(shape, dtype=dtype, device=device())


_WS = {}
_WS_LOCK = __import__("threading").Lock()


def workspace(nbytes, tag="default"):
    """Grow-only scratch buffer (uint8 tensor) per (tag, device, STREAM): two host threads that drive the library on two
    streams never share scratch memory (the library uses the buffer on the stream it is handed, so one buffer per stream
    is exactly what keeps concurrent calls apart; calls on one stream are ordered by the stream)."""
    t = torch()
    dev = device()
    key = (tag, dev.index, t.cuda.current_stream().cuda_stream)
    with _WS_LOCK:
        cur = _WS.get(key)
        if cur is not None and cur.numel() >= nbytes:
            return cur
        _WS[key] = None                          # drop the old buffer BEFORE the new one is allocated (they can be > 100 GB)
    cur = None
    # allocated OUTSIDE the lock (an allocation of tens of GB, or the empty_cache below, held up every other
    # thread's scratch lookup); the key is (tag, device, stream): only this stream's caller can be here for it
    try:
        cur = t.empty(int(nbytes), dtype=t.uint8, device=dev)
    except t.OutOfMemoryError:
        # the scratch is ONE block: free memory that the caching allocator holds in fragments cannot serve it.
        # Hand the cache back to the driver once and try again (slow -- seconds for > 100 GB -- hence only here)
        t.cuda.empty_cache()
        try:
            cur = t.empty(int(nbytes), dtype=t.uint8, device=dev)
        except t.OutOfMemoryError:
            # last resort: the scratch of the OTHER stages of this (device, stream) -- tens of GB that sit idle between their
            # calls (the Farneback workspace of an earlier create_flow) -- goes back as well; they grow again on their next use
            with _WS_LOCK:
                for k in [k for k in _WS if k[1] == key[1] and k[2] == key[2] and k != key]:
                    del _WS[k]
            try:                                # ... and the device twins remembered for host arrays (they are a convenience)
                from tobac_flow_amd import _staging
                _staging.clear(trim=False)
            except Exception:                   # noqa: BLE001
                pass
            t.cuda.empty_cache()
            cur = t.empty(int(nbytes), dtype=t.uint8, device=dev)
    with _WS_LOCK:
        _WS[key] = cur
    return cur


def borrow_workspace(tag):
    """the scratch buffer this (device, stream) currently holds under `tag`, or None: a caller that knows the owner is idle
    (the Farneback scratch after create_flow has returned) may lend it to another stage instead of allocating beside it"""
    t = torch()
    key = (tag, device().index, t.cuda.current_stream().cuda_stream)
    with _WS_LOCK:
        return _WS.get(key)


def workspace_bytes(tag):
    """bytes of scratch this (device, stream) holds under `tag` and its sub-tags `tag_*` (0 if none)"""
    t = torch()
    dev, stream = device().index, t.cuda.current_stream().cuda_stream
    with _WS_LOCK:
        return sum(int(v.numel()) for k, v in _WS.items()
                   if v is not None and k[1] == dev and k[2] == stream and (k[0] == tag or k[0].startswith(tag + "_")))


def release_workspaces(tag=None, stream=None):
    """Hand the scratch buffers (all, or those of one tag and of its sub-tags `tag_*`; of every stream, or of one
    `torch.cuda.Stream` / raw stream handle) back to torch's caching allocator.  The buffers are keyed by the calling
    stream and never dropped on their own: a caller that works on short-lived streams releases theirs when it is done
    with them."""
    handle = None if stream is None else int(getattr(stream, "cuda_stream", stream))
    with _WS_LOCK:
        for key in [k for k in _WS if (tag is None or k[0] == tag or k[0].startswith(tag + "_")) and (handle is None or k[2] == handle)]:
            del _WS[key]


def profile_enable(on=True):
    lib().tf_profile_enable(1 if on else 0)


def profile_collect():
    """{kernel name: (calls, total ms, total algorithmic bytes)} since the last collect."""
    L = lib()
    n = L.tf_profile_kernel_count()
    calls = np.zeros(n, np.int64)
    ms = np.zeros(n, np.float64)
    by = np.zeros(n, np.float64)
    L.tf_profile_collect(calls.ctypes.data_as(_P), ms.ctypes.data_as(_P), by.ctypes.data_as(_P))
    return {L.tf_profile_kernel_name(i).decode(): (int(calls[i]), float(ms[i]), float(by[i])) for i in range(n) if calls[i]}
