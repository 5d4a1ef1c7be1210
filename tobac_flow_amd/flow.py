"""Flow object API on the MI355X (mirrors /root/reference/tobac_flow/flow.py).

Same public names, signatures, defaults and exceptions as the reference: `create_flow`,
`calculate_flow`, `calculate_flow_2`, `calculate_flow_frame`, `smooth_flow_step`, `Flow` and the
diagnostics.  cv2's Farnebaeck / remap calls are replaced by the HIP library
(include/tobac_flow_hip.h); numpy inputs give numpy outputs, torch GPU tensors stay on the device.

cv2.VariationalRefinement (`vr_steps > 0`, flow.py:359, 513-519) is tf_varref (csrc/varref.hip); like the reference,
any vr_steps > 0 runs exactly ONE refinement per direction.
"""
import contextlib
import ctypes
import os
import warnings
from datetime import datetime
from typing import Callable

import numpy as np
from scipy import ndimage as ndi

from tobac_flow_amd import _lib
from tobac_flow_amd.convolve import convolve, tag_func
from tobac_flow_amd.core import AbstractFlow
from tobac_flow_amd.label import flow_label, flow_link_overlap
from tobac_flow_amd.sobel import sobel
from tobac_flow_amd.utils import mse, select_normalisation_method, select_of_model, to_8bit, warp_flow  # noqa: F401
from tobac_flow_amd.utils.flow_utils import select_interp_mode
from tobac_flow_amd.utils.normalisation_utils import linear_norm, to_8bit_pair_dev
from tobac_flow_amd.watershed import watershed


def _is_dataarray(x):
    return hasattr(x, "dims") and hasattr(x, "to_numpy")


def _unwrap_device_field(x):
    """detection.DeviceField (a device tensor + its time coordinate) -> the tensor"""
    inner = getattr(x, "data", None)
    return inner if isinstance(inner, _lib.torch().Tensor) and not isinstance(x, _lib.torch().Tensor) else x


def create_flow(data, model: str = "Farneback", vr_steps: int = 0, smoothing_passes: int = 0,
                interp_method: str = "linear", max_value=20, on_frames_ready=None, workspace_gb=None,
                split_parts=None, deferred_check=None) -> "Flow":
    """Forward and backward optical flow along the leading dimension of `data`, clipped to
    +-`max_value` pixels, wrapped in a Flow object (reference: flow.py:23-65).

    on_frames_ready (not in the reference; device-resident input only): `f(flow, n)` is called after every batch of frame
    pairs with the Flow object under construction and the number n of leading frames whose windows are complete:
    `flow.window_view(a, b)` / `flow.window(a, b)` with b <= n equal those of the finished object, bit for bit (the flow of
    a frame pair does not depend on the rest of the stack).  A long stack processed as overlapping time windows can so
    start on its first windows -- Sobel, seeds, flood, host work -- while the device computes the flow of the later frames
    (parallel.detect_stack_windows).  The calls are made on the calling thread, in order; n == len(data) in the last one.

    workspace_gb / split_parts (not in the reference; scheduling only, the flow is the same bits): the scratch budget of the
    Farneback batches in GB (default 115, and never more than 60 % of the free device memory) and the number of parts a batch's
    two finest pyramid levels, refinement and smoothing are run in (default 1; a caller that runs other work beside the
    flow -- detect_stack_windows -- asks for less scratch and for 2 parts).  The environment variables TF_FLOW_WORKSPACE_GB /
    TF_FLOW_SPLIT (development switches) override them.

    deferred_check (not in the reference; device-resident input only): create_flow on a device tensor returns while the device
    is still computing, so the library's report on those launches (a starved chain of the iteration kernel -> NaN rows,
    TF_ESTARVED) arrives later.  False (the default without on_frames_ready): the first method of the returned Flow that is
    called WAITS for the flow's launches and raises if they failed -- a plain caller can never compute from NaN rows unnoticed
    (the reference's call is synchronous).  True (the default with on_frames_ready): the Flow's methods only poll,
    and the caller -- a pipeline that keeps the host ahead of the device, parallel.detect_stack_windows -- calls `flow.check()`
    before it hands results on."""
    # the clip of flow.py:60-61 is applied by the same kernel that mirrors the end frames (tf_flow_finalize);
    # clipping commutes with the sign-flipped mirror
    extra = {}
    if on_frames_ready is not None:
        holder = {}

        def on_batch(forward, backward, pairs_done, n_pairs):
            if "flow" not in holder:
                holder["flow"] = Flow(forward, backward)
            on_frames_ready(holder["flow"], pairs_done + 1 if pairs_done < n_pairs else n_pairs + 1)
        extra["_on_batch"] = on_batch
    # host (numpy / DataArray) input: the flow stays in HBM, where every Flow method works, and the object's numpy arrays
    # `forward_flow` / `backward_flow` are downloaded when somebody reads them -- the drop-in scripts never do
    # (scripts/dcc_detect_goes.py:164-303 only pass the object on), and 2 x 7.5 GB per 16 x 5424^2 window would cross PCIe
    # twice otherwise (down here, up again at the first Flow method)
    extra["_device_out"] = True
    checks = []
    extra["_check_out"] = checks                # device output: the starved-chain check arrives as a poll bound to an event
    if workspace_gb is not None:
        extra["_workspace_gb"] = float(workspace_gb)
    if split_parts is not None:
        extra["_split_parts"] = int(split_parts)
    data = _unwrap_device_field(data)
    forward_flow, backward_flow = calculate_flow(data, model=model, vr_steps=vr_steps,
                                                 smoothing_passes=smoothing_passes, interp_method=interp_method,
                                                 _max_value=float(max_value), **extra)
    t = _lib.torch()
    if on_frames_ready is not None and "flow" in holder:
        flow = holder["flow"]
    elif isinstance(data, t.Tensor):
        flow = Flow(forward_flow, backward_flow)
    else:
        flow = Flow._lazy_host(forward_flow, backward_flow)
    if checks:
        if isinstance(data, t.Tensor):
            flow._pending_check = checks[0]       # looked at by every use of the object (Flow._dev_flows); Flow.check() waits for it
            flow._check_blocking = not (deferred_check if deferred_check is not None else on_frames_ready is not None)
        else:
            checks[0](block=True)                # host input: the reference's call is synchronous, so is this one
    return flow


@tag_func(_lib.FUNC_DIFF)
def _diff_func(x):
    """flow.py:180-184: centred semi-Lagrangian difference, one-sided where a neighbour is missing."""
    return (np.nansum([x[2] - x[1], x[1] - x[0]], axis=0) * 1
            / np.maximum(np.sum([np.isfinite(x[2]), np.isfinite(x[0])], 0), 1))


class Flow(AbstractFlow):
    """Semi-Lagrangian operations using optical flow vectors (reference: flow.py:68-355).

    `forward_flow` / `backward_flow`: (T, H, W, 2) arrays (numpy, or torch tensors on the GPU).
    A float32 device copy is cached on first use; do not mutate the arrays afterwards.
    """

    def __init__(self, forward_flow, backward_flow) -> None:
        if tuple(forward_flow.shape) != tuple(backward_flow.shape):
            raise ValueError("Forward and backward flow vector arrays must have the same shape")
        if forward_flow.shape[-1] != 2:
            raise ValueError("Flow vectors must have a size of 2 in the trailing dimension")
        self.shape = tuple(forward_flow.shape[:-1])
        self._fw = forward_flow
        self._bw = backward_flow
        self._dev = None
        self._pending_check = None
        self._check_blocking = True               # a pending check is WAITED for by the first method called (create_flow: deferred_check)

    def check(self):
        """Wait for the launches that produced this object and raise if one of them reported a starved chain (create_flow on
        device-resident input returns before the device is done: its check is deferred to the object's uses and to this call)."""
        if self._pending_check is not None:
            poll, self._pending_check = self._pending_check, None
            poll(block=True)

    @classmethod
    def _lazy_host(cls, forward_dev, backward_dev) -> "Flow":
        """The Flow create_flow returns for host input: the vectors live on the device (what every method works on);
        the numpy arrays `forward_flow` / `backward_flow` of the reference's object are materialised when first read."""
        obj = cls(forward_dev, backward_dev)
        obj._dev = (forward_dev, backward_dev)
        obj._fw = obj._bw = None
        return obj

    def _host_arrays(self):
        if self._pending_check is not None and self._check_blocking:
            self.check()                          # (the vectors themselves are a result: never handed out unchecked)
        if self._fw is None:
            self._fw, self._bw = _lib.to_host(self._dev[0]), _lib.to_host(self._dev[1])
        return self._fw, self._bw

    @property
    def forward_flow(self):
        return self._host_arrays()[0]

    @forward_flow.setter
    def forward_flow(self, value):
        self._host_arrays()
        self._fw, self._dev = value, None

    @property
    def backward_flow(self):
        return self._host_arrays()[1]

    @backward_flow.setter
    def backward_flow(self, value):
        self._host_arrays()
        self._bw, self._dev = value, None

    @property
    def flow(self):
        return self.forward_flow, self.backward_flow

    def __getitem__(self, items) -> "Flow":
        return Flow(self.forward_flow[items], self.backward_flow[items])

    def _dev_flows(self):
        if self._pending_check is not None:
            if self._check_blocking:             # plain callers: like the reference's synchronous call, never results from NaN rows
                self.check()
            else:                                # pipelined callers (deferred_check): reports once the producing launches have finished
                self._pending_check()
        if self._dev is None:
            t = _lib.torch()
            self._dev = (_lib.to_dev(self._fw, t.float32), _lib.to_dev(self._bw, t.float32))
        return self._dev

    def convolve(self, data, structure=ndi.generate_binary_structure(3, 1), method: str = "linear",
                 fill_value: float = np.nan, dtype: type = np.float32, func: Callable | None = None):
        assert tuple(data.shape) == self.shape, "Data input must have the same shape as the Flow object"
        return convolve(data, self._fw, self._bw, structure=structure, method=method,
                        dtype=dtype, fill_value=fill_value, func=func, _dev_flows=self._dev_flows())

    def diff(self, data, method: str = "linear", dtype: type = np.float32):
        diff_struct = np.zeros([3, 3, 3])
        diff_struct[:, 1, 1] = 1
        return self.convolve(data, structure=diff_struct, func=_diff_func, method=method, dtype=dtype)

    def sobel(self, data, method: str = "linear", dtype: type = None, fill_value: float = np.nan,
              direction: str | None = None):
        return sobel(data, self._fw, self._bw, method=method, dtype=dtype,
                     fill_value=fill_value, direction=direction, _dev_flows=self._dev_flows())

    def window(self, start: int, stop: int) -> "Flow":
        """The Flow object create_flow(data[start:stop]) would return, bit for bit, cut out of this one: the flow of a
        frame pair does not depend on the stack it is part of, only the two END frames of a stack are special (their
        missing neighbour is mirrored: forward[-1] = -backward[-1], backward[0] = -forward[0], flow.py:425-426).  A long
        stack processed as overlapping time windows (scripts/dcc_detect_goes.py:153) therefore needs ONE flow
        computation, not one per window: the frames two windows share are not computed twice.  (Not in the reference.)"""
        T = self.shape[0]
        start, stop, _ = slice(start, stop).indices(T)
        if stop - start < 1:
            raise ValueError("empty window")
        t = _lib.torch()
        on_device = isinstance(self.forward_flow, t.Tensor)
        if on_device:
            fw, bw = self.forward_flow[start:stop], self.backward_flow[start:stop]
            if stop < T or start > 0:
                fw, bw = fw.clone(), bw.clone()
                if stop - start > 1:
                    fw[-1] = -bw[-1]
                    bw[0] = -fw[0]
                else:
                    fw[-1].fill_(float("nan"))
                    bw[0].fill_(float("nan"))
        else:
            fw, bw = np.array(self.forward_flow[start:stop]), np.array(self.backward_flow[start:stop])
            if stop - start > 1:
                fw[-1] = -bw[-1]
                bw[0] = -fw[0]
            elif stop < T or start > 0:
                fw[-1], bw[0] = np.nan, np.nan
        return Flow(fw, bw)

    @contextlib.contextmanager
    def window_view(self, start: int, stop: int):
        """`with flow.window_view(a, b) as w:` -- window(a, b) WITHOUT the copy (7.5 GB per 16 x 5424^2 window): `w` views this
        object's arrays, whose two frames at the window's ends are overwritten with the mirrored values for the duration of
        the block and restored, bit for bit, when it is left (also by an exception).  Inside the block neither this
        object nor another window of it may be used, and `w` must not be used after it.  (Not in the reference.)"""
        T = self.shape[0]
        start, stop, _ = slice(start, stop).indices(T)
        if stop - start < 1:
            raise ValueError("empty window")
        fw, bw = self.forward_flow[start:stop], self.backward_flow[start:stop]
        copy = (lambda a: a.clone()) if isinstance(fw, _lib.torch().Tensor) else (lambda a: a.copy())
        saved_f = copy(fw[-1]) if stop < T else None
        saved_b = copy(bw[0]) if start > 0 else None
        try:
            if stop - start > 1:
                if saved_f is not None:
                    fw[-1] = -bw[-1]
                if saved_b is not None:
                    bw[0] = -fw[0]
            else:
                if saved_f is not None or saved_b is not None:
                    saved_f = copy(fw[-1]) if saved_f is None else saved_f
                    saved_b = copy(bw[0]) if saved_b is None else saved_b
                    fw[-1] = float("nan")
                    bw[0] = float("nan")
            yield Flow(fw, bw)
        finally:
            if saved_f is not None:
                fw[-1] = saved_f
            if saved_b is not None:
                bw[0] = saved_b

    def watershed(self, field, markers, mask=None, connectivity=1, **kwargs):
        """reference: flow.py (Flow.watershed).  Extra keywords (`on_ambiguous`, `return_ambiguous`, `chain_depth`,
        `max_chain_depth`: the exactness contract of tobac_flow_amd.watershed.watershed) are passed through."""
        return watershed(self._fw, self._bw, field, markers, mask=mask,
                         connectivity=connectivity, _dev_flows=self._dev_flows(), **kwargs)

    def label(self, data, structure=ndi.generate_binary_structure(3, 1), dtype: type = np.int32,
              overlap: float = 0, absolute_overlap: int = 1, subsegment_shrink: float = 0,
              peak_min_distance: int = 5):
        return flow_label(self, data, structure=structure, dtype=dtype, overlap=overlap,
                          absolute_overlap=absolute_overlap, subsegment_shrink=subsegment_shrink,
                          peak_min_distance=peak_min_distance)

    def link_overlap(self, data, structure=ndi.generate_binary_structure(3, 1), dtype: type = np.int32,
                     overlap: float = 0, absolute_overlap: int = 1):
        return flow_link_overlap(self, data, structure=structure, dtype=dtype, overlap=overlap,
                                 absolute_overlap=absolute_overlap)


class VariationalRefinement:
    """Stand-in for the object cv2.VariationalRefinement.create() returns (reference: flow.py:359): `.calc(I0, I1,
    flow)` refines a dense flow field by OpenCV's variational refinement (tf_varref, include/tobac_flow_hip.h; defaults
    fixedPointIterations 5, sorIterations 5, alpha 20, delta 5, gamma 10, omega 1.6 -- the getters / setters OpenCV
    exposes are plain attributes here).  Like OpenCV it updates `flow` in place and also returns it (numpy in -> the
    numpy array is written back; device tensor in -> refined on the device)."""

    def __init__(self):
        p = _lib.VarRefParams()
        _lib.lib().tf_varref_default_params(ctypes.byref(p))
        self.fixedPointIterations, self.sorIterations = p.fixed_point_iterations, p.sor_iterations
        self.alpha, self.delta, self.gamma, self.omega = p.alpha, p.delta, p.gamma, p.omega
        # not in OpenCV: True = hardware reciprocals instead of correctly rounded divisions / square roots
        # (TF_VR_FAST_DIVIDE, include/tobac_flow_hip.h: within 1e-4 px of the default, ~25 % faster); default False
        self.fastDivide = os.environ.get("TF_VR_FAST_DIVIDE", "0") == "1"
        # ... in the SOR sweeps only (TF_VR_FAST_SOR: the system assembly keeps its correctly rounded divisions)
        self.fastSor = os.environ.get("TF_VR_FAST_SOR", "0") == "1"

    @classmethod
    def create(cls):
        return cls()

    def _params(self):
        return _lib.VarRefParams(int(self.fixedPointIterations), int(self.sorIterations), float(self.alpha),
                                 float(self.delta), float(self.gamma), float(self.omega))

    def calc_dev(self, i0, i1, flow):
        """uint8 device frames (H, W), float32 device flow (H, W, 2) refined in place (may be a view with contiguous
        rows, e.g. one frame of a (T, H, W, 2) array)."""
        L = _lib.lib()
        H, W = i0.shape
        assert flow.is_contiguous() and tuple(flow.shape) == (H, W, 2)
        ws = _lib.workspace(L.tf_varref_workspace_bytes(H, W), "varref")
        p = self._params()
        _lib.check(L.tf_varref_ex(_lib.ptr(i0), _lib.ptr(i1), H, W, ctypes.byref(p), _lib.ptr(flow), (1 if self.fastDivide else 0) | (2 if self.fastSor else 0),
                                  _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "tf_varref")
        return flow

    def calc_batch_dev(self, i0, i1, flow, rounds=32):
        """The refinement of B images at once (tf_varref_batch: the grid gains an image dimension, the bits are calc_dev's):
        i0, i1 (B, H, W) uint8 device tensors, flow (B, H, W, 2) float32 refined in place -- frames with contiguous rows, any
        stride between them (views into the stack's flow arrays).  Images go out in groups that fill ~`rounds` rounds of the
        fused SOR kernel's tiles on the device's CUs: at 1500 x 2500 (282 tiles) all 23 pairs of BASELINE config C in one set
        of 11 launches; at 5424^2 (2279 tiles, 8.9 rounds by itself) one image per launch, as before round 6."""
        L = _lib.lib()
        t = _lib.torch()
        B, H, W = i0.shape
        assert tuple(i1.shape) == (B, H, W) and tuple(flow.shape) == (B, H, W, 2)
        assert i0.is_contiguous() and i1.is_contiguous() and flow[0].is_contiguous()
        if B == 1:
            self.calc_dev(i0[0], i1[0], flow[0])
            return flow
        tiles = -(-W // 108) * -(-H // 84)
        cus = t.cuda.get_device_properties(i0.device).multi_processor_count
        group = int(max(1, min(B, -(-rounds * cus // tiles))))
        # ... but only where ONE image does not fill the chip for several rounds by itself (fewer tiles than 4 x the CUs) and within
        # 6.5 GB of scratch (72 B per pixel and image).  At 5424^2 an image is 8.9 rounds; three per launch were measured 3 %
        # slower in bench.py's timed region (vr_sor 925 -> 952 ms per step: longer launches interleave worse with the floods' kernels)
        if tiles >= 4 * cus and rounds > 0:
            # ... unless its last round is poorly filled: the smallest group of up to four images that wastes < 3 % of its rounds
            # (3712^2: 1575 tiles = 6.15 rounds, 14 % of 7 rounds idle by itself, 5 % of 13 in pairs, 3 % of 19 in threes)
            def waste(g):
                r = g * tiles / cus
                return -(-g * tiles // cus) / r - 1.0
            group = next((g for g in range(1, min(B, 4) + 1) if waste(g) < 0.03), min(range(1, min(B, 4) + 1), key=waste))
        group = int(max(1, min(group, 6.5e9 // max(1, L.tf_varref_workspace_bytes(H, W)))))
        if os.environ.get("TF_VR_GROUP"):                    # development switch: images per launch
            group = max(1, min(B, int(os.environ["TF_VR_GROUP"])))
        ws = _lib.workspace(L.tf_varref_workspace_bytes_batch(group, H, W), "varref")
        p = self._params()
        flags = (1 if self.fastDivide else 0) | (2 if self.fastSor else 0)
        for b0 in range(0, B, group):
            n = min(group, B - b0)
            _lib.check(L.tf_varref_batch(_lib.ptr(i0[b0]), _lib.ptr(i1[b0]), n, H * W, H, W, ctypes.byref(p), _lib.ptr(flow[b0]),
                                         flow.stride(0) if B > 1 else H * W * 2, flags, _lib.ptr(ws), ws.numel(), _lib.stream_ptr()),
                       "tf_varref_batch")
        return flow

    def calc(self, I0, I1, flow):
        t = _lib.torch()
        i0, i1 = _lib.to_dev(I0), _lib.to_dev(I1)
        if i0.dtype != t.uint8 or i1.dtype != t.uint8:
            raise ValueError("VariationalRefinement input frames must be uint8")
        if i0.dim() != 2 or i0.shape != i1.shape:
            raise ValueError("I0 and I1 must be 2-D arrays of the same shape")
        if tuple(flow.shape) != tuple(i0.shape) + (2,):
            raise ValueError("flow must have shape I0.shape + (2,)")
        if isinstance(flow, t.Tensor):
            f = flow if (flow.is_cuda and flow.dtype == t.float32 and flow.is_contiguous()) else _lib.to_dev(flow, t.float32)
            self.calc_dev(i0, i1, f)
            if f is not flow:
                flow.copy_(f)
            return flow
        f = self.calc_dev(i0, i1, _lib.to_dev(np.asarray(flow), t.float32).clone()).cpu().numpy()
        if isinstance(flow, np.ndarray) and flow.dtype == np.float32 and flow.flags.writeable:
            flow[...] = f
            return flow
        return f


class _LazyVariationalRefinement:
    """`vr_model` (flow.py:359) is created at import time in the reference; here the library is loaded on first use so
    that importing the module does not need the built .so."""
    _obj = None

    def __getattr__(self, name):
        if _LazyVariationalRefinement._obj is None:
            _LazyVariationalRefinement._obj = VariationalRefinement()
        return getattr(_LazyVariationalRefinement._obj, name)

    def __setattr__(self, name, value):                      # vr_model.fastDivide = True, vr_model.alpha = ... reach the object
        if _LazyVariationalRefinement._obj is None:
            _LazyVariationalRefinement._obj = VariationalRefinement()
        setattr(_LazyVariationalRefinement._obj, name, value)


vr_model = _LazyVariationalRefinement()


def _pair_flows_dev(prev8, next8, of_model, vr_steps, smoothing_steps, interp_method, tag="farneback"):
    """calculate_flow_frame on device uint8 tensors; returns device (fwd, bwd)."""
    L = _lib.lib()
    t = _lib.torch()
    interp = select_interp_mode(interp_method) if smoothing_steps > 0 else 1
    fwd, bwd = of_model.calc_pair_dev(prev8, next8, tag=tag)
    if vr_steps > 0:                             # flow.py:513-519: ONE refinement per direction whatever the value
        fwd = vr_model.calc_dev(prev8, next8, fwd)
        bwd = vr_model.calc_dev(next8, prev8, bwd)
    H, W = prev8.shape
    for _ in range(smoothing_steps):
        f2, b2 = t.empty_like(fwd), t.empty_like(bwd)
        _lib.check(L.tf_smooth_flow_step(_lib.ptr(fwd), _lib.ptr(bwd), H, W, interp, _lib.ptr(f2), _lib.ptr(b2),
                                         _lib.stream_ptr()), "tf_smooth_flow_step")
        fwd, bwd = f2, b2
    return fwd, bwd


_SIDE_STREAMS = {}


def _side_stream(main):
    """one extra stream per (device, calling stream), created on first use"""
    t = _lib.torch()
    key = (main.device.index, main.cuda_stream)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = t.cuda.Stream(device=main.device)
    return _SIDE_STREAMS[key]


def _calculate_flow_impl(frame_pairs, T, shape, of_model, vr_steps, smoothing_passes, interp_method,
                         norm_name, norm_method, normalisation_kwargs, on_device, max_value=float("inf"), on_batch=None,
                         workspace_gb=None, split_parts=None, check_out=None):
    t = _lib.torch()
    L = _lib.lib()
    H, W = shape
    # the iteration kernel's chains in two parts when nothing is going to run beside the flow (no windows handed out while
    # the later frames are computed): same bits, 8 % faster alone, but it fills the CUs' LDS (csrc/farneback.hip).  A field of
    # THIS call's parameter block (select_of_model returns a fresh object per call), not a process-wide switch
    if hasattr(of_model, "params") and hasattr(of_model.params, "chain_form"):
        of_model.params.chain_form = _lib.FB_CHAIN_ONE_LANE if on_batch is not None else _lib.FB_CHAIN_TWO_PART
    forward = _lib.empty((T, H, W, 2), t.float32)
    backward = _lib.empty((T, H, W, 2), t.float32)
    # every frame is written below except the two mirrored ends, which tf_flow_finalize derives from their neighbours
    # (NaN there if the stack has a single frame and there is nothing to mirror)
    forward[T - 1].fill_(float("nan"))
    backward[0].fill_(float("nan"))
    # Frame pairs are independent units: they are processed in batches of TF_FLOW_BATCH pairs per set of
    # kernel launches (tf_farneback_batch), which keeps the coarse pyramid levels busy on all CUs.
    linear = norm_name == "linear" and not normalisation_kwargs
    interp = select_interp_mode(interp_method) if smoothing_passes > 0 else 1
    # The fused iteration kernel walks whole columns (OpenCV's running column sums cannot be split over rows,
    # csrc/farneback.hip), so its parallelism is strips x directions x PAIRS and a launch costs a whole number of rounds
    # of resident workgroups -- at every pyramid level.  Within the scratch budget the library picks the batch whose
    # rounds are fullest summed over the levels (tf_farneback_batch_hint: 42 pairs at 5424^2 -- full rounds at levels 0, 1
    # and 2 -- rather than the 54 that fit); what is left at the end goes into one batch, which never costs more rounds
    # than splitting it.  TF_FLOW_BATCH fixes the batch size (development switch).
    n_pairs = T - 1
    sizes = []
    if "TF_FLOW_BATCH" in os.environ or n_pairs <= 0 or not hasattr(of_model, "params"):
        chunk = max(1, int(os.environ.get("TF_FLOW_BATCH", "16")))
        n_b = max(1, -(-n_pairs // chunk))
        edges = [round(k * n_pairs / n_b) for k in range(n_b + 1)]
        sizes = [b - a for a, b in zip(edges[:-1], edges[1:]) if b > a]
    else:
        # budget: TF_FLOW_WORKSPACE_GB (default 115), and never more than 60 % of what the device has free now (the stages
        # after the flow need room too)
        budget = float(os.environ.get("TF_FLOW_WORKSPACE_GB", workspace_gb if workspace_gb is not None else "115")) * 1e9
        held = _lib.workspace_bytes("farneback")           # scratch kept from the previous call: reusable as it is
        free = t.cuda.mem_get_info()[0] + (t.cuda.memory_reserved() - t.cuda.memory_allocated())
        budget = int(max(min(budget, held + 0.6 * free), 1))   # (_lib.workspace empties the allocator's cache if fragments are in the way)
        # scratch of the library + the batch's 8-bit frames and raw flow vectors (this function's own buffers)
        lib_per_pair = max(1, int(L.tf_farneback_workspace_bytes_batch(1, H, W, ctypes.byref(of_model.params))))
        # TF_FLOW_SPLIT=<parts> (default 1): a batch of B pairs runs its pyramid levels >= 2 for all pairs at once and the two
        # finest levels in `parts` parts (tf_farneback_batch_split): full-size library scratch for B / parts pairs only, so a
        # budget that holds the full-size scratch of 21 pairs still fills the coarse levels' launches with 42
        split_parts = max(1, int(os.environ.get("TF_FLOW_SPLIT", split_parts if split_parts is not None else "1")))
        per_pair = lib_per_pair // split_parts + H * W * (2 + 16)
        cap = max(1, budget // per_pair)
        # the library's hint counts ITS scratch only: hand it the share of the budget that is the library's, so that both
        # sides price a pair the same way (the hint could return a batch ~25 % over the budget, which then went
        # through the halving path -- release, empty_cache, half batches for the rest of the call)
        lib_budget = max(1, int(budget * ((lib_per_pair // split_parts) / per_pair)))
        left = n_pairs
        while left > 0:
            if left <= cap:
                B = left
            else:       # the hint sizes the FULL-resolution launches: B / parts pairs of them
                b_fine = max(1, int(L.tf_farneback_batch_hint(H, W, ctypes.byref(of_model.params), -(-left // split_parts), lib_budget)))
                B = min(cap, left, b_fine * split_parts)
            sizes.append(B)
            left -= B
    if os.environ.get("TF_FLOW_DEBUG"):
        print("flow: %d pairs in batches %s" % (n_pairs, sizes), flush=True)
    n_batches = len(sizes)
    starts = [0]
    for B in sizes:
        starts.append(starts[-1] + B)
    # TF_FLOW_OVERLAP=1: a second stream for the per-pair stages after the Farneback batch.  Off by default: measured on
    # config F (144 x 5424^2) it gains 1.5 % (5033 vs 5110 ms per step) -- every kernel involved fills the GPU by itself
    # -- and the HIP-event kernel timing of bench.py cannot attribute time to kernels that run side by side.
    main = t.cuda.current_stream()
    side = None
    if n_batches > 1 and (vr_steps > 0 or smoothing_passes > 0) and os.environ.get("TF_FLOW_OVERLAP", "0") == "1":
        side = _side_stream(main)
        forward.record_stream(side)
        backward.record_stream(side)

    # The 8-bit frames and the raw (unsmoothed) flow vectors of a batch -- 2 x 1.6 GB and 2 x 9.9 GB for 42 pairs at 5424^2
    # -- live in buffers that stay allocated for the next batch and the next call, like the library's scratch: as fresh
    # tensors per batch they were carved out of the caching allocator's blocks in a different way in every call, and the
    # second call of a process paid two more device allocations (25 GB, 0.65 s: bench.py's first timed step).  With the
    # second stream (TF_FLOW_OVERLAP) two batches are in flight and each needs its own: fresh tensors then.
    pool = {"B_max": max(sizes) if sizes else 0}             # (shrinks when a batch has to be halved)
    pooled = side is None and pool["B_max"] > 0

    def batch_buffers(B):
        n8, nraw, B_max = H * W, H * W * 2 * 4, max(pool["B_max"], B)
        if pooled:
            u8 = _lib.workspace(2 * B_max * n8, "farneback_u8")
            prev8, next8 = u8[:B * n8].view(B, H, W), u8[B_max * n8:(B_max + B) * n8].view(B, H, W)
        else:
            prev8, next8 = _lib.empty((B, H, W), t.uint8), _lib.empty((B, H, W), t.uint8)
        if smoothing_passes == 0:
            return prev8, next8, None, None
        if pooled:
            raw = _lib.workspace(2 * B_max * nraw, "farneback_raw")
            f = raw[:B * nraw].view(t.float32).view(B, H, W, 2)
            bk = raw[B_max * nraw:(B_max + B) * nraw].view(t.float32).view(B, H, W, 2)
        else:
            f, bk = _lib.empty((B, H, W, 2), t.float32), _lib.empty((B, H, W, 2), t.float32)
        return prev8, next8, f, bk

    def split_parts_used(B):
        parts = max(1, int(os.environ.get("TF_FLOW_SPLIT", split_parts if split_parts is not None else "1")))
        if parts == 1 or not hasattr(of_model, "params") or "TF_FLOW_BATCH" in os.environ or B < 2 * parts:
            return 1
        # a launch of the iteration kernel costs whole rounds of resident workgroups: parts that no longer fill a round at
        # the full resolution each cost what the whole batch costs (config C, 23 pairs of 1500 x 2500 = one round:
        # two parts took 2 x 2.3 ms per iteration instead of 2.6)
        if os.environ.get("TF_FLOW_SPLIT_FORCE"):            # (tests: split whatever the size)
            return parts
        resident = ctypes.c_int64(0)
        wgs = int(L.tf_farneback_iteration_workgroups(H, W, ctypes.byref(of_model.params), -(-B // parts), ctypes.byref(resident)))
        return parts if wgs >= 0.9 * max(1, resident.value) else 1

    def run_batch(i0, B):
        """GENERATOR over the library work of one batch: yields the number of leading frame pairs that are final after each
        part of a split batch / after the batch.  Everything that allocates or launches for the batch happens inside
        `next()`; what the caller does with the frames (on_frames_ready) happens between two `next()` calls, OUTSIDE the
        out-of-memory handler of the loop below."""
        if B <= 0:
            return
        prev8, next8, f, bk = batch_buffers(B)
        for b in range(B):
            fa, fb = frame_pairs(i0 + b)
            if linear:
                to_8bit_pair_dev(fa, fb, out=(prev8[b], next8[b]))
            else:   # other normalisations are host glue (not on the production path)
                pair = np.stack([fa.cpu().numpy(), fb.cpu().numpy()], 0)
                p8 = to_8bit(norm_method(pair, **normalisation_kwargs), 0, 1)
                prev8[b].copy_(_lib.to_dev(p8[0]))
                next8[b].copy_(_lib.to_dev(p8[1]))
        if smoothing_passes == 0:
            f, bk = forward[i0:i0 + B], backward[i0 + 1:i0 + 1 + B]
        parts = split_parts_used(B)
        by_part = parts > 1 and side is None and of_model.can_split(H, W)
        if by_part:
            # SPLIT BATCH, part by part: the coarse pyramid levels for all B pairs at once (their launches are half empty with
            # B / parts pairs), then per part the two finest levels, the refinement, the smoothing -- and the caller's callback:
            # the frames are handed out with the granularity of a part
            per = -(-B // parts)
            of_model.calc_phase_dev(prev8, next8, f, bk, 1, (B, per))
        else:
            of_model.calc_batch_dev(prev8, next8, f, bk, parts=parts)
        if vr_steps == 0 and smoothing_passes == 0 and not by_part:
            yield i0 + B
            return

        def refine_and_smooth(i0=i0, B=B, prev8=prev8, next8=next8, f=f, bk=bk):
            if vr_steps > 0:                     # flow.py:513-519, before the smoothing (flow.py:521-525)
                # (round 6: all pairs of the batch per set of launches, one direction after the other -- tf_varref_batch)
                vr_model.calc_batch_dev(prev8, next8, f)
                vr_model.calc_batch_dev(next8, prev8, bk)
            for b in range(B if smoothing_passes > 0 else 0):
                fi, bi = f[b], bk[b]
                for k in range(smoothing_passes):
                    last = k == smoothing_passes - 1
                    fo = forward[i0 + b] if last else t.empty_like(fi)
                    bo = backward[i0 + 1 + b] if last else t.empty_like(bi)
                    # the clip of create_flow rides on the store of the LAST pass (elementwise: same values as a pass of its
                    # own over both arrays afterwards, which cost 23 ms per 144 x 5424^2 stack)
                    _lib.check(L.tf_smooth_flow_step_clip(_lib.ptr(fi), _lib.ptr(bi), H, W, interp, _lib.ptr(fo), _lib.ptr(bo),
                                                          max_value if last else float("inf"), _lib.stream_ptr()), "tf_smooth_flow_step")
                    fi, bi = fo, bo

        if by_part:
            for b0 in range(0, B, per):
                b1 = min(B, b0 + per)
                of_model.calc_phase_dev(prev8[b0:b1], next8[b0:b1], f[b0:b1], bk[b0:b1], 2, (B, per))
                if vr_steps > 0 or smoothing_passes > 0:
                    refine_and_smooth(i0 + b0, b1 - b0, prev8[b0:b1], next8[b0:b1], f[b0:b1], bk[b0:b1])
                yield i0 + b1
            return
        elif side is None:
            refine_and_smooth()
        else:
            # The refinement / smoothing of this batch runs on a second stream while the main stream goes on with the
            # NEXT batch's pyramids and iterations: the iteration kernel leaves CUs idle (coarse levels, the last round of
            # a launch), the refinement kernels fill them.  Same kernels on the same data -- the results do not change.
            ready = main.record_event()
            for x in (prev8, next8, f, bk):
                x.record_stream(side)            # allocated on `main`, last used on `side`
            with t.cuda.stream(side):
                side.wait_event(ready)
                refine_and_smooth()
        yield i0 + B
    # a batch that does not fit (the budget above is an estimate; a caching allocator's free memory can be fragmented) is
    # halved and tried again, and so are the batches after it
    def finalize_ends():
        # flow.py:425-426 (mirror the end frames); max_value = inf -> no clipping.  With smoothing the interior is already
        # clipped (run_batch): only the two end frames are left to write
        if smoothing_passes > 0 or max_value == float("inf"):
            _lib.check(L.tf_flow_finalize_ends(_lib.ptr(forward), _lib.ptr(backward), T, H, W, max_value, _lib.stream_ptr()),
                       "tf_flow_finalize_ends")
        else:
            _lib.check(L.tf_flow_finalize(_lib.ptr(forward), _lib.ptr(backward), T, H, W, max_value, _lib.stream_ptr()),
                       "tf_flow_finalize")

    if on_batch is not None and (side is not None or not on_device or not (smoothing_passes > 0 or max_value == float("inf"))):
        raise ValueError("on_frames_ready needs device-resident input, the default single-stream schedule and either smoothing or no clipping")
    pending = [(a_, b_ - a_) for a_, b_ in zip(starts[:-1], starts[1:])]
    while pending:
        i0, B = pending.pop(0)
        work = run_batch(i0, B)
        done = i0                                             # leading pairs that are final (and, with on_batch, handed out)
        while True:
            # ONLY the batch's own allocations and launches sit inside the handler: an out-of-memory error raised by the
            # caller's callback (bench.py's allocates edge fields and flood scratch on a nearly full device) is the caller's,
            # it must not be taken for "this batch's scratch did not fit" -- which used to release the workspace, recompute
            # the batch at half size over frames already handed out and re-enter the callback with a smaller n
            try:
                done_now = next(work, None)
            except t.OutOfMemoryError:
                if os.environ.get("TF_FLOW_DEBUG"):
                    print("flow: batch of %d pairs does not fit (free %.1f GB, cached %.1f GB): halving" % (
                        B, t.cuda.mem_get_info()[0] / 1e9, (t.cuda.memory_reserved() - t.cuda.memory_allocated()) / 1e9), flush=True)
                if B <= 1:
                    raise
                work.close()
                _lib.release_workspaces("farneback")
                t.cuda.empty_cache()
                half = B // 2
                pool["B_max"] = half
                # never re-run pairs whose frames have been handed out: the parts of this batch that are done stay done, the
                # rest of it (from `done` on) and the batches after it are cut to the new size
                todo, pending = [(done, i0 + B - done)] + pending, []
                for j0, Bj in todo:
                    while Bj > half:
                        pending.append((j0, half))
                        j0, Bj = j0 + half, Bj - half
                    if Bj > 0:
                        pending.append((j0, Bj))
                break
            if done_now is None:
                break
            done = done_now
            if on_batch is not None and done < n_pairs:
                # the first done + 1 frames are final but for the stack's own first frame, whose backward vectors are the
                # mirror of its forward ones: written now (and, with the still unknown other end, once more at the end)
                finalize_ends()
                on_batch(forward, backward, done, n_pairs)
    if side is not None:
        main.wait_stream(side)
    if n_batches > 0 and sizes and max(sizes) > 16:
        # A large batch's scratch (tens of GB) stays allocated for the next call (carving one block of that size out of a
        # caching allocator's free fragments fails more often than not: measured, bench.py fell back to half batches from
        # its second step on) -- unless the device is nearly full: then the stages after the flow (Sobel, watershed, the
        # labels of a whole stack) need the memory more
        free_now, total = t.cuda.mem_get_info()
        if free_now + (t.cuda.memory_reserved() - t.cuda.memory_allocated()) < 0.2 * total:
            _lib.release_workspaces("farneback")
    finalize_ends()
    # A stack whose flow holds NaN rows (a starved chain of the iteration kernel) must not be returned as if nothing had
    # happened.  Host output: the download below synchronises anyway -- check behind it.  Device output: the caller's pipeline is
    # asynchronous; the check is DEFERRED to an event recorded here (`deferred`): the Flow object polls it at every use and
    # create_flow's callers with a synchronisation of their own (detect_stack_windows) wait for it there
    deferred = of_model.deferred_check("create_flow / calculate_flow") if hasattr(of_model, "deferred_check") else None
    if on_batch is not None:
        on_batch(forward, backward, n_pairs, n_pairs)
    if on_device:
        if check_out is not None and deferred is not None:
            check_out.append(deferred)
        elif deferred is not None:
            deferred(block=True)                  # (a caller that cannot take the deferred check gets the synchronous one)
        return forward, backward
    out = _lib.to_host(forward), _lib.to_host(backward)
    if deferred is not None:
        deferred(block=True)
    return out


def calculate_flow(data, model: str = "Farneback", vr_steps: int = 0, smoothing_passes: int = 0,
                   interp_method: str = "linear", normalisation_method: str = "linear", **normalisation_kwargs):
    """Forward / backward flow for every consecutive frame pair of `data` (reference: flow.py:362-428).
    forward[i] = flow i -> i+1, backward[i+1] = flow i+1 -> i; the end frames are mirrored."""
    max_value = normalisation_kwargs.pop("_max_value", float("inf"))
    on_batch = normalisation_kwargs.pop("_on_batch", None)
    device_out = normalisation_kwargs.pop("_device_out", False)
    workspace_gb = normalisation_kwargs.pop("_workspace_gb", None)
    split_parts = normalisation_kwargs.pop("_split_parts", None)
    check_out = normalisation_kwargs.pop("_check_out", None)
    of_model = select_of_model(model)
    norm_method = select_normalisation_method(normalisation_method)
    t = _lib.torch()
    data = _unwrap_device_field(data)
    if _is_dataarray(data):
        data = data.compute().data if hasattr(data, "compute") else data.to_numpy()
    on_device = isinstance(data, t.Tensor) or device_out
    d = _lib.to_dev(data, t.float32, share=True)          # (only read: the same array usually goes to detect_cores next)
    if d.dim() != 3:
        raise ValueError("data must have three dimensions (t, y, x)")
    T = d.shape[0]
    return _calculate_flow_impl(lambda i: (d[i], d[i + 1]), T, tuple(d.shape[1:]), of_model, vr_steps,
                                smoothing_passes, interp_method, normalisation_method, norm_method,
                                normalisation_kwargs, on_device, max_value, on_batch, workspace_gb, split_parts, check_out)


def calculate_flow_2(a, b, model: str = "Farneback", vr_steps: int = 0, smoothing_passes: int = 0,
                     normalisation_method: str = "linear", **normalisation_kwargs):
    """Flow between the frames of two stacks, a[i] -> b[i] (reference: flow.py:431-496)."""
    of_model = select_of_model(model)
    norm_method = select_normalisation_method(normalisation_method)
    t = _lib.torch()
    a, b = _unwrap_device_field(a), _unwrap_device_field(b)
    if _is_dataarray(a):
        a = a.compute().data if hasattr(a, "compute") else a.to_numpy()
    if _is_dataarray(b):
        b = b.compute().data if hasattr(b, "compute") else b.to_numpy()
    on_device = isinstance(a, t.Tensor)
    da, db = _lib.to_dev(a, t.float32), _lib.to_dev(b, t.float32)
    T = da.shape[0]
    return _calculate_flow_impl(lambda i: (da[i], db[i]), T, tuple(da.shape[1:]), of_model, vr_steps,
                                smoothing_passes, "linear", normalisation_method, norm_method,
                                normalisation_kwargs, on_device)


def calculate_flow_frame(prev_frame, next_frame, of_model, vr_steps: int = 0, smoothing_steps: int = 0,
                         interp_method: str = "linear"):
    """Forward and backward flow between two uint8 images (reference: flow.py:499-527)."""
    t = _lib.torch()
    on_device = isinstance(prev_frame, t.Tensor)
    p, n = _lib.to_dev(prev_frame), _lib.to_dev(next_frame)
    if p.dtype != t.uint8 or n.dtype != t.uint8:
        raise ValueError("frames must be uint8 (see to_8bit)")
    f, b = _pair_flows_dev(p, n, of_model, vr_steps, smoothing_steps, interp_method)
    if hasattr(of_model, "check_launches"):
        of_model.check_launches("calculate_flow_frame")
    return (f, b) if on_device else (_lib.to_host(f), _lib.to_host(b))


def smooth_flow_step(forward_flow, backward_flow, method: str = "linear"):
    """f' = nanmean(f, -warp(b by f)), b' = nanmean(b, -warp(f by b)) (reference: flow.py:530-568)."""
    t = _lib.torch()
    L = _lib.lib()
    interp = select_interp_mode(method)
    on_device = isinstance(forward_flow, t.Tensor)
    f, b = _lib.to_dev(forward_flow, t.float32), _lib.to_dev(backward_flow, t.float32)
    H, W = f.shape[:2]
    f2, b2 = t.empty_like(f), t.empty_like(b)
    _lib.check(L.tf_smooth_flow_step(_lib.ptr(f), _lib.ptr(b), H, W, interp, _lib.ptr(f2), _lib.ptr(b2),
                                     _lib.stream_ptr()), "tf_smooth_flow_step")
    return (f2, b2) if on_device else (_lib.to_host(f2), _lib.to_host(b2))


# ---- diagnostics (host numpy; reference: flow.py:571-666) -------------------------------------------
def combine_flow(*args):
    def blend(pick):
        mags = [((pick(f)[..., 0] ** 2 + pick(f)[..., 1] ** 2) ** 0.5)[..., np.newaxis] for f in args]
        return sum(pick(f) * m for f, m in zip(args, mags)) / sum(mags)
    return Flow(blend(lambda f: f.forward_flow), blend(lambda f: f.backward_flow))


def get_forward_warp(da, flow):
    forward_struct = np.zeros([3, 3, 3], dtype=bool)
    forward_struct[2, 1, 1] = True
    return flow.convolve(da.data, forward_struct)[0]


def flow_diff_mse_estimate(da, flow):
    fw = get_forward_warp(da, flow)
    cold = da.data < 273
    return mse(fw, da.data), mse(fw[cold], da.data[cold])


def get_flow_residual(da, flow, model="Farneback", vr_steps=1, smoothing_passes=1):
    new_flow, _ = calculate_flow_2(da.data, get_forward_warp(da, flow), model=model, vr_steps=vr_steps,
                                   smoothing_passes=smoothing_passes)
    return new_flow


def flow_magnitude(flow, direction="forward"):
    if direction == "forward":
        v = flow.forward_flow
    elif direction == "backward":
        v = flow.backward_flow
    else:
        raise ValueError("Direction must be one of 'forward', 'backward'")
    return (v[..., 0] ** 2 + v[..., 1] ** 2) ** 0.5


def flow_residual_mse_estimate(da, flow, model="Farneback", vr_steps=1, smoothing_passes=1):
    nf = get_flow_residual(da, flow, model=model, vr_steps=vr_steps, smoothing_passes=smoothing_passes)
    mag = ((nf[..., 0] ** 2 + nf[..., 1] ** 2) ** 0.5)[:, 20:-20, 20:-20]
    cold = da.data[:, 20:-20, 20:-20] < 273
    return mse(mag, np.zeros_like(mag)), mse(mag[cold], np.zeros_like(mag[cold]))


def time_flow(da, model="Farneback", vr_steps=1, smoothing_passes=1):
    start = datetime.now()
    create_flow(da, model=model, vr_steps=vr_steps, smoothing_passes=smoothing_passes)
    return (datetime.now() - start).total_seconds()
