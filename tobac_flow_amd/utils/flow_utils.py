"""Optical-flow model factory and single-image warp (mirrors
/root/reference/tobac_flow/utils/flow_utils.py with cv2 replaced by the HIP library)."""
import numpy as np

from tobac_flow_amd import _lib

border_modes = ("constant", "nearest", "reflect", "mirror", "wrap", "isolated", "transparent")
interp_modes = ("nearest", "linear", "cubic", "lanczos")
_MODELS = ("Farneback", "DeepFlow", "PCA", "SimpleFlow", "SparseToDense", "DIS", "DenseRLOF", "DualTVL1")


def select_border_mode(mode: str):
    if mode not in border_modes:
        raise ValueError("Invalid border mode")
    if mode != "constant":
        raise NotImplementedError(f"border mode '{mode}' has no HIP kernel (only 'constant')")
    return mode


def select_interp_mode(mode: str):
    if mode not in interp_modes:
        raise ValueError("Invalid border mode")
    return _lib.INTERP[mode]


class FarnebackFlow:
    """Stand-in for the object returned by cv2.optflow.createOptFlow_Farneback()
    (flow_utils.py:52-53): `.calc(prev, next, None)` -> (H, W, 2) float32 flow, OpenCV defaults."""

    def __init__(self, num_levels=5, pyr_scale=0.5, win_size=13, num_iters=10, poly_n=5, poly_sigma=1.1):
        self.params = _lib.FarnebackParams(num_levels, pyr_scale, win_size, num_iters, poly_n, poly_sigma, _lib.FB_CHAIN_DEFAULT, 0)
        self._slot_device = None

    def _own_status_word(self):
        """A status word of this object's own for the starved-chain report (tf_farneback_status_acquire; select_of_model returns
        a fresh object per create_flow / calculate_flow call, so: one word per flow).  Two flows in flight on one device -- the
        flood thread finishing stack k beside stack k + 1's flow, two host threads -- then neither consume nor get blamed for
        each other's report.  Taken on first use, on the device that is current then; 0 (the device's shared word)
        if the library has none left."""
        if self._slot_device is None:
            t = _lib.torch()
            self._slot_device = t.cuda.current_device()
            self.params.status_slot = int(_lib.lib().tf_farneback_status_acquire())
        return self.params.status_slot

    def _status(self):
        L = _lib.lib()
        return L.tf_farneback_status_check(self.params.status_slot) if self.params.status_slot else L.tf_farneback_check()

    def __del__(self):
        try:
            if self._slot_device is not None and self.params.status_slot:
                t = _lib.torch()
                with t.cuda.device(self._slot_device):
                    _lib.lib().tf_farneback_status_release(self.params.status_slot)
        except Exception:                                   # interpreter shutdown
            pass

    def check_launches(self, what="Farneback flow"):
        """The launches are asynchronous; what a launch found out arrives later.  Waits for the current stream and raises
        (TobacFlowHipError, a RuntimeError) if a row-sum chain of any iteration launch gave up waiting for its neighbour --
        its rows are NaN (tf_farneback_check; csrc/farneback.hip fb_chain_enter).  cv2's calc is synchronous and has no such
        state; this is where the asynchronous library reports like a synchronous one."""
        _lib.torch().cuda.current_stream().synchronize()
        _lib.check(self._status(), what)

    def deferred_check(self, what="Farneback flow"):
        """check_launches without stalling a device-resident pipeline: returns a callable `poll(block=False)` bound to an event
        recorded NOW on the current stream.  poll() reports (raises like check_launches) once the event has passed and is a
        no-op afterwards; poll(block=True) waits for it first.  A host that runs seconds ahead of the device -- enqueuing the
        next stages, allocating their scratch -- keeps doing so (a full synchronisation at the end of create_flow cost config
        F3 0.9 s per step of exposed allocator work), and the report still arrives: at the next use of the Flow, at the
        next tf_farneback_batch* call (which reports on entry), or at the caller's own synchronisation."""
        t = _lib.torch()
        ev = t.cuda.Event()
        ev.record()
        state = {"done": False}

        def poll(block=False):
            if state["done"]:
                return
            if block:
                ev.synchronize()
            elif not ev.query():
                return
            state["done"] = True
            _lib.check(self._status(), what)
        poll.status_slot = self.params.status_slot          # (tests: which word this flow's launches report to)
        return poll

    def calc_pair_dev(self, prev, nxt, want_fwd=True, want_bwd=True, tag="farneback"):
        """Both directions at once on device uint8 tensors (they share pyramid + expansion)."""
        import ctypes
        t = _lib.torch()
        L = _lib.lib()
        H, W = prev.shape
        self._own_status_word()
        fwd = _lib.empty((H, W, 2), t.float32) if want_fwd else None
        bwd = _lib.empty((H, W, 2), t.float32) if want_bwd else None
        nbytes = L.tf_farneback_workspace_bytes(H, W, ctypes.byref(self.params))
        ws = _lib.workspace(nbytes, tag)
        rc = L.tf_farneback_pair(_lib.ptr(prev), _lib.ptr(nxt), H, W, ctypes.byref(self.params),
                                 _lib.ptr(fwd), _lib.ptr(bwd), _lib.ptr(ws), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "tf_farneback_pair")
        return fwd, bwd

    def calc_batch_dev(self, prev, nxt, fwd_out, bwd_out, tag="farneback", parts=1):
        """B independent pairs at once: prev / nxt (B, H, W) uint8 device tensors; fwd_out / bwd_out are
        (B, H, W, 2) float32 device tensors (views into the big flow arrays are fine: only the batch stride
        may differ from H*W*2).  parts > 1: the pyramid levels >= 2 run for all B pairs at once, the two finest ones in
        `parts` parts (tf_farneback_batch_split: full-size scratch for B / parts pairs only; same results)."""
        import ctypes
        L = _lib.lib()
        B, H, W = prev.shape
        self._own_status_word()
        assert prev.is_contiguous() and nxt.is_contiguous()
        for o in (fwd_out, bwd_out):
            assert o.shape == (B, H, W, 2) and o[0].is_contiguous() and (B == 1 or o.stride(0) >= H * W * 2)
        parts = max(1, int(parts))
        nbytes = L.tf_farneback_workspace_bytes_split(B, parts, H, W, ctypes.byref(self.params))
        ws = _lib.workspace(nbytes, tag)
        stride = fwd_out.stride(0) if B > 1 else H * W * 2
        assert B == 1 or bwd_out.stride(0) == stride
        rc = L.tf_farneback_batch_split(_lib.ptr(prev), _lib.ptr(nxt), B, parts, H * W, H, W, ctypes.byref(self.params),
                                        _lib.ptr(fwd_out), _lib.ptr(bwd_out), stride, _lib.ptr(ws), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "tf_farneback_batch")

    def can_split(self, H, W):
        import ctypes
        return bool(_lib.lib().tf_farneback_can_split(H, W, ctypes.byref(self.params)))

    def calc_phase_dev(self, prev, nxt, fwd_out, bwd_out, phase, ws_pairs, tag="farneback"):
        """phase 1 / 2 of a split batch (tf_farneback_batch_phase) on the pairs handed in; the scratch is sized for the larger
        of phase 1 with `ws_pairs[0]` pairs and phase 2 with `ws_pairs[1]` pairs, so that both phases of a batch share it."""
        import ctypes
        L = _lib.lib()
        B, H, W = prev.shape
        self._own_status_word()
        nbytes = max(L.tf_farneback_workspace_bytes_phase(ws_pairs[0], H, W, ctypes.byref(self.params), 1),
                     L.tf_farneback_workspace_bytes_phase(ws_pairs[1], H, W, ctypes.byref(self.params), 2))
        ws = _lib.workspace(nbytes, tag)
        stride = fwd_out.stride(0) if B > 1 else H * W * 2
        assert B == 1 or bwd_out.stride(0) == stride
        rc = L.tf_farneback_batch_phase(_lib.ptr(prev), _lib.ptr(nxt), B, H * W, H, W, ctypes.byref(self.params), _lib.ptr(fwd_out),
                                        _lib.ptr(bwd_out), stride, _lib.ptr(ws), ws.numel(), _lib.stream_ptr(), phase)
        _lib.check(rc, "tf_farneback_batch_phase")

    def calc(self, prev, nxt, flow=None):
        t = _lib.torch()
        on_device = isinstance(prev, t.Tensor)
        p, n = _lib.to_dev(prev), _lib.to_dev(nxt)
        if p.dtype != t.uint8 or n.dtype != t.uint8:
            raise ValueError("Farneback input frames must be uint8")
        if p.shape != n.shape or p.dim() != 2:
            raise ValueError("prev and next must be 2-D arrays of the same shape")
        fwd, _ = self.calc_pair_dev(p, n, True, False)
        self.check_launches("FarnebackFlow.calc")
        return fwd if on_device else fwd.cpu().numpy()


def select_of_model(model: str):
    """Optical-flow model by name (reference: flow_utils.py:37-77).  Only 'Farneback' -- the model
    the production scripts use -- exists on the MI355X; the other OpenCV model names are recognised
    and raise NotImplementedError, unknown names raise ValueError like the reference."""
    if model == "Farneback":
        return FarnebackFlow()
    if model == "DenseRLOF":
        raise NotImplementedError("DenseRLOF requires multi-channel input which is currently not implemented")
    if model in _MODELS:
        raise NotImplementedError(f"optical-flow model '{model}' has no HIP implementation (only 'Farneback')")
    raise ValueError("'model' parameter must be one of: 'Farneback', 'DeepFlow', 'PCA', 'SimpleFlow', "
                     "'SparseToDense', 'DIS', 'DenseRLOF', 'DualTVL1'")


def warp_flow(img, flow, method: str = "linear", border: str = "constant"):
    """Warp an image by a set of flow vectors (reference: flow_utils.py:80-99; border value NaN)."""
    t = _lib.torch()
    L = _lib.lib()
    interp = select_interp_mode(method)
    select_border_mode(border)
    on_device = isinstance(img, t.Tensor)
    i, f = _lib.to_dev(img, t.float32), _lib.to_dev(flow, t.float32)
    H, W = f.shape[:2]
    if tuple(i.shape) != (H, W) or f.shape[-1] != 2:
        raise ValueError("img must be (H, W) and flow (H, W, 2)")
    out = _lib.empty((H, W), t.float32)
    _lib.check(L.tf_warp_flow(_lib.ptr(i), _lib.ptr(f), H, W, interp, _lib.ptr(out), _lib.stream_ptr()), "tf_warp_flow")
    return out if on_device else out.cpu().numpy()


__all__ = ("select_border_mode", "select_interp_mode", "select_of_model", "warp_flow")
