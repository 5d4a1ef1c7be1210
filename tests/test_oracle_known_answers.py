"""The reference's own known-answer tests, run against the oracle's restatement of the cv2-backed
operators (ports of /root/reference/tests/test_flow.py:53-194 and tests/test_detection.py:36-60).
These are the only golden values the reference holds at the cv2 boundary."""
import numpy as np

from oracle import np_ops


def test_to_8bit_known_answers():
    assert np.all(np_ops.to_8bit(np.zeros(5)) == 0)
    assert np.all(np_ops.to_8bit(np.ones(5)) == 0)
    assert np.all(np_ops.to_8bit(np.ones(5), vmin=0, vmax=1) == 255)
    arr = np.arange(256)
    assert np.all(np_ops.to_8bit(arr) == arr)
    assert np.all(np_ops.to_8bit(arr + 10, vmin=10, vmax=10 + 255) == arr)


def _valid(w, ref):
    ok = ~np.isnan(w)
    return w[ok], ref[ok]


def test_warp_flow_known_answers():
    arr = np.arange(15, dtype=np.float32).reshape(3, 5)
    fl = np.zeros(arr.shape + (2,), np.float32)
    a, b = _valid(np_ops.warp_flow_single(arr, fl), arr)
    assert np.all(a == b)
    fl[..., 0] = 1
    a, b = _valid(np_ops.warp_flow_single(arr, fl)[:, :-1], arr[:, 1:])
    assert np.all(a == b)
    fl[:] = 0
    fl[..., 1] = 1
    a, b = _valid(np_ops.warp_flow_single(arr, fl)[:-1], arr[1:])
    assert np.all(a == b)
    fl[:] = 1
    a, b = _valid(np_ops.warp_flow_single(arr, fl)[:-1, :-1], arr[1:, 1:])
    assert np.all(a == b)
    fl[:] = 0
    fl[..., 0] = 0.5
    a, b = _valid(np_ops.warp_flow_single(arr, fl)[:, :-1], (arr[:, 1:] + arr[:, :-1]) * 0.5)
    assert np.all(a == b)


def test_smooth_flow_step_known_answers():
    z, one = np.zeros([3, 5, 2], np.float32), np.ones([3, 5, 2], np.float32)
    assert np.all(np.stack(list(np_ops.smooth_flow_step(z, z))) == 0)
    f, b = np_ops.smooth_flow_step(one, -one)
    assert np.all(f == 1) and np.all(b == -1)
    f, b = np_ops.smooth_flow_step(one, z)
    assert np.all(f[:1, :3] == 0.5) and np.all(b[:2, :4] == -0.5)


def test_combined_edge_field_known_answer():
    field = np.zeros([1, 5, 5], np.float32)
    field[:, 3:] = 1
    z = np.zeros([1, 5, 5, 2], np.float32)

    def edge_field(f):
        e = np_ops.sobel(f, z, z, method="cubic", dtype=None, direction="uphill")
        e[e > 0] += 1
        e = e - f
        e[np.isnan(f)] = np.inf
        return e
    r = edge_field(field)
    assert r.dtype == np.float64
    assert np.all(r[:, 2] > 0) and np.all(r[:, :2] == 0) and np.all(r[:, 3:] == -1)
    field[:, :, 0] = np.nan
    assert np.all(np.isnan(field) == np.isinf(edge_field(field)))


def test_farneback_oracle_recovers_translation():
    import ctypes
    import scipy.ndimage as ndi
    from oracle import _lib as ol
    rng = np.random.default_rng(0)
    img = ndi.gaussian_filter(rng.normal(size=(160, 200)), 4)
    img = ((img - img.min()) / (img.max() - img.min()) * 255).astype(np.uint8)
    nxt = np.roll(img, (1, -2), (0, 1))
    out = np.zeros(img.shape + (2,), np.float32)
    L = ol.lib()
    L.oracle_farneback.restype = ctypes.c_int
    n = L.oracle_farneback(ol.ptr(img, ctypes.c_uint8), ol.ptr(np.ascontiguousarray(nxt), ctypes.c_uint8), 160, 200,
                           ol.ptr(out, ctypes.c_float), 5, ctypes.c_double(0.5), 13, 10, 5, ctypes.c_double(1.1))
    assert n == 3          # 160 x 200 -> 40 x 50, 80 x 100, full
    c = out[40:120, 50:150].mean((0, 1))
    assert abs(c[0] + 2) < 0.05 and abs(c[1] - 1) < 0.05
