"""Host staging (include/tobac_flow_hip.h "host staging", csrc/staging.hip, tobac_flow_amd/_staging.py): the containers of
the reference's interface (numpy / DataArray in and out: /root/reference/tobac_flow/decorators.py:21-61,
/root/reference/scripts/dcc_detect_goes.py:164-303) moved through pinned memory, and an array presented again recognised by
its CONTENT.  Bit-exact work: every transfer is compared byte for byte, the device checksum with the host checksum word for
word."""
import ctypes
import gc

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _hash_host(L, a):
    from tobac_flow_amd import _lib
    h = np.zeros(2, np.uint64)
    _lib.check(L.tf_hash_host(a.ctypes.data_as(_lib._P), a.nbytes, h.ctypes.data_as(_lib._P)), "tf_hash_host")
    return int(h[0]), int(h[1])


def _hash_dev(L, d, nbytes=None, offset=0):
    from tobac_flow_amd import _lib
    h = np.zeros(2, np.uint64)
    nbytes = d.numel() * d.element_size() - offset if nbytes is None else nbytes
    _lib.check(L.tf_hash_dev(ctypes.c_void_p(d.data_ptr() + offset), nbytes, h.ctypes.data_as(_lib._P), _lib.stream_ptr()), "tf_hash_dev")
    return int(h[0]), int(h[1])


def test_the_checksum_is_the_same_function_on_the_host_and_on_the_device():
    import torch
    from tobac_flow_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(3)
    seen = set()
    for n in (1, 7, 15, 16, 17, 31, 32, 4097, (1 << 20) + 3, (5 << 20) + 13, (40 << 20) + 5):      # below / above the threaded path (4 MiB)
        a = rng.integers(0, 256, size=n, dtype=np.uint8)
        hh = _hash_host(L, a)
        d = torch.from_numpy(a).cuda()
        assert _hash_dev(L, d) == hh, n
        seen.add(hh)
        # one flipped bit anywhere, one byte more or less: another checksum
        b = a.copy()
        b[n // 2] ^= 0x10
        assert _hash_host(L, b) != hh
        if n > 1:
            assert _hash_host(L, a[:-1].copy()) != hh
        # a misaligned device view (scalar loads) gives the host's value of the same bytes
        if n > 40:
            assert _hash_dev(L, d, offset=3) == _hash_host(L, a[3:].copy()), n
    assert len(seen) == 11
    z = np.zeros(1 << 16, np.uint8)                                      # zeros of different lengths differ (length and position are keyed)
    assert _hash_host(L, z) != _hash_host(L, z[:-16].copy())
    assert L.tf_hash_host(None, 16, None) == -1 and L.tf_hash_dev(None, 16, None, None) == -1


@pytest.mark.parametrize("n", [5, (1 << 20) - 1, (1 << 20) + 1, (8 << 20), (77 << 20) + 12345, (300 << 20) + 1])
def test_upload_and_download_move_every_byte(n):
    """pageable source through the ring (more chunks than the ring has slots at 300 MiB: 32 x 8 MiB), pinned source, the
    checksum computed by the copying threads; download into a pinned block"""
    import torch
    from tobac_flow_amd import _lib, _staging
    L = _lib.lib()
    rng = np.random.default_rng(n % 1000)
    a = rng.integers(0, 256, size=n, dtype=np.uint8)
    d = torch.zeros(n, dtype=torch.uint8, device="cuda")
    h = np.zeros(2, np.uint64)
    _lib.check(L.tf_upload(_lib.ptr(d), a.ctypes.data_as(_lib._P), n, h.ctypes.data_as(_lib._P), _lib.stream_ptr()), "tf_upload")
    torch.cuda.synchronize()
    assert torch.equal(d, torch.from_numpy(a).cuda())
    assert (int(h[0]), int(h[1])) == _hash_host(L, a) == _hash_dev(L, d)
    back = _staging.download(d, remember=False)
    assert back.dtype == np.uint8 and np.array_equal(back, a) and back.flags.writeable
    if n >= (1 << 20):
        assert L.tf_host_is_pinned(ctypes.c_void_p(back.ctypes.data), n) == 1
        # a pinned source goes out in one DMA
        d2 = torch.zeros_like(d)
        _lib.check(L.tf_upload(_lib.ptr(d2), ctypes.c_void_p(back.ctypes.data), n, None, _lib.stream_ptr()), "tf_upload")
        torch.cuda.synchronize()
        assert torch.equal(d2, d)
    assert L.tf_host_is_pinned(a.ctypes.data_as(_lib._P), n) == 0
    assert L.tf_upload(None, a.ctypes.data_as(_lib._P), n, None, None) == -1


def test_pinned_blocks_return_to_the_pool_with_their_last_view():
    import torch
    from tobac_flow_amd import _lib, _staging
    L = _lib.lib()
    live, cached = ctypes.c_int64(), ctypes.c_int64()
    _staging.clear()
    gc.collect()
    L.tf_host_pool_stats(ctypes.byref(live), ctypes.byref(cached))
    live0 = live.value
    d = torch.arange(3 << 20, dtype=torch.int32, device="cuda")
    out = _staging.download(d, remember=False)
    view = out[5:100]
    L.tf_host_pool_stats(ctypes.byref(live), ctypes.byref(cached))
    assert live.value >= live0 + out.nbytes
    ptr = out.ctypes.data
    del out
    gc.collect()
    assert L.tf_host_is_pinned(ctypes.c_void_p(ptr), 16) == 1            # a view still holds the block
    assert int(view[0]) == 5
    del view
    gc.collect()
    L.tf_host_pool_stats(ctypes.byref(live), ctypes.byref(cached))
    assert live.value == live0 and cached.value >= 12 << 20 and L.tf_host_is_pinned(ctypes.c_void_p(ptr), 16) == 0
    again = _staging.download(d, remember=False)                          # the same size class: the cached block is handed out again
    assert again.ctypes.data == ptr and np.array_equal(again, np.arange(3 << 20, dtype=np.int32))
    del again
    gc.collect()
    L.tf_host_pool_trim(0)
    L.tf_host_pool_stats(ctypes.byref(live), ctypes.byref(cached))
    assert cached.value == 0
    p = ctypes.c_void_p(12345)
    assert L.tf_host_free(p) == -1                                        # not a block of the pool


def test_an_array_presented_again_is_recognised_by_content_never_by_address(monkeypatch):
    import torch
    from tobac_flow_amd import _staging
    monkeypatch.delenv("TF_HOST_CACHE_GB", raising=False)
    _staging.clear()
    s0 = dict(_staging.stats)
    rng = np.random.default_rng(11)
    wvd = rng.normal(size=(6, 300, 400)).astype(np.float32)
    swd = rng.normal(size=(6, 300, 400)).astype(np.float32)
    d1 = _staging.upload(wvd)
    assert _staging.stats["uploads"] == s0["uploads"] + 1
    d2 = _staging.upload(wvd)                                             # the same object again
    assert d2 is d1 and _staging.stats["hits"] == s0["hits"] + 1 and _staging.stats["uploads"] == s0["uploads"] + 1
    e1 = _staging.upload(wvd - swd)                                       # two temporaries with the same values (dcc_detect_goes.py:227,241)
    e2 = _staging.upload(wvd - swd)
    assert e2 is e1 and torch.equal(e1.cpu(), torch.from_numpy(wvd - swd))
    # mutated in place: same address, same shape -- another content, another upload
    wvd[3, 100, 200] += 1.0
    d3 = _staging.upload(wvd)
    assert d3 is not d1 and torch.equal(d3.cpu(), torch.from_numpy(wvd)) and not torch.equal(d3, d1)
    # a recycled buffer: same address, other values
    buf = np.empty_like(swd)
    buf[...] = swd
    f1 = _staging.upload(buf)
    buf[...] = swd * 2
    f2 = _staging.upload(buf)
    assert f2 is not f1 and torch.equal(f2.cpu(), torch.from_numpy(swd * 2)) and torch.equal(f1.cpu(), torch.from_numpy(swd))
    # fresh=True: private, never shared
    g = _staging.upload(swd, fresh=True)
    assert g is not f1 and torch.equal(g, f1)
    # a result handed out and coming back (markers=): the device tensor it was downloaded from
    lab = torch.randint(0, 50, (6, 300, 400), dtype=torch.int32, device="cuda")
    host = _staging.download(lab)
    assert _staging.upload(host) is lab
    host[0, 0, 0] += 1                                                    # the caller edits the labels: uploaded, not trusted
    again = _staging.upload(host)
    assert again is not lab and int(again[0, 0, 0]) == int(lab[0, 0, 0]) + 1
    # a hit on another stream waits for the producing stream and is safe to use there
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        d4 = _staging.upload(wvd)
        total = float(d4.double().sum())
    assert d4 is d3 and abs(total - float(wvd.astype(np.float64).sum())) < 1e-6 * wvd.size
    # bool arrays travel as bytes; small arrays are never cached; the switch
    m = rng.random((5, 700, 900)) > 0.5
    dm = _staging.upload(m)
    assert dm.dtype == torch.uint8 and np.array_equal(dm.cpu().numpy().astype(bool), m)
    small = np.arange(10, dtype=np.float32)
    assert _staging.upload(small) is not _staging.upload(small)
    monkeypatch.setenv("TF_HOST_CACHE_GB", "0")
    assert _staging.upload(swd) is not _staging.upload(swd)
    _staging.clear()
    monkeypatch.setenv("TF_HOST_CACHE_GB", "0.02")                        # 20 MB: room for six 2.9 MB volumes -- the oldest goes first
    vols = [rng.normal(size=(6, 300, 400)).astype(np.float32) for _ in range(8)]
    devs = [_staging.upload(v) for v in vols]
    assert _staging.upload(vols[-1]) is devs[-1] and _staging.upload(vols[0]) is not devs[0]
    _staging.clear()


def test_to_device_and_to_host_helpers():
    import torch
    import tobac_flow_amd
    from tobac_flow_amd.detection import DeviceField
    from tools.synth import field_with_time
    rng = np.random.default_rng(2)
    bt = field_with_time(rng.normal(size=(5, 300, 400)).astype(np.float32) + 280, minutes=5)
    plain = rng.normal(size=(5, 300, 400)).astype(np.float32)
    b, p = tobac_flow_amd.to_device(bt, plain)
    assert isinstance(b, DeviceField) and b.data.is_cuda and np.array_equal(b.t.values, bt.t.values)
    assert isinstance(p, torch.Tensor) and p.is_cuda
    assert np.array_equal(tobac_flow_amd.to_host(b), np.asarray(bt)) and np.array_equal(tobac_flow_amd.to_host(p), plain)
    assert tobac_flow_amd.to_device(b) is b
    d = b - b
    assert isinstance(d, DeviceField) and float(d.data.abs().max()) == 0.0
    tobac_flow_amd.clear_device_cache()


def test_the_script_sequence_gives_the_same_labels_whatever_the_container_and_the_cache(monkeypatch):
    """The drop-in script's call sequence (scripts/dcc_detect_goes.py:164-303) on one scene, four ways: host containers with
    recognition by content switched off (every array uploaded), switched on (twice in a row: the second pass finds results of
    the first still remembered), and device-resident (tobac_flow_amd.to_device).  Every result equal, array for array --
    in particular no recipe writes into a device twin that a later call is served."""
    import warnings
    import torch
    import tobac_flow_amd
    import tobac_flow_amd.flow as tf
    from tobac_flow_amd import _staging
    from tobac_flow_amd.detection import detect_anvils, detect_cores, get_anvil_markers, relabel_anvils
    from tools.script_sequence import scene
    from tools.synth import field_with_time
    T, H, W, minutes = 10, 300, 420, 5
    bt_d, wvd_d, swd_d = scene(T, H, W, minutes, torch.device("cuda", 0))
    host = [field_with_time(x.cpu().numpy(), minutes=minutes) for x in (bt_d, wvd_d, swd_d)]
    assert host[0].nbytes >= 1 << 20                                    # (large enough to be remembered)

    def sequence(bt, wvd, swd):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            flow = tf.create_flow(bt, model="Farneback", vr_steps=1, smoothing_passes=1, interp_method="cubic")
            core = detect_cores(flow, bt, wvd, swd, wvd_threshold=0.25, bt_threshold=0.5, overlap=0.5, absolute_overlap=4,
                                subsegment_shrink=0.0, min_length=2, use_wvd=False)
            markers = get_anvil_markers(flow, wvd - swd, threshold=-5, overlap=0.5, absolute_overlap=4, subsegment_shrink=0.0, min_length=2)
            thick0 = detect_anvils(flow, wvd - swd, markers=markers, upper_threshold=-5, lower_threshold=-12.5, erode_distance=2, min_length=2)
            thick = relabel_anvils(flow, thick0, markers=markers, overlap=0.5, absolute_overlap=4, min_length=2)
            thin = detect_anvils(flow, wvd + swd, markers=thick, upper_threshold=0, lower_threshold=-7.5, erode_distance=2, min_length=2)
        return [tobac_flow_amd.to_host(x) if isinstance(x, torch.Tensor) else np.asarray(x) for x in (core, markers, thick0, thick, thin)]

    def host_fields():                                                   # (arithmetic on the plain arrays: a fresh temporary per evaluation, like the script's)
        return [field_with_time(np.asarray(h).copy(), minutes=minutes) for h in host]
    monkeypatch.setenv("TF_HOST_CACHE_GB", "0")
    _staging.clear()
    plain = sequence(*host_fields())
    assert int(plain[1].max()) >= 1 and int(plain[4].max()) >= 1, [int(p.max()) for p in plain]
    monkeypatch.delenv("TF_HOST_CACHE_GB")
    before = dict(_staging.stats)
    first = sequence(*host_fields())
    hits_first = _staging.stats["hits"] - before["hits"]
    second = sequence(*host_fields())
    assert hits_first >= 6                                               # bt, wvd - swd, the markers twice, the two anvil volumes
    assert _staging.stats["hits"] - before["hits"] >= 2 * hits_first     # (the second pass finds at least as much)
    device = sequence(*tobac_flow_amd.to_device(*host_fields()))
    for name, a, b, c, d in zip(("core", "markers", "thick0", "thick", "thin"), plain, first, second, device):
        assert a.dtype == b.dtype == c.dtype == d.dtype and np.array_equal(a, b) and np.array_equal(a, c) and np.array_equal(a, d), name
    _staging.clear()
