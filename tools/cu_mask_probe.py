"""Which XCD / CU does bit i of a hipExtStreamCreateWithCUMask mask select on this part?  Launches idle workgroups on masked
streams and prints where they ran (tf_debug_cu_histogram)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from tobac_flow_amd import _lib

# (round 6: these entry points are no longer in libtobac_flow_hip.so -- build tools/experiments/stream_experiments.hip, see its head)
_lib.lib()
L = ctypes.CDLL(os.environ.get("TF_STREAM_EXPERIMENTS_LIB", "gpurun_out/libstream_experiments.so"))
t = _lib.torch()
t.cuda.init()
t.zeros(1, device="cuda")


def probe(name, bits):
    words = np.zeros(8, np.uint32)
    for b in bits:
        words[b // 32] |= np.uint32(1 << (b % 32))
    s = ctypes.c_void_p()
    assert L.tf_stream_create_cu_mask(words.ctypes.data_as(_lib._P), 8, ctypes.byref(s)) == 0
    hist = np.zeros(2048, np.int32)
    assert L.tf_debug_cu_histogram(s, 4096, hist.ctypes.data_as(_lib._P)) == 0
    L.tf_stream_destroy(s)
    h = hist.reshape(8, 256)
    per_xcc = h.sum(1)
    cus = [(x, int(c)) for x in range(8) for c in np.nonzero(h[x])[0]]
    print("%-28s workgroups per XCC %s; %d distinct (xcc, se/sh/cu) slots" % (name, per_xcc.tolist(), len(cus)))
    if len(cus) <= 40:
        print("     ", ["x%d:%02x" % c for c in cus])


probe("bits 0..31", range(32))
probe("bits 0..63", range(64))
probe("bits i % 8 == 0", [i for i in range(256) if i % 8 == 0])
probe("bits i % 8 < 2", [i for i in range(256) if i % 8 < 2])
probe("bits 0..255", range(256))
probe("bit 0", [0])
probe("bit 1", [1])
probe("bit 8", [8])
