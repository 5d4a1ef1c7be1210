"""ctypes loader for oracle/_build/liboracle.so (ORACLE, test infrastructure only)."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        if os.environ.get("ORACLE_LIB"):               # another build of the same sources (the sanitizer job: make -C oracle asan-test)
            _LIB = ctypes.CDLL(os.environ["ORACLE_LIB"])
            return _LIB
        so = os.path.join(_HERE, "_build", "liboracle.so")
        srcs = [os.path.join(_HERE, "c", f) for f in os.listdir(os.path.join(_HERE, "c"))]
        if (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(["make", "-s", "-C", _HERE, "_build/liboracle.so"])
        _LIB = ctypes.CDLL(so)
    return _LIB


def ptr(a, ctype):
    return a.ctypes.data_as(ctypes.POINTER(ctype))
