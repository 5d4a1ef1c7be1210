"""ORACLE (test infrastructure): import the reference's own compiled flood kernel from oracle/_ref."""
import glob
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))


def available():
    tag = "py%d%d" % sys.version_info[:2]
    return bool(glob.glob(os.path.join(_HERE, "_ref", tag, "_watershed*.so")))


def ref_watershed_raveled():
    tag = "py%d%d" % sys.version_info[:2]
    d = os.path.join(_HERE, "_ref", tag)
    if not available():
        from . import build_ref
        if build_ref.build() is None:
            raise ImportError("oracle/_ref not built and /root/reference absent")
    if d not in sys.path:
        sys.path.insert(0, d)
    import _watershed
    return _watershed.watershed_raveled
