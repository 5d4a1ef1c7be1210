"""Build the reference's own Cython flood kernel into oracle/_ref/ (TEST INFRASTRUCTURE ONLY).

The only first-party native component of the reference is
/root/reference/tobac_flow/_watershed.pyx.  This recipe compiles it
*from where it lies* -- nothing from the reference is copied into the repository; only the
compiled extension module stays in oracle/_ref/ (git-ignored); the generated C file is deleted.

Usage:  python oracle/build_ref.py [python-executable]
        (default interpreter: the one running this script)

The resulting module is importable as ``_watershed`` after adding oracle/_ref/<tag>/ to
sys.path (see oracle/ref_loader.py).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may use it, and only as the checker.
"""
import os
import subprocess
import sys
import sysconfig

REF_PYX = "/root/reference/tobac_flow/_watershed.pyx"
HERE = os.path.dirname(os.path.abspath(__file__))


def build(python=sys.executable, quiet=True):
    if not os.path.exists(REF_PYX):
        return None  # GPU box: the reference is absent, prebuilt files (if any) are used
    tag = subprocess.check_output(
        [python, "-c", "import sys;print('py%d%d' % sys.version_info[:2])"], text=True).strip()
    out = os.path.join(HERE, "_ref", tag)
    os.makedirs(out, exist_ok=True)
    cfile = os.path.join(out, "_watershed.c")
    ext = subprocess.check_output(
        [python, "-c", "import sysconfig;print(sysconfig.get_config_var('EXT_SUFFIX'))"],
        text=True).strip()
    so = os.path.join(out, "_watershed" + ext)
    if os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(REF_PYX):
        return so
    inc = subprocess.check_output(
        [python, "-W", "ignore", "-c",
         "import sysconfig, numpy;print(sysconfig.get_paths()['include']);print(numpy.get_include())"],
        text=True).split()
    subprocess.check_call([python, "-W", "ignore", "-m", "cython", "-3", "-o", cfile, REF_PYX],
                          stdout=subprocess.DEVNULL if quiet else None,
                          stderr=subprocess.DEVNULL if quiet else None)
    cmd = ["gcc", "-O2", "-fPIC", "-shared", "-w", "-I" + inc[0], "-I" + inc[1],
           "-DNPY_NO_DEPRECATED_API=0", cfile, "-o", so]
    subprocess.check_call(cmd)
    # the generated C file quotes the .pyx line by line in its comments: it is an intermediate, not a deliverable, and
    # must not travel with the tree (the compiled module is all the checker needs)
    os.remove(cfile)
    return so


if __name__ == "__main__":
    py = sys.argv[1] if len(sys.argv) > 1 else sys.executable
    print(build(py, quiet=False))
