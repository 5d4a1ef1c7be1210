"""tobac_flow_amd -- the hot path of tobac-flow (dense optical flow -> semi-Lagrangian Sobel /
growth detection -> marker-controlled watershed) on AMD MI355X (gfx950).

Module layout and public names mirror the reference package `tobac_flow`; the numeric back ends
(cv2 Farnebaeck / remap, the Cython heap flood) are replaced by hand-written HIP kernels behind the
C ABI of include/tobac_flow_hip.h.  There is no CPU fallback: compute entry points need the built
library (tobac_flow_amd/csrc/libtobac_flow_hip.so) and a HIP device.
"""
__version__ = "0.1.0"
