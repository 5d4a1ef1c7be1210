"""tobac_flow_amd -- the hot path of tobac-flow (dense optical flow -> semi-Lagrangian Sobel /
growth detection -> marker-controlled watershed) on AMD MI355X (gfx950).

Module layout and public names mirror the reference package `tobac_flow`; the numeric back ends
(cv2 Farnebaeck / remap, the Cython heap flood) are replaced by hand-written HIP kernels behind the
C ABI of include/tobac_flow_hip.h.  There is no CPU fallback: compute entry points need the built
library (tobac_flow_amd/csrc/libtobac_flow_hip.so) and a HIP device.
"""
__version__ = "0.1.0"


def to_device(*fields):
    """`bt, wvd, swd = tobac_flow_amd.to_device(bt, wvd, swd)`: upload the fields once and run the entry points
    device-resident (detection.DeviceField = tensor + time coordinate); see _staging.to_device and INTEGRATION.md."""
    from tobac_flow_amd._staging import to_device as f
    return f(*fields)


def to_host(x):
    """device tensor / DeviceField -> numpy array (in a pooled pinned block)"""
    from tobac_flow_amd._staging import to_host as f
    return f(x)


def clear_device_cache():
    """drop the device twins remembered for host arrays (and the pinned pool's free blocks): _staging.clear"""
    from tobac_flow_amd._staging import clear
    clear()
