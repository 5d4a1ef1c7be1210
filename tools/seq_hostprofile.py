"""Where the HOST time of the drop-in sequence goes (device-resident mode: wall - kernels is all host): cProfile over the
second pass of tools/script_sequence.py's calls.  python tools/seq_hostprofile.py [--config C]"""
import cProfile
import io
import os
import pstats
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tobac_flow_amd.flow as tf  # noqa: E402
from tobac_flow_amd.detection import detect_anvils, detect_cores, get_anvil_markers, relabel_anvils  # noqa: E402
from tools.script_sequence import scene  # noqa: E402
from tools.synth import field_with_time  # noqa: E402

cfg_c = "--config" in sys.argv and sys.argv[sys.argv.index("--config") + 1] == "C"
T, H, W, minutes = (24, 1500, 2500, 5) if cfg_c else (16, 5424, 5424, 10)
bt_d, wvd_d, swd_d = scene(T, H, W, minutes, torch.device("cuda", 0))
bt, wvd, swd = (field_with_time(x, minutes=minutes) for x in (bt_d, wvd_d, swd_d))
wd, ws = field_with_time(wvd_d - swd_d, minutes=minutes), field_with_time(wvd_d + swd_d, minutes=minutes)


def sequence():
    flow = tf.create_flow(bt, model="Farneback", vr_steps=1, smoothing_passes=1, interp_method="cubic")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        core = detect_cores(flow, bt, wvd, swd, wvd_threshold=0.25, bt_threshold=0.5, overlap=0.5, absolute_overlap=4, subsegment_shrink=0.0, min_length=3, use_wvd=False)
        markers = get_anvil_markers(flow, wd, threshold=-5, overlap=0.5, absolute_overlap=4, subsegment_shrink=0.0, min_length=3)
        thick0 = detect_anvils(flow, wd, markers=markers, upper_threshold=-5, lower_threshold=-12.5, erode_distance=2, min_length=3)
        thick = relabel_anvils(flow, thick0, markers=markers, overlap=0.5, absolute_overlap=4, min_length=3)
        thin = detect_anvils(flow, ws, markers=thick, upper_threshold=0, lower_threshold=-7.5, erode_distance=2, min_length=3)
    torch.cuda.synchronize()
    return core, thin


sequence()
pr = cProfile.Profile()
pr.enable()
sequence()
pr.disable()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(35)
print(out.getvalue()[:9000])
