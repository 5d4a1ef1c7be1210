"""Development probe (round 5): could the tie value of a flood be BOUNDED before the flood runs?

A label can hang on the reference heap's order of equal-valued markers only where two relevant markers (markers with a
floodable neighbour) of DIFFERENT labels carry the same value.  The largest such shared value is an upper bound of the
tie value the flood finds after its root phase (k_ws_tie_value_max) -- if it is tight on the benchmark's windows, the
export for the host replay could always happen on it at `begin` (no guess from history, no second export).  Prints, per
window of the benchmark's config-F sequence: the bound, the number of shared values above the background's value, and
the tie value the flood reports.

    python tools/tie_bound_probe.py [n_windows] [frames_per_window]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import tobac_flow_amd.flow as tf                                             # noqa: E402
from tobac_flow_amd.detection import get_combined_edge_field                 # noqa: E402
from tobac_flow_amd.watershed import neighbour_offsets, watershed_dev        # noqa: E402
from tools.synth import anvil_seeds, blob_stack                              # noqa: E402


def key_to_float(k):
    k = int(k) & 0xFFFFFFFF
    u = (k & 0x7FFFFFFF) if (k & 0x80000000) else (~k & 0xFFFFFFFF)
    return float(np.array([u], np.uint32).view(np.float32)[0])


def main():
    n_win = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    H = W = int(os.environ.get("PROBE_SIZE", "5424"))
    nbr = neighbour_offsets(1)
    for w in range(int(os.environ.get('PROBE_FIRST', '0')), n_win):
        lo = w * 12
        bt = blob_stack(L, H, W, seed=20240601, t0=lo)
        flow = tf.create_flow(bt, model="Farneback", vr_steps=1, smoothing_passes=1, interp_method="cubic")
        lin, seeds = anvil_seeds(bt)
        e = get_combined_edge_field(flow, lin, dtype=np.float32)
        fw, bw = flow._dev_flows()
        st = {}
        watershed_dev(fw, bw, e, seeds, None, nbr, 3, stats=st, on_ambiguous="reference")
        det = st.get("reference_order_detail", {})
        # relevant markers (un-displaced neighbourhood: a probe): markers with a floodable face neighbour
        flood = (seeds == 0)
        near = torch.zeros_like(flood)
        near[1:] |= flood[:-1]; near[:-1] |= flood[1:]
        near[:, 1:] |= flood[:, :-1]; near[:, :-1] |= flood[:, 1:]
        near[:, :, 1:] |= flood[:, :, :-1]; near[:, :, :-1] |= flood[:, :, 1:]
        rel = near & (seeds != 0)
        del near, flood
        v = e[rel].contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
        lab = seeds[rel].to(torch.int64) & 0xFFFFFFFF
        key = torch.where((v & 0x80000000) != 0, (~v) & 0xFFFFFFFF, v | 0x80000000)
        pairs = torch.unique((key << 32) | lab)                              # sorted distinct (value key, label)
        k = (pairs >> 32) & 0xFFFFFFFF                                      # (the shift is arithmetic: keys of positive values have bit 31 set)
        shared = k[1:][k[1:] == k[:-1]]                                       # value keys carried by >= 2 labels
        shared = torch.unique(shared)
        bg_key = 0x80000000
        above = shared[shared > bg_key]
        bound = int(shared.max()) if shared.numel() else -1
        print("window %2d: relevant markers %8d, shared values %6d (%d above the background's), bound %s; flood: tie key %s (%s), form %s, "
              "guessed %s covered %s" % (
                  w, int(rel.sum()), int(shared.numel()), int(above.numel()),
                  "%.6g" % key_to_float(bound) if bound >= 0 else "none",
                  det.get("tie_key"), "%.6g" % key_to_float(det["tie_key"]) if det.get("tie_key") else "-",
                  det.get("replay_form"), det.get("guessed"), det.get("guess_covered_the_tie")), flush=True)
        if det.get("tie_key") and key_to_float(det["tie_key"]) > 0:
            tv = torch.tensor(key_to_float(det["tie_key"]), dtype=torch.float32, device=e.device)
            hit = (e == tv)
            idx = hit.nonzero()
            print("           voxels AT the tie value: %d (markers among them: %d, relevant by the probe: %d)" % (int(hit.sum()), int((hit & (seeds != 0)).sum()), int((hit & rel).sum())), flush=True)
            for t, y, x in idx[:16].tolist():
                print("             (t %d, y %d, x %d) seed %d  lin %.6g  neighbours' seeds: %s" % (
                    t, y, x, int(seeds[t, y, x]), float(lin[t, y, x]),
                    [int(seeds[t, max(y - 1, 0), x]), int(seeds[t, min(y + 1, H - 1), x]), int(seeds[t, y, max(x - 1, 0)]), int(seeds[t, y, min(x + 1, W - 1)])]), flush=True)
        if above.numel():
            print("           shared values above the background's: " + ", ".join("%.6g" % key_to_float(int(a)) for a in above[:12].tolist()), flush=True)
        del flow, fw, bw, e, lin, seeds, bt


if __name__ == "__main__":
    main()
