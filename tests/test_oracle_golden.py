"""The oracle pinned against the reference's own outputs (tests/golden/watershed_ref.npz was produced
by the reference itself: tests/golden/make_watershed_golden.py) and, when oracle/_ref is built (i.e.
when /root/reference is present), live against the reference's compiled Cython kernel."""
import numpy as np
import pytest

from oracle import ref_loader, ws_oracle

CASES = ["A_cont_c1", "B_cont_mask_c2", "B_cont_mask_c3", "C_quant4_c1", "C_quant32_c1", "D_anvil_like_c1",
         "E_const_plateau_c1", "F_zero_flow_c1", "G_big_flow_c1"]


@pytest.mark.parametrize("name", CASES)
def test_c_twin_matches_reference_golden(golden_ws, name):
    c = golden_ws[name]
    got = ws_oracle.watershed(c["fwd"], c["bwd"], c["field"], c["markers"], c.get("mask"), int(c["conn"]))
    assert got.dtype == np.int32
    assert np.array_equal(got, c["labels"])


@pytest.mark.parametrize("name", CASES)
def test_c_twin_matches_live_reference_kernel(golden_ws, name):
    if not ref_loader.available():
        pytest.skip("oracle/_ref not built (reference absent on this machine)")
    c = golden_ws[name]
    conn = int(c["conn"])
    a = ws_oracle.watershed(c["fwd"], c["bwd"], c["field"], c["markers"], c.get("mask"), conn)
    b = ws_oracle.watershed(c["fwd"], c["bwd"], c["field"], c["markers"], c.get("mask"), conn, use_ref=True)
    assert np.array_equal(a, b)


def test_idealised_order_characterisation(golden_ws):
    """Where the reference's output depends on heap-internal order of equal-valued markers: the
    idealised (value, age, push sequence) queue differs ONLY on the tie-heavy cases, by a known amount."""
    expect = {"A_cont_c1": 0, "B_cont_mask_c2": 0, "B_cont_mask_c3": 0, "C_quant4_c1": 21, "C_quant32_c1": 0,
              "D_anvil_like_c1": 0, "E_const_plateau_c1": 37, "F_zero_flow_c1": 0, "G_big_flow_c1": 0}
    for name, n in expect.items():
        c = golden_ws[name]
        ideal = ws_oracle.watershed(c["fwd"], c["bwd"], c["field"], c["markers"], c.get("mask"), int(c["conn"]), tie_mode=1)
        assert int((ideal != c["labels"]).sum()) == n, name


def test_neighbour_orders_match_fixture():
    import os
    from tobac_flow_amd.watershed import neighbour_offsets
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "watershed_ref.npz"))
    for k in (1, 2, 3):
        want = z["neighbour_order/conn%d" % k]
        assert np.array_equal(neighbour_offsets(k), want)
        assert np.array_equal(ws_oracle.neighbour_offsets(k), want)
        import scipy.ndimage as ndi
        assert np.array_equal(neighbour_offsets(ndi.generate_binary_structure(3, k)), want)


def test_parallel_model_equals_idealised_oracle(golden_ws):
    """The chain-key formulation implemented by the HIP kernels (tools/ws_parallel_model.py is its
    numpy model) reproduces the sequential flood exactly."""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    import ws_parallel_model as M
    for name, depth in (("A_cont_c1", 1), ("D_anvil_like_c1", 3), ("C_quant4_c1", 3), ("G_big_flow_c1", 1)):
        c = golden_ws[name]
        conn = int(c["conn"])
        got, _ = M.run(c["fwd"], c["bwd"], c["field"], c["markers"], c.get("mask"), conn, depth=depth)
        ideal = ws_oracle.watershed(c["fwd"], c["bwd"], c["field"], c["markers"], c.get("mask"), conn, tie_mode=1)
        assert np.array_equal(got, ideal), name


@pytest.mark.parametrize("name", CASES)
def test_exactness_report_of_the_parallel_model_covers_every_difference_from_the_reference(golden_ws, name):
    """The contract the HIP flood reports against (include/tobac_flow_hip.h, tf_watershed_ex2), checked on its numpy
    model: (1) whatever the depth, every pixel that differs from the REFERENCE's labels is reported as depending on
    a last-resort tie-break; (2) no origin left by the depth cut-off <=> equal to the idealised-order flood;
    (3) deepening on its own ends with the reference's labels everywhere outside the reported marker ties."""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    import ws_parallel_model as M
    c = golden_ws[name]
    conn = int(c["conn"])
    args = (c["fwd"], c["bwd"], c["field"], c["markers"], c.get("mask"), conn)
    ideal = ws_oracle.watershed(*args, tie_mode=1)
    for depth in (1, 2, 3):
        got, info = M.run(*args, depth=depth)
        rep = info["report"]
        assert not ((got != c["labels"]) & ((rep & 1) == 0)).any(), depth
        assert not ((got != ideal) & ((rep & 1) == 0)).any(), depth
        if not (rep & 4).any():
            assert np.array_equal(got, ideal), depth
    got, info = M.run_auto(*args, depth=1)
    assert np.array_equal(got, ideal)
    assert not (info["report"] & 4).any()
    assert info["depth"] == {"C_quant32_c1": 6, "C_quant4_c1": 3, "D_anvil_like_c1": 3, "E_const_plateau_c1": 2}.get(name, 1)
    differs = got != c["labels"]
    assert not (differs & ((info["report"] & 1) == 0)).any()
    if name not in ("C_quant4_c1", "C_quant32_c1", "E_const_plateau_c1"):
        assert not info["report"].any() and not differs.any()
