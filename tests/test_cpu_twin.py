"""CPU: the C++ sparse-formulation twin (include/csmpn_cpu.h) against the golden vectors recorded from
the imported reference - the same fixtures that pin the PyTorch oracle. A second, torch-free ground
truth for the kernels and bench.py's strong CPU baseline; never on the product path."""
import os

import numpy as np
import pytest

from oracle import cpu_twin

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ALGS = ["cl20", "cl30", "cl50", "cl41"]
EGCL_TAGS = ["sum_res1_ag0", "sum_res1_ag1", "sum_res0_ag0", "mean_res1_ag0", "mean_res1_ag1", "mean_res0_ag0", "noattr"]


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("name", ALGS)
@pytest.mark.parametrize("variant", EGCL_TAGS)
def test_twin_egcl_vs_reference_fixture(name, variant):
    g = np.load(os.path.join(GOLD, f"egcl_{name}.npz"))
    metric = np.load(os.path.join(GOLD, f"tables_{name}.npz"))["metric"]
    f32, f64 = f"f32/{variant}", f"f64/{variant}"
    p = {k[len(f32) + 3:]: g[k] for k in g.files if k.startswith(f32 + "/p/")}
    noattr = variant == "noattr"
    ag = variant.endswith("ag1")
    res = cpu_twin.egcl_layer(metric, p, g[f"{f32}/h"], g[f"{f32}/edge_index"],
                              None if noattr else g[f"{f32}/edge_attr"], None if noattr else g[f"{f32}/node_attr"],
                              aggr="mean" if noattr else variant.split("_")[0], residual="res0" not in variant,
                              gout=g[f"{f32}/gout"], want_attr_grads=ag, threads=2)
    # float32 against the float64 reference run; the reference's own float32 run is the yardstick
    slack = 20.0 if (metric < 0).any() else 4.0   # null-cone norms: ill-conditioned in any float32 evaluation
    bound = lambda k: max(1e-5, slack * rel(g[f"{f32}/{k}"], g[f"{f64}/{k}"]))
    assert rel(res["out"], g[f"{f64}/y"]) <= bound("y")
    assert rel(res["gh"], g[f"{f64}/gh"]) <= bound("gh")
    if ag:
        assert rel(res["g_edge_attr"], g[f"{f64}/g_edge_attr"]) <= bound("g_edge_attr")
        assert rel(res["g_node_attr"], g[f"{f64}/g_node_attr"]) <= bound("g_node_attr")
    for k, v in res["grads"].items():
        assert rel(v, g[f"{f64}/g/{k}"]) <= bound(f"g/{k}"), k


@pytest.mark.parametrize("name", ALGS)
@pytest.mark.parametrize("variant", ["sum_res1_ag1", "mean_res0_ag0", "noattr"])
def test_twin_float64_build_vs_reference_float64_fixture(name, variant):
    """The float64 build of the twin (oracle/_build/libcsmpn_cpu64.so: the truth of tests/test_full_size_twin.py) against
    the reference's own float64 run: same parameters and inputs (the float32 ones, widened), agreement to rounding."""
    g = np.load(os.path.join(GOLD, f"egcl_{name}.npz"))
    metric = np.load(os.path.join(GOLD, f"tables_{name}.npz"))["metric"]
    f32, f64 = f"f32/{variant}", f"f64/{variant}"
    p = {k[len(f32) + 3:]: g[k] for k in g.files if k.startswith(f32 + "/p/")}
    noattr = variant == "noattr"
    ag = variant.endswith("ag1")
    for k in ("h", "gout"):
        assert np.array_equal(g[f"{f32}/{k}"].astype(np.float64), g[f"{f64}/{k}"]), k
    res = cpu_twin.egcl_layer(metric, p, g[f"{f32}/h"], g[f"{f32}/edge_index"],
                              None if noattr else g[f"{f32}/edge_attr"], None if noattr else g[f"{f32}/node_attr"],
                              aggr="mean" if noattr else variant.split("_")[0], residual="res0" not in variant,
                              gout=g[f"{f32}/gout"], want_attr_grads=ag, threads=2, real64=True)
    assert res["out"].dtype == np.float64
    tol = 1e-7 if (metric < 0).any() else 1e-10
    assert rel(res["out"], g[f"{f64}/y"]) <= tol
    assert rel(res["gh"], g[f"{f64}/gh"]) <= tol
    if ag:
        assert rel(res["g_edge_attr"], g[f"{f64}/g_edge_attr"]) <= tol
        assert rel(res["g_node_attr"], g[f"{f64}/g_node_attr"]) <= tol
    for k, v in res["grads"].items():
        assert v.dtype == np.float64 and rel(v, g[f"{f64}/g/{k}"]) <= tol, k


@pytest.mark.parametrize("name", ALGS)
@pytest.mark.parametrize("tag", ["cemlp1_C3", "cemlp2_C8"])
def test_twin_cemlp_vs_reference_fixture(name, tag):
    g = np.load(os.path.join(GOLD, f"layers_{name}.npz"))
    metric = np.load(os.path.join(GOLD, f"tables_{name}.npz"))["metric"]
    p = {k[len(tag) + 3:]: g[k] for k in g.files if k.startswith(tag + "/p/")}
    y, gx, grads = cpu_twin.cemlp(metric, p, g[f"{tag}/x"], g[f"{tag}/gout"], threads=1)
    tol = 2e-4 if (metric < 0).any() else 2e-5     # float32 twin against the float32 fixture
    assert rel(y, g[f"{tag}/y"]) <= tol
    assert rel(gx, g[f"{tag}/gx"]) <= 10 * tol
    for k, v in grads.items():
        assert rel(v, g[f"{tag}/g/{k}"]) <= 10 * tol, k


def test_twin_rejects_bad_indices():
    g = np.load(os.path.join(GOLD, "egcl_cl30.npz"))
    f32 = "f32/noattr"
    p = {k[len(f32) + 3:]: g[k] for k in g.files if k.startswith(f32 + "/p/")}
    ei = g[f"{f32}/edge_index"].copy()
    ei[1, 0] = 10_000
    with pytest.raises(RuntimeError, match="outside"):
        cpu_twin.egcl_layer([1.0, 1.0, 1.0], p, g[f"{f32}/h"], ei)


@pytest.mark.parametrize("tag", ["S1", "M32"])
def test_twin_float64_vs_reference_fixture_full_size(tag):
    """The float64 build of the twin - the truth of tests/test_full_size_twin.py, and the only truth of S2 / H28 where the
    reference's dense formulation does not fit in memory - against the imported reference's own float64 run at FULL size
    (100 k edges; tests/golden/make_fullsize_golden.py): y and d/dh on the stored node subsample, every parameter gradient."""
    import importlib
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    T = importlib.import_module("test_full_size_twin")
    g = np.load(os.path.join(GOLD, f"fullsize_{tag}.npz"))
    metric, C, aggr, h, ei, ea, na, p, gout, t64, _t32 = T._case(tag)
    np.testing.assert_allclose(T._input_checksums(h, ei, ea, na, p, gout), g["checksums"], rtol=1e-12, atol=0)
    st = int(g["node_stride"])
    assert rel(t64["out"][::st], g["f64/y"]) < 1e-10 and rel(t64["gh"][::st], g["f64/gh"]) < 1e-10
    for k, v in t64["grads"].items():
        assert rel(v, g[f"f64/g/{k}"]) < 1e-10, k
