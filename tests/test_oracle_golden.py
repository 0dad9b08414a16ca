"""CPU: pin the oracle (oracle/tables.py, oracle/ref_path.py) against the golden
vectors captured from the imported reference (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_path as O
from oracle.tables import AlgebraTables

ALGS = ["cl20", "cl30", "cl50", "cl41"]


def load(golden_dir, kind, name):
    return np.load(os.path.join(golden_dir, f"{kind}_{name}.npz"))


@pytest.mark.parametrize("name", ALGS)
def test_tables_bit_exact(golden_dir, name):
    g = load(golden_dir, "tables", name)
    t = AlgebraTables(g["metric"].tolist())
    assert np.array_equal(t.index_to_bitmap, g["index_to_bitmap"])
    assert np.array_equal(t.bitmap_to_index, g["bitmap_to_index"])
    assert np.array_equal(t.grades, g["grades"])
    assert np.array_equal(t.subspaces, g["subspaces"])
    assert np.array_equal(t.cayley, g["cayley"])          # exact: entries are +-1, 0
    assert np.array_equal(t.paths, g["paths"])
    # exactly one non-zero per (left, right) pair
    assert ((t.cayley != 0).sum(axis=1) == 1).all()


@pytest.mark.parametrize("name", ALGS)
def test_algebra_ops(golden_dir, name):
    g = load(golden_dir, "algebra", name)
    t = load(golden_dir, "tables", name)
    alg = O.Algebra(t["metric"].tolist())
    a, b, x = (torch.from_numpy(g[k]) for k in ("a", "b", "x"))
    np.testing.assert_allclose(O.geometric_product(alg, a, b).numpy(), g["gp"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(O.q_all(alg, x).numpy(), g["q"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(O.norm_all(alg, x).numpy(), g["norm"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(torch.cat(O.q_grades(alg, x), -1).numpy(), g["qs"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(torch.cat(O.norm_grades(alg, x), -1).numpy(), g["norms"], rtol=1e-6, atol=1e-6)
    # closed form used by the HIP kernels: q_g = sum_d qsign[d] x_d^2
    qs = torch.stack([(alg.qsign[s] * x[..., s] ** 2).sum(-1) for s in alg.grade_slices], -1)
    np.testing.assert_allclose(qs.numpy(), g["qs"], rtol=1e-5, atol=1e-6)


def _layer_params(g, tag):
    pre = f"{tag}/p/"
    return {k[len(pre):]: torch.from_numpy(g[k]).clone().requires_grad_(True) for k in g.files if k.startswith(pre)}


LAYER_FNS = {
    "mvlinear": lambda alg, x, p: O.mv_linear(alg, x, p["weight"], p.get("bias")),
    "mvlinear_nosub": lambda alg, x, p: O.mv_linear(alg, x, p["weight"], p.get("bias")),
    "mvlinear_nobias": lambda alg, x, p: O.mv_linear(alg, x, p["weight"], None),
    "mvsilu": lambda alg, x, p: O.mv_silu(alg, x, p["a"], p["b"]),
    "norm": lambda alg, x, p: O.normalization(alg, x, p["a"]),
    "sgp": lambda alg, x, p: O.steerable_gp(alg, x, p, ""),
    "mvlayernorm": lambda alg, x, p: O.mv_layernorm(alg, x, p["a"]),
    "cemlp1": lambda alg, x, p: O.cemlp(alg, x, p),
    "cemlp2": lambda alg, x, p: O.cemlp(alg, x, p),
}


@pytest.mark.parametrize("name", ALGS)
@pytest.mark.parametrize("C", [3, 8])
@pytest.mark.parametrize("layer", sorted(LAYER_FNS))
def test_layers_fwd_bwd(golden_dir, name, C, layer):
    g = load(golden_dir, "layers", name)
    t = load(golden_dir, "tables", name)
    alg = O.Algebra(t["metric"].tolist())
    tag = f"{layer}_C{C}"
    p = _layer_params(g, tag)
    x = torch.from_numpy(g[f"{tag}/x"]).clone().requires_grad_(True)
    y = LAYER_FNS[layer](alg, x, p)
    np.testing.assert_allclose(y.detach().numpy(), g[f"{tag}/y"], rtol=2e-5, atol=2e-6)
    (y * torch.from_numpy(g[f"{tag}/gout"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), g[f"{tag}/gx"], rtol=1e-4, atol=1e-5)
    for k, v in p.items():
        ref = g[f"{tag}/g/{k}"]
        np.testing.assert_allclose(v.grad.numpy(), ref, rtol=1e-4, atol=1e-5 * max(1.0, np.abs(ref).max()))


def egcl_case(g, tag, dtype):
    p = {k[len(tag) + 3:]: torch.from_numpy(g[k]).to(dtype).requires_grad_(True)
         for k in g.files if k.startswith(f"{tag}/p/")}
    get = lambda k: torch.from_numpy(g[f"{tag}/{k}"]) if f"{tag}/{k}" in g.files else None
    return p, get


EGCL_TAGS = ["sum_res1_ag0", "sum_res1_ag1", "sum_res0_ag0", "mean_res1_ag0", "mean_res1_ag1", "mean_res0_ag0", "noattr"]


@pytest.mark.parametrize("name", ALGS)
@pytest.mark.parametrize("variant", EGCL_TAGS)
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_egcl_fwd_bwd(golden_dir, name, variant, prec):
    g = load(golden_dir, "egcl", name)
    t = load(golden_dir, "tables", name)
    dtype = torch.float32 if prec == "f32" else torch.float64
    alg = O.Algebra(t["metric"].tolist(), dtype)
    tag = f"{prec}/{variant}"
    p, get = egcl_case(g, tag, dtype)
    h = get("h").clone().requires_grad_(True)
    ea, na = get("edge_attr"), get("node_attr")
    ag = variant.endswith("ag1")
    if ag:
        ea = ea.clone().requires_grad_(True)
        na = na.clone().requires_grad_(True)
    aggr = "mean" if variant == "noattr" else variant.split("_")[0]
    residual = "res0" not in variant
    y = O.egcl(alg, h, get("edge_index"), ea, na, p, aggr=aggr, residual=residual)
    rt, at = (1e-4, 2e-5) if prec == "f32" else (1e-9, 1e-10)
    np.testing.assert_allclose(y.detach().numpy(), get("y").numpy(), rtol=rt, atol=at)
    (y * get("gout")).sum().backward()
    scale = lambda r: max(1.0, float(np.abs(r).max()))
    np.testing.assert_allclose(h.grad.numpy(), get("gh").numpy(), rtol=rt * 10, atol=at * 10 * scale(get("gh").numpy()))
    if ag:
        np.testing.assert_allclose(ea.grad.numpy(), get("g_edge_attr").numpy(), rtol=rt * 10, atol=at * 10)
        np.testing.assert_allclose(na.grad.numpy(), get("g_node_attr").numpy(), rtol=rt * 10, atol=at * 10)
    for k, v in p.items():
        ref = g[f"{tag}/g/{k}"]
        np.testing.assert_allclose(v.grad.numpy(), ref, rtol=rt * 10, atol=at * 10 * scale(ref))


@pytest.mark.parametrize("name", ALGS)
def test_egcl_fixture_runs_share_parameters(golden_dir, name):
    """The f32 and f64 EGCL fixture runs must describe the same layer (round-1 fixtures did not)."""
    g = load(golden_dir, "egcl", name)
    n = 0
    for k in g.files:
        if k.startswith("f32/") and "/p/" in k:
            assert np.abs(g[k].astype(np.float64) - g["f64/" + k[4:]]).max() <= 1e-6, k
            n += 1
    assert n > 0
    for variant in EGCL_TAGS:
        y32, y64 = g[f"f32/{variant}/y"], g[f"f64/{variant}/y"]
        assert np.abs(y32 - y64).max() / np.abs(y64).max() < 1e-5


@pytest.mark.parametrize("name", ALGS)
def test_embed_grade_fixture(pkg, golden_dir, name):
    """a6: embed_grade / get_grade of the product algebra against the reference fixture
    (cliffordalgebra.py:105-117)."""
    g = load(golden_dir, "algebra", name)
    t = load(golden_dir, "tables", name)
    alg = pkg.CliffordAlgebra(tuple(t["metric"].tolist()))
    ref = torch.from_numpy(g["embed_grade1"])
    n = len(t["metric"])
    v = ref[..., 1:1 + n]
    out = alg.embed_grade(v, 1)
    assert out.shape == ref.shape and torch.equal(out, ref)
    assert torch.equal(alg.get_grade(out, 1), v)
    assert torch.equal(alg.embed(v, tuple(range(1, 1 + n))), ref)
    assert torch.equal(alg.get(out, tuple(range(1, 1 + n))), v)
