"""CPU (no GPU needed): the C-ABI library loads and exports every symbol
include/csmpn_hip.h declares; host-side tables are bit-exact against the golden
vectors; the nn.Module surface matches the reference's state_dict contract; the
product path refuses CPU tensors instead of falling back."""
import os
import re

import numpy as np
import pytest
import torch

ALGS = ["cl20", "cl30", "cl50", "cl41"]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(pkg):
    from csmpn_hip import native
    header = open(os.path.join(ROOT, "include", "csmpn_hip.h")).read()
    declared = set(re.findall(r"\b(csmpn_[a-z0-9_]+)\s*\(", header))
    assert declared == set(native.EXPORTS), declared ^ set(native.EXPORTS)
    lib = native.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.csmpn_abi_version() == native.ABI_VERSION == 2
    assert lib.csmpn_build_target() == b"gfx950"
    # ... and NOTHING else: the dynamic symbol table (defined function symbols) is exactly the header's list - no csmpn::
    # launchers, no template instances, no diagnostic hooks (csrc/exports.map)
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", native.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1].split("@")[0] for l in nm.splitlines() if len(l.split()) >= 3 and l.split()[-2] in "TtWw"}
    assert exported == declared, sorted(exported ^ declared)


@pytest.mark.parametrize("name", ALGS)
def test_host_tables_bit_exact(pkg, golden_dir, name):
    from csmpn.algebra.metric import native_tables
    g = np.load(os.path.join(golden_dir, f"tables_{name}.npz"))
    t = native_tables(g["metric"].tolist())
    assert np.array_equal(t["cayley"], g["cayley"])
    assert np.array_equal(t["index_to_bitmap"], g["index_to_bitmap"])
    assert np.array_equal(t["bitmap_to_index"], g["bitmap_to_index"])
    assert np.array_equal(t["grades"], g["grades"])
    assert np.array_equal(t["subspaces"], g["subspaces"])
    assert np.array_equal(t["paths"].astype(bool), g["paths"])


def test_general_metric_tables(pkg):
    """Non +-1 metric entries (e.g. PGA's degenerate generator) are supported by the
    host tables even though the device kernels reject them."""
    from csmpn.algebra.metric import native_tables
    from oracle.tables import AlgebraTables
    from csmpn_hip import native
    for metric in ([0.0, 1.0, 1.0, 1.0], [2.0, -0.5, 1.0]):
        t = native_tables(metric)
        assert np.array_equal(t["cayley"], AlgebraTables(metric).cayley)
        assert native.lib().csmpn_metric_supported(native.metric_array(metric), len(metric)) == 0


@pytest.mark.parametrize("name", ALGS)
def test_clifford_algebra_surface(pkg, golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"tables_{name}.npz"))
    a = np.load(os.path.join(golden_dir, f"algebra_{name}.npz"))
    alg = pkg.CliffordAlgebra(tuple(g["metric"].tolist()))
    assert list(dict(alg.named_buffers())) == ["metric", "subspaces", "bbo_grades", "even_grades", "odd_grades", "cayley"]
    assert np.array_equal(alg.cayley.numpy(), g["cayley"])
    assert np.array_equal(alg.geometric_product_paths.numpy(), g["paths"])
    assert alg.dim == len(g["metric"]) and alg.n_blades == 2 ** alg.dim
    x = torch.from_numpy(a["x"])
    # helper methods run on any device (plain tensor code, not the hot path)
    np.testing.assert_allclose(alg.q(x).numpy(), a["q"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(alg.norm(x).numpy(), a["norm"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(torch.cat(alg.qs(x), -1).numpy(), a["qs"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(torch.cat(alg.norms(x), -1).numpy(), a["norms"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(alg.beta(torch.from_numpy(a["a"])).numpy(), a["beta"])
    np.testing.assert_allclose(alg.alpha(torch.from_numpy(a["a"])).numpy(), a["alpha"])
    np.testing.assert_allclose(alg.gamma(torch.from_numpy(a["a"])).numpy(), a["gamma"])
    np.testing.assert_allclose(alg.b(x, x).numpy(), a["q"], rtol=1e-5, atol=1e-6)
    # dense einsum form of geometric_product (CPU tensors: table lookup path)
    np.testing.assert_allclose(alg.geometric_product(torch.from_numpy(a["a"]), torch.from_numpy(a["b"])).numpy(),
                               a["gp"], rtol=1e-5, atol=1e-5)


def test_egcl_state_dict_contract(pkg, golden_dir):
    alg = pkg.CliffordAlgebra((1.0, 1.0, 1.0))
    layer = pkg.EGCL(alg, 8, 8, 8, edge_attr_features=6, node_attr_features=3)
    ours = {f"{k} {tuple(v.shape)} {v.dtype}" for k, v in layer.state_dict().items()}
    ref = set(open(os.path.join(golden_dir, "egcl_state_dict_keys.txt")).read().split("\n")) - {""}
    assert ours == ref
    assert sum(p.numel() for p in layer.parameters()) == 4736


def test_initialisers_match_reference_statistics(pkg):
    torch.manual_seed(0)
    alg = pkg.CliffordAlgebra((1.0, 1.0, 1.0))
    m = pkg.CEMLP(alg, 64, 64, 64)
    blk = m.layers[0]
    assert abs(blk[0].weight.std().item() - 1 / 8) < 0.01       # N(0, 1/sqrt(in))
    assert torch.all(blk[0].bias == 0) and torch.all(blk[1].a == 1) and torch.all(blk[1].b == 0)
    assert abs(blk[2].weight.std().item() - 0.5) < 0.05         # N(0, 1/sqrt(dim+1))
    assert torch.all(blk[2].normalization.a == 0) and torch.all(blk[3].a == 1)


def test_product_path_refuses_cpu_tensors(pkg):
    alg = pkg.CliffordAlgebra((1.0, 1.0, 1.0))
    layer = pkg.EGCL(alg, 4, 4, 4)
    h = torch.randn(5, 4, 8)
    ei = torch.randint(0, 5, (2, 7))
    with pytest.raises(RuntimeError, match="GPU|MI355X"):
        layer(h, ei)
    with pytest.raises(RuntimeError, match="MI355X"):
        layer.edge_model(torch.randn(5, 4, 8))


def test_unsupported_configs_fail_loudly(pkg):
    from csmpn_hip import native, ops
    b = ops.CemlpBinding((1.0, 0.0, 1.0), [dict(in_features=2, out_features=2)])
    assert not b.supported()
    with pytest.raises(native.CsmpnError):
        ops.EgclSpec(None, None, 1, 1, 0, 0, "max", True)


def test_modules_copy_and_pickle_after_binding(pkg):
    """The cached ctypes bindings (raw pointers) must not break deepcopy / torch.save of a module
    that has already run (EMA, best-model snapshots): they are dropped and rebuilt on demand."""
    import copy
    import io
    layer = pkg.EGCL(pkg.CliffordAlgebra((1.0, 1.0, 1.0)), 4, 4, 4, edge_attr_features=6, node_attr_features=3)
    layer.spec()
    layer.edge_model.binding()
    clone = copy.deepcopy(layer)
    assert clone._spec is None and clone.edge_model._binding is None
    assert clone.spec() is not None
    buf = io.BytesIO()
    torch.save(layer, buf)
    buf.seek(0)
    back = torch.load(buf, weights_only=False)
    assert sorted(back.state_dict()) == sorted(layer.state_dict())


def test_flat_parameters_guard_against_set_to_none(pkg):
    """flatten_parameters keeps its aliasing in the .grad attributes: zero_grad(set_to_none=True) - PyTorch's default -
    breaks it silently (round-4 advice). flat_gradients_intact() sees it, zero_flat_grad() is the safe fill."""
    from csmpn_hip.graphed import flat_gradients_intact, flatten_parameters, zero_flat_grad
    m = torch.nn.Sequential(torch.nn.Linear(3, 5), torch.nn.Linear(5, 2))
    flat = flatten_parameters(m)
    opt = torch.optim.SGD([flat], lr=0.1)
    m(torch.randn(4, 3)).sum().backward()
    assert flat_gradients_intact(flat) and float(flat.grad.abs().sum()) > 0
    before = flat.detach().clone()
    opt.step()
    assert not torch.equal(before, flat.detach())          # every module parameter moved through the ONE flat update
    assert torch.equal(m[0].weight.detach().reshape(-1), flat.detach()[:15])
    zero_flat_grad(flat)
    assert flat_gradients_intact(flat) and float(m[1].bias.grad.abs().sum()) == 0.0
    opt.zero_grad()                                          # set_to_none=True: the aliasing is gone
    assert not flat_gradients_intact(flat)


def test_saved_buffer_sizes_of_the_standalone_32_channel_cemlps(pkg):
    """csmpn_cemlp_saved_floats is the sizing contract of the saved buffer (host-side query, no GPU): for the standalone
    32-channel Cl(3,0) CEMLPs that the 16-row-tile family serves (the md17 embeddings 60 -> 32, 90 -> 32 -> 32 and the head
    32 -> 32) CSMPN_FLAG_SAVE_STATE adds three state regions per block of whole 16-row tiles; one-block modules have nothing
    else in the buffer, two-block modules the block-1 inputs + the hand-over rows in front - at every row count, also below
    the general kernels' phased-backward threshold. Shapes outside the family are not touched by the flag."""
    from csmpn_hip import native, ops
    lib = native.lib()
    metric = (1.0, 1.0, 1.0)
    rows = 1001
    tiles16 = (rows + 15) // 16 * 16
    for in_f, nblk in ((60, 1), (90, 2), (32, 1)):
        specs = [dict(in_features=in_f if k == 0 else 32, out_features=32) for k in range(nblk)]
        b = ops.CemlpBinding(metric, specs)
        plain = int(lib.csmpn_cemlp_saved_floats(b.n, b.params, b.nblk, rows, 0))
        state = int(lib.csmpn_cemlp_saved_floats(b.n, b.params, b.nblk, rows, native.FLAG_SAVE_STATE))
        base = 2 * 32 * 8 * rows if nblk == 2 else 0
        assert plain == base and state == base + 3 * nblk * 32 * 8 * tiles16, (in_f, nblk, plain, state)
        assert int(lib.csmpn_cemlp_saved_floats_per_row(b.n, b.params, b.nblk)) > 0
        assert int(lib.csmpn_cemlp_workspace_bytes(b.n, b.params, b.nblk)) > 768 * 4 * 32 * 48   # tables + one slice per workgroup
    # a 16-channel standalone CEMLP (the motion model's): no state regions, whatever the flag
    b = ops.CemlpBinding(metric, [dict(in_features=40, out_features=16), dict(in_features=16, out_features=16)])
    assert lib.csmpn_cemlp_saved_floats(b.n, b.params, b.nblk, rows, 0) == lib.csmpn_cemlp_saved_floats(b.n, b.params, b.nblk, rows, native.FLAG_SAVE_STATE)
