"""The four small Clifford layers used on their own — MVSiLU, NormalizationLayer,
SteerableGeometricProductLayer, MVLayerNorm (reference: csmpn/models/cegnn_utils.py:34-155) —
against the fixtures recorded from the imported reference (tests/golden/layers_*.npz:
`mvsilu_C*`, `norm_C*`, `sgp_C*`, `mvlayernorm_C*`; forward, d/dx and every parameter
gradient).

CPU: the package's host formulation of the standalone `forward` (tensor ops; CEMLP / EGCL never
call it). GPU (`-m gpu`): the same modules on device tensors, which routes them through the
standalone HIP entry points of include/csmpn_hip.h (csmpn_mvsilu_*, csmpn_mvnorm_*,
csmpn_mvlayernorm_*, csmpn_wgp_*)."""
import os

import numpy as np
import pytest
import torch

ALGS = ["cl20", "cl30", "cl50", "cl41"]
LAYERS = ["mvsilu", "norm", "sgp", "mvlayernorm"]


def _build(pkg, layer, alg, C):
    from csmpn.models import cegnn_utils as cu
    if layer == "mvsilu":
        return cu.MVSiLU(alg, C)
    if layer == "norm":
        return cu.NormalizationLayer(alg, C)
    if layer == "sgp":
        return cu.SteerableGeometricProductLayer(alg, C)
    return cu.MVLayerNorm(alg, C)


def _run(pkg, golden_dir, name, C, layer, device, fwd_tol, bwd_tol):
    g = np.load(os.path.join(golden_dir, f"layers_{name}.npz"))
    t = np.load(os.path.join(golden_dir, f"tables_{name}.npz"))
    alg = pkg.CliffordAlgebra(tuple(t["metric"].tolist()))
    tag = f"{layer}_C{C}"
    mod = _build(pkg, layer, alg, C)
    pre = f"{tag}/p/"
    state = {k[len(pre):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(pre)}
    missing = mod.load_state_dict(state, strict=False)
    assert not missing.unexpected_keys and all(".algebra." in "." + k for k in missing.missing_keys), missing
    mod = mod.to(device)
    x = torch.from_numpy(g[f"{tag}/x"]).to(device).requires_grad_(True)
    y = mod(x)
    scale = float(np.abs(g[f"{tag}/y"]).max())
    np.testing.assert_allclose(y.detach().cpu().numpy(), g[f"{tag}/y"], rtol=fwd_tol, atol=fwd_tol * scale)
    (y * torch.from_numpy(g[f"{tag}/gout"]).to(device)).sum().backward()
    gs = float(np.abs(g[f"{tag}/gx"]).max())
    np.testing.assert_allclose(x.grad.cpu().numpy(), g[f"{tag}/gx"], rtol=bwd_tol, atol=bwd_tol * gs)
    params = dict(mod.named_parameters())
    gpre = f"{tag}/g/"
    checked = 0
    for k in g.files:
        if not k.startswith(gpre):
            continue
        p = params[k[len(gpre):]]
        assert p.grad is not None, k
        ps = max(float(np.abs(g[k]).max()), 1e-30)
        np.testing.assert_allclose(p.grad.cpu().numpy(), g[k], rtol=bwd_tol, atol=bwd_tol * ps, err_msg=k)
        checked += 1
    assert checked == len(params)


@pytest.mark.parametrize("name", ALGS)
@pytest.mark.parametrize("C", [3, 8])
@pytest.mark.parametrize("layer", LAYERS)
def test_small_layers_host(pkg, golden_dir, name, C, layer):
    _run(pkg, golden_dir, name, C, layer, torch.device("cpu"), 2e-5, 1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ALGS)
@pytest.mark.parametrize("C", [3, 8])
@pytest.mark.parametrize("layer", LAYERS)
def test_small_layers_hip(pkg, golden_dir, name, C, layer):
    from csmpn_hip import ops
    before = ops.small_layer_launches()
    _run(pkg, golden_dir, name, C, layer, torch.device("cuda:0"), 2e-5, 1e-4)
    assert ops.small_layer_launches() > before, "the standalone HIP entry point did not run"


@pytest.mark.gpu
@pytest.mark.parametrize("name,C,rows", [("cl30", 28, 1000), ("cl50", 28, 257), ("cl41", 8, 1031), ("cl20", 40, 513)])
@pytest.mark.parametrize("layer", LAYERS)
def test_small_layers_hip_vs_host_fp64(pkg, golden_dir, name, C, rows, layer):
    """Shapes beyond the fixtures (many workgroups, row tails, the hulls width): the HIP entry points against the host
    formulation of the same module in float64 on the CPU (forward, d/dx, parameter gradients)."""
    import copy
    t = np.load(os.path.join(golden_dir, f"tables_{name}.npz"))
    metric = tuple(t["metric"].tolist())
    torch.manual_seed(1234)
    alg = pkg.CliffordAlgebra(metric)
    mod = _build(pkg, layer, alg, C)
    with torch.no_grad():
        for p in mod.parameters():
            p.add_(0.3 * torch.randn_like(p))
    ref = copy.deepcopy(mod).double()
    host32 = copy.deepcopy(mod)            # yardstick: the same formulation in float32 on the CPU
    x = torch.randn(rows, C, 1 << len(metric))
    gout = torch.randn(rows, C, 1 << len(metric))
    xr = x.double().requires_grad_(True)
    yr = ref(xr)
    (yr * gout.double()).sum().backward()
    xh = x.clone().requires_grad_(True)
    yh = host32(xh)
    (yh * gout).sum().backward()
    dev = torch.device("cuda:0")
    mod = mod.to(dev)
    xd = x.to(dev).requires_grad_(True)
    from csmpn_hip import ops
    before = ops.small_layer_launches()
    yd = mod(xd)
    (yd * gout.to(dev)).sum().backward()
    assert ops.small_layer_launches() > before

    def close(a, h, b, what):
        # bound: 5e-5 of the tensor's scale, or 4x the float32 host formulation's own error where the layer is
        # ill-conditioned (norms near the null cone of an indefinite metric)
        b = b.float()
        scale = float(b.abs().max().clamp(min=1e-30))
        err = float((a.cpu() - b).abs().max()) / scale
        yard = float((h - b).abs().max()) / scale
        assert err < max(5e-5, 4 * yard), f"{what}: rel err {err:.2e} (float32 host formulation: {yard:.2e})"

    close(yd.detach(), yh.detach(), yr.detach(), "y")
    close(xd.grad, xh.grad, xr.grad, "gx")
    for (k, p), (_, h), (_, q) in zip(mod.named_parameters(), host32.named_parameters(), ref.named_parameters()):
        close(p.grad, h.grad, q.grad, k)
