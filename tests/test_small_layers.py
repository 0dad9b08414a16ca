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
