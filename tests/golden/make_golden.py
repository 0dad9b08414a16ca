"""Generate golden input/output vectors from the IMPORTED reference.

Runs only in the build container (needs /root/reference); the resulting
``*.npz`` files are data (inputs + expected outputs) and are committed, the
reference itself never travels. Usage:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

For every metric in {Cl(2,0), Cl(3,0), Cl(5,0), Cl(4,1)} it records
  tables_<alg>.npz   cayley, blade order, grades, subspaces, grade paths
  algebra_<alg>.npz  geometric_product, q, norm, qs, norms on random inputs
  layers_<alg>.npz   MVLinear(subspaces/no-subspaces/bias) , MVSiLU, NormalizationLayer,
                     SteerableGeometricProductLayer, MVLayerNorm, CEMLP(1|2 layers):
                     input, randomised parameters, output, d(input), d(parameters)
  egcl_<alg>.npz     EGCL fwd + all grads at N=12, E=40 (duplicate edges,
                     self loops, one isolated node), aggr in {sum, mean},
                     residual on/off, attr grads, fp32 and fp64 runs
  egcl8_<alg>.npz    (`make_golden.py egcl8`, Cl(3,0) and Cl(4,1)) the same cases at 8 channels
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np
import torch

import pyg_standin

pyg_standin.install()
REF = os.environ.get("CSMPN_REFERENCE", "/root/reference")
if not os.path.isdir(REF):
    print("reference not present: nothing to do")
    sys.exit(0)
sys.path.insert(0, REF)

from csmpn.algebra.cliffordalgebra import CliffordAlgebra  # noqa: E402
from csmpn.models import cegnn_utils as R  # noqa: E402

ALGEBRAS = {
    "cl20": (1.0, 1.0),
    "cl30": (1.0, 1.0, 1.0),
    "cl50": (1.0, 1.0, 1.0, 1.0, 1.0),
    "cl41": (1.0, 1.0, 1.0, 1.0, -1.0),
}


def npy(t):
    return t.detach().cpu().numpy()


def randomize_(module, gen):
    """Move every parameter away from its 0/1 initial value."""
    with torch.no_grad():
        for name, prm in module.named_parameters():
            leaf = name.split(".")[-1]
            if leaf in ("a", "b", "bias"):
                prm.add_(0.3 * torch.randn(prm.shape, generator=gen, dtype=torch.float32).to(prm.dtype))


def params_of(module):
    return {k: npy(v) for k, v in module.state_dict().items() if ".algebra." not in k and not k.startswith("algebra.")}


def run_layer(out, tag, module, x, gen):
    x = x.clone().requires_grad_(True)
    y = module(x)
    gout = torch.randn(y.shape, generator=gen, dtype=torch.float32).to(y.dtype)
    (y * gout).sum().backward()
    out[f"{tag}/x"] = npy(x)
    out[f"{tag}/y"] = npy(y)
    out[f"{tag}/gout"] = npy(gout)
    out[f"{tag}/gx"] = npy(x.grad)
    for k, v in params_of(module).items():
        out[f"{tag}/p/{k}"] = v
    for k, v in module.named_parameters():
        out[f"{tag}/g/{k}"] = npy(v.grad)


def make_tables(name, metric):
    alg = CliffordAlgebra(metric)
    np.savez_compressed(
        os.path.join(HERE, f"tables_{name}.npz"),
        metric=np.asarray(metric, dtype=np.float32),
        cayley=npy(alg.cayley),
        index_to_bitmap=npy(alg.bbo.index_to_bitmap),
        bitmap_to_index=npy(alg.bbo.bitmap_to_index),
        grades=npy(alg.bbo.grades),
        subspaces=npy(alg.subspaces),
        paths=npy(alg.geometric_product_paths),
    )


def make_algebra(name, metric):
    gen = torch.Generator().manual_seed(11)
    alg = CliffordAlgebra(metric)
    D = alg.n_blades
    a = torch.randn(7, D, generator=gen)
    b = torch.randn(7, D, generator=gen)
    x = torch.randn(5, 3, D, generator=gen)
    np.savez_compressed(
        os.path.join(HERE, f"algebra_{name}.npz"),
        a=npy(a), b=npy(b), x=npy(x),
        gp=npy(alg.geometric_product(a, b)),
        q=npy(alg.q(x)), norm=npy(alg.norm(x)),
        qs=npy(torch.cat(alg.qs(x), dim=-1)),
        norms=npy(torch.cat(alg.norms(x), dim=-1)),
        beta=npy(alg.beta(a)), alpha=npy(alg.alpha(a)), gamma=npy(alg.gamma(a)),
        embed_grade1=npy(alg.embed_grade(torch.randn(4, alg.dim, generator=gen), 1)),
    )


def make_layers(name, metric):
    gen = torch.Generator().manual_seed(23)
    torch.manual_seed(23)
    alg = CliffordAlgebra(metric)
    D = alg.n_blades
    out = {}
    B = 5
    for C in (3, 8):
        x = torch.randn(B, C, D, generator=gen)
        mods = {
            f"mvlinear_C{C}": R.MVLinear(alg, C, C + 2),
            f"mvlinear_nosub_C{C}": R.MVLinear(alg, C, C + 1, subspaces=False),
            f"mvlinear_nobias_C{C}": R.MVLinear(alg, C, C, bias=False),
            f"mvsilu_C{C}": R.MVSiLU(alg, C),
            f"norm_C{C}": R.NormalizationLayer(alg, C, init=0.2),
            f"sgp_C{C}": R.SteerableGeometricProductLayer(alg, C),
            f"mvlayernorm_C{C}": R.MVLayerNorm(alg, C),
            f"cemlp1_C{C}": R.CEMLP(alg, C, 5, 4, n_layers=1),
            f"cemlp2_C{C}": R.CEMLP(alg, C, 5, 4, n_layers=2),
        }
        for tag, m in mods.items():
            randomize_(m, gen)
            run_layer(out, tag, m, x, gen)
    np.savez_compressed(os.path.join(HERE, f"layers_{name}.npz"), **out)


def _load_f32_params(layer64, layer32):
    """The float64 layer gets the float32 layer's parameters (cast), so that both fixture runs
    describe the SAME function; the algebra buffers keep the float64 layer's own values."""
    sd = layer64.state_dict()
    for k, v in layer32.state_dict().items():
        if ".algebra." in k or k.startswith("algebra."):
            continue
        sd[k] = v.detach().to(torch.float64)
    layer64.load_state_dict(sd, strict=True)


def make_egcl(name, metric, C=4, kind="egcl"):
    out = {}
    N, E, T = 12, 40, 3
    # inputs are drawn once in float32 and cast; layers are initialised once in float32 and the
    # float64 run loads the same parameters (round 1 drew them twice under different default
    # dtypes: the f32 and f64 fixtures then described different layers)
    torch.set_default_dtype(torch.float32)
    gen = torch.Generator().manual_seed(37)
    ei = torch.randint(0, N - 1, (2, E), generator=gen)  # node N-1 is isolated
    ei[:, 5] = ei[:, 4]            # duplicate edge
    ei[:, 9] = ei[:, 4]            # triplicate
    ei[1, 12] = ei[0, 12]          # self loop
    ei[1, 20:28] = 3               # high in-degree node
    types = torch.randint(0, T, (N,), generator=gen)
    D = 1 << len(metric)
    h0_32 = torch.randn(N, C, D, generator=gen, dtype=torch.float32)

    def run(dt_name, dtype, layer, tag, h0, edge_attr, node_attr, attr_grad):
        h = h0.clone().requires_grad_(True)
        if attr_grad:
            edge_attr = edge_attr.clone().requires_grad_(True)
            node_attr = node_attr.clone().requires_grad_(True)
        y = layer(h, ei, edge_attr, node_attr) if edge_attr is not None else layer(h, ei)
        gout = torch.randn(y.shape, generator=torch.Generator().manual_seed(47), dtype=torch.float32).to(dtype)
        (y * gout).sum().backward()
        t = f"{dt_name}/{tag}"
        out[f"{t}/h"] = npy(h)
        out[f"{t}/edge_index"] = npy(ei)
        if edge_attr is not None:
            out[f"{t}/edge_attr"] = npy(edge_attr)
            out[f"{t}/node_attr"] = npy(node_attr)
        out[f"{t}/y"] = npy(y)
        out[f"{t}/gout"] = npy(gout)
        out[f"{t}/gh"] = npy(h.grad)
        if attr_grad:
            out[f"{t}/g_edge_attr"] = npy(edge_attr.grad)
            out[f"{t}/g_node_attr"] = npy(node_attr.grad)
        for k, v in params_of(layer).items():
            out[f"{t}/p/{k}"] = v
        for k, v in layer.named_parameters():
            out[f"{t}/g/{k}"] = npy(v.grad)

    variants = []
    for aggr in ("sum", "mean"):
        for residual in (True, False):
            for attr_grad in (False, True):
                if attr_grad and not residual:
                    continue
                variants.append((f"{aggr}_res{int(residual)}_ag{int(attr_grad)}", aggr, residual, attr_grad))
    variants.append(("noattr", "mean", True, False))

    for tag, aggr, residual, attr_grad in variants:
        noattr = tag == "noattr"
        torch.set_default_dtype(torch.float32)
        alg32 = CliffordAlgebra(metric)
        g2 = torch.Generator().manual_seed(41)
        torch.manual_seed(41)
        if noattr:
            layer32 = R.EGCL(alg32, C, C + 1, C, aggr="mean")
        else:
            layer32 = R.EGCL(alg32, C, C, C, edge_attr_features=2 * T, node_attr_features=T,
                             residual=residual, aggr=aggr)
        randomize_(layer32, g2)
        na0 = alg32.embed_grade(torch.nn.functional.one_hot(types, T).float()[..., None], 0)
        if noattr:
            ea32 = na32 = None
        elif attr_grad:
            na32 = na0 + 0.1 * torch.randn(na0.shape, generator=g2, dtype=torch.float32)
            ea32 = torch.randn(E, 2 * T, D, generator=g2, dtype=torch.float32)
        else:
            na32 = na0.clone()
            ea32 = torch.cat([na32[ei[0]], na32[ei[1]]], dim=1)
        run("f32", torch.float32, layer32, tag, h0_32, ea32, na32, attr_grad)
        torch.set_default_dtype(torch.float64)
        alg64 = CliffordAlgebra(metric)
        if noattr:
            layer64 = R.EGCL(alg64, C, C + 1, C, aggr="mean")
        else:
            layer64 = R.EGCL(alg64, C, C, C, edge_attr_features=2 * T, node_attr_features=T,
                             residual=residual, aggr=aggr)
        _load_f32_params(layer64, layer32)
        run("f64", torch.float64, layer64, tag, h0_32.double(),
            None if ea32 is None else ea32.double(), None if na32 is None else na32.double(), attr_grad)
        torch.set_default_dtype(torch.float32)
    np.savez_compressed(os.path.join(HERE, f"{kind}_{name}.npz"), **out)


def make_state_dict_keys():
    alg = CliffordAlgebra((1.0, 1.0, 1.0))
    layer = R.EGCL(alg, 8, 8, 8, edge_attr_features=6, node_attr_features=3)
    keys = [f"{k} {tuple(v.shape)} {v.dtype}" for k, v in layer.state_dict().items()]
    with open(os.path.join(HERE, "egcl_state_dict_keys.txt"), "w") as f:
        f.write("\n".join(keys) + "\n")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "egcl8":
        # round 3: the same EGCL cases at 8 channels - the width the lane kernels serve ((row, channel)-per-lane kernels
        # for Cl(3,0), parity-lane kernels for Cl(4,1)), so that the deterministic mode can be held to the fixture
        for name in ("cl30", "cl41"):
            make_egcl(name, ALGEBRAS[name], C=8, kind="egcl8")
            print("8-channel EGCL vectors written for", name)
        sys.exit(0)
    for name, metric in ALGEBRAS.items():
        make_tables(name, metric)
        make_algebra(name, metric)
        make_layers(name, metric)
        make_egcl(name, metric)
        print("golden vectors written for", name)
    make_state_dict_keys()
