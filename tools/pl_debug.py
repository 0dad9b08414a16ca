"""PL kernels against the parity-split kernels (CSMPN_NO_PL=1 in a second process is the reference): stage outputs."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("clifford-group-equivariant-simplicial-message-passing-networks_amd")
from oracle import ref_path as O
from csmpn_hip import ops
dev = torch.device("cuda:0")
N, E, C = int(os.environ.get("N", 120)), int(os.environ.get("E", 1001)), 8
METRIC = [1.0, 1.0, 1.0, 1.0, -1.0]
torch.manual_seed(0)
layer = pkg.EGCL(pkg.CliffordAlgebra(tuple(METRIC)), C, C, C, edge_attr_features=6, node_attr_features=3, aggr="mean").to(dev)
h, ei, ea, na = (t.to(dev) for t in O.synthetic_complex(O.Algebra(METRIC), N, E, C, seed=0))
be, spec = ops.HipBackend, layer.spec()
csr = ops.get_csr(ei, N)
pe, pn = layer.edge_model.flat_params(), layer.node_model.flat_params()
gout = torch.randn(N, C, 32, device=dev)
agg, st_e = be.edge_forward(spec, csr, h, ea, pe)
out, st_n = be.node_forward(spec, csr.deg, h, agg, na, pn)
gh, g_agg, g_na, gn = be.node_backward(spec, csr.deg, h, agg, na, pn, gout, True, st_n)
g_ea, ge = be.edge_backward(spec, csr, h, ea, pe, g_agg, gh, True, st_e)
torch.cuda.synchronize()
res = {"agg": agg, "saved_e": st_e[1], "out": out, "saved_n": st_n[1], "gh": gh, "g_agg": g_agg, "g_na": g_na, "g_ea": g_ea}
for i, v in enumerate(gn): res[f"gn{i}"] = v
for i, v in enumerate(ge): res[f"ge{i}"] = v
path = os.environ.get("DUMP")
if path:
    torch.save({k: v.cpu() for k, v in res.items() if v is not None}, path)
ref = os.environ.get("REF")
for k, v in res.items():
    if v is None: continue
    line = f"{k:8s} nan={int(torch.isnan(v).sum())} max={float(v.abs().max()):.4g}"
    if ref:
        r = torch.load(ref)[k].to(dev)
        line += f" relerr={float((v - r).abs().max() / r.abs().max().clamp(min=1e-30)):.3e}"
    print(line)
