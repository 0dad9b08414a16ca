import importlib, sys, os, torch
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("clifford-group-equivariant-simplicial-message-passing-networks_amd")
from oracle import ref_path as O
from csmpn_hip import ops
dev = torch.device('cuda:0')
o32 = O.Algebra([1.,1.,1.])
alg = pkg.CliffordAlgebra((1.,1.,1.))
torch.manual_seed(0)
for I, rows in ((19, 5), (19, 300), (14, 300), (20, 300), (35, 300)):
    m = pkg.CEMLP(alg, I, 8, 8)
    p = {k: v.detach().clone() for k, v in m.named_parameters()}
    x = torch.randn(rows, I, 8)
    y = m.to(dev)(x.to(dev)).cpu()
    ref = O.cemlp(o32, x, p)
    print(f"PLAIN I={I} rows={rows}: rel err {float((y-ref).abs().max()/ref.abs().max()):.2e}", flush=True)
# NODE without node_attr / with
for T in (0, 3):
    N, E, C = 300, 2999, 8
    h, ei, ea, na = O.synthetic_complex(o32, N, E, C, seed=5)
    layer = pkg.EGCL(alg, C, C, C, edge_attr_features=6, node_attr_features=T, aggr="sum")
    p = {k: v.detach().clone() for k, v in layer.named_parameters()}
    layer = layer.to(dev)
    out = layer(h.to(dev), ei.to(dev), ea.to(dev), na.to(dev) if T else None).cpu()
    ref = O.egcl(o32, h, ei, ea, na if T else None, p, aggr="sum")
    print(f"EGCL T={T}: rel err {float((out-ref).abs().max()/ref.abs().max()):.2e}", flush=True)
