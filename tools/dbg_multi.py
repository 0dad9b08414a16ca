import importlib, sys, os, torch
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("clifford-group-equivariant-simplicial-message-passing-networks_amd")
from oracle import ref_path as O
dev = torch.device('cuda:0')
def run(metric, C, hidden, N=60, E=300):
    o32 = O.Algebra(metric)
    h, ei, ea, na = O.synthetic_complex(o32, N, E, C, seed=5)
    layer = pkg.EGCL(pkg.CliffordAlgebra(tuple(metric)), C, hidden, C, edge_attr_features=6, node_attr_features=3, aggr="sum").to(dev)
    hd = h.to(dev).requires_grad_(True)
    y = layer(hd, ei.to(dev), ea.to(dev), na.to(dev))
    torch.cuda.synchronize(); print("fwd ok", metric, C, float(y.abs().max()))
    y.sum().backward()
    torch.cuda.synchronize(); print("bwd ok")
run([1.,1.], 40, 40)
run([1.,1.,1.,1.,1.], 28, 28)
