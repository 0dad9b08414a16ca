import importlib, sys, os, torch
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("clifford-group-equivariant-simplicial-message-passing-networks_amd")
from csmpn_hip import ops
dev = torch.device('cuda:0')
alg = pkg.CliffordAlgebra((1.,1.,1.))
torch.manual_seed(0)
N, C = 40, 8
layer = pkg.EGCL(alg, C, C, C, edge_attr_features=0, node_attr_features=0, aggr="sum").to(dev)
h = torch.randn(N, C, 8, device=dev); agg = torch.randn(N, C, 8, device=dev)
deg = torch.ones(N, dtype=torch.int32, device=dev)
out, _ = ops.HipBackend.node_forward(layer.spec(), deg, h, agg, None, layer.node_model.flat_params(), save=False)
ref = h + layer.node_model(torch.cat([h, agg], 1))
print("node vs plain: max abs diff", float((out - ref).abs().max()))
# which part is wrong: zero agg / zero h
out0, _ = ops.HipBackend.node_forward(layer.spec(), deg, h, torch.zeros_like(agg), None, layer.node_model.flat_params(), save=False)
ref0 = h + layer.node_model(torch.cat([h, torch.zeros_like(agg)], 1))
print("agg=0: diff", float((out0 - ref0).abs().max()))
outh, _ = ops.HipBackend.node_forward(layer.spec(), deg, torch.zeros_like(h), agg, None, layer.node_model.flat_params(), save=False)
refh = layer.node_model(torch.cat([torch.zeros_like(h), agg], 1))
print("h=0: diff", float((outh - refh).abs().max()))
# permuted agg rows?
for shift in (1, 2, 16):
    refs = h + layer.node_model(torch.cat([h, torch.roll(agg, shift, 0)], 1))
    print("shift", shift, float((out - refs).abs().max()))
