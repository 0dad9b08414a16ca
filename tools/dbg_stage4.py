import importlib, sys, os, torch
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("clifford-group-equivariant-simplicial-message-passing-networks_amd")
from csmpn_hip import ops, native
dev = torch.device('cuda:0')
alg = pkg.CliffordAlgebra((1.,1.,1.))
torch.manual_seed(0)
N, C = 40, 8
layer = pkg.EGCL(alg, C, C, C, edge_attr_features=0, node_attr_features=0, aggr="sum", residual=False).to(dev)
h = torch.randn(N, C, 8, device=dev); agg = torch.randn(N, C, 8, device=dev)
cat = torch.cat([h, agg], 1).contiguous()
deg = torch.ones(N, dtype=torch.int32, device=dev)
nd = layer.spec().node
pn = layer.node_model.flat_params()
nd.bind(pn)
ref = layer.node_model(cat)
def call(hh, ch, ag, ach):
    out = torch.empty(N, 8, 8, device=dev)
    ws = nd.workspace(dev)
    native.check(native.lib().csmpn_egcl_node_forward(nd.metric_arr, nd.n, nd.params, nd.nblk, hh.data_ptr(), ch, ag.data_ptr(), ach,
        None, 0, deg.data_ptr(), 0, 0, N, out.data_ptr(), None, ws.data_ptr(), ws.numel(), 0, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return out
for (ch, ach) in ((16, 0), (8, 8), (12, 4), (4, 12)):
    a = cat[:, :ch].contiguous(); b = cat[:, ch:].contiguous() if ach else cat
    out = call(a, ch, b, ach)
    print(f"split {ch}+{ach}: max diff {float((out - ref).abs().max()):.3e}")
