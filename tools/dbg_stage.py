import importlib, sys, os, torch
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("clifford-group-equivariant-simplicial-message-passing-networks_amd")
from oracle import ref_path as O
from csmpn_hip import ops
dev = torch.device('cuda:0')
def run(metric, C, N, E):
    o32 = O.Algebra(metric)
    h, ei, ea, na = O.synthetic_complex(o32, N, E, C, seed=5)
    torch.manual_seed(0)
    layer = pkg.EGCL(pkg.CliffordAlgebra(tuple(metric)), C, C, C, edge_attr_features=6, node_attr_features=3, aggr="sum")
    p = {k: v.detach().clone() for k, v in layer.named_parameters()}
    layer = layer.to(dev)
    spec = layer.spec(); be = ops.HipBackend
    csr = ops.get_csr(ei.to(dev), N)
    pe, pn = layer.edge_model.flat_params(), layer.node_model.flat_params()
    for save in (False, True):
        agg, st = be.edge_forward(spec, csr, h.to(dev), ea.to(dev), pe, save=save)
        torch.cuda.synchronize()
        # oracle agg
        x = torch.cat([h[ei[1]] - h[ei[0]], ea], 1)
        msg = O.cemlp(o32, x, p, "edge_model.")
        ref = O.scatter_rows(msg.reshape(E, -1), ei[1], N, "sum").reshape(N, C, -1)
        err = (agg.cpu() - ref).abs().max() / ref.abs().max()
        print(f"metric={metric} C={C} N={N} E={E} save={save}: agg rel err {err:.2e}", flush=True)
        out, sn = be.node_forward(spec, csr.deg, h.to(dev), agg, na.to(dev), pn, save=save)
        torch.cuda.synchronize()
        ref_out = O.egcl(o32, h, ei, ea, na, p, aggr="sum")
        print("   out rel err", float((out.cpu() - ref_out).abs().max() / ref_out.abs().max()), flush=True)
run([1.,1.,1.], 8, 12, 40)
run([1.,1.,1.], 8, 300, 2999)
run([1.,1.,1.], 16, 300, 2999)
