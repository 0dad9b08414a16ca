"""Tile-quantisation check (round-2 review, item 2): S1's edge stages at 98 304 edges (a whole number of tiles per wave) and at
100 000 edges; prints the HIP-event time per launch and the ratio against the edge ratio."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "clifford-group-equivariant-simplicial-message-passing-networks_amd"
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module(PKG)
from csmpn_hip import ops
import bench

def stage_times(E, reps=200):
    dev = torch.device("cuda:0")
    metric, C, N, _ = bench.WORKLOADS["S1"]
    (h, ei, ea, na), _ = bench.make_inputs(metric, C, N, E, 0, E, dev)
    torch.manual_seed(0)
    layer = pkg.EGCL(pkg.CliffordAlgebra(metric), C, C, C, edge_attr_features=6, node_attr_features=3, aggr="mean").to(dev)
    be, spec = ops.HipBackend, layer.spec()
    csr = ops.get_csr(ei, N)
    pe, pn = layer.edge_model.flat_params(), layer.node_model.flat_params()
    gout = torch.ones(N, C, 8, device=dev)
    agg, se = be.edge_forward(spec, csr, h, ea, pe)
    out, sn = be.node_forward(spec, csr.deg, h, agg, na, pn)
    gh, g_agg, _, _ = be.node_backward(spec, csr.deg, h, agg, na, pn, gout, False, sn)
    res = {}
    for name, fn in (("edge_fwd", lambda: be.edge_forward(spec, csr, h, ea, pe)),
                     ("edge_bwd", lambda: be.edge_backward(spec, csr, h, ea, pe, g_agg, gh, False, se))):
        for _ in range(10):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) * 1e3 / reps
    return res

if __name__ == "__main__":
    a, b = stage_times(98_304), stage_times(100_000)
    for k in a:
        print(f"{k}: {a[k]:.1f} us at 98 304 edges, {b[k]:.1f} us at 100 000 edges: x{b[k] / a[k]:.3f} for x{100000 / 98304:.3f} the edges")
