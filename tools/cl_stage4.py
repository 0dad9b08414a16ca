"""Runs the four EGCL stages of a bench workload a few times each (for rocprofv3 --kernel-trace --stats: per-kernel
averages of whichever library CSMPN_LIB points to). Diagnostic aid; not part of the bench contract."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "clifford-group-equivariant-simplicial-message-passing-networks_amd"
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module(PKG)
from csmpn_hip import ops
import bench

def main(workload="S1", reps=20):
    dev = torch.device("cuda:0")
    metric, C, N, E = bench.WORKLOADS[workload]
    (h, ei, ea, na), _ = bench.make_inputs(metric, C, N, E, 0, E, dev)
    torch.manual_seed(0)
    layer = pkg.EGCL(pkg.CliffordAlgebra(metric), C, C, C, edge_attr_features=6, node_attr_features=3, aggr="mean").to(dev)
    be, spec = ops.HipBackend, layer.spec()
    csr = ops.get_csr(ei, N)
    pe, pn = layer.edge_model.flat_params(), layer.node_model.flat_params()
    gout = torch.ones(N, C, 1 << len(metric), device=dev)
    for _ in range(reps):
        agg, se = be.edge_forward(spec, csr, h, ea, pe)
        out, sn = be.node_forward(spec, csr.deg, h, agg, na, pn)
        gh, g_agg, _, _ = be.node_backward(spec, csr.deg, h, agg, na, pn, gout, False, sn)
        be.edge_backward(spec, csr, h, ea, pe, g_agg, gh, False, se)
    torch.cuda.synchronize()

if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "S1")
