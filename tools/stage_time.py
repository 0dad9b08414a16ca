"""Times the four EGCL stages separately (HIP events, median), with / without saved block
inputs. Diagnostic aid for kernel work; not part of the bench contract."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "clifford-group-equivariant-simplicial-message-passing-networks_amd"
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module(PKG)
from csmpn_hip import ops
import bench


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def main(workload="S1"):
    dev = torch.device("cuda:0")
    metric, C, N, E = bench.WORKLOADS[workload]
    (h, ei, ea, na), _ = bench.make_inputs(metric, C, N, E, 0, E, dev)
    torch.manual_seed(0)
    layer = pkg.EGCL(pkg.CliffordAlgebra(metric), C, C, C, edge_attr_features=6, node_attr_features=3, aggr="mean").to(dev)
    be, spec = ops.HipBackend, layer.spec()
    csr = ops.get_csr(ei, N)
    pe, pn = layer.edge_model.flat_params(), layer.node_model.flat_params()
    gout = torch.ones(N, C, 1 << len(metric), device=dev)
    agg, se = be.edge_forward(spec, csr, h, ea, pe)
    out, sn = be.node_forward(spec, csr.deg, h, agg, na, pn)
    gh, g_agg, _, _ = be.node_backward(spec, csr.deg, h, agg, na, pn, gout, False, sn)
    _, se0 = be.edge_forward(spec, csr, h, ea, pe, save=False)
    _, sn0 = be.node_forward(spec, csr.deg, h, agg, na, pn, save=False)
    stages = {
        "edge_fwd(save)": lambda: be.edge_forward(spec, csr, h, ea, pe),
        "edge_fwd(nosave)": lambda: be.edge_forward(spec, csr, h, ea, pe, save=False),
        "node_fwd(save)": lambda: be.node_forward(spec, csr.deg, h, agg, na, pn),
        "node_fwd(nosave)": lambda: be.node_forward(spec, csr.deg, h, agg, na, pn, save=False),
        "node_bwd(saved)": lambda: be.node_backward(spec, csr.deg, h, agg, na, pn, gout, False, sn),
        "node_bwd(recompute)": lambda: be.node_backward(spec, csr.deg, h, agg, na, pn, gout, False, sn0),
        "edge_bwd(saved)": lambda: be.edge_backward(spec, csr, h, ea, pe, g_agg, gh, False, se),
        "edge_bwd(recompute)": lambda: be.edge_backward(spec, csr, h, ea, pe, g_agg, gh, False, se0),
        "empty_launch": lambda: torch.empty(1, device=dev).zero_(),
    }
    for name, fn in stages.items():
        print(f"{name:22s} {timeit(fn) * 1e3:8.1f} us")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "S1")
