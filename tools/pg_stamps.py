"""Per-phase shader-clock breakdown of the pg kernels (diagnostic stamps build: tools/pg_stamps_build.sh). Shares, not
durations: the stamp fences forbid overlaps across phases; a phase is charged with the waits it executes."""
import ctypes, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "clifford-group-equivariant-simplicial-message-passing-networks_amd"
os.environ["CSMPN_LIB"] = os.path.join(ROOT, "tools", "_bin", "libcsmpn_hip_stamps.so")
os.environ["CSMPN_PG"] = "1"
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module(PKG)
from csmpn_hip import native, ops
import bench

FWD = ["prologue", "tile bookkeeping + input staging (+barrier)", "b0 MIX W1", "barrier waits (b0)", "b0 ROW gates", "b0 MIX R/L", "b0 ROW norm+gp",
       "b0 layer norm + out", "b1 MIX W1 (+in1 copy)", "barrier waits (b1)", "b1 ROW gates", "b1 MIX R/L", "b1 ROW norm+gp", "b1 layer norm + out",
       "rows out / scatter"]


BWD = ["prologue", "bookkeeping + d/d(out) staging (+barrier)", "ROW z, layer-norm backward -> ggp", "MIX WL^T + d/dWL", "ROW gp + norm backward",
       "MIX WR^T + d/dWR", "input staging + ROW silu backward", "MIX W1^T + d/dW1", "rows out / scatter", "slice store"]


def main(workload="H28", which="edge_fwd"):
    dev = torch.device("cuda:0")
    metric, C, N, E = bench.WORKLOADS[workload]
    (h, ei, ea, na), _ = bench.make_inputs(metric, C, N, E, 0, E, dev)
    torch.manual_seed(0)
    layer = pkg.EGCL(pkg.CliffordAlgebra(metric), C, C, C, edge_attr_features=6, node_attr_features=3, aggr="mean").to(dev)
    lib = native.lib()
    lib.csmpn_debug_set_stamps.argtypes = [ctypes.c_void_p]
    lib.csmpn_debug_set_stamps.restype = None
    st = torch.zeros(25, dtype=torch.int64, device=dev)
    be, spec = ops.HipBackend, layer.spec()
    csr = ops.get_csr(ei, N)
    pe, pn = layer.edge_model.flat_params(), layer.node_model.flat_params()
    agg, se = be.edge_forward(spec, csr, h, ea, pe)
    out, sn = be.node_forward(spec, csr.deg, h, agg, na, pn)
    gout = torch.ones_like(out)
    gh, g_agg, _, _ = be.node_backward(spec, csr.deg, h, agg, na, pn, gout, False, sn)
    stages = {"edge_fwd": lambda: be.edge_forward(spec, csr, h, ea, pe), "node_fwd": lambda: be.node_forward(spec, csr.deg, h, agg, na, pn),
              "node_bwd": lambda: be.node_backward(spec, csr.deg, h, agg, na, pn, gout, False, sn),
              "edge_bwd": lambda: be.edge_backward(spec, csr, h, ea, pe, g_agg, gh, False, se)}
    for name, fn in stages.items():
        fn(); torch.cuda.synchronize()
        st.zero_(); torch.cuda.synchronize()
        lib.csmpn_debug_set_stamps(st.data_ptr())
        fn(); torch.cuda.synchronize()
        lib.csmpn_debug_set_stamps(None)
        v = st.cpu().tolist()
        waves, tot = v[24], sum(v[:24])
        print(f"== {name}: {waves} waves, {tot / max(waves,1) / 1e3:.1f} kcycles per wave  [{lib.csmpn_last_kernel().decode()}]")
        for i, nm in enumerate(FWD if name.endswith("fwd") else BWD):
            if v[i]:
                print(f"   {nm:48s} {v[i] / waves / 1e3:9.1f} kcyc  {100.0 * v[i] / tot:5.1f}%")


if __name__ == "__main__":
    main(*sys.argv[1:])
