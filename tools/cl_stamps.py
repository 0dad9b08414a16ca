"""Per-phase shader-clock breakdown of the four EGCL stages on the (row, channel)-per-lane kernels (diagnostic stamps
build only: tools/_bin/libcsmpn_hip_stamps.so, see tools/cl_stamps_build.sh). Shares, not durations: the stamp fences
forbid overlaps across phases. Backward stages: both block launches add into the same slots."""
import ctypes, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "clifford-group-equivariant-simplicial-message-passing-networks_amd"
os.environ["CSMPN_LIB"] = os.path.join(ROOT, "tools", "_bin", "libcsmpn_hip_stamps.so")
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module(PKG)
from csmpn_hip import native, ops
import bench

FWD = ["setup(stage tables)", "loads(wait)", "W1 mix b0", "silu b0", "linR/L b0", "norm b0", "gp b0", "layernorm b0",
       "W1 mix b1", "silu b1", "linR/L b1", "norm b1", "gp b1", "layernorm b1", "store/scatter"]
BWD = ["setup(stage tables)", "loads(wait)", "r:W1 mix", "r:silu", "r:linR/L", "r:norm", "r:gp", "r:layernorm",
       "b:layernorm", "b:linL^T", "b:gp", "b:norm", "b:linR^T", "b:wgradRL(mfma)", "b:silu", "b:wgradW1(mfma)",
       "b:W1^T+store/scatter", "end-of-kernel sums"]

CM_FWD = ["setup(stage tables)", "loads(wait)", "W1 mix b0", "silu b0", "linR/L b0", "norm+gp b0", "layernorm b0", "", "W1 mix b1",
          "silu b1", "linR/L b1", "norm+gp b1", "layernorm b1", "", "issue next + store/scatter"]
# channel-MFMA backward, parked form (cemlp_cmb.hpp, round 4)
CMB_BWD = ["setup(stage tables)", "input loads + W1 mix", "r:silu, z->P, gout->W, linR/L, norm+gp", "b:layernorm (ggp->W)",
           "b:wgrad WL (slots)", "b:gp+norm per channel", "b:WL^T, WR^T, wgrad WR, gz->W", "input again + W1 mix, x->P",
           "b:silu in place (W)", "b:wgrad W1 (slots)", "b:W1^T + store/scatter", "", "", "", "", "", "", "end-of-block sums", "tile loop, waves 0-3 (x2: per wave)", "tile loop, waves 4-7 (x2: per wave)"]
# pair backward of the 32-channel layers (cemlp_cmp.hpp, round 4)
CMP_BWD = ["setup(stage tables)", "input -> slots, W1 mix", "r:silu, z->slots, linR/L, gout load, norm+gp, row sums", "b:layernorm, ggp->slots",
           "b:wgrad WL, WL^T", "b:gp+norm per channel", "b:gR->slots, input again, wgrad WR, WR^T", "input -> slots, W1 mix (y again)",
           "b:silu, gy->slots, wgrad W1, W1^T", "", "next tile + store/scatter", "waiting at the pair's rendezvous (all phases)", "", "", "", "", "", "end-of-block sums"]


def main(workload="S1", family="cl", nodes=0, edges=0):
    dev = torch.device("cuda:0")
    metric, C, N, E = bench.WORKLOADS[workload]
    if nodes and edges:   # the workload's layer shape at another size (e.g. an md17 batch: 940 11266)
        N, E = nodes, edges
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
    gout = torch.ones(N, C, 1 << len(metric), device=dev)
    agg, se = be.edge_forward(spec, csr, h, ea, pe)
    out, sn = be.node_forward(spec, csr.deg, h, agg, na, pn)
    gh, g_agg, _, _ = be.node_backward(spec, csr.deg, h, agg, na, pn, gout, False, sn)
    stages = {
        "edge_fwd": lambda: be.edge_forward(spec, csr, h, ea, pe),
        "node_fwd": lambda: be.node_forward(spec, csr.deg, h, agg, na, pn),
        "node_bwd": lambda: be.node_backward(spec, csr.deg, h, agg, na, pn, gout, False, sn),
        "edge_bwd": lambda: be.edge_backward(spec, csr, h, ea, pe, g_agg, gh, False, se),
    }
    for name, fn in stages.items():
        fn(); torch.cuda.synchronize()
        st.zero_(); torch.cuda.synchronize()
        lib.csmpn_debug_set_stamps(st.data_ptr())
        fn(); torch.cuda.synchronize()
        lib.csmpn_debug_set_stamps(None)
        v = st.cpu().tolist()
        waves, tot = v[24], sum(v[:24])
        print(f"== {name}: {waves} waves, {tot / max(waves,1) / 1e3:.1f} kcycles per wave")
        names = (CM_FWD if family == "cm" else FWD) if name.endswith("fwd") else ((CMP_BWD if C == 32 else CMB_BWD) if family == "cm" else BWD)
        for i, nm in enumerate(names):
            if v[i] and nm:
                print(f"   {nm:24s} {v[i] / waves / 1e3:9.1f} kcyc  {100.0 * v[i] / tot:5.1f}%")

if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "S1", sys.argv[2] if len(sys.argv) > 2 else "cl",
         int(sys.argv[3]) if len(sys.argv) > 4 else 0, int(sys.argv[4]) if len(sys.argv) > 4 else 0)
