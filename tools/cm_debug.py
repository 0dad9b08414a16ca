"""Debug aid: edge / node forward of a small 16-channel Cl(3,0) layer in this process (channel-MFMA kernels) against a
child process with CSMPN_NO_CM=1 (row-per-lane kernels); prints the rows that differ."""
import importlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "clifford-group-equivariant-simplicial-message-passing-networks_amd"
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module(PKG)
from csmpn_hip import ops
import bench

def run(N, E, C=16):
    dev = torch.device("cuda:0")
    metric = (1.0, 1.0, 1.0)
    (h, ei, ea, na), _ = bench.make_inputs(metric, C, N, E, 0, E, dev)
    torch.manual_seed(0)
    layer = pkg.EGCL(pkg.CliffordAlgebra(metric), C, C, C, edge_attr_features=6, node_attr_features=3, aggr="mean").to(dev)
    be, spec = ops.HipBackend, layer.spec()
    csr = ops.get_csr(ei, N)
    pe, pn = layer.edge_model.flat_params(), layer.node_model.flat_params()
    agg, se = be.edge_forward(spec, csr, h, ea, pe)
    out, sn = be.node_forward(spec, csr.deg, h, agg, na, pn)
    torch.cuda.synchronize()
    return agg.cpu(), out.cpu(), csr

if __name__ == "__main__":
    N, E = int(sys.argv[1]), int(sys.argv[2])
    agg, out, csr = run(N, E)
    if os.environ.get("CSMPN_NO_CM"):
        torch.save((agg, out), "/tmp/cm_dbg_ref.pt")
        sys.exit(0)
    subprocess.run([sys.executable, __file__, str(N), str(E)], env=dict(os.environ, CSMPN_NO_CM="1"), check=True)
    ragg, rout = torch.load("/tmp/cm_dbg_ref.pt")
    da = (agg - ragg).abs().flatten(1).max(1).values
    do = (out - rout).abs().flatten(1).max(1).values
    print("agg rows differing:", torch.nonzero(da > 1e-4).flatten().tolist()[:40], "max", float(da.max()))
    print("out rows differing:", torch.nonzero(do > 1e-4).flatten().tolist()[:40], "max", float(do.max()))
    print("dst of sorted edges:", csr.dst_sorted.tolist()[:40] if hasattr(csr, "dst_sorted") else "n/a")
