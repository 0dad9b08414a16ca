"""Dispatch / timing check of the row-per-lane kernels on the S1 shape (run on the GPU box)."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("clifford-group-equivariant-simplicial-message-passing-networks_amd")
from oracle import ref_path as O
from csmpn_hip import ops
dev = torch.device("cuda:0")
N, E, C = int(os.environ.get("N", 10000)), int(os.environ.get("E", 100000)), int(os.environ.get("C", 8))
METRIC = {"cl30": [1.0, 1.0, 1.0], "cl50": [1.0] * 5, "cl41": [1.0, 1.0, 1.0, 1.0, -1.0]}[os.environ.get("ALG", "cl30")]
torch.manual_seed(0)
layer = pkg.EGCL(pkg.CliffordAlgebra(tuple(METRIC)), C, C, C, edge_attr_features=6, node_attr_features=3, aggr="mean").to(dev)
h, ei, ea, na = (t.to(dev) for t in O.synthetic_complex(O.Algebra(METRIC), N, E, C, seed=0))
be, spec = ops.HipBackend, layer.spec()
csr = ops.get_csr(ei, N)
pe, pn = layer.edge_model.flat_params(), layer.node_model.flat_params()
gout = torch.randn(N, C, 1 << len(METRIC), device=dev)
agg, st_e = be.edge_forward(spec, csr, h, ea, pe)
out, st_n = be.node_forward(spec, csr.deg, h, agg, na, pn)
gh, g_agg, _, gn = be.node_backward(spec, csr.deg, h, agg, na, pn, gout, False, st_n)
_, ge = be.edge_backward(spec, csr, h, ea, pe, g_agg, gh, False, st_e)
torch.cuda.synchronize()
print("saved:", None if st_e[1] is None else st_e[1].shape, None if st_n[1] is None else st_n[1].shape)
stages = {
    "edge_fwd": lambda: be.edge_forward(spec, csr, h, ea, pe),
    "node_fwd": lambda: be.node_forward(spec, csr.deg, h, agg, na, pn),
    "node_bwd": lambda: be.node_backward(spec, csr.deg, h, agg, na, pn, gout, False, st_n),
    "edge_bwd": lambda: be.edge_backward(spec, csr, h, ea, pe, g_agg, gh, False, st_e),
}
os.environ.pop("CSMPN_DEBUG", None)
for name, fn in stages.items():
    fn(); torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    print(f"{name}: median {ts[len(ts)//2]*1e3:.1f} us  min {ts[0]*1e3:.1f} us")
