"""Debug driver of the 16-row-tile MFMA-mixing kernels (cemlp_pg.hpp): one EGCL layer (Cl(5,0) / Cl(4,1), wide) forward
+ backward against the oracle on a small complex; prints per-tensor errors and the dispatched kernels (CSMPN_DEBUG=1)."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("clifford-group-equivariant-simplicial-message-passing-networks_amd")
from oracle import ref_path as O
from csmpn_hip import ops, native

def main():
    C = int(os.environ.get("PG_C", "28"))
    N = int(os.environ.get("PG_N", "37"))
    E = int(os.environ.get("PG_E", "203"))
    metric = {"0": (1.0,) * 5, "1": (1.0, 1.0, 1.0, 1.0, -1.0), "3": (1.0, 1.0, 1.0)}[os.environ.get("PG_M", "0")]   # 3: Cl(3,0) (cemlp_pq.hpp)
    aggr = os.environ.get("PG_AGGR", "mean")
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    o = O.Algebra(list(metric))
    gen = torch.Generator().manual_seed(12)
    p = O.init_egcl_params(o, C, C, C, 6, 3, gen=gen, randomize=True)
    layer = pkg.EGCL(pkg.CliffordAlgebra(metric), C, C, C, edge_attr_features=6, node_attr_features=3, aggr=aggr)
    sd = layer.state_dict(); sd.update(p); layer.load_state_dict(sd, strict=True)
    layer = layer.to(dev)
    h, ei, ea, na = O.synthetic_complex(o, N, E, C, seed=5)
    gout = torch.randn(N, C, 1 << len(metric), generator=gen)
    hd = h.to(dev).requires_grad_(True)
    y = layer(hd, ei.to(dev), ea.to(dev), na.to(dev))
    (y * gout.to(dev)).sum().backward()
    torch.cuda.synchronize()
    pr = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    hc = h.clone().requires_grad_(True)
    yo = O.egcl(o, hc, ei, ea, na, pr, aggr=aggr)
    (yo * gout).sum().backward()
    rel = lambda a, b: float((a.cpu() - b).abs().max() / b.abs().max().clamp(min=1e-30))
    print("y", rel(y.detach(), yo.detach()), "gh", rel(hd.grad, hc.grad))
    worst = 0
    for k, prm in layer.named_parameters():
        e = rel(prm.grad, pr[k].grad)
        worst = max(worst, e)
        if e > 1e-4: print("  grad", k, e)
    print("worst param grad", worst)
    if os.environ.get("PG_TIME"):
        Nn, Ee = 10000, 100000
        h, ei, ea, na = (t.to(dev) for t in O.synthetic_complex(o, Nn, Ee, C, seed=6))
        be, spec = ops.HipBackend, layer.spec()
        csr = ops.get_csr(ei, Nn)
        pe, pn = layer.edge_model.flat_params(), layer.node_model.flat_params()
        for name, fn in (("edge_fwd", lambda: be.edge_forward(spec, csr, h, ea, pe)),):
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10): fn()
            torch.cuda.synchronize()
            print(name, (time.perf_counter() - t0) / 10 * 1e3, "ms", native.lib().csmpn_last_kernel().decode())
        agg, st = be.edge_forward(spec, csr, h, ea, pe)
        fn = lambda: be.node_forward(spec, csr.deg, h, agg, na, pn)
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize()
        print("node_fwd", (time.perf_counter() - t0) / 10 * 1e3, "ms", native.lib().csmpn_last_kernel().decode())
main()
