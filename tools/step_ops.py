"""Diagnostic: which aten ops / kernels one eager EGCL training step (bench.py's `step`) launches."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("clifford-group-equivariant-simplicial-message-passing-networks_amd")
import bench
dev = torch.device("cuda:0")
metric, C, N, E = bench.WORKLOADS["S1"]
(h, ei, ea, na), _ = bench.make_inputs(metric, C, N, E, 0, E, dev)
torch.manual_seed(0)
layer = pkg.EGCL(pkg.CliffordAlgebra(metric), C, C, C, edge_attr_features=6, node_attr_features=3, aggr="mean").to(dev)
params = list(layer.parameters())
h.requires_grad_(True)
gout = torch.ones(N, C, 1 << len(metric), device=dev)
def step():
    y = layer(h, ei, ea, na)
    return torch.autograd.grad(y, [h] + params, gout)
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
for ev in prof.events():
    if ev.device_type.name == "CUDA" or "fill" in ev.name.lower() or "copy" in ev.name.lower() or "zero" in ev.name.lower():
        print(f"{ev.device_type.name:5s} {ev.name[:90]:90s} {getattr(ev, 'input_shapes', '')}")
