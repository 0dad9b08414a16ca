"""GPU parity at BASELINE size against something that is NOT the HIP path (round-3 review, item 4).

The lane kernels are persistent: at 100 k / 1 M edges a wave walks many row tiles, prefetches tile t + 1 / t + 2 and (backward)
hands d/d(block input) rows from block to block through L2. The oracle-compared cases elsewhere stay under 3 k edges, i.e. one
tile per wave. Here the whole EGCL layer (csmpn/models/cegnn_utils.py:254-284) runs at the full S1 / S3 / M32 / S2 sizes and
y, d/dh and EVERY parameter gradient are held to max(1e-5, 4 x yardstick) - tensor-level and element-wise (`check`) - against
the C++ twin (oracle/cpu_twin, pinned to the reference's fixtures by tests/test_cpu_twin.py):

    truth     = the twin's float64 build (libcsmpn_cpu64.so; agrees with the reference's own float64 run to 1e-10),
    yardstick = the twin's float32 build against that truth (what float32 arithmetic costs at this size),

on the atomic path and under CSMPN_FLAG_DETERMINISTIC. The twin is built by __graft_entry__.build() (make -C oracle) and
travels as oracle/_build/*.so; nothing here reads /root/reference.
"""
import importlib
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import ref_path as O
from test_hip_parity import check, deterministic_aggregation, dev, relmax

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

#            tag    metric                      C   N        E          aggr    neg_scale
WORKLOADS = {
    "S1": ((1.0, 1.0, 1.0), 8, 10_000, 100_000, "mean", None),
    "S3": ((1.0, 1.0, 1.0, 1.0, -1.0), 8, 10_000, 100_000, "mean", 0.02),
    "M32": ((1.0, 1.0, 1.0), 32, 10_000, 100_000, "sum", None),
    # BASELINE config 5 on the RAW generator inputs (no taming of the negative-signature blades: null-cone norms, the
    # reference's own float32 run is 1e-3 .. 1e-1 off its float64 run on some tensors - reported, bounded by that yardstick)
    "S3raw": ((1.0, 1.0, 1.0, 1.0, -1.0), 8, 10_000, 100_000, "mean", None),
    "S2": ((1.0, 1.0, 1.0), 16, 100_000, 1_000_000, "mean", None),
    # the convex-hulls width (hulls_cssmpnn.py:16-28: Cl(5,0), 28 channels) at S1's size: the wide parity-lane kernels
    "H28": ((1.0, 1.0, 1.0, 1.0, 1.0), 28, 10_000, 100_000, "mean", None),
}
# workloads with a fixture of the imported reference's own float64 / float32 run at full size
# (tests/golden/make_fullsize_golden.py; the reference's dense formulation needs ~35 GB for S3, more for S2 / H28)
REFERENCE_FIXTURES = ("S1", "M32", "S3", "S3raw")
_cache = {}


def _twin():
    from oracle import cpu_twin
    if not (os.path.exists(cpu_twin.LIB_PATH) and os.path.exists(cpu_twin.LIB64_PATH)):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    return cpu_twin


def _case(tag):
    """inputs, parameters and the twin's float64 / float32 results of one workload (computed once per session)"""
    if tag in _cache:
        return _cache[tag]
    metric, C, N, E, aggr, neg_scale = WORKLOADS[tag]
    o32 = O.Algebra(metric, torch.float32)
    h, ei, ea, na = O.synthetic_complex(o32, N, E, C, seed=11)
    if neg_scale is not None:   # stay off the null cone of an indefinite metric (as test_egcl_cl41_well_conditioned)
        neg_bits = sum(1 << i for i, m in enumerate(metric) if m < 0)
        mask = torch.from_numpy(((np.asarray(o32.t.index_to_bitmap) & neg_bits) != 0).astype(np.float32))
        h = h * (1.0 - mask + neg_scale * mask)
    gen = torch.Generator().manual_seed(12)
    p = O.init_egcl_params(o32, C, C, C, 6, 3, gen=gen, randomize=True)
    gout = torch.randn(N, C, o32.D if hasattr(o32, "D") else 2 ** len(metric), generator=gen)
    tw = _twin()
    args = (np.asarray(metric, np.float32), {k: v.numpy() for k, v in p.items()}, h.numpy(), ei.numpy(), ea.numpy(), na.numpy())
    t64 = tw.egcl_layer(*args, aggr=aggr, gout=gout.numpy(), real64=True)
    t32 = tw.egcl_layer(*args, aggr=aggr, gout=gout.numpy(), real64=False)
    _cache[tag] = (metric, C, aggr, h, ei, ea, na, p, gout, t64, t32)
    return _cache[tag]


def _hip(tag, deterministic):
    metric, C, aggr, h, ei, ea, na, p, gout, _, _ = _case(tag)
    pkg = importlib.import_module("clifford-group-equivariant-simplicial-message-passing-networks_amd")
    layer = pkg.EGCL(pkg.CliffordAlgebra(tuple(metric)), C, C, C, edge_attr_features=6, node_attr_features=3, aggr=aggr)
    sd = layer.state_dict()
    sd.update(p)
    layer.load_state_dict(sd, strict=True)
    layer = layer.to(dev())
    hd = h.to(dev()).requires_grad_(True)

    def run():
        y = layer(hd, ei.to(dev()), ea.to(dev()), na.to(dev()))
        (y * gout.to(dev())).sum().backward()
        torch.cuda.synchronize()
        return y

    if deterministic:
        with deterministic_aggregation():
            y = run()
    else:
        y = run()
    return y.detach().cpu().numpy(), hd.grad.cpu().numpy(), {k: v.grad.cpu().numpy() for k, v in layer.named_parameters()}


@pytest.mark.parametrize("deterministic", [True, False], ids=["deterministic", "atomic"])
@pytest.mark.parametrize("tag", [t for t in WORKLOADS if t != "S3raw"])
def test_full_size_layer_against_cpu_twin(tag, deterministic):
    *_, t64, t32 = _case(tag)
    y, gh, grads = _hip(tag, deterministic)
    slack = 4.0
    errs = {"y": check("y", y, t64["out"], t32["out"], slack=slack), "gh": check("gh", gh, t64["gh"], t32["gh"], slack=slack)}
    assert set(grads) == set(t64["grads"])
    for k, g in grads.items():
        errs[k] = check("g." + k, g, t64["grads"][k], t32["grads"][k], slack=slack)
    yard = max([relmax(t32["out"], t64["out"]), relmax(t32["gh"], t64["gh"])] +
               [relmax(t32["grads"][k], t64["grads"][k]) for k in grads])
    print(f"{tag} {'det' if deterministic else 'atomic'}: worst HIP err {max(errs.values()):.2e}, float32 yardstick {yard:.2e}")


def _input_checksums(h, ei, ea, na, p, gout):   # = tests/golden/make_fullsize_golden.py::checksums
    return np.asarray([h.double().sum().item(), h.double().abs().sum().item(), float(ei.sum().item()),
                       float((ei[0] * 7 + ei[1]).remainder(1000003).sum().item()), ea.double().sum().item(),
                       na.double().sum().item(), gout.double().sum().item(),
                       sum(v.double().abs().sum().item() for v in p.values())], dtype=np.float64)


@pytest.mark.parametrize("deterministic", [True, False], ids=["deterministic", "atomic"])
@pytest.mark.parametrize("tag", REFERENCE_FIXTURES)
def test_full_size_layer_against_reference_fixture(tag, deterministic, golden_dir):
    """The same full-size runs against the IMPORTED REFERENCE (round-4 review, item 5): its own EGCL.forward + autograd in
    float64 (truth) and float32 (yardstick) on these very inputs, recorded in the build container by
    tests/golden/make_fullsize_golden.py - y and d/dh on a 1-in-16 node subsample, every parameter gradient whole. A defect
    shared by the HIP kernels and this repository's C++ twin (index width, degree handling, tile bookkeeping at scale)
    would pass the twin test above and fail here."""
    path = os.path.join(golden_dir, f"fullsize_{tag}.npz")
    if not os.path.exists(path):
        pytest.skip(f"no reference fixture for {tag}")
    g = np.load(path)
    metric, C, aggr, h, ei, ea, na, p, gout, _, _ = _case(tag)
    # the inputs are regenerated from the seed, not stored: a drifting generator must not compare different problems
    np.testing.assert_allclose(_input_checksums(h, ei, ea, na, p, gout), g["checksums"], rtol=1e-12, atol=0)
    st = int(g["node_stride"])
    y, gh, grads = _hip(tag, deterministic)
    # raw Cl(4,1) inputs sit on the null cone: the factor the golden Cl(4,1) cases use (test_hip_parity.py: 10 atomic, 6
    # deterministic) against the reference's own float32 error; every other workload: 4
    slack = (6.0 if deterministic else 10.0) if tag == "S3raw" else 4.0
    errs, report = {}, {}
    # tensor-level scale = the FULL tensor's maximum (stored), not the subsample's
    for name, arr in (("y", y), ("gh", gh)):
        truth, ref32 = g[f"f64/{name}"], g[f"f32/{name}"]
        scale = float(g[f"f64/{name}_absmax"])
        yard = float(np.abs(ref32 - truth).max() / scale)
        err = float(np.abs(arr[::st] - truth).max() / scale)
        bound = max(1e-5, slack * yard)
        assert err <= bound, f"{tag} {name}: HIP {err:.2e} vs reference float64, bound {bound:.2e} (reference float32: {yard:.2e})"
        errs[name] = err
        report[name] = (err, yard)
    keys = [k[len("f64/g/"):] for k in g.files if k.startswith("f64/g/")]
    assert set(keys) == set(grads)
    for k in keys:
        errs[k] = check("g." + k, grads[k], g[f"f64/g/{k}"], g[f"f32/g/{k}"], slack=slack)
        report[k] = (errs[k], relmax(g[f"f32/g/{k}"], g[f"f64/g/{k}"]))
    print(f"{tag} {'det' if deterministic else 'atomic'} vs reference fixture: worst HIP err {max(errs.values()):.2e}")
    # per-tensor table (HIP error, the reference's own float32 error; both against the reference's float64 run) for BASELINE.md
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        import json
        with open(os.path.join(out_dir, f"fullsize_errors_{tag}_{'det' if deterministic else 'atomic'}.json"), "w") as f:
            json.dump({k: {"hip": v[0], "reference_float32": v[1]} for k, v in report.items()}, f, indent=1)
