"""Deterministic aggregation (CSMPN_FLAG_DETERMINISTIC): the reference trains under
torch.use_deterministic_algorithms(True) (engineer/utils/seed.py:30). With the flag the edge
kernels write their rows to an [E, C, D] table and csmpn_segment_reduce sums them in a fixed
order; parameter gradients come from the per-workgroup slices + fixed-order reduction of the
row-per-lane kernels. Checked here: known answers of the two new entry points, bit-identical
repeated runs (outputs, d/dh, d/d attributes and every parameter gradient), parity against the
float64 oracle in that mode, and the behaviour on shapes the mode does not cover.
"""
import numpy as np
import pytest
import torch

from test_hip_parity import _oracle_egcl_case, dev

pytestmark = pytest.mark.gpu


@pytest.fixture
def det(pkg):
    from csmpn_hip import ops
    ops.set_deterministic(True)
    yield ops
    ops.set_deterministic(None)


def test_segment_reduce_known_answer(pkg):
    from csmpn_hip import ops
    # 3 nodes, 5 rows of 4 floats; add segments {0: rows 0,1; 1: -; 2: rows 2,3,4}, sub segments through
    # an order table {0: row 4; 1: rows 0,2; 2: -}
    rows = torch.tensor([[1., 2., 3., 4.], [10., 20., 30., 40.], [100., 200., 300., 400.],
                         [0.5, 0.25, 0.125, 1.], [7., 7., 7., 7.]], device=dev())
    add_ptr = torch.tensor([0, 2, 2, 5], dtype=torch.int32, device=dev())
    sub_ptr = torch.tensor([0, 1, 3, 3], dtype=torch.int32, device=dev())
    sub_ord = torch.tensor([4, 0, 2], dtype=torch.int32, device=dev())
    out = torch.full((3, 4), 1000.0, device=dev())
    ops.segment_reduce(rows, out, add=(add_ptr, None), sub=(sub_ptr, sub_ord), accumulate=False)
    want = np.array([[11. - 7., 22. - 7., 33. - 7., 44. - 7.],
                     [-101., -202., -303., -404.],
                     [107.5, 207.25, 307.125, 408.]], dtype=np.float32)
    assert np.array_equal(out.cpu().numpy(), want)
    ops.segment_reduce(rows, out, add=(add_ptr, None), accumulate=True)
    want2 = want + np.array([[11., 22., 33., 44.], [0, 0, 0, 0], [107.5, 207.25, 307.125, 408.]], dtype=np.float32)
    assert np.array_equal(out.cpu().numpy(), want2)


def test_source_order(pkg):
    from csmpn_hip import ops
    g = torch.Generator().manual_seed(3)
    N, E = 50, 3000
    ei = torch.randint(0, N, (2, E), generator=g)
    ei[0, 100:900] = 7   # a hub source
    csr = ops.Csr(ei.to(dev()), N)
    rp, order = csr.source_order()
    src, order, rp = csr.src.cpu().numpy(), order.cpu().numpy(), rp.cpu().numpy()
    assert sorted(order.tolist()) == list(range(E))
    s = src[order]
    assert np.all(np.diff(s) >= 0)
    for v in range(N):
        seg = order[rp[v]:rp[v + 1]]
        assert np.all(src[seg] == v)
        assert np.all(np.diff(seg) > 0)        # stable: ascending sorted position inside a segment
    assert rp[0] == 0 and rp[-1] == E


def _run_layer(pkg, C, seed, N=2000, E=20000, metric=(1.0, 1.0, 1.0)):
    alg = pkg.CliffordAlgebra(tuple(metric))
    D = 1 << len(metric)
    torch.manual_seed(seed)
    layer = pkg.EGCL(alg, C, C, C, edge_attr_features=6, node_attr_features=3, aggr="mean").to(dev())
    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, N, (2, E), generator=g)
    ei[1, : E // 4] = 5                          # hub: a quarter of the edges share one target
    ei[0, E // 4: E // 2] = 9                    # and a quarter share one source
    ei[:, -50:] = ei[:, -100:-50]                # duplicates
    ei = ei.to(dev())
    h = torch.randn(N, C, D, generator=g).to(dev())
    ea = torch.randn(E, 6, D, generator=g).to(dev())
    na = torch.randn(N, 3, D, generator=g).to(dev())
    gout = torch.randn(N, C, D, generator=g).to(dev())

    def once():
        hh, e2, n2 = h.clone().requires_grad_(True), ea.clone().requires_grad_(True), na.clone().requires_grad_(True)
        for p in layer.parameters():
            p.grad = None
        y = layer(hh, ei, e2, n2)
        (y * gout).sum().backward()
        return [y.detach().clone(), hh.grad.clone(), e2.grad.clone(), n2.grad.clone()] + \
               [p.grad.clone() for p in layer.parameters()]

    return once


@pytest.mark.parametrize("C,metric,N,E", [(8, (1.0, 1.0, 1.0), 2000, 20000), (16, (1.0, 1.0, 1.0), 2000, 20000),
                                          # the wide parity-lane kernels: the convex-hulls width, and Cl(4,1)
                                          (28, (1.0,) * 5, 400, 4000), (16, (1.0, 1.0, 1.0, 1.0, -1.0), 300, 3000),
                                          # 8 channels of D = 32: routed to the one-group wide kernels in this mode
                                          (8, (1.0, 1.0, 1.0, 1.0, -1.0), 500, 5000),
                                          # round 3: the general row-tile kernels in their deterministic form (one row tile per
                                          # workgroup, mirror slices + fixed-order reduction): the md17 width (Cl(3,0), 32 channels),
                                          # the NBA width (Cl(2,0), 40 channels), an odd narrow width
                                          (32, (1.0, 1.0, 1.0), 600, 6000), (40, (1.0, 1.0), 400, 4000), (5, (1.0, 1.0, 1.0), 700, 5000)])
def test_bit_reproducible(pkg, det, C, metric, N, E):
    once = _run_layer(pkg, C, seed=11, N=N, E=E, metric=metric)
    a = once()
    for _ in range(3):
        b = once()
        for i, (x, y) in enumerate(zip(a, b)):
            assert torch.equal(x, y), f"tensor {i} differs between two runs in deterministic mode"


@pytest.mark.parametrize("C,metric", [(8, (1.0, 1.0, 1.0)), (16, (1.0, 1.0, 1.0)), (32, (1.0, 1.0, 1.0)), (40, (1.0, 1.0))])
def test_matches_atomic_mode(pkg, C, metric):
    from csmpn_hip import ops
    once = _run_layer(pkg, C, seed=12, metric=metric)
    ops.set_deterministic(False)
    try:
        a = once()
        ops.set_deterministic(True)
        b = once()
    finally:
        ops.set_deterministic(None)
    for i, (x, y) in enumerate(zip(a, b)):
        scale = float(y.abs().max()) + 1e-30
        assert float((x - y).abs().max()) <= 2e-5 * scale, f"tensor {i}"


@pytest.mark.parametrize("metric,C,aggr,N,E", [((1.0, 1.0, 1.0), 8, "mean", 300, 2999), ((1.0, 1.0, 1.0), 16, "sum", 300, 2999),
                                               ((1.0,) * 5, 28, "mean", 120, 1001),
                                               ((1.0, 1.0, 1.0), 32, "sum", 200, 1501), ((1.0, 1.0), 40, "mean", 150, 999)])
def test_parity_vs_oracle(pkg, det, metric, C, aggr, N, E):
    _oracle_egcl_case(list(metric), N, E, C, C, aggr, seed=5)


def test_unsupported_shape(pkg, monkeypatch):
    from csmpn_hip import native, ops
    alg = pkg.CliffordAlgebra((1.0, 1.0, 1.0, 1.0, 1.0))
    layer = pkg.EGCL(alg, 12, 12, 12, aggr="mean").to(dev())   # no attributes, 12 channels: general kernels
    g = torch.Generator().manual_seed(1)
    ei = torch.randint(0, 20, (2, 100), generator=g).to(dev())
    h = torch.randn(20, 12, 32, generator=g).to(dev())
    want = layer(h, ei)
    ops.set_deterministic(True)
    try:
        with pytest.raises(native.CsmpnError, match="DETERMINISTIC"):
            layer(h, ei)
    finally:
        ops.set_deterministic(None)
    # inherited from torch.use_deterministic_algorithms: warn once, fall back to the atomic path
    monkeypatch.setattr(torch, "are_deterministic_algorithms_enabled", lambda: True)
    monkeypatch.setattr(ops, "_warned_soft_det", False)
    with pytest.warns(UserWarning, match="deterministic aggregation is not available"):
        got = layer(h, ei)
    assert float((got - want).detach().abs().max()) <= 1e-5 * float(want.detach().abs().max())
