"""CPU, world_size 2, gloo: the edge-sharded EGCL (csmpn_hip.sharded) reproduces the
unsharded layer — outputs, d/dh and every parameter gradient — with the oracle injected
as the compute backend (the product backend is the HIP C-ABI; collectives are the same)."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "clifford-group-equivariant-simplicial-message-passing-networks_amd"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, aggr, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pkg = importlib.import_module(PKG)
        from csmpn_hip import sharded
        from oracle import ref_path as O
        from oracle_backend import OracleBackend
        torch.manual_seed(0)
        alg = pkg.CliffordAlgebra((1.0, 1.0, 1.0))
        layer = pkg.EGCL(alg, 4, 5, 4, edge_attr_features=6, node_attr_features=3, aggr=aggr)
        o = O.Algebra([1.0, 1.0, 1.0])
        N, E = 40, 333
        h, ei, ea, na = O.synthetic_complex(o, N, E, 4, seed=1)
        gout = torch.randn(N, 4, 8, generator=torch.Generator().manual_seed(2))
        lo, hi = sharded.shard_bounds(E, world, rank)
        sl = sharded.ShardedEGCL(layer, backend=OracleBackend)
        plan = sl.plan(ei[:, lo:hi].contiguous(), N)
        hh = h.clone().requires_grad_(True)
        eal = ea[lo:hi].clone().requires_grad_(True)
        y = sl(hh, plan, eal, na)
        y.backward(gout)
        # unsharded truth on every rank
        p = {k: v.detach().clone().requires_grad_(True) for k, v in layer.named_parameters()}
        h2 = h.clone().requires_grad_(True)
        ea2 = ea.clone().requires_grad_(True)
        y2 = O.egcl(o, h2, ei, ea2, na, p, aggr=aggr)
        y2.backward(gout)
        res = {"y": (y.detach() - y2.detach()).abs().max().item(),
               "gh": (hh.grad - h2.grad).abs().max().item(),
               "gea": (eal.grad - ea2.grad[lo:hi]).abs().max().item(),
               "deg": int((plan.deg.long() - torch.bincount(ei[1], minlength=N)).abs().max())}
        for k, prm in layer.named_parameters():
            res["g." + k] = (prm.grad - p[k].grad).abs().max().item() / max(p[k].grad.abs().max().item(), 1e-6)
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("aggr", ["mean", "sum"])
def test_sharded_matches_unsharded_world2(aggr):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, aggr, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, res in results:
        assert res["deg"] == 0
        for k, v in res.items():
            assert v < 5e-5, (rank, k, v)


def test_shard_bounds_cover_everything():
    sys.path.insert(0, ROOT)
    importlib.import_module(PKG)
    from csmpn_hip.sharded import shard_bounds
    for E in (0, 1, 7, 100, 1001):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(E, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == E
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


# ----------------------------------------------------------------------------- partitioning B

def _worker_dst(rank, world, port, aggr, q):
    """Destination-partitioned layer: every rank owns N/W nodes and all edges into
    them; all-gather of the output slices forward, reduce-scatter of d/dh backward."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pkg = importlib.import_module(PKG)
        from csmpn_hip import sharded
        from oracle import ref_path as O
        from oracle_backend import OracleBackend
        torch.manual_seed(0)
        alg = pkg.CliffordAlgebra((1.0, 1.0, 1.0))
        layer = pkg.EGCL(alg, 4, 5, 4, edge_attr_features=6, node_attr_features=3, aggr=aggr)
        o = O.Algebra([1.0, 1.0, 1.0])
        N, E = 12 * world + 1, 401           # N does not divide by the world size
        h, ei, ea, na = O.synthetic_complex(o, N, E, 4, seed=1)
        ei[1, :40] = 3                       # a hub: slices are cut by in-degree, not by node count
        gout = torch.randn(N, 4, 8, generator=torch.Generator().manual_seed(2))
        part = sharded.DstPartitionedEGCL(layer, backend=OracleBackend)
        plan = part.plan(ei, N)
        assert plan.cuts[0] == 0 and plan.cuts[-1] == N and sorted(plan.cuts) == plan.cuts
        assert sum(plan.edges_per_rank) == E
        assert max(plan.edges_per_rank) <= E // world + 40 + N   # balanced by incoming edges (the hub is indivisible)
        hh = h.clone().requires_grad_(True)
        eal = ea[plan.edge_ids].clone().requires_grad_(True)
        naa = na.clone().requires_grad_(True)
        y = part(hh, plan, eal, naa)
        y.backward(gout)
        p = {k: v.detach().clone().requires_grad_(True) for k, v in layer.named_parameters()}
        h2 = h.clone().requires_grad_(True)
        ea2 = ea.clone().requires_grad_(True)
        na2 = na.clone().requires_grad_(True)
        y2 = O.egcl(o, h2, ei, ea2, na2, p, aggr=aggr)
        y2.backward(gout)
        # every edge belongs to exactly one rank
        cnt = torch.zeros(E)
        cnt[plan.edge_ids] = 1
        dist.all_reduce(cnt)
        res = {"y": (y.detach() - y2.detach()).abs().max().item(),
               "gh": (hh.grad - h2.grad).abs().max().item(),
               "gea": (eal.grad - ea2.grad[plan.edge_ids]).abs().max().item(),
               "gna": (naa.grad - na2.grad).abs().max().item(),
               "cover": float((cnt - 1).abs().max())}
        for k, prm in layer.named_parameters():
            res["g." + k] = (prm.grad - p[k].grad).abs().max().item() / max(p[k].grad.abs().max().item(), 1e-6)
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("aggr,world", [("mean", 2), ("sum", 3)])
def test_dst_partitioned_matches_unsharded(aggr, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_dst, args=(r, world, port, aggr, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, res in results:
        assert res["cover"] == 0
        for k, v in res.items():
            assert v < 5e-5, (rank, k, v)


def _worker_stack(rank, world, port, q, overlap=False):
    """Three chained layers on one destination partition: output, d/dh, attribute and every
    parameter gradient equal the unsharded chain; N not divisible by the world size, a hub node, an isolated node."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pkg = importlib.import_module(PKG)
        from csmpn_hip import sharded
        from oracle import ref_path as O
        from oracle_backend import OracleBackend
        torch.manual_seed(0)
        alg = pkg.CliffordAlgebra((1.0, 1.0, 1.0))
        layers = [pkg.EGCL(alg, 4, 5, 4, edge_attr_features=6, node_attr_features=3, aggr=a) for a in ("mean", "sum", "mean")]
        o = O.Algebra([1.0, 1.0, 1.0])
        N, E = 10 * world + 3, 333
        h, ei, ea, na = O.synthetic_complex(o, N, E, 4, seed=3)
        ei[1, :60] = N - 2                   # hub near the end
        ei[:, ei[1] == 5] = torch.tensor([[1], [6]])   # node 5 receives nothing (isolated as a target)
        gout = torch.randn(N, 4, 8, generator=torch.Generator().manual_seed(4))
        stack = sharded.DstPartitionedStack(layers, backend=OracleBackend, overlap=overlap)
        plan = stack.plan(ei, N)
        hh = h.clone().requires_grad_(True)
        eal = ea[plan.edge_ids].clone().requires_grad_(True)
        naa = na.clone().requires_grad_(True)
        y = stack(hh, plan, eal, naa)
        y.backward(gout)
        ps = [{k: v.detach().clone().requires_grad_(True) for k, v in l.named_parameters()} for l in layers]
        h2 = h.clone().requires_grad_(True)
        ea2 = ea.clone().requires_grad_(True)
        na2 = na.clone().requires_grad_(True)
        x = h2
        for l, p, a in zip(layers, ps, ("mean", "sum", "mean")):
            x = O.egcl(o, x, ei, ea2, na2, p, aggr=a)
        x.backward(gout)
        res = {"y": (y.detach() - x.detach()).abs().max().item() / x.detach().abs().max().item(),
               "gh": (hh.grad - h2.grad).abs().max().item() / h2.grad.abs().max().item(),
               "gea": (eal.grad - ea2.grad[plan.edge_ids]).abs().max().item() / ea2.grad.abs().max().item(),
               "gna": (naa.grad - na2.grad).abs().max().item() / na2.grad.abs().max().item()}
        for i, (l, p) in enumerate(zip(layers, ps)):
            for k, prm in l.named_parameters():
                res[f"g{i}." + k] = (prm.grad - p[k].grad).abs().max().item() / max(p[k].grad.abs().max().item(), 1e-6)
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,overlap", [(2, False), (3, False), (2, True), (3, True)])
def test_dst_partitioned_stack_matches_unsharded_chain(world, overlap):
    """overlap=True: the collectives of the chain in flight under the local-source edges (two edge launches per stage)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_stack, args=(r, world, port, q, overlap)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, res in results:
        for k, v in res.items():
            assert v < 1e-4, (rank, k, v)


def test_balanced_node_cuts():
    sys.path.insert(0, ROOT)
    importlib.import_module(PKG)
    from csmpn_hip import sharded
    deg = torch.tensor([0, 5, 5, 0, 50, 1, 1, 1, 1, 0])
    for w in (1, 2, 3, 4, 8):
        cuts = sharded.balanced_node_cuts(deg, w)
        assert len(cuts) == w + 1 and cuts[0] == 0 and cuts[-1] == 10 and cuts == sorted(cuts)
    assert sharded.balanced_node_cuts(torch.zeros(7, dtype=torch.int64), 3) == [0, 3, 5, 7]
    spans = [sharded.node_bounds(11, 4, r) for r in range(4)]
    assert spans[0][0] == 0 and spans[-1][1] == 11 and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def _worker_agree(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        importlib.import_module(PKG)
        from csmpn_hip import sharded
        # a target-sorted edge list cut into 3 contiguous shards: only the middle shard straddles N / 2, so only
        # rank 1 could split its edge forward in two - the collective decision must be "nobody splits"
        N, E = 30, 300
        dst = torch.sort(torch.randint(0, N, (E,), generator=torch.Generator().manual_seed(0))).values
        lo, hi = sharded.shard_bounds(E, world, rank)
        mine = dst[lo:hi]
        e1 = int((mine < N // 2).sum())
        local_ok = 0 < e1 < mine.numel()
        res = {"local": bool(local_ok), "all": sharded.agree_all(local_ok), "all_true": sharded.agree_all(True)}
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_split_decision_is_collective_world3():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_agree, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [results[r]["local"] for r in range(3)] == [False, True, False]   # the ranks disagree locally ...
    assert all(results[r]["all"] is False for r in range(3))                 # ... and agree collectively
    assert all(results[r]["all_true"] is True for r in range(3))


# ----------------------------------------------------------------------------- partition B: where the sources live (round 4)

def _batch_like(kind, n_graphs, seed):
    """A collated batch of lifted complexes shaped like the task datasets: convex hulls of 8 points in R^5 (hulls.py) or
    Vietoris-Rips complexes of 21 points in R^3 (md17's molecules). Returns (edge_index, ptr, n_rows)."""
    import numpy as np
    sys.path.insert(0, ROOT)
    importlib.import_module(PKG)
    from csmpn.data import complexes
    rng = np.random.default_rng(seed)
    if kind == "hulls":
        cs = [complexes.hull_complex(rng.normal(size=(8, 5))) for _ in range(n_graphs)]
    else:
        cs = [complexes.rips_complex(rng.normal(size=(21, 3)), dis=1.6) for _ in range(n_graphs)]
    b = complexes.collate(cs)
    return b.edge_index, b.ptr, int(b.node_types.shape[0])


def _worker_locality(rank, world, port, kind, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pkg = importlib.import_module(PKG)
        from csmpn_hip import sharded
        from oracle import ref_path as O
        from oracle_backend import OracleBackend
        ei, ptr, N = _batch_like(kind, 7, seed=5)
        torch.manual_seed(0)
        alg = pkg.CliffordAlgebra((1.0, 1.0, 1.0))
        layers = [pkg.EGCL(alg, 4, 4, 4, edge_attr_features=0, node_attr_features=0, aggr=a) for a in ("mean", "sum")]
        o = O.Algebra([1.0, 1.0, 1.0])
        h = torch.randn(N, 4, 8, generator=torch.Generator().manual_seed(1))
        gout = torch.randn(N, 4, 8, generator=torch.Generator().manual_seed(2))
        stack = sharded.DstPartitionedStack(layers, backend=OracleBackend, overlap=True)
        free = stack.plan(ei, N)                      # cuts by in-degree alone
        plan = stack.plan(ei, N, boundaries=ptr)      # ... moved to graph boundaries
        hh = h.clone().requires_grad_(True)
        y = stack(hh, plan, None, None)
        y.backward(gout)
        ps = [{k: v.detach().clone().requires_grad_(True) for k, v in l.named_parameters()} for l in layers]
        h2 = h.clone().requires_grad_(True)
        x = h2
        for l, p, a in zip(layers, ps, ("mean", "sum")):
            x = O.egcl(o, x, ei, None, None, p, aggr=a)
        x.backward(gout)
        res = {"y": (y.detach() - x.detach()).abs().max().item() / x.detach().abs().max().item(),
               "gh": (hh.grad - h2.grad).abs().max().item() / h2.grad.abs().max().item()}
        for i, (l, p) in enumerate(zip(layers, ps)):
            for k, prm in l.named_parameters():
                res[f"g{i}." + k] = (prm.grad - p[k].grad).abs().max().item() / max(p[k].grad.abs().max().item(), 1e-6)
        q.put((rank, res, free.local_share, plan.local_share, plan.cuts, [int(v) for v in ptr]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["hulls", "md17"])
@pytest.mark.parametrize("world", [2, 3])
def test_graph_aligned_cuts_keep_every_source_local(kind, world):
    """Round-3 review, item 5: on collated batches of small complexes (the hulls and md17 datasets) the nodes of a graph are
    contiguous, so node slices cut AT GRAPH BOUNDARIES own the source of every adjacency they own: local share 1.0 on every
    rank (against < 1 for cuts by in-degree alone whenever a graph straddles a cut) - the overlapped stack then has no
    remote-source launch at all, and its results still equal the unsharded chain."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_locality, args=(r, world, port, kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    free_shares = []
    for rank, res, free_share, share, cuts, ptr in results:
        assert share == 1.0, (rank, share)
        assert all(c in ptr for c in cuts), (cuts, ptr)
        free_shares.append(free_share)
        for k, v in res.items():
            assert v < 1e-4, (rank, k, v)
    assert min(free_shares) < 1.0, free_shares   # the unaligned cuts do split a graph somewhere


def test_locality_order_on_a_geometric_complex():
    """ONE large geometric complex (k-nearest-neighbour adjacency of random points, node ids shuffled): in the given order
    a contiguous node slice owns the source of ~1/W of its adjacencies; after sharded.locality_order (reverse Cuthill-McKee)
    most sources are local - the share of the edge work that hides the collectives of DstPartitionedStack(overlap=True)."""
    import numpy as np
    sys.path.insert(0, ROOT)
    importlib.import_module(PKG)
    from csmpn_hip import sharded
    rng = np.random.default_rng(0)
    N, k, W = 4000, 8, 4
    pts = rng.uniform(size=(N, 3))
    from scipy.spatial import cKDTree
    _, nb = cKDTree(pts).query(pts, k=k + 1)
    src = torch.from_numpy(nb[:, 1:].reshape(-1).astype(np.int64))
    dst = torch.arange(N).repeat_interleave(k)
    shuffle = torch.from_numpy(rng.permutation(N))
    ei = torch.stack([shuffle[src], shuffle[dst]])

    def share(e):
        deg = torch.bincount(e[1], minlength=N)
        cuts = sharded.balanced_node_cuts(deg, W)
        owner = torch.bucketize(e, torch.tensor(cuts[1:-1]), right=True)
        return float((owner[0] == owner[1]).float().mean())

    before = share(ei)
    perm = sharded.locality_order(ei, N)
    assert sorted(perm.tolist()) == list(range(N))
    after = share(perm[ei])
    assert before < 0.35 and after > 0.75, (before, after)
