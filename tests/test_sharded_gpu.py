"""GPU: the edge-sharded layer with the HIP backend (two processes sharing cuda:0, gloo as the
transport because RCCL refuses two ranks on one device) equals the unsharded HIP layer."""
import importlib
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "clifford-group-equivariant-simplicial-message-passing-networks_amd"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, metric=(1.0, 1.0, 1.0), C=8):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pkg = importlib.import_module(PKG)
        from csmpn_hip import sharded
        from oracle import ref_path as O
        dev = torch.device("cuda:0")
        torch.manual_seed(0)
        alg = pkg.CliffordAlgebra(tuple(metric))
        layer = pkg.EGCL(alg, C, C, C, edge_attr_features=6, node_attr_features=3, aggr="mean").to(dev)
        N, E = 500, 7001
        h, ei, ea, na = (t.to(dev) for t in O.synthetic_complex(O.Algebra(list(metric)), N, E, C, seed=1))
        gout = torch.randn(N, C, 1 << len(metric), generator=torch.Generator().manual_seed(2)).to(dev)
        lo, hi = sharded.shard_bounds(E, world, rank)
        sl = sharded.ShardedEGCL(layer)
        plan = sl.plan(ei[:, lo:hi].contiguous(), N)
        hh = h.clone().requires_grad_(True)
        y = sl(hh, plan, ea[lo:hi].contiguous(), na)
        params = list(layer.parameters())
        gs = torch.autograd.grad(y, [hh] + params, gout)
        h2 = h.clone().requires_grad_(True)
        y2 = layer(h2, ei, ea, na)
        g2 = torch.autograd.grad(y2, [h2] + params, gout)
        torch.cuda.synchronize()
        rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))
        res = {"y": rel(y, y2), "deg": int((plan.deg.long() - torch.bincount(ei[1], minlength=N)).abs().max())}
        for i, (a, b) in enumerate(zip(gs, g2)):
            res[f"g{i}"] = rel(a, b)
        # the fixed-buffer step (compute in two HIP graphs, the collectives eager between them)
        st = sharded.GraphedShardedStep(sl, plan, h, ea[lo:hi].contiguous(), na, gout)
        for _ in range(2):
            st.run()
        torch.cuda.synchronize()
        out, gh, ge, gn = st.results()
        res["graph.y"], res["graph.gh"] = rel(out, y2.detach()), rel(gh, g2[0])
        flat = layer.edge_model.flat_params() + layer.node_model.flat_params()
        by_id = {id(p): g for p, g in zip(params, g2[1:])}
        for j, (p, g) in enumerate(zip(flat, ge + gn)):
            if p is not None:
                res[f"graph.g{j}"] = rel(g, by_id[id(p)])
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("metric,C", [((1.0, 1.0, 1.0), 8),      # row-per-lane kernels
                                      ((1.0,) * 5, 8),           # parity-lane kernels (saved inputs + backward scratch region)
                                      ((1.0,) * 5, 16)])         # wide parity-lane kernels
def test_sharded_hip_matches_unsharded(metric, C):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, metric, C)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, res in results:
        assert res["deg"] == 0
        for k, v in res.items():
            assert v < 2e-5, (rank, k, v)


def _worker_dst(rank, world, port, q, metric=(1.0, 1.0, 1.0), C=8):
    """Partitioning B with the HIP backend: nodes partitioned, edges by target."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pkg = importlib.import_module(PKG)
        from csmpn_hip import sharded
        from oracle import ref_path as O
        dev = torch.device("cuda:0")
        torch.manual_seed(0)
        alg = pkg.CliffordAlgebra(tuple(metric))
        layer = pkg.EGCL(alg, C, C, C, edge_attr_features=6, node_attr_features=3, aggr="mean").to(dev)
        N, E = 500, 7001
        h, ei, ea, na = (t.to(dev) for t in O.synthetic_complex(O.Algebra(list(metric)), N, E, C, seed=1))
        gout = torch.randn(N, C, 1 << len(metric), generator=torch.Generator().manual_seed(2)).to(dev)
        part = sharded.DstPartitionedEGCL(layer)
        plan = part.plan(ei, N)
        eal = ea[plan.edge_ids].contiguous()
        hh = h.clone().requires_grad_(True)
        y = part(hh, plan, eal, na)
        params = list(layer.parameters())
        gs = torch.autograd.grad(y, [hh] + params, gout)
        h2 = h.clone().requires_grad_(True)
        y2 = layer(h2, ei, ea, na)
        g2 = torch.autograd.grad(y2, [h2] + params, gout)
        torch.cuda.synchronize()
        rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))
        res = {"y": rel(y, y2)}
        for i, (a, b) in enumerate(zip(gs, g2)):
            res[f"g{i}"] = rel(a, b)
        st = sharded.GraphedDstStep(part, plan, h, eal, na, gout)
        for _ in range(2):
            st.run()
        torch.cuda.synchronize()
        lo, hi = plan.lo, plan.hi
        res["graph.y"] = rel(st.out, y2.detach())
        res["graph.gh"] = rel(st.gh_loc, g2[0][lo:hi])
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("metric,C", [((1.0, 1.0, 1.0), 8), ((1.0,) * 5, 8), ((1.0,) * 5, 28)])
def test_dst_partitioned_hip_matches_unsharded(metric, C):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_dst, args=(r, 2, port, q, metric, C)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, res in results:
        for k, v in res.items():
            assert v < 2e-5, (rank, k, v)


def _worker_stack_overlap(rank, world, port, q):
    """Two chained layers, partitioning B with the collectives in flight under the local-source edges (HIP backend)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pkg = importlib.import_module(PKG)
        from csmpn_hip import sharded
        from oracle import ref_path as O
        dev = torch.device("cuda:0")
        torch.manual_seed(0)
        metric, C = (1.0, 1.0, 1.0), 8
        alg = pkg.CliffordAlgebra(metric)
        layers = [pkg.EGCL(alg, C, C, C, edge_attr_features=6, node_attr_features=3, aggr=a).to(dev) for a in ("mean", "sum")]
        N, E = 501, 7001
        h, ei, ea, na = (t.to(dev) for t in O.synthetic_complex(O.Algebra(list(metric)), N, E, C, seed=1))
        gout = torch.randn(N, C, 8, generator=torch.Generator().manual_seed(2)).to(dev)
        stack = sharded.DstPartitionedStack(layers, overlap=True)
        plan = stack.plan(ei, N)
        assert plan.idx_loc.numel() > 0 and plan.idx_rem.numel() > 0
        eal = ea[plan.edge_ids].contiguous().requires_grad_(True)
        hh = h.clone().requires_grad_(True)
        y = stack(hh, plan, eal, na)
        params = [p for l in layers for p in l.parameters()]
        gs = torch.autograd.grad(y, [hh, eal] + params, gout)
        h2, ea2 = h.clone().requires_grad_(True), ea.clone().requires_grad_(True)
        x = h2
        for l in layers:
            x = l(x, ei, ea2, na)
        g2 = list(torch.autograd.grad(x, [h2, ea2] + params, gout))
        g2[1] = g2[1][plan.edge_ids]
        torch.cuda.synchronize()
        rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))
        res = {"y": rel(y, x)}
        for i, (a, b) in enumerate(zip(gs, g2)):
            res[f"g{i}"] = rel(a, b)
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_dst_partitioned_stack_overlap_hip():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_stack_overlap, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, res in results:
        for k, v in res.items():
            assert v < 5e-5, (rank, k, v)


def _worker_graphed_stack(rank, world, port, q):
    """GraphedDstStackStep (2 L graph segments + collectives) against the autograd chain of the same stack."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pkg = importlib.import_module(PKG)
        from csmpn_hip import sharded
        from oracle import ref_path as O
        dev = torch.device("cuda:0")
        torch.manual_seed(0)
        metric, C = (1.0, 1.0, 1.0), 8
        alg = pkg.CliffordAlgebra(metric)
        layers = [pkg.EGCL(alg, C, C, C, edge_attr_features=6, node_attr_features=3, aggr=a).to(dev) for a in ("mean", "sum", "mean")]
        N, E = 403, 5003
        h, ei, ea, na = (t.to(dev) for t in O.synthetic_complex(O.Algebra(list(metric)), N, E, C, seed=5))
        gout = torch.randn(N, C, 8, generator=torch.Generator().manual_seed(6)).to(dev)
        stack = sharded.DstPartitionedStack(layers)
        plan = stack.plan(ei, N)
        eal = ea[plan.edge_ids].contiguous()
        hh = h.clone().requires_grad_(True)
        y = stack(hh, plan, eal, na)
        params = [p for l in layers for p in l.edge_model.flat_params() + l.node_model.flat_params() if p is not None]
        gs = torch.autograd.grad(y, [hh] + params, gout)
        step = sharded.GraphedDstStackStep(stack, plan, h, eal, na, gout)
        for _ in range(2):
            step.run()
        torch.cuda.synchronize()
        rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))
        lo, hi = plan.lo, plan.hi
        flat_ref = torch.cat([g.reshape(-1) for g in gs[1:]])
        res = {"y": rel(step.out, y.detach()), "gh": rel(step.gh_loc, gs[0][lo:hi]), "params": rel(step.flat, flat_ref),
               "n_flat": float(step.flat.numel() != flat_ref.numel())}
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_graphed_dst_stack_step_matches_autograd():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_graphed_stack, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, res in results:
        for k, v in res.items():
            assert v < 5e-5, (rank, k, v)


def test_graphed_dst_stack_step_single_process():
    """World size 1 (no process group): the collectives of GraphedDstStackStep are copies; same check as above."""
    sys.path.insert(0, ROOT)
    pkg = importlib.import_module(PKG)
    from csmpn_hip import sharded
    from oracle import ref_path as O
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    metric, C = (1.0, 1.0, 1.0), 16
    alg = pkg.CliffordAlgebra(metric)
    layers = [pkg.EGCL(alg, C, C, C, edge_attr_features=6, node_attr_features=3, aggr="mean").to(dev) for _ in range(2)]
    N, E = 300, 4001
    h, ei, ea, na = (t.to(dev) for t in O.synthetic_complex(O.Algebra(list(metric)), N, E, C, seed=7))
    gout = torch.randn(N, C, 8, generator=torch.Generator().manual_seed(8)).to(dev)
    stack = sharded.DstPartitionedStack(layers)
    plan = stack.plan(ei, N)
    hh = h.clone().requires_grad_(True)
    y = stack(hh, plan, ea, na)
    params = [p for l in layers for p in l.edge_model.flat_params() + l.node_model.flat_params() if p is not None]
    gs = torch.autograd.grad(y, [hh] + params, gout)
    step = sharded.GraphedDstStackStep(stack, plan, h, ea, na, gout)
    for _ in range(2):
        step.run()
    torch.cuda.synchronize()
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))
    assert rel(step.out, y.detach()) < 2e-5 and rel(step.gh_loc, gs[0]) < 5e-5
    assert rel(step.flat, torch.cat([g.reshape(-1) for g in gs[1:]])) < 5e-5


# ----------------------------------------------------------------------------- data parallel over graphs

def _worker_ddp(rank, world, port, q):
    """The reference's multi-GPU mode (csmpn/md17.py:15-20, engineer/trainer/trainer.py:342-343): whole-model
    DistributedDataParallel, every rank on its own graphs. Here: the md17 task model on the HIP layers, two ranks on
    cuda:0 over gloo; the averaged gradients equal the single-process gradients of the mean of the two losses."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import numpy as np
        pkg = importlib.import_module(PKG)
        import test_model_harness as H
        dev = torch.device("cuda:0")
        g = np.load(os.path.join(ROOT, "tests", "golden", "model_md17.npz"))
        model = H.build(pkg, "md17", g, dev)
        batch = H.load_batch(pkg, g, device=dev)
        # two different batches of the same topology: the fixture's batch, and the same complexes with perturbed positions
        torch.manual_seed(7)
        second = {k: getattr(batch, k) for k in batch._names}
        for k in ("loc", "vel", "y"):
            second[k] = second[k] + 0.05 * torch.randn_like(second[k])
        batches = [batch, type(batch)(**second)]
        # single process, both batches
        model.zero_grad(set_to_none=True)
        total = 0.0
        for b in batches:
            loss, _ = model(b)
            (loss / world).backward()
            total += float(loss.detach()) / world
        ref = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
        # data parallel: this rank's batch only
        model.zero_grad(set_to_none=True)
        ddp = torch.nn.parallel.DistributedDataParallel(model)
        loss, _ = ddp(batches[rank])
        loss.backward()
        torch.cuda.synchronize()
        res = {}
        for k, p in model.named_parameters():
            if p.grad is not None:
                res[k] = float((p.grad - ref[k]).abs().max() / ref[k].abs().max().clamp(min=1e-12))
        lt = torch.tensor([float(loss.detach())], device=dev)
        dist.all_reduce(lt)
        res["loss"] = abs(float(lt) / world - total) / max(abs(total), 1e-12)
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_ddp_over_graphs_md17():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_ddp, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, res in results:
        assert len(res) > 50
        for k, v in res.items():
            assert v < 2e-4, (rank, k, v)


# ----------------------------------------------------------------------------- the nccl (= RCCL) branch, world size 1

def _worker_rccl_world1(port, q):
    """Round-3 review, item 5: a one-GPU box cannot hold two RCCL ranks, so the `nccl` branch of sharded.py had never run. A
    world-size-1 NCCL group + CSMPN_FORCE_COLLECTIVES=1 posts every collective of the sharded paths to RCCL (all_reduce,
    all_gather_into_tensor, reduce_scatter_tensor, their async forms, the padded layouts) exactly as on N ranks; the
    results must equal the unsharded layer / chain."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CSMPN_FORCE_COLLECTIVES="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        pkg = importlib.import_module(PKG)
        from csmpn_hip import sharded
        from oracle import ref_path as O
        dev = torch.device("cuda:0")
        torch.manual_seed(0)
        metric, C = (1.0, 1.0, 1.0), 8
        alg = pkg.CliffordAlgebra(metric)
        layers = [pkg.EGCL(alg, C, C, C, edge_attr_features=6, node_attr_features=3, aggr=a).to(dev) for a in ("mean", "sum")]
        N, E = 403, 5003
        h, ei, ea, na = (t.to(dev) for t in O.synthetic_complex(O.Algebra(list(metric)), N, E, C, seed=5))
        gout = torch.randn(N, C, 8, generator=torch.Generator().manual_seed(6)).to(dev)
        rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))
        res = {"forced": float(not sharded._multi(None)), "fused": float(not sharded._fused_collectives(None))}
        # unsharded references: one layer, and the chain of two
        layer = layers[0]
        p1 = list(layer.parameters())
        h1 = h.clone().requires_grad_(True)
        y1 = layer(h1, ei, ea, na)
        g1 = torch.autograd.grad(y1, [h1] + p1, gout)
        pall = [p for l in layers for p in l.parameters()]
        h2, ea2 = h.clone().requires_grad_(True), ea.clone().requires_grad_(True)
        x = h2
        for l in layers:
            x = l(x, ei, ea2, na)
        g2 = list(torch.autograd.grad(x, [h2, ea2] + pall, gout))
        # A: edge shards + all_reduce
        sl = sharded.ShardedEGCL(layer)
        planA = sl.plan(ei, N)
        hh = h.clone().requires_grad_(True)
        yA = sl(hh, planA, ea, na)
        gA = torch.autograd.grad(yA, [hh] + p1, gout)
        res["A.y"] = rel(yA, y1)
        for i, (a, b) in enumerate(zip(gA, g1)):
            res[f"A.g{i}"] = rel(a, b)
        st = sharded.GraphedShardedStep(sl, planA, h, ea, na, gout)
        for _ in range(2):
            st.run()
        out, gh, _, _ = st.results()
        res["A.graph.y"], res["A.graph.gh"] = rel(out, y1.detach()), rel(gh, g1[0])
        # B: destination partition, all_gather_into_tensor / reduce_scatter_tensor on the padded layout
        part = sharded.DstPartitionedEGCL(layer)
        planB = part.plan(ei, N)
        assert planB.multi and planB.edges_per_rank == [E]
        eal = ea[planB.edge_ids].contiguous()
        hh = h.clone().requires_grad_(True)
        yB = part(hh, planB, eal, na)
        gB = torch.autograd.grad(yB, [hh] + p1, gout)
        res["B.y"] = rel(yB, y1)
        for i, (a, b) in enumerate(zip(gB, g1)):
            res[f"B.g{i}"] = rel(a, b)
        # B, two chained layers with the asynchronous collectives of the overlap form
        stack = sharded.DstPartitionedStack(layers, overlap=True)
        planS = stack.plan(ei, N)
        eas = ea[planS.edge_ids].contiguous().requires_grad_(True)
        hh = h.clone().requires_grad_(True)
        yS = stack(hh, planS, eas, na)
        gS = torch.autograd.grad(yS, [hh, eas] + pall, gout)
        g2o = list(g2)
        g2o[1] = g2o[1][planS.edge_ids]
        res["S.y"] = rel(yS, x)
        for i, (a, b) in enumerate(zip(gS, g2o)):
            res[f"S.g{i}"] = rel(a, b)
        # the graphed chain: 2 L HIP graphs, one RCCL collective between consecutive graphs
        stack2 = sharded.DstPartitionedStack(layers)
        planG = stack2.plan(ei, N)
        eag = ea[planG.edge_ids].contiguous()
        flatp = [p for l in layers for p in l.edge_model.flat_params() + l.node_model.flat_params() if p is not None]
        by_id = {id(p): g for p, g in zip(pall, g2[2:])}
        step = sharded.GraphedDstStackStep(stack2, planG, h, eag, na, gout)
        for _ in range(2):
            step.run()
        torch.cuda.synchronize()
        res["G.y"], res["G.gh"] = rel(step.out, x.detach()), rel(step.gh_loc, g2[0])
        res["G.params"] = rel(step.flat, torch.cat([by_id[id(p)].reshape(-1) for p in flatp]))
        q.put(res)
    except Exception as e:   # the parent must not wait for a result that will never come
        import traceback
        q.put({"error": f"{type(e).__name__}: {e}\n{traceback.format_exc()}"})
    finally:
        dist.destroy_process_group()


def test_rccl_world1_sharded_paths():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_rccl_world1, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=300)
    p.join(timeout=120)
    assert "error" not in res, res["error"]
    assert p.exitcode == 0
    assert res.pop("forced") == 0.0 and res.pop("fused") == 0.0
    for k, v in res.items():
        assert v < 5e-5, (k, v)


def _worker_stack_reference(rank, world, port, q):
    """The hulls model's three EGCL layers (parameters of model_hulls.npz) as a destination-partitioned stack over two ranks,
    cuts aligned to the graphs of the batch, collectives in flight under the local-source edges, and the same chain replayed
    from HIP-graph segments."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import numpy as np
        pkg = importlib.import_module(PKG)
        import test_model_harness as H
        from csmpn.models import simplicial_mpnn as M
        from csmpn_hip import sharded
        dev = torch.device("cuda:0")
        g = np.load(os.path.join(H.GOLD, "model_hulls.npz"))
        model = H.build(pkg, "hulls", g, dev)
        batch = H.load_batch(pkg, g, device=dev)
        with torch.no_grad():
            B, n = batch.num_graphs, model.algebra.dim
            plan_b = batch.plan(model.max_dim)
            vr = plan_b["vertex_rows"]
            pos = batch.input.index_select(0, vr).reshape(B, -1, n)
            inp = batch.input.index_copy(0, vr, (pos - pos.mean(dim=1, keepdim=True)).reshape(-1, n))
            x0 = model._embed(batch, [(inp.unsqueeze(1), 1)])
            types = torch.nn.functional.one_hot(batch.node_types, model.num_node_type).float()
            node_attr, edge_attr = M.type_attributes(model.algebra, types, batch.edge_index)
        N = int(batch.node_types.shape[0])
        out = {}
        for overlap in (False, True):
            stack = sharded.DstPartitionedStack(list(model.layers), overlap=overlap)
            plan = stack.plan(batch.edge_index, N, boundaries=batch.ptr)
            eal = edge_attr[plan.edge_ids].contiguous()
            with torch.no_grad():
                out[f"stack_overlap{int(overlap)}"] = stack(x0, plan, eal, node_attr).cpu()
        stack = sharded.DstPartitionedStack(list(model.layers))
        plan = stack.plan(batch.edge_index, N, boundaries=batch.ptr)
        eal = edge_attr[plan.edge_ids].contiguous()
        step = sharded.GraphedDstStackStep(stack, plan, x0, eal, node_attr, torch.ones_like(x0))
        step.run()
        torch.cuda.synchronize()
        out["graphed"] = step.out.detach().cpu()
        q.put((rank, {k: v.numpy() for k, v in out.items()}, float(plan.local_share)))
    finally:
        dist.destroy_process_group()


def test_dst_stack_against_reference_stage_fixture():
    """SURVEY.md §8(f)-4 pinned to the REFERENCE (round-4 review: the sharded / graphed multi-layer tests compared HIP with HIP):
    the output of the hulls model's 3-layer chain computed by DstPartitionedStack (blocking and overlapped) and by
    GraphedDstStackStep on two ranks equals x behind the reference model's last layer - stages_hulls.npz, recorded from the
    imported reference (every 8th row whole, norm and a seeded projection of the full tensor; its float32 run = yardstick)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_model_harness as H
    st = np.load(os.path.join(H.GOLD, "stages_hulls.npz"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_stack_reference, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    k = "layer2"
    rows64, rows32 = st[f"f64/{k}/rows"].astype(np.float64), st[f"f32/{k}/rows"].astype(np.float64)
    scale = np.abs(rows64).max()
    yard = np.abs(rows32 - rows64).max() / scale
    n64, p64 = st[f"f64/{k}/np"]
    n32, p32 = st[f"f32/{k}/np"]
    for rank, outs, share in results:
        assert share == 1.0            # graph-aligned cuts: no adjacency crosses ranks, the exchanges carry nothing needed
        for name, t in outs.items():
            err = np.abs(t[::8].astype(np.float64) - rows64).max() / scale
            assert err <= max(1e-5, 4 * yard), (rank, name, err, yard)
            tt = torch.from_numpy(t).double()
            nrm, prj = float(tt.norm()), float((tt * H.direction_for(k, tt.shape)).sum())
            assert abs(nrm - n64) <= max(1e-5, 4 * abs(n32 - n64) / n64) * n64, (rank, name, nrm, n64)
            assert abs(prj - p64) <= max(1e-5, 4 * abs(p32 - p64) / n64) * n64, (rank, name, prj, p64)
