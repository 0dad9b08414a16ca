#!/usr/bin/env python3
"""bench.py — simplicial edges/sec (forward + backward) of one shared simplicial
message-passing layer (EGCL) on MI355X, with its HBM-roofline fraction and the
reference CPU path timed beside it.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one EGCL layer (edge CEMLP + scatter + node CEMLP), forward + backward
(gradients w.r.t. h and every parameter; attributes without gradient, hulls-style),
on synthetic input already resident in HBM.

Workload (BASELINE.json configs[1]): S1 = Cl(3,0), 8 channels,
10 000 nodes, 100 000 directed adjacencies per GPU, aggr=mean, seeded generator.
With N > 1 GPUs the adjacency list of an N x 100k-edge complex over the same 10k nodes is
sharded (100k edges per rank, weak scaling). `python bench.py --gpus N` without a launcher starts the N ranks itself
(child processes of torch.distributed.run, before this process touches a GPU). Default partitioning B
(csmpn_hip/sharded.py; SURVEY.md §8(e)'s recommendation): every rank owns a node slice (cut by in-degree) and all edges
into it - all-gather of the updated node slices forward, reduce-scatter of d/dh backward, all-reduce of the parameter
gradients. `--partition A` (BASELINE.json's wording): contiguous edge shards with an all-reduce of the per-node aggregate
(forward) and of [d/dh | edge-model gradients] (backward); its replicated node stage bounds the strong-scaling speed-up
at 4.0x on 8 GPUs. `--scaling strong --workload S2` shards ONE 1M-edge complex (north_star's multi-GPU
configuration) instead; both report the compute-only rate and the bus bandwidth (partition B).
"""
import argparse
import importlib
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "clifford-group-equivariant-simplicial-message-passing-networks_amd"

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FP32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: vector = matrix fp32 peak

WORKLOADS = {
    # name: (metric, channels, nodes, edges per GPU)
    "S1": ((1.0, 1.0, 1.0), 8, 10_000, 100_000),
    "S2": ((1.0, 1.0, 1.0), 16, 100_000, 1_000_000),
    "S3": ((1.0, 1.0, 1.0, 1.0, -1.0), 8, 10_000, 100_000),
    # not a BASELINE throughput config: one layer of the convex-hulls width (Cl(5,0), 28 channels) at S1's size
    "H28": ((1.0, 1.0, 1.0, 1.0, 1.0), 28, 10_000, 100_000),
    # not a BASELINE config: the two-wave variant of the wide parity-lane kernels (Cl(5,0), 16 channels) at S1's size
    "H16": ((1.0, 1.0, 1.0, 1.0, 1.0), 16, 10_000, 100_000),
    # BASELINE config 3's layer shape (md17_cssmpnn.py: Cl(3,0), 32 channels, aggr = sum) at S1's size
    "M32": ((1.0, 1.0, 1.0), 32, 10_000, 100_000),
}
WORKLOAD_AGGR = {"M32": "sum"}   # every other workload: mean (hulls_cssmpnn.py)


def algorithmic_bytes(C, D, A=6, T=3):
    """algorithmic HBM bytes per edge / per node of each stage."""
    row, attr, nattr = C * D * 4, A * D * 4, T * D * 4
    return {
        "edge_fwd": 8 + 3 * row + attr,
        "edge_bwd": 8 + 5 * row + attr,
        "node_fwd": 3 * row + nattr,
        "node_bwd": 5 * row + nattr,
    }


def kernel_of(stage, name):
    """Does the rocprofv3 kernel name belong to this stage? (MODE 1 = edge, 2 = node; last template
    argument = backward.) cemlp_pl_kernel<Alg, MODE, NBLK, I0, BWD>,
    cemlp_kernel<Alg, MODE, ...>, cemlp_ps_kernel<Alg, MODE, BWD>."""
    import re
    # cemlp_cl_{fwd,bwd}_kernel / cemlp_cm_{fwd,bwd}_kernel<Alg, C, MODE, NBLK, NA> (round 3: all blocks of a backward in one launch)
    m = re.search(r"cemlp_c[lm]_(fwd|bwd)_kernel<csmpn::Alg<[^>]*>, ([^>]*)>", name)
    if m:
        args = [a.strip() for a in m.group(2).split(",")]
        return args[1] == ("1" if stage.startswith("edge") else "2") and (m.group(1) == "bwd") == stage.endswith("bwd")
    # round 4: cemlp_cmb_kernel / cemlp_cmp_kernel<Alg, C, MODE, NBLK, NA>: the channel-MFMA backward (16 / 32 channels)
    m = re.search(r"cemlp_cm[bp]_kernel<csmpn::Alg<[^>]*>, ([^>]*)>", name)
    if m:
        args = [a.strip() for a in m.group(1).split(",")]
        return args[1] == ("1" if stage.startswith("edge") else "2") and stage.endswith("bwd")
    m = re.search(r"cemlp(_ps|_pl)?_kernel<csmpn::Alg<[^>]*>, ([^>]*)>", name)
    if not m:
        return False
    args = [a.strip() for a in m.group(2).split(",")]
    mode = args[0]
    return mode == ("1" if stage.startswith("edge") else "2") and args[-1] == ("true" if stage.endswith("bwd") else "false")


def pmc_traffic(stage, workload, kernel_name=None):
    """(HBM bytes per launch of the dominant kernel, file it came from) from the committed rocprofv3 PMC summary of THIS
    workload (separate --pmc passes, tools/profile_r04.sh -> profiles/r04_<workload>_pmc_summary.json):
    2 x FETCH_SIZE (gfx950 counts a wide coalesced read at half its bytes, MI355X_MICROARCH.md HBM
    section) + WRITE_SIZE, KB -> B. The entry must carry the name of the kernel this run DISPATCHED (csmpn_last_kernel):
    a summary recorded for another kernel - a stale profile - gives None, not its number. None too when no summary of this
    workload is committed (counters cannot be read from inside the timed run)."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{workload}_pmc_summary.json")))
    if not found:
        return None, None
    norm = lambda n: n.replace("void ", "").split("(")[0].replace(" ", "")
    import re
    try:
        with open(found[-1]) as f:
            summary = json.load(f)
        # families whose entry point reports an abbreviated name ("csmpn::cemlp_pg_bwd_kernel<Alg, ...> (mode M, ...)": the wide
        # D = 32 kernels) run a stage as SEVERAL launches (one per block): the stage's traffic is the sum over the launches
        # of that family and mode in the summary
        m = re.match(r"(csmpn::cemlp_(plw|pg)_(fwd|bwd)_kernel)<.*\.\.\.>\s*\(mode (\d)", kernel_name or "")
        if m:
            fam, which, mode = m.group(1), m.group(2), m.group(4)
            total = 0
            for name, c in summary.items():
                if fam + "<" not in name.replace("void ", "") or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
                    continue
                cfg = re.search(r"P(lw|g)Cfg<csmpn::Alg<[^>]*>, ([^>]*)>", name)
                if not cfg:
                    continue
                args = [a_.strip() for a_ in cfg.group(2).split(",")]
                if args[2 if which == "plw" else 1] == mode:
                    total += int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
            return (total, os.path.relpath(found[-1], ROOT)) if total else (None, None)
        if True:
            best = None
            for name, c in summary.items():
                if kernel_name is not None and norm(name) != norm(kernel_name):
                    continue
                if kernel_of(stage, name) and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                    t = int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
                    best = t if best is None or t > best else best
            return best, (os.path.relpath(found[-1], ROOT) if best is not None else None)
    except OSError:
        return None, None


def make_inputs(metric, C, N, E_total, lo, hi, device):
    """Seeded S-series generator; returns this rank's shard [lo, hi)."""
    D = 1 << len(metric)
    g = torch.Generator().manual_seed(0)
    edge_index = torch.randint(0, N, (2, E_total), generator=g, dtype=torch.int64)
    node_types = torch.randint(0, 3, (N,), generator=g)
    node_attr = torch.zeros(N, 3, D)
    node_attr[torch.arange(N), node_types, 0] = 1.0            # one-hot type on the scalar blade
    h = torch.randn(N, C, D, generator=g)
    ei = edge_index[:, lo:hi].contiguous()
    edge_attr = torch.cat([node_attr[ei[0]], node_attr[ei[1]]], dim=1)
    return (h.to(device), ei.to(device), edge_attr.to(device).contiguous(), node_attr.to(device)), \
           (h, ei, edge_attr, node_attr)


def cpu_baseline(metric, C, state, cpu_inputs, budget_s=20.0, aggr="mean"):
    """The reference CPU path (oracle restatement: dense-einsum formulation, PyTorch CPU)
    on a BOUNDED sample of the same workload: the first E_s edges of the edge list, node ids folded
    onto the first N_s = N E_s / E nodes (same edge / node ratio as the workload), fwd+bwd. E_s is sized from a small probe so that the timed runs take
    about `budget_s` seconds; threads = min(host cores, 32) (more threads make the many
    small einsum/bmm calls slower, measured: 256 threads -> 235 s per 100k edges)."""
    from oracle import ref_path as O
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    alg = O.Algebra(list(metric))
    h, ei, ea, na = cpu_inputs
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in state.items()}

    E_all, N_all = ei.shape[1], h.shape[0]

    def nodes_for(ne):   # the sample keeps the workload's edge / node ratio (node ids folded onto the first N_s)
        return N_all if ne >= E_all else max(16, -(-N_all * ne // E_all))

    def run(ne):
        ns = nodes_for(ne)
        hh = h[:ns].clone().requires_grad_(True)
        es = ei[:, :ne] if ns == N_all else ei[:, :ne] % ns
        t0 = time.perf_counter()
        y = O.egcl(alg, hh, es, ea[:ne], na[:ns], p, aggr=aggr)
        y.backward(torch.ones_like(y))
        dt = time.perf_counter() - t0
        for v in p.values():
            v.grad = None
        return dt

    probe_e = min(2000, ei.shape[1])
    run(probe_e)                      # warm-up (thread pool, allocator)
    t_probe = run(probe_e)
    reps = 5
    ne = int(min(ei.shape[1], max(probe_e, probe_e * budget_s / (reps * max(t_probe, 1e-4)))))
    times = [run(ne) for _ in range(reps)]
    med = statistics.median(times)
    twin = None
    try:   # the C++ sparse-formulation twin (oracle/cpu_twin): the strong CPU baseline, all host cores
        from oracle import cpu_twin
        pn = {k: v.detach().cpu().numpy() for k, v in state.items()}
        gout = torch.ones(h.shape[0], C, 1 << len(metric)).numpy()
        args_t = (list(metric), pn, h.numpy(), ei.numpy(), ea.numpy(), na.numpy())
        cpu_twin.egcl_layer(*args_t, aggr=aggr, gout=gout)
        tt = []
        for _ in range(5):
            t0 = time.perf_counter()
            cpu_twin.egcl_layer(*args_t, aggr=aggr, gout=gout)
            tt.append(time.perf_counter() - t0)
        tm = statistics.median(tt)
        twin = {"value": ei.shape[1] / tm, "unit": "edges/s", "cores": os.cpu_count(), "kind": "port",
                "sample": f"C++/OpenMP twin (sparse sign-table formulation, oracle/cpu_twin), all {ei.shape[1]} edges, "
                          f"fwd+bwd, median of 5 runs ({tm:.3f} s each), {os.cpu_count()} threads"}
    except Exception as exc:   # the twin is optional infrastructure
        twin = {"error": f"{type(exc).__name__}: {exc}"}
    return {"value": ne / med, "unit": "edges/s", "cores": cores, "kind": "port", "host_cores": os.cpu_count(),
            "strong_cpu_twin": twin,
            "sample": f"{'all' if ne == ei.shape[1] else 'first'} {ne} of {ei.shape[1]} edges over "
                      f"{'the same' if ne == ei.shape[1] else 'the first'} {nodes_for(ne)} of {h.shape[0]} nodes "
                      f"(same edge/node ratio), fwd+bwd, median of {reps} runs ({med:.3f} s each), torch "
                      f"{torch.__version__} CPU threads={cores} of {os.cpu_count()} host cores"}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py
    <the same arguments>` as a child process on a free loopback port. Every line the ranks print is relayed as it comes;
    rank 0's JSON line is held back and printed LAST on stdout. Returns the child's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line_json = None
    for line in proc.stdout:
        if line.startswith('{"metric"'):
            line_json = line
        else:
            sys.stdout.write(line)
            sys.stdout.flush()
    rc = proc.wait()
    if line_json is not None:
        sys.stdout.write(line_json)
        sys.stdout.flush()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="S1", choices=sorted(WORKLOADS))
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--segments", action="store_true",
                    help="run the N>1 launch path (graph segments + eager collectives) on one GPU too")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of CPU-baseline work")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = the workload's edges PER rank; strong = ONE complex sharded over the ranks")
    ap.add_argument("--partition", default="auto", choices=["auto", "A", "B"],
                    help="N > 1: A = edge shards + all-reduce of the per-node aggregate (BASELINE.json's wording; its "
                         "replicated node stage bounds the strong-scaling speed-up at 4.0x on 8 GPUs, DESIGN.md §5); "
                         "B = destination-partitioned (all-gather / reduce-scatter, half the bytes, no replicated "
                         "stage). auto = B")
    ap.add_argument("--deterministic", action="store_true",
                    help="atomic-free aggregation (CSMPN_FLAG_DETERMINISTIC): edge rows to a table + fixed-order segmented sums")
    ap.add_argument("--rehearse-backend", default=None, metavar="MODULE:ATTR",
                    help="launch-path rehearsal where there is no GPU (tests only): CPU tensors, gloo, the sharded step "
                         "computed by the injected backend object instead of the HIP C-ABI. The line is marked "
                         "'rehearsal' and its value is NOT a measurement")
    ap.add_argument("--rehearse-size", default="64,512", metavar="NODES,EDGES",
                    help="with --rehearse-backend: nodes, edges per rank instead of the workload's")
    args = ap.parse_args()

    if args.partition == "auto":
        args.partition = "B"
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # Plain `python bench.py --gpus N`: start the N ranks as CHILD processes (torch.distributed.run) before anything
        # in this process has touched the GPU (importing torch does not), relay their output and exit with their code.
        raise SystemExit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}: the line would report a wrong n_gpus")
    import torch.distributed as dist
    # one rank per GPU; the modulo only matters for rehearsing the N > 1 path on a box with fewer
    # GPUs than ranks (--dist-backend gloo: RCCL refuses two ranks on one device)
    rehearsal = args.rehearse_backend is not None
    if rehearsal:
        device = torch.device("cpu")
        args.dist_backend = "gloo" if args.dist_backend == "nccl" else args.dist_backend
        args.no_graph = True
    else:
        device = torch.device("cuda", local_rank % max(torch.cuda.device_count(), 1))
        torch.cuda.set_device(device)
    sync = (lambda: None) if rehearsal else torch.cuda.synchronize
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)

    pkg = importlib.import_module(PKG)
    from csmpn_hip import native, ops, sharded
    if args.deterministic:
        ops.set_deterministic(True)

    metric, C, N, E_per = WORKLOADS[args.workload]
    backend = ops.HipBackend
    if rehearsal:
        mod, attr = args.rehearse_backend.split(":")
        backend = getattr(importlib.import_module(mod), attr)
        N, E_per = (int(v) for v in args.rehearse_size.split(","))
    D = 1 << len(metric)
    E_total = E_per * world if args.scaling == "weak" else E_per
    part_b = world > 1 and args.partition == "B"
    if part_b:
        lo, hi = 0, E_total            # replicated topology; every rank selects the edges into its nodes
    else:
        lo, hi = sharded.shard_bounds(E_total, world, rank)
    (h, ei, ea, na), cpu_inputs = make_inputs(metric, C, N, E_total, lo, hi, device)

    torch.manual_seed(0)
    aggr = WORKLOAD_AGGR.get(args.workload, "mean")
    layer = pkg.EGCL(pkg.CliffordAlgebra(metric), C, C, C, edge_attr_features=6, node_attr_features=3, aggr=aggr)
    state = {k: v.detach().clone() for k, v in layer.named_parameters()}
    layer = layer.to(device)
    params = list(layer.parameters())
    h.requires_grad_(True)
    gout = torch.ones(N, C, D, device=device)

    segments = None
    if part_b:
        sl = sharded.DstPartitionedEGCL(layer, backend=backend)
        plan = sl.plan(ei, N)
        ea = ea[plan.edge_ids].contiguous()

        def step():
            y = sl(h, plan, ea, na)
            return torch.autograd.grad(y, [h] + params, gout)
    elif world > 1 or args.segments or rehearsal:
        sl = sharded.ShardedEGCL(layer, backend=backend)
        plan = sl.plan(ei, N)

        def step():
            y = sl(h, plan, ea, na)
            return torch.autograd.grad(y, [h] + params, gout)
    else:
        def step():
            y = layer(h, ei, ea, na)
            return torch.autograd.grad(y, [h] + params, gout)

    # warm-up (builds the CSR once, sets kernel attributes)
    for _ in range(max(args.warmup, 1)):
        step()
    sync()

    use_graph = (world == 1) and not args.no_graph and not args.segments and not rehearsal
    graph = None
    if (world > 1 or args.segments) and not args.no_graph:
        # compute stages as two HIP graphs, the two collectives eager between them
        try:
            segments = (sharded.GraphedDstStep if part_b else sharded.GraphedShardedStep)(sl, plan, h, ea, na, gout)
            segments.run()
            sync()
        except Exception as exc:
            print(f"[bench] graph segments unavailable ({type(exc).__name__}: {exc}); launching eagerly", file=sys.stderr)
            segments = None
            sync()
    if use_graph:
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                graph_out = step()
            graph.replay()
            sync()
        except Exception as exc:   # capture is an optimisation of the launch path only
            print(f"[bench] HIP-graph capture unavailable ({type(exc).__name__}: {exc}); launching eagerly",
                  file=sys.stderr)
            graph = None
            sync()
    run = graph.replay if graph is not None else (segments.run if segments is not None else step)
    for _ in range(2):
        run()

    edges_per_rank = None
    if world > 1:
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
        cnt = torch.zeros(world, dtype=torch.int64, device=device)
        cnt[rank] = int(plan.csr.n_edges)
        dist.all_reduce(cnt)
        edges_per_rank = cnt.tolist()
        assert sum(edges_per_rank) == E_total, (edges_per_rank, E_total)   # every adjacency on exactly one rank
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    sync()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = E_total * args.steps / elapsed
    # N > 1, partitioning B: the same steps without the collectives (compute-only rate, bus bandwidth)
    compute_only = None
    if part_b and segments is not None:
        dist.barrier()
        sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            segments.run(compute_only=True)
        sync()
        dist.barrier()
        tc = torch.tensor([time.perf_counter() - t1], device=device, dtype=torch.float64)
        dist.all_reduce(tc, op=dist.ReduceOp.MAX)
        tc = float(tc.item())
        coll_s = max(elapsed - tc, 1e-9) / args.steps
        compute_only = {"value": round(E_total * args.steps / tc, 1), "ms_per_step": round(1e3 * tc / args.steps, 4),
                        "collective_ms_per_step": round(1e3 * coll_s, 4),
                        "bytes_sent_per_rank_per_step": segments.bytes_per_step,
                        "bus_GBps_per_rank": round(segments.bytes_per_step / coll_s / 1e9, 2)}

    result = None
    if rank == 0 and rehearsal:
        result = {"metric": "launch-path rehearsal (CPU tensors, injected backend): NOT a measurement",
                  "value": round(value, 1), "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                  "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": args.scaling,
                  "vs_baseline": None, "dtype": "f32", "data": "rehearsal",
                  "config": {"workload": f"rehearsal of {args.workload}: {N} nodes, {E_per} edges per rank",
                             "partition": None if world == 1 else args.partition, "edges_per_rank": edges_per_rank,
                             "pad_ratio": round(plan.pad_ratio, 4) if part_b else None,
                             "local_share": round(plan.local_share, 4) if part_b else None,
                             "backend": args.rehearse_backend, "dist_backend": args.dist_backend},
                  "roofline": None, "cpu_baseline": None}
    elif rank == 0:
        # ---- per-stage kernel time (HIP events on the launch stream, weights pre-packed)
        be = ops.HipBackend
        spec = layer.spec()
        csr = ops.get_csr(ei, N) if world == 1 else plan.csr
        csr_first_ms = getattr(csr, "build_ms", None)   # includes loading the sort kernels' code objects
        if world == 1:                                     # steady state: what a new batch pays (one-time per complex)
            csr_build_ms = statistics.median(ops.Csr(ei, N).build_ms for _ in range(5))
        else:
            csr_build_ms = csr_first_ms
        deg = csr.deg if world == 1 else plan.deg
        pe, pn = layer.edge_model.flat_params(), layer.node_model.flat_params()
        hd = h.detach()
        agg, st_e = be.edge_forward(spec, csr, hd, ea, pe)
        out, st_n = be.node_forward(spec, deg, hd, agg, na, pn)
        gh, g_agg, _, _ = be.node_backward(spec, deg, hd, agg, na, pn, gout, False, st_n)
        stages = {
            "edge_fwd": lambda: be.edge_forward(spec, csr, hd, ea, pe),   # incl. saving block inputs
            "node_fwd": lambda: be.node_forward(spec, deg, hd, agg, na, pn),
            "node_bwd": lambda: be.node_backward(spec, deg, hd, agg, na, pn, gout, False, st_n),
            "edge_bwd": lambda: be.edge_backward(spec, csr, hd, ea, pe, g_agg, gh, False, st_e),
        }
        reps = max(10, min(args.steps, 50))
        stage_ms, stage_kernel = {}, {}
        for name, fn in stages.items():
            fn()
            sync()
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for a, b in evs:
                a.record()
                fn()
                b.record()
            sync()
            stage_ms[name] = statistics.median(a.elapsed_time(b) for a, b in evs)
            stage_kernel[name] = native.lib().csmpn_last_kernel().decode()   # what the entry point dispatched
        ab = algorithmic_bytes(C, D)
        units = {"edge_fwd": E_per, "edge_bwd": E_per, "node_fwd": N, "node_bwd": N}
        dom = max(stage_ms, key=stage_ms.get)
        alg_bytes = ab[dom] * units[dom]
        achieved = alg_bytes / (stage_ms[dom] * 1e-3) / 1e9
        bytes_per_edge = ab["edge_fwd"] + ab["edge_bwd"] + (N / E_per) * (ab["node_fwd"] + ab["node_bwd"])
        traffic, traffic_source = pmc_traffic(dom, args.workload, stage_kernel[dom])
        roofline = {
            "bound": "hbm", "kernel": stage_kernel[dom], "stage": dom,
            "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
            # `frac` prices the ALGORITHMIC bytes (no reuse assumed for the gathers); the counters see fewer
            # bytes because h is L2-resident: the same kernel time against the MEASURED traffic of the committed PMC summary
            "frac_of_measured_traffic": None if traffic is None else round(traffic / (stage_ms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            "traffic_source": traffic_source,
            "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": round(stage_ms[dom], 4),
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "layer_bytes_per_edge": round(bytes_per_edge, 1),
            "layer_frac_of_hbm_roofline": round(value / world * bytes_per_edge / (HBM_PEAK_GBS * 1e9), 5),
        }
        # the second roofline: dense mixings D C (6C + A) MAC per edge (node program: D C (7C + T) per node), x 3
        # with the backward; geometric product 2 C D^2 sign-table terms per row, x 4 with its backward; against MI355X's fp32 peak,
        # ONE pool for vector and matrix instructions (tools/mfma_valu_overlap_probe.hip)
        mac_edge = 3 * D * C * (6 * C + 6) + 4 * 2 * C * D * D
        mac_node = 3 * D * C * (7 * C + 3) + 4 * 2 * C * D * D
        mac_per_edge = mac_edge + (N / E_per) * mac_node
        roofline["fp32_alu"] = {"mac_per_edge": round(mac_per_edge, 1), "peak_tflops": FP32_PEAK_TFLOPS,
                                "layer_frac_of_fp32_peak": round(value / world * 2 * mac_per_edge / (FP32_PEAK_TFLOPS * 1e12), 5)}
        result = {
            "metric": "simplicial edges/sec (fwd+bwd) on Cl(3,0) 8-ch multivectors; % HBM roofline"
                      if args.workload == "S1" else f"simplicial edges/sec (fwd+bwd), workload {args.workload}",
            "value": round(value, 1), "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: EGCL layer fwd+bwd, Cl{tuple(int(m) for m in metric)}, "
                                   f"{C} channels, {N} nodes, "
                                   + (f"{E_per} edges/GPU x {world} GPU" if args.scaling == "weak" else
                                      f"{E_total} edges sharded over {world} GPU")
                                   + f", aggr={aggr}, edge_attr 6ch, node_attr 3ch",
                       "csr_build_ms": None if csr_build_ms is None else round(csr_build_ms, 3),
                       "csr_first_build_ms": None if csr_first_ms is None else round(csr_first_ms, 3),
                       "host_cores": os.cpu_count(),
                       "aggregation": "deterministic (row table + fixed-order segmented sums)" if args.deterministic else "float atomics",
                       "launch": ("hip-graph replay" if graph is not None else
                                  "hip-graph segments + eager collectives" if segments is not None else "eager"),
                       "sharding": ("none" if world == 1 else
                                    "B: nodes partitioned, edges by target; all-gather(out) fwd, reduce-scatter(d/dh) + "
                                    "all-reduce(param grads) bwd" if part_b else
                                    "A: edge list sharded, all-reduce(agg) fwd + all-reduce([dh|edge grads]) bwd"),
                       "edges_per_rank": edges_per_rank,
                       "partition": None if world == 1 else args.partition,
                       "pad_ratio": round(plan.pad_ratio, 4) if part_b else None,
                       "local_share": round(plan.local_share, 4) if part_b else None,
                       "compute_only": compute_only},
            "roofline": roofline,
        }
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and not rehearsal:
            result["cpu_baseline"] = cpu_baseline(metric, C, state, cpu_inputs, args.cpu_budget, aggr=aggr)
        print(json.dumps(result))


if __name__ == "__main__":
    main()
