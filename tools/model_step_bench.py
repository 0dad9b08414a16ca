"""Whole-model training step, eager vs one HIP graph (csmpn_hip.graphed.GraphedTrainStep).

    python tools/model_step_bench.py [--batch 16] [--steps 30]

The convex-hulls task model at the reference's configuration (csmpn/configs/hulls.yaml: Cl(5,0), 28
hidden channels, 3 layers, batch size 16, 8 points per hull, Adam lr 1e-3) on synthetic hulls. Prints one
JSON line with the step times and the simplices / adjacencies of the batch.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "clifford-group-equivariant-simplicial-message-passing-networks_amd"


def op_sites(eager, steps=2):
    """GPU time of an eager step by (innermost frame of this repo, autograd node, kernel): where the small launches
    come from. Chrome trace of torch.profiler: kernels -> launching runtime call by correlation id -> enclosing
    python_function / cpu_op events of the same thread by time."""
    import bisect
    import collections
    import tempfile
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(steps):
            eager()
        torch.cuda.synchronize()
    path = os.path.join(tempfile.mkdtemp(), "trace.json")
    prof.export_chrome_trace(path)
    ev = json.load(open(path))["traceEvents"]
    launches, spans = {}, collections.defaultdict(list)
    for e in ev:
        if e.get("ph") != "X":
            continue
        cat = e.get("cat", "")
        if cat in ("cuda_runtime", "cuda_driver") and "correlation" in e.get("args", {}):
            launches[e["args"]["correlation"]] = e
        elif cat in ("python_function", "cpu_op", "user_annotation"):
            spans[(e["pid"], e["tid"])].append((e["ts"], e["ts"] + e["dur"], cat, e["name"]))
    for v in spans.values():
        v.sort()
    starts = {k: [s[0] for s in v] for k, v in spans.items()}
    table = collections.defaultdict(lambda: [0.0, 0])
    for e in ev:
        if e.get("ph") != "X" or e.get("cat") not in ("kernel", "gpu_memcpy", "gpu_memset"):
            continue
        la = launches.get(e.get("args", {}).get("correlation"))
        site, node = "?", ""
        if la is not None:
            key = (la["pid"], la["tid"])
            v, ts = spans.get(key, []), la["ts"]
            i = bisect.bisect_right(starts.get(key, []), ts)
            best_py, best_op = None, None
            for s0, s1, cat, name in reversed(v[max(0, i - 4000):i]):
                if s1 < ts:
                    continue
                if cat == "python_function" and best_py is None and ("csmpn" in name or "tools/" in name) and "<" not in name.split(":")[-1][:1]:
                    best_py = name
                if cat == "cpu_op" and (name.startswith("autograd::engine") or best_op is None):
                    best_op = name
            site = (best_py or "?").split("networks_amd/")[-1]
            node = (best_op or "").replace("autograd::engine::evaluate_function: ", "bwd:")
        k = (site[-70:], node[:40], e["name"][:60])
        table[k][0] += e["dur"]
        table[k][1] += 1
    tot = sum(v[0] for v in table.values())
    print(f"GPU time per step {tot / steps:.0f} us, {sum(v[1] for v in table.values()) / steps:.0f} launches")
    for k, (us, n) in sorted(table.items(), key=lambda kv: -kv[1][0])[:90]:
        print(f"{us / steps:8.1f} us {n / steps:5.1f}x  {k[0]:70s} {k[1]:40s} {k[2]}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--points", type=int, default=8)
    ap.add_argument("--model", default="hulls", choices=["hulls", "md17", "motion", "nba"])
    ap.add_argument("--no-fused-adam", action="store_true", help="torch.optim.Adam(foreach) instead of fused=True")
    ap.add_argument("--no-fused-grads", action="store_true", help="hand parameter gradients to autograd (one add kernel per tensor)")
    ap.add_argument("--profile-ops", action="store_true", help="torch.profiler over 3 eager steps: the small GPU kernels by Python call site")
    ap.add_argument("--no-flat-params", action="store_true", help="optimizer over the module's ~1 000 parameter tensors instead of ONE flat buffer (csmpn_hip.graphed.flatten_parameters)")
    ap.add_argument("--op-sites", action="store_true", help="GPU kernels of one eager step by the package line that launched them")
    args = ap.parse_args()
    importlib.import_module(PKG)
    from csmpn.data import complexes as cx
    from csmpn.models import simplicial_mpnn as M
    from csmpn_hip.graphed import GraphedTrainStep
    from csmpn_hip import ops
    ops.set_fused_grad_accumulation(not args.no_fused_grads)

    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    if args.model == "hulls":
        batch = cx.collate([cx.hulls_example(rng.standard_normal((args.points, 5)).astype(np.float32))
                            for _ in range(args.batch)]).to(dev)
        features = ["input", "target"]
    elif args.model == "motion":
        # motion-capture-shaped batch (motion_cssmpnn.py: Cl(3,0), 16 channels, 4 layers, aggr = mean): 31 joints per
        # skeleton, Vietoris-Rips complex of the joint positions, one frame of positions / velocities
        V, graphs = 31, []
        for _ in range(args.batch):
            base = (1.6 * rng.standard_normal((V, 3))).astype(np.float32)
            c = cx.rips_complex(base, dis=1.6, max_dim=2)
            S = c.n_simplices
            pos, vel = torch.zeros(S, 3), torch.zeros(S, 3)
            pos[:V] = torch.from_numpy(base)
            vel[:V] = 0.1 * torch.from_numpy(rng.standard_normal((V, 3)).astype(np.float32))
            c.features.update(pos=pos, vel=vel)
            graphs.append(c)
        batch = cx.collate(graphs)
        batch.y = torch.cat([g.features["pos"][: g.n_vertices] + 0.1 * torch.randn(g.n_vertices, 3) for g in graphs], dim=0)
        batch._names.append("y")
        batch = batch.to(dev)
        features = ["pos", "vel", "y"]
    elif args.model == "nba":
        # NBA-shaped batch (nba_cssmpnn.py:12-190: Cl(2,0), 40 channels, 4 layers, aggr = sum): 6 agents in the plane (5 players +
        # the ball), 10 frames, Vietoris-Rips complex of the first frame; the target: 40 future frames of the 5 players
        V, F, graphs, ys = 6, 10, [], []
        for _ in range(args.batch):
            base = rng.standard_normal((V, 2)).astype(np.float32)
            c = cx.rips_complex(base, dis=1.6, max_dim=2)
            S = c.n_simplices
            pos, vel = torch.zeros(S, F, 2), torch.zeros(S, F, 2)
            pos[:V] = torch.from_numpy(base)[:, None, :] + 0.05 * torch.from_numpy(rng.standard_normal((V, F, 2)).astype(np.float32))
            vel[:V] = 0.1 * torch.from_numpy(rng.standard_normal((V, F, 2)).astype(np.float32))
            c.features.update(pos=pos, vel=vel)
            graphs.append(c)
            ys.append(torch.from_numpy(rng.standard_normal((V - 1, 4 * F, 2)).astype(np.float32)))
        batch = cx.collate(graphs)
        batch.y = torch.cat(ys, dim=0)
        batch._names.append("y")
        batch = batch.to(dev)
        features = ["pos", "vel", "y"]
    else:
        # MD17-shaped batch (md17_cssmpnn.py; csmpn/configs/md17.yaml: Cl(3,0), 32 channels, 5 layers): 21 atoms
        # (aspirin), 10 frames, Vietoris-Rips complex of the first frame, random positions / velocities / charges
        V, F, graphs = 21, 10, []
        for _ in range(args.batch):
            base = (1.6 * rng.standard_normal((V, 3))).astype(np.float32)
            c = cx.rips_complex(base, dis=1.8, max_dim=2)
            S = c.n_simplices
            loc = torch.zeros(S, F, 3); vel = torch.zeros(S, F, 3); ch = torch.zeros(S, F, 1)
            loc[:V] = torch.from_numpy(base)[:, None, :] + 0.05 * torch.from_numpy(rng.standard_normal((V, F, 3)).astype(np.float32))
            vel[:V] = 0.1 * torch.from_numpy(rng.standard_normal((V, F, 3)).astype(np.float32))
            ch[:V] = torch.from_numpy(rng.integers(1, 9, size=(V, 1, 1)).astype(np.float32)).expand(V, F, 1)
            c.features.update(loc=loc, vel=vel, charges=ch)
            graphs.append(c)
        batch = cx.collate(graphs)
        batch.y = torch.cat([g.features["loc"][: g.n_vertices] + 0.1 * torch.randn(g.n_vertices, F, 3) for g in graphs], dim=0)
        batch._names.append("y")
        batch = batch.to(dev)
        features = ["loc", "vel", "charges", "y"]
    torch.manual_seed(0)
    model = {"hulls": M.HullsSimplicialMPNN, "md17": M.MD17SimplicialMPNN, "motion": M.MotionSimplicialMPNN,
             "nba": M.NBASimplicialMPNN}[args.model]().to(dev)
    # the reference trains with torch.optim.Adam (csmpn/configs/hulls.yaml); fused=True is the same update in one
    # multi-tensor kernel (the default foreach path with capturable=True issues ~300 per-tensor div kernels: 1.5 ms)
    from csmpn_hip.graphed import flatten_parameters
    opt_params = model.parameters() if args.no_flat_params else [flatten_parameters(model)]
    opt = torch.optim.Adam(opt_params, lr=1e-3, capturable=True,
                           **({"foreach": True} if args.no_fused_adam else {"fused": True}))

    def eager():
        opt.zero_grad(set_to_none=False)
        loss, _ = model(batch)
        loss.backward()
        opt.step()
        return loss

    for _ in range(5):
        eager()
    torch.cuda.synchronize()
    if args.op_sites:
        op_sites(eager)
        return
    if args.profile_ops:
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
            for _ in range(3):
                eager()
            torch.cuda.synchronize()
        print(prof.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=50,
                                                            max_src_column_width=110))
        return
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eager()
    torch.cuda.synchronize()
    eager_ms = (time.perf_counter() - t0) * 1e3 / args.steps

    gs = GraphedTrainStep(model, opt, batch, features)
    for _ in range(5):
        gs.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gs.step()
    torch.cuda.synchronize()
    graph_ms = (time.perf_counter() - t0) * 1e3 / args.steps
    label = {"hulls": "hulls (Cl(5,0), 28 channels, 3 layers)", "md17": "md17 (Cl(3,0), 32 channels, 5 layers)",
             "motion": "motion (Cl(3,0), 16 channels, 4 layers)", "nba": "nba (Cl(2,0), 40 channels, 4 layers)"}[args.model]
    print(json.dumps({"model": label, "graphs_per_batch": args.batch,
                      "simplices": int(batch.x_ind.shape[0]), "adjacencies": int(batch.edge_index.shape[1]),
                      "fused_grad_accumulation": not args.no_fused_grads, "fused_adam": not args.no_fused_adam, "flat_parameters": not args.no_flat_params, "eager_ms_per_step": round(eager_ms, 3), "graphed_ms_per_step": round(graph_ms, 3),
                      "loss": float(gs.loss.detach())}))


if __name__ == "__main__":
    main()
