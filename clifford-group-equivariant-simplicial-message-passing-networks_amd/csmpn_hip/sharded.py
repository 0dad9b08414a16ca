"""Edge-sharded EGCL over the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference never shards a layer (its only multi-GPU mode is whole-model DDP,
csmpn/md17.py:15-20); this is the partitioning BASELINE.json's north_star names:

  * the edge / simplex-adjacency list is split into contiguous shards, one per rank;
    node features h, node_attr and all parameters are replicated;
  * forward: every rank runs the fused edge kernel on its shard -> partial
    agg[N, O, D]; ONE all-reduce (sum) over the per-node aggregated features; the
    node update then runs replicated (identical on every rank);
  * backward: node backward replicated; edge backward on the shard -> partial
    d/dh[N, C, D] and partial edge-model parameter gradients; ONE all-reduce over
    [d/dh | edge-model gradients] packed in a single buffer.
  * mean aggregation uses the GLOBAL in-degree (all-reduced once per complex).

The data path of a shard has no other exchange step. The compute backend is
injectable so that the collective plumbing is testable on CPU with gloo (tests
inject the oracle there); the default backend is the HIP C-ABI.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import ops


def shard_bounds(n_edges: int, world: int, rank: int):
    """Contiguous, balanced shard [lo, hi) of the edge list for `rank`."""
    base, rem = divmod(n_edges, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ShardPlan:
    """Per-complex state of one rank: local CSR + global in-degree."""

    def __init__(self, edge_index_local, n_nodes, backend=ops.HipBackend, group=None):
        self.csr = backend.build_csr(edge_index_local, n_nodes)
        deg = self.csr.deg.clone()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(deg, op=dist.ReduceOp.SUM, group=group)
        self.deg = deg
        self.n_nodes = n_nodes


class _ShardedEgclFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, edge_attr, node_attr, spec, plan: ShardPlan, backend, group, *params):
        h = h.contiguous()
        ne = spec.edge.nblk * ops.NP
        pe, pn = params[:ne], params[ne:]
        agg, st_e = backend.edge_forward(spec, plan.csr, h, edge_attr, pe)
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(agg, op=dist.ReduceOp.SUM, group=group)
        out, st_n = backend.node_forward(spec, plan.deg, h, agg, node_attr, pn)
        ctx.st_e, ctx.st_n = st_e, st_n
        ctx.spec, ctx.plan, ctx.backend, ctx.group = spec, plan, backend, group
        ctx.has_ea, ctx.has_na = edge_attr is not None, node_attr is not None
        ctx.mask = [p is not None for p in params]
        saved = [h, agg] + ([edge_attr] if ctx.has_ea else []) + ([node_attr] if ctx.has_na else [])
        ctx.save_for_backward(*saved, *[p for p in params if p is not None])
        return out

    @staticmethod
    def backward(ctx, gout):
        spec, plan, backend, group = ctx.spec, ctx.plan, ctx.backend, ctx.group
        saved = list(ctx.saved_tensors)
        h, agg = saved[0], saved[1]
        pos = 2
        edge_attr = node_attr = None
        if ctx.has_ea:
            edge_attr = saved[pos]; pos += 1
        if ctx.has_na:
            node_attr = saved[pos]; pos += 1
        it = iter(saved[pos:])
        params = [next(it) if m else None for m in ctx.mask]
        ne = spec.edge.nblk * ops.NP
        pe, pn = params[:ne], params[ne:]
        gout = gout.contiguous()
        gh_node, g_agg, g_na, views_n = backend.node_backward(spec, plan.deg, h, agg, node_attr, pn, gout,
                                                              ctx.needs_input_grad[2], ctx.st_n)
        # partial d/dh of this shard starts from zero so that the all-reduce sums shards only
        gh_edge = torch.zeros_like(h)
        g_ea, views_e = backend.edge_backward(spec, plan.csr, h, edge_attr, pe, g_agg, gh_edge,
                                              ctx.needs_input_grad[1], ctx.st_e)
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            # one collective: [d/dh | edge-model parameter gradients]
            pieces = [gh_edge.reshape(-1)] + [v.reshape(-1) for v in views_e if v is not None]
            packed = torch.cat(pieces)
            dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
            off = gh_edge.numel()
            gh_edge = packed[:off].view_as(h)
            out_views = []
            for v in views_e:
                if v is None:
                    out_views.append(None)
                else:
                    out_views.append(packed[off:off + v.numel()].view(v.shape))
                    off += v.numel()
            views_e = out_views
        gh = gh_node + gh_edge
        return (gh, g_ea, g_na, None, None, None, None, *views_e, *views_n)


class ShardedEGCL(torch.nn.Module):
    """Wraps an EGCL module; forward takes this rank's shard of the edge list."""

    def __init__(self, layer, backend=ops.HipBackend, group=None):
        super().__init__()
        self.layer = layer
        self.backend = backend
        self.group = group

    def plan(self, edge_index_local, n_nodes) -> ShardPlan:
        return ShardPlan(edge_index_local, n_nodes, self.backend, self.group)

    def forward(self, h, plan: ShardPlan, edge_attr_local=None, node_attr=None):
        layer = self.layer
        params = layer.edge_model.flat_params() + layer.node_model.flat_params()
        return _ShardedEgclFn.apply(h, edge_attr_local, node_attr, layer.spec(), plan, self.backend, self.group,
                                    *params)


class GraphedShardedStep:
    """Forward + backward of the sharded layer on FIXED buffers, for steady-state training loops
    and the multi-GPU benchmark: the compute stages are captured in two HIP graphs, the two
    collectives of the partitioning are launched eagerly between them (no collective inside a
    captured graph):

        [graph 1: edge forward] -> all_reduce(agg)
        -> [graph 2: node forward, node backward, edge backward, pack] -> all_reduce([d/dh | edge grads])
        -> d/dh = node part + reduced edge part

    Same stage calls, same collectives and same results as `_ShardedEgclFn`; what it removes is
    the per-step Python / autograd launch path (0.54 ms per step eager vs 0.47 ms of GPU time on
    S1), which would otherwise bound the N-GPU step. Inputs are read from the tensors given here:
    update them in place between `run()` calls."""

    def __init__(self, sharded_layer: "ShardedEGCL", plan: ShardPlan, h, edge_attr_local, node_attr, gout):
        layer, be = sharded_layer.layer, sharded_layer.backend
        self.group = sharded_layer.group
        self.spec = spec = layer.spec()
        self.pe = layer.edge_model.flat_params()
        self.pn = layer.node_model.flat_params()
        self.h, self.ea, self.na, self.gout = h.detach(), edge_attr_local, node_attr, gout
        self._multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1
        h_, ea, na, pe, pn = self.h, self.ea, self.na, self.pe, self.pn

        # N > 1: the edge forward runs as two launches over the two halves of the target-sorted
        # shard (targets < N/2 | >= N/2), so that the all-reduce of the first half of the aggregate
        # travels while the second half is still being computed. Needs the saved block inputs to be
        # one [rows, O, D] array (two-block edge model) and the default backend.
        import os
        self._split = None
        if (self._multi and be is ops.HipBackend and spec.edge.nblk == 2
                and os.environ.get("CSMPN_SPLIT_FWD", "1") != "0"):
            n_half = plan.n_nodes // 2
            e1 = int(plan.csr.row_ptr[n_half].item())
            if 0 < e1 < plan.csr.n_edges:
                self._split = (n_half, e1)

        def part1():
            return be.edge_forward(spec, plan.csr, h_, ea, pe)

        def part1_lo():
            n_half, e1 = self._split
            spec.edge.bind(pe)
            agg = torch.zeros(h_.shape[0], spec.O, spec.edge.D, dtype=torch.float32, device=h_.device)
            saved = spec.edge.new_saved(plan.csr.n_edges, h_.device)
            per_row = saved.numel() // plan.csr.n_edges
            _, (ws, _s) = be.edge_forward(spec, ops.CsrSlice(plan.csr, 0, e1), h_, ea, pe, agg=agg,
                                          saved=saved[:e1 * per_row])
            return agg, (ws, saved), per_row

        def part1_hi(agg, saved, per_row):
            n_half, e1 = self._split
            be.edge_forward(spec, ops.CsrSlice(plan.csr, e1, plan.csr.n_edges), h_, ea, pe, agg=agg,
                            saved=saved[e1 * per_row:])

        def part2(agg, st_e):
            out, st_n = be.node_forward(spec, plan.deg, h_, agg, na, pn)
            gh_node, g_agg, _g_na, views_n = be.node_backward(spec, plan.deg, h_, agg, na, pn, self.gout, False, st_n)
            gh_edge = torch.zeros_like(h_)
            _g_ea, views_e = be.edge_backward(spec, plan.csr, h_, ea, pe, g_agg, gh_edge, False, st_e)
            packed = torch.cat([gh_edge.reshape(-1)] + [v.reshape(-1) for v in views_e if v is not None])
            return out, gh_node, packed, views_e, views_n

        # warm-up outside capture (kernel attributes, workspaces, allocator)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            if self._split:
                agg, st_e, per_row = part1_lo()
                part1_hi(agg, st_e[1], per_row)
            else:
                agg, st_e = part1()
            part2(agg, st_e)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()

        self.g1, self.g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        # thread-local capture mode: the process group's watchdog thread polls events while we capture
        self.g1b = None
        with torch.cuda.graph(self.g1, capture_error_mode="thread_local"):
            if self._split:
                self.agg, self._st_e, per_row = part1_lo()
            else:
                self.agg, self._st_e = part1()
        if self._split:
            self.g1b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g1b, pool=self.g1.pool(), capture_error_mode="thread_local"):
                part1_hi(self.agg, self._st_e[1], per_row)
        with torch.cuda.graph(self.g2, pool=self.g1.pool(), capture_error_mode="thread_local"):
            self.out, self.gh_node, self.packed, views_e, self.views_n = part2(self.agg, self._st_e)
        self._edge_shapes = [None if v is None else tuple(v.shape) for v in views_e]
        self.gh = torch.empty_like(h_)

    def run(self):
        self.g1.replay()
        if self._split:
            n_half = self._split[0]
            w_lo = dist.all_reduce(self.agg[:n_half], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.g1b.replay()   # writes rows >= n_half only
            w_hi = dist.all_reduce(self.agg[n_half:], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            w_lo.wait()
            w_hi.wait()
        elif self._multi:
            dist.all_reduce(self.agg, op=dist.ReduceOp.SUM, group=self.group)
        self.g2.replay()
        if self._multi:
            dist.all_reduce(self.packed, op=dist.ReduceOp.SUM, group=self.group)
        n = self.h.numel()
        torch.add(self.gh_node, self.packed[:n].view_as(self.h), out=self.gh)

    def results(self):
        """(out, d/dh, edge-model parameter gradients, node-model parameter gradients) of the last run()."""
        off, views_e = self.h.numel(), []
        for shp in self._edge_shapes:
            if shp is None:
                views_e.append(None)
            else:
                cnt = 1
                for s in shp:
                    cnt *= s
                views_e.append(self.packed[off:off + cnt].view(shp))
                off += cnt
        return self.out, self.gh, views_e, list(self.views_n)
