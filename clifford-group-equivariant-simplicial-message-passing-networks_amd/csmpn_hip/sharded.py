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

Partitioning B (`DstPartitionedEGCL`): rank r owns the nodes
[r N/W, (r+1) N/W) and ALL edges into them. Its aggregate is complete without any
reduction and the node update runs on its own nodes only, so nothing is computed
twice (partitioning A replicates the node update: an Amdahl bound of ~4x at 8 GPUs
with the S2 stage times). Exchanges per layer: forward ONE all-gather of the updated
node slices, backward ONE reduce-scatter of d/dh (the -g -> source half lands on any
node) plus an all-reduce of the parameter gradients (a few thousand floats) - half
the bytes of A, and both collectives use all xGMI links at once.
"""
from __future__ import annotations

import os

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
        if _multi(group):
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
        if _multi(group):
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
        if _multi(group):
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
        self._multi = _multi(self.group)
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
            # the choice changes which collectives a rank issues (two half-size all-reduces or one
            # whole one): it must be the same on every rank - split only if EVERY shard straddles N/2
            if agree_all(0 < e1 < plan.csr.n_edges, self.group, h.device):
                self._split = (n_half, e1)

        def part1():
            return be.edge_forward(spec, plan.csr, h_, ea, pe)

        def part1_lo():
            n_half, e1 = self._split
            spec.edge.bind(pe)
            agg = torch.zeros(h_.shape[0], spec.O, spec.edge.D, dtype=torch.float32, device=h_.device)
            saved = spec.edge.new_saved(plan.csr.n_edges, h_.device)
            # floats of one saved block-1 input row (NOT numel / rows: the buffer may carry a backward scratch region
            # behind the [rows, O, D] inputs, see csmpn_cemlp_saved_floats_per_row)
            per_row = int(spec.edge.params[0].out_features) * spec.edge.D
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


# ===================================================================================== partitioning B


def node_bounds(n_nodes: int, world: int, rank: int):
    """Equal-width node slice of `rank` (the last slices are one shorter when N does not divide by the world size)."""
    base, rem = divmod(n_nodes, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def balanced_node_cuts(in_degree: torch.Tensor, world: int, boundaries=None):
    """W + 1 cut points of the node range such that every slice receives about E / W adjacencies (prefix sum of the
    in-degree): the edge stage is the expensive one, so slices are balanced by incoming edges, not by node count. A hub
    may leave a slice empty; cuts are non-decreasing and cover [0, N].
    boundaries (sorted node positions, e.g. the `ptr` of a collated batch of complexes): every inner cut moves to the
    nearest allowed position - no graph straddles two slices, so every adjacency has its source in the slice of its
    target and partition B exchanges nothing between layers (DstPlan.local_share == 1)."""
    n = int(in_degree.shape[0])
    if world == 1:
        return [0, n]
    csum = torch.cumsum(in_degree.to(torch.int64).cpu(), 0)
    total = int(csum[-1]) if n else 0
    if total == 0:
        return [node_bounds(n, world, r)[0] for r in range(world)] + [n]
    targets = torch.tensor([total * r // world for r in range(1, world)], dtype=torch.int64)
    inner = torch.searchsorted(csum, targets, right=False).tolist()   # first node whose prefix reaches the target
    cuts = [0] + [min(max(int(c), 0), n) for c in inner] + [n]
    if boundaries is not None:
        b = torch.as_tensor(boundaries, dtype=torch.int64).cpu().flatten()
        b = torch.unique(torch.cat([b, torch.tensor([0, n], dtype=torch.int64)]))
        for i in range(1, len(cuts) - 1):
            j = int(torch.searchsorted(b, torch.tensor(cuts[i])))
            lo, hi = int(b[max(j - 1, 0)]), int(b[min(j, len(b) - 1)])
            cuts[i] = lo if cuts[i] - lo <= hi - cuts[i] else hi
    for i in range(1, len(cuts)):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return cuts


def locality_order(edge_index: torch.Tensor, n_nodes: int) -> torch.Tensor:
    """A node order for partition B on ONE large complex: reverse Cuthill-McKee over the (symmetrised) adjacency, so that
    the two ends of an adjacency get nearby numbers and contiguous node slices keep most sources local (the overlapped
    stack hides the exchange under exactly those edges). Returns perm with new_id = perm[old_id]; apply it to edge_index
    (perm[edge_index]) and scatter the node rows (h_new[perm] = h). Host-side, once per complex (scipy). Collated batches
    of small complexes need none of this: their nodes are grouped by graph already - cut at graph boundaries instead."""
    import numpy as np
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    ei = edge_index.detach().cpu().numpy()
    a = coo_matrix((np.ones(ei.shape[1], dtype=np.int8), (ei[0], ei[1])), shape=(n_nodes, n_nodes)).tocsr()
    order = reverse_cuthill_mckee((a + a.T).tocsr(), symmetric_mode=True)   # order[new] = old
    perm = np.empty(n_nodes, dtype=np.int64)
    perm[order] = np.arange(n_nodes, dtype=np.int64)
    return torch.from_numpy(perm)


def _world(group):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def _multi(group=None) -> bool:
    """True when the collectives have to be issued: a process group of more than one rank - or of ONE rank with
    CSMPN_FORCE_COLLECTIVES=1. The second form is the RCCL rehearsal of tests/test_sharded_gpu.py (round 4): a one-GPU box
    cannot hold two RCCL ranks, but with a world-size-1 `nccl` group every all_reduce / all_gather_into_tensor /
    reduce_scatter_tensor of this module is posted to RCCL exactly as it would be on N ranks (padded layouts included)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("CSMPN_FORCE_COLLECTIVES", "0") == "1"


def _collective_device(group, fallback):
    """Device a small bookkeeping tensor must live on to go through the group's collectives: NCCL = RCCL moves device
    memory only (a CPU edge_index would otherwise fail in DstPlan / ShardPlan), gloo takes either."""
    if dist.is_available() and dist.is_initialized() and "nccl" in str(dist.get_backend(group)).lower():
        return torch.device("cuda", torch.cuda.current_device())
    return fallback


def _fused_collectives(group) -> bool:
    """The single-tensor collectives (all_gather_into_tensor / reduce_scatter_tensor) exist on NCCL = RCCL; gloo takes the
    list / all-reduce forms. Decided ONCE from the backend name: a per-call try / except would let a rank-local
    error (an OOM, an asynchronous failure) send one rank to a different collective than its peers."""
    return "nccl" in str(dist.get_backend(group)).lower()


def _all_gather_rows(full, part, group):
    """full[W * n] <- concat over ranks of part[n] (rows)."""
    if _fused_collectives(group):
        dist.all_gather_into_tensor(full, part, group=group)
    else:
        chunks = list(full.chunk(dist.get_world_size(group), dim=0))
        dist.all_gather(chunks, part, group=group)


def _reduce_scatter_rows(part, full, group):
    """part[n] <- this rank's row slice of the sum over ranks of full[W * n]."""
    if _fused_collectives(group):
        dist.reduce_scatter_tensor(part, full, op=dist.ReduceOp.SUM, group=group)
    else:   # gloo has no reduce-scatter: all-reduce and slice
        dist.all_reduce(full, op=dist.ReduceOp.SUM, group=group)
        w, r = dist.get_world_size(group), dist.get_rank(group)
        part.copy_(full.chunk(w, dim=0)[r])


class _Pending:
    """Handle of a collective in flight (async_op=True): wait() orders the current stream behind it, then runs the
    epilogue the blocking form would have run (gloo's reduce-scatter stand-in slices the all-reduced rows)."""

    def __init__(self, works, epilogue=None):
        self.works, self.epilogue = works, epilogue

    def wait(self):
        for w in self.works:
            w.wait()
        if self.epilogue is not None:
            self.epilogue()


def _all_gather_rows_async(full, part, group) -> _Pending:
    if _fused_collectives(group):
        return _Pending([dist.all_gather_into_tensor(full, part, group=group, async_op=True)])
    chunks = list(full.chunk(dist.get_world_size(group), dim=0))
    return _Pending([dist.all_gather(chunks, part, group=group, async_op=True)])


def _reduce_scatter_rows_async(part, full, group) -> _Pending:
    if _fused_collectives(group):
        return _Pending([dist.reduce_scatter_tensor(part, full, op=dist.ReduceOp.SUM, group=group, async_op=True)])
    w, r = dist.get_world_size(group), dist.get_rank(group)
    return _Pending([dist.all_reduce(full, op=dist.ReduceOp.SUM, group=group, async_op=True)],
                    lambda: part.copy_(full.chunk(w, dim=0)[r]))


def agree_all(local_ok: bool, group=None, device=None) -> bool:
    """True iff the condition holds on EVERY rank (all-reduce MIN of a flag). For choices that change which
    collectives a rank issues: they must come out the same everywhere, or the ranks post mismatched collectives."""
    if not _multi(group):
        return bool(local_ok)
    flag = torch.tensor([1 if local_ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return int(flag.item()) == 1


class DstPlan:
    """Per-complex state of one rank for partitioning B: the edges into its node slice (global node ids), their ids in
    the caller's edge order (for edge_attr), the slice cuts of every rank (balanced by in-degree unless `balance=False`)
    and the index tables that move rows between the natural layout [N] and the padded layout [W * per] the
    equal-size collectives need (per = longest slice; pad rows read a zero row / are dropped)."""

    def __init__(self, edge_index, n_nodes, backend=ops.HipBackend, group=None, balance=True, max_pad=1.5, boundaries=None):
        self.world, self.rank = _world(group)
        self.multi = _multi(group)      # collectives are issued (world > 1, or the forced world-size-1 rehearsal)
        dev = edge_index.device
        dst = edge_index[1]
        if balance and self.world > 1:
            deg_all = torch.bincount(dst, minlength=n_nodes)
            self.cuts = balanced_node_cuts(deg_all, self.world, boundaries)
            # The collectives move the padded layout W x per (per = longest slice in NODES): on complexes whose in-degree
            # is skewed (vertex / edge / triangle rows) cuts by in-degree alone let per grow towards N. Bound it: no slice
            # longer than max_pad x N / W nodes (round-3 ADVICE) - the edge balance gives way first.
            cap = max(1, int(-(-n_nodes * max_pad // self.world)))
            if boundaries is None:   # (graph-aligned cuts stay where the graphs end)
                for r in range(1, self.world):          # left to right: a slice that is too long hands nodes to the next one
                    self.cuts[r] = min(self.cuts[r], self.cuts[r - 1] + cap)
                for r in range(self.world - 1, 0, -1):  # ... and right to left for the last slices
                    self.cuts[r] = max(self.cuts[r], self.cuts[r + 1] - cap)
        else:
            self.cuts = [node_bounds(n_nodes, self.world, r)[0] for r in range(self.world)] + [n_nodes]
        self.lo, self.hi = self.cuts[self.rank], self.cuts[self.rank + 1]
        self.per = max(1, max(self.cuts[r + 1] - self.cuts[r] for r in range(self.world)))
        self.pad_ratio = self.world * self.per / max(1, n_nodes)   # rows the collectives move / rows that exist
        mine = (dst >= self.lo) & (dst < self.hi)
        self.edge_ids = torch.nonzero(mine, as_tuple=False).squeeze(1)
        self.csr = backend.build_csr(edge_index[:, self.edge_ids].contiguous(), n_nodes)
        self.deg = self.csr.deg     # every edge into an owned node is local: the local in-degree is the global one
        self.n_nodes = n_nodes
        # The owned edges split by SOURCE locality (positions in the owned list): an edge whose source is owned too needs
        # no row of another rank - the layer's edge stage on these runs under the all-gather of the previous layer's
        # output, their d/dh stays out of the reduce-scatter (DstPartitionedStack(overlap=True))
        src_own = edge_index[0, self.edge_ids]
        loc = (src_own >= self.lo) & (src_own < self.hi)
        self.idx_loc = torch.nonzero(loc, as_tuple=False).squeeze(1)
        self.idx_rem = torch.nonzero(~loc, as_tuple=False).squeeze(1)
        # share of the owned adjacencies whose source row is owned too (1.0: the layer needs no row of another rank)
        self.local_share = float(self.idx_loc.numel()) / max(1, int(self.edge_ids.numel()))
        self._split = None
        self._backend = backend
        self._edge_index_owned = edge_index[:, self.edge_ids]
        # padded position of every node row / node row (or N = the zero row) of every padded position
        pos = torch.empty(n_nodes, dtype=torch.int64)
        src = torch.full((self.world * self.per,), n_nodes, dtype=torch.int64)
        for r in range(self.world):
            lo, hi = self.cuts[r], self.cuts[r + 1]
            pos[lo:hi] = torch.arange(r * self.per, r * self.per + (hi - lo))
            src[r * self.per: r * self.per + (hi - lo)] = torch.arange(lo, hi)
        self.pos_of_node = pos.to(dev)
        self.node_of_pos = src.to(dev)
        self.edges_per_rank = None
        if self.multi:   # a collective: plan() must be called on every rank of the group
            cnt = torch.zeros(self.world, dtype=torch.int64, device=_collective_device(group, dev))
            cnt[self.rank] = int(self.edge_ids.numel())
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)
            self.edges_per_rank = cnt.tolist()

    def split(self):
        """(csr of the local-source edges, csr of the remote-source edges), built on first use."""
        if self._split is None:
            ei = self._edge_index_owned
            self._split = (self._backend.build_csr(ei[:, self.idx_loc].contiguous(), self.n_nodes),
                           self._backend.build_csr(ei[:, self.idx_rem].contiguous(), self.n_nodes))
        return self._split

    # ---- rows between the layouts (one index kernel each)
    def pad_slice(self, rows_loc):
        """[hi - lo, ...] -> [per, ...] (zero pad rows)."""
        n = self.hi - self.lo
        if n == self.per:
            return rows_loc.contiguous()
        out = rows_loc.new_zeros((self.per,) + tuple(rows_loc.shape[1:]))
        out[:n] = rows_loc
        return out

    def to_padded(self, rows_all):
        """[N, ...] -> [W * per, ...] (pad positions zero): the input layout of the reduce-scatter."""
        z = torch.cat([rows_all, rows_all.new_zeros((1,) + tuple(rows_all.shape[1:]))], dim=0)
        return z.index_select(0, self.node_of_pos)

    def from_padded(self, padded):
        """[W * per, ...] -> [N, ...]: the output layout of the all-gather."""
        return padded.index_select(0, self.pos_of_node)


class _DstPartStackFn(torch.autograd.Function):
    """L chained EGCL layers, destination-partitioned. h [N, C, D] replicated in, the last layer's output replicated out.
    Forward, per layer: edge stage on the owned edges -> node update of the owned slice -> ONE all-gather. Backward, per
    layer: node backward on the slice -> edge backward on the owned edges -> ONE reduce-scatter of d/dh, whose result
    (+ the node stage's d/dh) is exactly the slice of d/d(previous layer's output) the next step needs - no all-gather
    between chained layers; the replicated d/dh of the chain's input is all-gathered once, if anybody asks for it. The
    parameter gradients of all layers travel in ONE all-reduce."""

    @staticmethod
    def forward(ctx, h, edge_attr_local, node_attr, specs, plan: DstPlan, backend, group, counts, *params):
        lo, hi = plan.lo, plan.hi
        na_loc = None if node_attr is None else node_attr[lo:hi].contiguous()
        deg_loc = plan.deg[lo:hi].contiguous()
        hs, aggs, st = [], [], []
        x = h.contiguous()
        off = 0
        for spec, cnt in zip(specs, counts):
            lp = params[off:off + cnt]
            off += cnt
            ne = spec.edge.nblk * ops.NP
            pe, pn = lp[:ne], lp[ne:]
            agg, st_e = backend.edge_forward(spec, plan.csr, x, edge_attr_local, pe)      # complete on [lo, hi)
            out_loc, st_n = backend.node_forward(spec, deg_loc, x[lo:hi].contiguous(), agg[lo:hi].contiguous(), na_loc, pn)
            hs.append(x); aggs.append(agg); st.append((st_e, st_n))
            if plan.multi:
                padded = out_loc.new_empty((plan.world * plan.per,) + tuple(out_loc.shape[1:]))
                _all_gather_rows(padded, plan.pad_slice(out_loc), group)
                x = plan.from_padded(padded)
            else:
                x = out_loc
        ctx.st = st
        ctx.specs, ctx.plan, ctx.backend, ctx.group, ctx.counts = specs, plan, backend, group, counts
        ctx.has_ea, ctx.has_na = edge_attr_local is not None, node_attr is not None
        ctx.mask = [p is not None for p in params]
        ctx.n_layers = len(specs)
        saved = hs + aggs + ([edge_attr_local] if ctx.has_ea else []) + ([node_attr] if ctx.has_na else [])
        ctx.save_for_backward(*saved, *[p for p in params if p is not None])
        return x

    @staticmethod
    def backward(ctx, gout):
        specs, plan, backend, group, counts = ctx.specs, ctx.plan, ctx.backend, ctx.group, ctx.counts
        L = ctx.n_layers
        saved = list(ctx.saved_tensors)
        hs, aggs = saved[:L], saved[L:2 * L]
        pos = 2 * L
        edge_attr = node_attr = None
        if ctx.has_ea:
            edge_attr = saved[pos]; pos += 1
        if ctx.has_na:
            node_attr = saved[pos]; pos += 1
        it = iter(saved[pos:])
        params = [next(it) if m else None for m in ctx.mask]
        lo, hi = plan.lo, plan.hi
        na_loc = None if node_attr is None else node_attr[lo:hi].contiguous()
        deg_loc = plan.deg[lo:hi].contiguous()
        offs = [0]
        for cnt in counts:
            offs.append(offs[-1] + cnt)
        g_loc = gout[lo:hi].contiguous()          # d/d(output slice) of the layer being processed
        views_all = [None] * len(params)
        g_ea_total, g_na_loc_total = None, None
        gh_full_single = None
        for k in range(L - 1, -1, -1):
            spec = specs[k]
            lp = params[offs[k]:offs[k + 1]]
            ne = spec.edge.nblk * ops.NP
            pe, pn = lp[:ne], lp[ne:]
            h, agg = hs[k], aggs[k]
            st_e, st_n = ctx.st[k]
            gh_node, g_agg_loc, g_na_loc, views_n = backend.node_backward(
                spec, deg_loc, h[lo:hi].contiguous(), agg[lo:hi].contiguous(), na_loc, pn, g_loc,
                ctx.needs_input_grad[2], st_n)
            g_agg = torch.zeros_like(agg)
            g_agg[lo:hi] = g_agg_loc
            gh_edge = torch.zeros_like(h)          # +g -> owned targets, -g -> any source
            g_ea, views_e = backend.edge_backward(spec, plan.csr, h, edge_attr, pe, g_agg, gh_edge,
                                                  ctx.needs_input_grad[1], st_e)
            for i, v in enumerate(list(views_e) + list(views_n)):
                views_all[offs[k] + i] = v
            if g_ea is not None:
                g_ea_total = g_ea if g_ea_total is None else g_ea_total + g_ea
            if g_na_loc is not None:
                g_na_loc_total = g_na_loc if g_na_loc_total is None else g_na_loc_total + g_na_loc
            if plan.multi:
                g_pad = gh_edge.new_empty((plan.per,) + tuple(gh_edge.shape[1:]))
                _reduce_scatter_rows(g_pad, plan.to_padded(gh_edge), group)
                g_loc = g_pad[:hi - lo] + gh_node
            else:
                gh_edge[lo:hi] += gh_node
                gh_full_single = gh_edge
                g_loc = gh_edge[lo:hi]
        gh = None
        if ctx.needs_input_grad[0]:
            if plan.multi:
                padded = g_loc.new_empty((plan.world * plan.per,) + tuple(g_loc.shape[1:]))
                _all_gather_rows(padded, plan.pad_slice(g_loc), group)
                gh = plan.from_padded(padded)
            else:
                gh = gh_full_single
        g_na = None
        if plan.multi:
            live = [v for v in views_all if v is not None]
            if live:
                flat = torch.cat([v.reshape(-1) for v in live])
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
                off, red = 0, []
                for v in views_all:
                    if v is None:
                        red.append(None)
                    else:
                        red.append(flat[off:off + v.numel()].view(v.shape))
                        off += v.numel()
                views_all = red
            if g_na_loc_total is not None:
                g_na = torch.zeros_like(node_attr)
                g_na[lo:hi] = g_na_loc_total
                dist.all_reduce(g_na, op=dist.ReduceOp.SUM, group=group)
        else:
            g_na = g_na_loc_total
        return (gh, g_ea_total, g_na, None, None, None, None, None, *views_all)


class _DstPartStackOverlapFn(torch.autograd.Function):
    """_DstPartStackFn with the collectives of the chain hidden under edge work (world > 1). Every layer's edge stage
    runs in two launches: the owned edges whose source is owned too (they read rows of the own slice only) and the rest.
    Forward: the all-gather of layer k's output slice is in flight while layer k + 1 runs its local-source edges on the
    slice it already has; the remote-source edges follow the wait. Backward: the remote-source edges go first, the
    reduce-scatter of THEIR d/dh (the only part with rows of other ranks) travels while the local-source edges run; the
    local part lands on owned rows and is added behind the wait. Same results as the blocking chain up to the order of
    two float additions per row."""

    @staticmethod
    def forward(ctx, h, edge_attr_local, node_attr, specs, plan: DstPlan, backend, group, counts, *params):
        lo, hi = plan.lo, plan.hi
        csr_l, csr_r = plan.split()
        ea_l = None if edge_attr_local is None else edge_attr_local.index_select(0, plan.idx_loc)
        ea_r = None if edge_attr_local is None else edge_attr_local.index_select(0, plan.idx_rem)
        na_loc = None if node_attr is None else node_attr[lo:hi].contiguous()
        deg_loc = plan.deg[lo:hi].contiguous()
        hs, aggs, st = [], [], []
        x = h.contiguous()
        pending, padded, x_own = None, None, None
        off = 0
        for spec, cnt in zip(specs, counts):
            lp = params[off:off + cnt]
            off += cnt
            ne = spec.edge.nblk * ops.NP
            pe, pn = lp[:ne], lp[ne:]
            if pending is None:
                agg_l, st_l = backend.edge_forward(spec, csr_l, x, ea_l, pe)
            else:
                # rows outside [lo, hi) of x_own are never read by the local-source edges
                agg_l, st_l = backend.edge_forward(spec, csr_l, x_own, ea_l, pe)
                pending.wait()
                x = plan.from_padded(padded)
                pending = None
            agg_r, st_r = backend.edge_forward(spec, csr_r, x, ea_r, pe)
            agg = agg_l + agg_r                                                       # complete on [lo, hi)
            out_loc, st_n = backend.node_forward(spec, deg_loc, x[lo:hi].contiguous(), agg[lo:hi].contiguous(), na_loc, pn)
            hs.append(x); aggs.append(agg); st.append((st_l, st_r, st_n))
            padded = out_loc.new_empty((plan.world * plan.per,) + tuple(out_loc.shape[1:]))
            pending = _all_gather_rows_async(padded, plan.pad_slice(out_loc), group)
            x_own = out_loc.new_zeros((plan.n_nodes,) + tuple(out_loc.shape[1:]))
            x_own[lo:hi] = out_loc
        pending.wait()
        x = plan.from_padded(padded)
        ctx.st = st
        ctx.specs, ctx.plan, ctx.backend, ctx.group, ctx.counts = specs, plan, backend, group, counts
        ctx.has_ea, ctx.has_na = edge_attr_local is not None, node_attr is not None
        ctx.mask = [p is not None for p in params]
        ctx.n_layers = len(specs)
        saved = hs + aggs + ([ea_l, ea_r] if ctx.has_ea else []) + ([node_attr] if ctx.has_na else [])
        ctx.save_for_backward(*saved, *[p for p in params if p is not None])
        return x

    @staticmethod
    def backward(ctx, gout):
        specs, plan, backend, group, counts = ctx.specs, ctx.plan, ctx.backend, ctx.group, ctx.counts
        L = ctx.n_layers
        saved = list(ctx.saved_tensors)
        hs, aggs = saved[:L], saved[L:2 * L]
        pos = 2 * L
        ea_l = ea_r = node_attr = None
        if ctx.has_ea:
            ea_l, ea_r = saved[pos], saved[pos + 1]; pos += 2
        if ctx.has_na:
            node_attr = saved[pos]; pos += 1
        it = iter(saved[pos:])
        params = [next(it) if m else None for m in ctx.mask]
        lo, hi = plan.lo, plan.hi
        csr_l, csr_r = plan.split()
        na_loc = None if node_attr is None else node_attr[lo:hi].contiguous()
        deg_loc = plan.deg[lo:hi].contiguous()
        offs = [0]
        for cnt in counts:
            offs.append(offs[-1] + cnt)
        g_loc = gout[lo:hi].contiguous()
        views_all = [None] * len(params)
        g_ea_l_tot = g_ea_r_tot = g_na_loc_total = None

        def acc(a, b):
            return b if a is None else (a if b is None else a + b)

        for k in range(L - 1, -1, -1):
            spec = specs[k]
            lp = params[offs[k]:offs[k + 1]]
            ne = spec.edge.nblk * ops.NP
            pe, pn = lp[:ne], lp[ne:]
            h, agg = hs[k], aggs[k]
            st_l, st_r, st_n = ctx.st[k]
            gh_node, g_agg_loc, g_na_loc, views_n = backend.node_backward(
                spec, deg_loc, h[lo:hi].contiguous(), agg[lo:hi].contiguous(), na_loc, pn, g_loc,
                ctx.needs_input_grad[2], st_n)
            g_agg = torch.zeros_like(agg)
            g_agg[lo:hi] = g_agg_loc
            gh_r = torch.zeros_like(h)             # +g -> owned targets, -g -> sources on OTHER ranks
            g_ea_r, views_r = backend.edge_backward(spec, csr_r, h, ea_r, pe, g_agg, gh_r, ctx.needs_input_grad[1], st_r)
            g_pad = gh_r.new_empty((plan.per,) + tuple(gh_r.shape[1:]))
            pending = _reduce_scatter_rows_async(g_pad, plan.to_padded(gh_r), group)
            gh_l = torch.zeros_like(h)             # both ends owned: rows of [lo, hi) only
            g_ea_l, views_l = backend.edge_backward(spec, csr_l, h, ea_l, pe, g_agg, gh_l, ctx.needs_input_grad[1], st_l)
            views_e = [acc(a, b) for a, b in zip(views_l, views_r)]
            for i, v in enumerate(list(views_e) + list(views_n)):
                views_all[offs[k] + i] = v
            g_ea_l_tot, g_ea_r_tot = acc(g_ea_l_tot, g_ea_l), acc(g_ea_r_tot, g_ea_r)
            g_na_loc_total = acc(g_na_loc_total, g_na_loc)
            pending.wait()
            g_loc = g_pad[:hi - lo] + gh_l[lo:hi] + gh_node
        gh = None
        if ctx.needs_input_grad[0]:
            padded = g_loc.new_empty((plan.world * plan.per,) + tuple(g_loc.shape[1:]))
            _all_gather_rows(padded, plan.pad_slice(g_loc), group)
            gh = plan.from_padded(padded)
        g_ea_total = None
        if g_ea_l_tot is not None or g_ea_r_tot is not None:
            ref = g_ea_l_tot if g_ea_l_tot is not None else g_ea_r_tot
            g_ea_total = ref.new_zeros((plan.edge_ids.numel(),) + tuple(ref.shape[1:]))
            if g_ea_l_tot is not None:
                g_ea_total[plan.idx_loc] = g_ea_l_tot
            if g_ea_r_tot is not None:
                g_ea_total[plan.idx_rem] = g_ea_r_tot
        live = [v for v in views_all if v is not None]
        if live:
            flat = torch.cat([v.reshape(-1) for v in live])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
            off, red = 0, []
            for v in views_all:
                if v is None:
                    red.append(None)
                else:
                    red.append(flat[off:off + v.numel()].view(v.shape))
                    off += v.numel()
            views_all = red
        g_na = None
        if g_na_loc_total is not None:
            g_na = torch.zeros_like(node_attr)
            g_na[lo:hi] = g_na_loc_total
            dist.all_reduce(g_na, op=dist.ReduceOp.SUM, group=group)
        return (gh, g_ea_total, g_na, None, None, None, None, None, *views_all)


class DstPartitionedEGCL(torch.nn.Module):
    """Wraps an EGCL module (partitioning B). `plan(edge_index, n_nodes)` takes the WHOLE edge list
    (replicated topology); forward takes the whole h / node_attr and this rank's rows of edge_attr
    (`edge_attr[plan.edge_ids]`)."""

    def __init__(self, layer, backend=ops.HipBackend, group=None, balance=True):
        super().__init__()
        self.layer = layer
        self.backend = backend
        self.group = group
        self.balance = balance

    def plan(self, edge_index, n_nodes, boundaries=None) -> DstPlan:
        """boundaries: allowed cut positions (the `ptr` of a collated batch): graphs stay whole, see balanced_node_cuts."""
        return DstPlan(edge_index, n_nodes, self.backend, self.group, balance=self.balance, boundaries=boundaries)

    def forward(self, h, plan: DstPlan, edge_attr_local=None, node_attr=None):
        layer = self.layer
        params = layer.edge_model.flat_params() + layer.node_model.flat_params()
        return _DstPartStackFn.apply(h, edge_attr_local, node_attr, (layer.spec(),), plan, self.backend, self.group,
                                     (len(params),), *params)


class DstPartitionedStack(torch.nn.Module):
    """L chained EGCL layers on ONE destination partition of the complex: h stays resident, every
    layer costs one all-gather forward and one reduce-scatter backward, the parameter gradients of all layers one
    all-reduce. All layers share edge_attr / node_attr, as the reference's models do (hulls_cssmpnn.py:89-94).
    overlap=True: the collectives travel under the edges whose source is owned too (two edge launches per stage)."""

    def __init__(self, layers, backend=ops.HipBackend, group=None, balance=True, overlap=False):
        super().__init__()
        self.layers = torch.nn.ModuleList(layers)
        self.backend = backend
        self.group = group
        self.balance = balance
        self.overlap = overlap     # collectives in flight under the local-source edges (_DstPartStackOverlapFn)

    def plan(self, edge_index, n_nodes, boundaries=None) -> DstPlan:
        """boundaries: allowed cut positions (the `ptr` of a collated batch): graphs stay whole, see balanced_node_cuts."""
        return DstPlan(edge_index, n_nodes, self.backend, self.group, balance=self.balance, boundaries=boundaries)

    def forward(self, h, plan: DstPlan, edge_attr_local=None, node_attr=None):
        params, counts, specs = [], [], []
        for layer in self.layers:
            lp = layer.edge_model.flat_params() + layer.node_model.flat_params()
            params += lp
            counts.append(len(lp))
            specs.append(layer.spec())
        fn = _DstPartStackOverlapFn if (self.overlap and plan.multi) else _DstPartStackFn
        return fn.apply(h, edge_attr_local, node_attr, tuple(specs), plan, self.backend, self.group, tuple(counts), *params)


class GraphedDstStep:
    """Forward + backward of the destination-partitioned layer on fixed buffers (multi-GPU benchmark):
    [graph 1: edge forward on the owned targets, node forward on the owned nodes, slice padded] -> all-gather(out)
    -> [graph 2: rows back to the natural layout, node backward, edge backward, d/dh rows to the padded layout] ->
    reduce-scatter(d/dh) + all-reduce(parameter gradients). `compute_only=True` skips the collectives (the
    compute-only rate of the bench)."""

    def __init__(self, part: "DstPartitionedEGCL", plan: DstPlan, h, edge_attr_local, node_attr, gout):
        layer, be = part.layer, part.backend
        self.group, self.plan = part.group, plan
        self.spec = spec = layer.spec()
        pe, pn = layer.edge_model.flat_params(), layer.node_model.flat_params()
        self.h = h_ = h.detach()
        lo, hi = plan.lo, plan.hi
        self._multi = plan.multi
        na_loc = None if node_attr is None else node_attr[lo:hi].contiguous()
        h_loc, deg_loc, gout_loc = h_[lo:hi], plan.deg[lo:hi].contiguous(), gout[lo:hi].contiguous()
        self.out_all = h_.new_empty((plan.world * plan.per,) + tuple(h_.shape[1:]))   # all-gather target (padded layout)

        def part1():
            agg, st_e = be.edge_forward(spec, plan.csr, h_, edge_attr_local, pe)
            out_loc, st_n = be.node_forward(spec, deg_loc, h_loc, agg[lo:hi], na_loc, pn)
            return agg, st_e, plan.pad_slice(out_loc), st_n

        def part2(agg, st_e, st_n):
            out = plan.from_padded(self.out_all) if self._multi else None   # the layer's replicated output
            gh_node, g_agg_loc, _g, views_n = be.node_backward(spec, deg_loc, h_loc, agg[lo:hi], na_loc, pn, gout_loc, False, st_n)
            g_agg = torch.zeros_like(agg)
            g_agg[lo:hi] = g_agg_loc
            gh_edge = torch.zeros_like(h_)
            _g_ea, views_e = be.edge_backward(spec, plan.csr, h_, edge_attr_local, pe, g_agg, gh_edge, False, st_e)
            flat = torch.cat([v.reshape(-1) for v in list(views_e) + list(views_n) if v is not None])
            return gh_node, gh_edge, plan.to_padded(gh_edge) if self._multi else None, flat, out

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            a, se, ol, sn = part1()
            part2(a, se, sn)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.g1, self.g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g1, capture_error_mode="thread_local"):
            self.agg, self._st_e, self.out_pad, self._st_n = part1()
        with torch.cuda.graph(self.g2, pool=self.g1.pool(), capture_error_mode="thread_local"):
            self.gh_node, self.gh_edge, self.gh_pad, self.flat, self.out = part2(self.agg, self._st_e, self._st_n)
        self.gh_rs = h_.new_empty((plan.per,) + tuple(h_.shape[1:]))   # reduce-scatter target: this rank's padded slice
        self.gh_loc = torch.empty_like(self.gh_node)
        self.bytes_per_step = 0
        if self._multi:
            w = plan.world
            # bytes every rank sends per step: all-gather and reduce-scatter move (W-1)/W of the padded tensor
            self.bytes_per_step = int(2 * w * plan.per * h_[0].numel() * 4 * (w - 1) / w + 2 * self.flat.numel() * 4 * (w - 1) / w)

    def run(self, compute_only=False):
        self.g1.replay()
        if self._multi and not compute_only:
            _all_gather_rows(self.out_all, self.out_pad, self.group)
        self.g2.replay()
        n = self.plan.hi - self.plan.lo
        if self._multi and not compute_only:
            _reduce_scatter_rows(self.gh_rs, self.gh_pad, self.group)
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            torch.add(self.gh_rs[:n], self.gh_node, out=self.gh_loc)
        else:
            torch.add(self.gh_edge[self.plan.lo:self.plan.hi], self.gh_node, out=self.gh_loc)


class GraphedDstStackStep:
    """Forward + backward of L chained destination-partitioned layers on fixed buffers, h resident:
    2 L HIP graphs with one collective between consecutive graphs and nothing else on the host -
      forward  k = 0 .. L-1:  [rows of layer k's input back from the padded layout (k > 0), edge forward on the owned
                               edges, node forward on the owned slice, slice padded]  -> all-gather
      backward k = L-1 .. 0:  [d/d(output slice) = reduce-scattered rows + the node stage's d/dh of layer k + 1, node
                               backward, edge backward, d/dh rows to the padded layout] -> reduce-scatter
    and ONE all-reduce of the parameter gradients of all layers at the end. `out` = the replicated output of the last
    layer, `gh_loc` = this rank's slice of d/dh, `flat` = the parameter gradients (layer by layer, edge model then
    node model, in flat_params() order). World size 1: the collectives are copies."""

    def __init__(self, stack: "DstPartitionedStack", plan: DstPlan, h, edge_attr_local, node_attr, gout):
        be, self.group, self.plan = stack.backend, stack.group, plan
        layers = list(stack.layers)
        self.L = L = len(layers)
        specs = [l.spec() for l in layers]
        pes = [l.edge_model.flat_params() for l in layers]
        pns = [l.node_model.flat_params() for l in layers]
        lo, hi = plan.lo, plan.hi
        n = hi - lo
        self._multi = plan.multi
        h0 = h.detach().contiguous()
        na_loc = None if node_attr is None else node_attr[lo:hi].contiguous()
        deg_loc = plan.deg[lo:hi].contiguous()
        gout_loc = gout[lo:hi].contiguous()
        row = tuple(h0.shape[1:])
        self.out_all = [h0.new_empty((plan.world * plan.per,) + row) for _ in range(L)]   # all-gather targets
        self.gh_rs = [h0.new_empty((plan.per,) + row) for _ in range(L)]                  # reduce-scatter targets

        def fwd(k):
            x = h0 if k == 0 else plan.from_padded(self.out_all[k - 1])
            agg, st_e = be.edge_forward(specs[k], plan.csr, x, edge_attr_local, pes[k])
            out_loc, st_n = be.node_forward(specs[k], deg_loc, x[lo:hi].contiguous(), agg[lo:hi].contiguous(), na_loc, pns[k])
            return x, agg, st_e, st_n, plan.pad_slice(out_loc)

        def bwd(k, x, agg, st_e, st_n, gh_node_next):
            g_loc = gout_loc if k == L - 1 else self.gh_rs[k + 1][:n] + gh_node_next
            gh_node, g_agg_loc, _g, views_n = be.node_backward(specs[k], deg_loc, x[lo:hi].contiguous(), agg[lo:hi].contiguous(),
                                                               na_loc, pns[k], g_loc, False, st_n)
            g_agg = torch.zeros_like(agg)
            g_agg[lo:hi] = g_agg_loc
            gh_edge = torch.zeros_like(x)
            _g_ea, views_e = be.edge_backward(specs[k], plan.csr, x, edge_attr_local, pes[k], g_agg, gh_edge, False, st_e)
            flat = torch.cat([v.reshape(-1) for v in list(views_e) + list(views_n) if v is not None])
            return gh_node, plan.to_padded(gh_edge), flat

        def gather(k, pad):
            if self._multi:
                _all_gather_rows(self.out_all[k], pad, self.group)
            else:
                self.out_all[k].copy_(pad)

        def scatter(k, pad):
            if self._multi:
                _reduce_scatter_rows(self.gh_rs[k], pad, self.group)
            else:
                self.gh_rs[k].copy_(pad)
        self._gather, self._scatter = gather, scatter

        # warm-up on a side stream (allocations, lazy kernel loads), then one capture per segment
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            st = []
            for k in range(L):
                st.append(fwd(k))
                gather(k, st[k][4])
            ghn = None
            for k in range(L - 1, -1, -1):
                ghn, pad, _f = bwd(k, *st[k][:4], ghn)
                scatter(k, pad)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.gf, self.gb = [], [None] * L
        self._st, self._bw = [], [None] * L
        pool = None
        for k in range(L):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                self._st.append(fwd(k))
            pool = pool or g.pool()
            self.gf.append(g)
            gather(k, self._st[k][4])      # the next segment's capture reads a buffer of the right content
        ghn = None
        for k in range(L - 1, -1, -1):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                self._bw[k] = bwd(k, *self._st[k][:4], ghn)
            self.gb[k] = g
            ghn = self._bw[k][0]
            scatter(k, self._bw[k][1])
        self.flat = torch.cat([self._bw[k][2] for k in range(L)])
        self.gh_loc = torch.empty_like(self._bw[0][0])
        self.out = None
        self.bytes_per_step = 0
        if self._multi:
            w = plan.world
            self.bytes_per_step = int(L * 2 * w * plan.per * h0[0].numel() * 4 * (w - 1) / w + 2 * self.flat.numel() * 4 * (w - 1) / w)

    def run(self):
        L, n = self.L, self.plan.hi - self.plan.lo
        for k in range(L):
            self.gf[k].replay()
            self._gather(k, self._st[k][4])
        for k in range(L - 1, -1, -1):
            self.gb[k].replay()
            self._scatter(k, self._bw[k][1])
        torch.add(self.gh_rs[0][:n], self._bw[0][0], out=self.gh_loc)
        torch.cat([self._bw[k][2] for k in range(L)], out=self.flat)
        if self._multi:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.out = self.plan.from_padded(self.out_all[L - 1])
