"""Callers either side of the shared simplicial message-passing path (SURVEY.md §8(f)-1,2):
the simplex feature embedding in front of the EGCL stack and the readout + loss behind it, as
the reference's task models compose them, on the HIP-backed layers of this package.

    SimplexEmbedding   embed_simplicial_complex (hulls_cssmpnn.py:96-125, md17_cssmpnn.py:85-120):
                       vertex features of every d-simplex in all (d+1)! vertex orders -> grade
                       embedding -> MVLinear (d = 0) / CEMLP(n_layers = d) -> sum over the orders
    type_attributes    embed_simplex_types (hulls_cssmpnn.py:127-140, md17_cssmpnn.py:122-133)
    HullsSimplicialMPNN  convex-hull volume model: 3 x EGCL -> MVLinear -> scalar blade ->
                       mean over the simplices of a graph -> MSE (hulls_cssmpnn.py:89-164)
    MD17SimplicialMPNN   trajectory model: MVLinear featurisation, 5 x EGCL (aggr = sum, learned
                       type attributes), CEMLP + MVLinear head on the vertices, vector blades ->
                       MSE / ADE / FDE (md17_cssmpnn.py:135-176)
    MotionSimplicialMPNN motion-capture model: 4 x EGCL (Cl(3,0), 16 channels, aggr = mean), MVLinear head, positions
                       relative to the graph's mean (motion_cssmpnn.py:13-163)
    NBASimplicialMPNN    NBA trajectory model: Cl(2,0), 40 channels, its own embedding modules, 4 x EGCL (aggr = sum),
                       ADE / FDE over the players (nba_cssmpnn.py:12-190)

Attribute names follow the reference so that its checkpoints load (`state_dict` keys are the
contract); the bodies are this package's own. Batches are
csmpn.data.complexes.SimplicialBatch (no torch_geometric).
"""
from __future__ import annotations

import itertools
import math
from typing import List, Sequence, Tuple

import torch
from torch import nn

from csmpn.algebra.cliffordalgebra import CliffordAlgebra
from csmpn.models.cegnn_utils import CEMLP, EGCL, MVLinear


def _fused_traj_readout(head, x) -> bool:
    """The trajectory heads go through csmpn_readout_traj_* on the device (CSMPN_NO_FUSED_READOUT=1: composed ops)."""
    import os
    if not x.is_cuda or os.environ.get("CSMPN_NO_FUSED_READOUT", "0") not in ("", "0"):
        return False
    from csmpn_hip import ops
    return ops.readout_traj_supported(head, x)


def _zero_grad_of(head):
    """0 * sum(bias): the head's bias sits on blade 0 and does not reach the vector blades the fused readout reads - its
    gradient is exactly zero, but it must EXIST: under DistributedDataParallel (the reference's multi-GPU mode,
    csmpn/md17.py:15-20) a parameter without a gradient leaves its whole bucket un-reduced."""
    b = getattr(head, "bias", None)
    return 0.0 * b.sum() if b is not None else 0.0


def segment_mean(x: torch.Tensor, index: torch.Tensor, n: int) -> torch.Tensor:
    """global_mean_pool: mean of the rows of x per segment id (sum / clamp(count, 1))."""
    out = x.new_zeros((n,) + tuple(x.shape[1:]))
    out.index_add_(0, index, x)
    cnt = x.new_zeros(n)
    cnt.index_add_(0, index, torch.ones_like(index, dtype=x.dtype))
    return out / cnt.clamp(min=1).reshape((-1,) + (1,) * (x.dim() - 1))


def simplex_vertex_rows(batch) -> torch.Tensor:
    """Row (in the batch) of every vertex of every simplex: x_ind is local to its graph."""
    start = batch.x_ind_ptr[:-1][batch.x_ind_batch]
    return batch.x_ind.long() + start.unsqueeze(-1)


class SimplexEmbedding(nn.Module):
    """cl_feature_embedding of the task models: module d embeds the d-simplices."""

    def __init__(self, algebra: CliffordAlgebra, in_features: int, hidden_features: int, max_dim: int = 2):
        super().__init__()
        self.algebra = algebra
        self.hidden_features = hidden_features
        self.max_dim = max_dim
        self.cl_feature_embedding = nn.ModuleList(
            [MVLinear(algebra, in_features, hidden_features, subspaces=False)]
            + [CEMLP(algebra, (d + 1) * in_features, hidden_features, hidden_features, n_layers=d, normalization_init=0)
               for d in range(1, max_dim + 1)])

    def forward(self, batch, vertex_blocks: Sequence[Tuple[torch.Tensor, int]]) -> torch.Tensor:
        """vertex_blocks: [(tensor [S, K, n_g], grade g)]: K channels per vertex, embedded as grade g.
        Channel order of a d-simplex row: block by block, inside a block vertex by vertex.
        The index tables come from batch.plan() (computed once per batch): no data-dependent shapes
        here, so a whole model step can be captured in a HIP graph."""
        plan = batch.plan(self.max_dim)
        n = self.algebra.dim
        D = 2 ** n
        dev = batch.x_ind.device
        out = torch.zeros(batch.x_ind.shape[0], self.hidden_features, D, device=dev)
        fused = dev.type == "cuda" and not any(t.requires_grad for t, _ in vertex_blocks)
        # Fused embedding (round 3, csmpn_embed_cemlp_*): for d >= 1 the vertex features are gathered in every vertex order
        # inside the CEMLP kernel and the sum over the orders is taken there - neither the [n_d (d+1)!, (d+1) K, D] input
        # rows nor the per-order outputs exist in memory. Served for one feature block on the wide parity-lane shapes (the
        # convex-hulls model); everything else composes simplex_rows + CEMLP + reshape-sum as before.
        emb_feat = None
        import os
        if fused and len(vertex_blocks) == 1 and os.environ.get("CSMPN_NO_FUSED_EMBED", "0") in ("", "0"):
            from csmpn_hip import ops
            t0, g0 = vertex_blocks[0]
            if all(ops.embed_cemlp_supported(self.cl_feature_embedding[d].binding(), d + 1, t0.shape[1])
                   for d in range(1, self.max_dim + 1)):
                emb_feat = self.algebra.embed_grade(t0, g0).contiguous()      # [S, K, D]: no vertex-order blow-up
                if "verts_i32" not in plan:
                    plan["verts_i32"] = [v.to(torch.int32).contiguous() for v in plan["verts"]]
                    # range check once per batch (the C-ABI would repeat it on every forward, with a host round trip)
                    S = int(emb_feat.shape[0])
                    for v in plan["verts"]:
                        if v.numel() and not (0 <= int(v.min()) and int(v.max()) < S):
                            raise IndexError(f"simplex vertex rows outside [0, {S}) (x_ind / x_ind_ptr of the batch)")
                    plan["verts_rows_checked"] = S
        # (Tried in round 4 and dropped: the modules of the dimensions as parallel branches on side streams - forward and,
        # through autograd's stream rule, backward. Two long independent chains overlap in a replayed HIP graph (1 618 ->
        # 851 us in a probe), these short forks do not pay for their joins: md17 step 3.13 -> 3.27 ms, hulls 4.80 -> 4.91.)
        for d in range(self.max_dim + 1):
            rows, pv, nperm = plan["rows"][d], plan["verts"][d], plan["nperm"][d]
            if rows.shape[0] == 0:
                continue
            out = out.index_copy(0, rows, self._embed_dim(d, rows, pv, nperm, vertex_blocks, emb_feat, plan, fused, D))
        return out

    def _embed_dim(self, d, rows, pv, nperm, vertex_blocks, emb_feat, plan, fused, D):
        """Embedded d-simplices [len(rows), hidden, D] (summed over the vertex orders)."""
        n = self.algebra.dim
        if emb_feat is not None and d >= 1:
            from csmpn_hip import ops
            mod = self.cl_feature_embedding[d]
            return ops.embed_cemlp_apply(emb_feat, plan["verts_i32"][d], nperm, mod.binding(), mod.flat_params(),
                                         validated=plan.get("verts_rows_checked") == int(emb_feat.shape[0]))
        if fused:
            from csmpn_hip import ops
            x = ops.simplex_rows(n, [(t, g) for t, g in vertex_blocks], pv)   # one gather + embed kernel
        else:
            feats = []
            for t, grade in vertex_blocks:
                g = t[pv]                                   # [rows, d+1, K, n_g]
                g = g.reshape(g.shape[0], (d + 1) * t.shape[1], t.shape[2])
                feats.append(self.algebra.embed_grade(g, grade))
            x = (feats[0] if len(feats) == 1 else torch.cat(feats, dim=1)).contiguous()
        e = self.cl_feature_embedding[d](x)
        return e.reshape(rows.shape[0], nperm, self.hidden_features, D).sum(dim=1) if nperm > 1 else e


def type_attributes(algebra: CliffordAlgebra, type_features: torch.Tensor, edge_index: torch.Tensor):
    """node_attr = per-simplex type features as scalar-blade multivectors, edge_attr = (source, target)."""
    node_attr = algebra.embed_grade(type_features.unsqueeze(-1), 0)
    # index_select, not node_attr[index]: the backward of advanced indexing sorts the index (11 launches, 155 us per md17
    # step); index_select's is one index_add_
    edge_attr = torch.cat((node_attr.index_select(0, edge_index[0]), node_attr.index_select(0, edge_index[1])), dim=1)
    return node_attr, edge_attr


def type_embedding(table: nn.Embedding, types: torch.Tensor) -> torch.Tensor:
    """sim_type_embedding(node_types) (md17_cssmpnn.py:122-133) as a row gather of the table: the same values; the backward
    is one index_add_ instead of embedding_dense_backward (82 us for a 3 x 3 table on the md17 batch)."""
    return table.weight.index_select(0, types)


def embedded_type_attributes(algebra: CliffordAlgebra, table: nn.Embedding, batch, max_dim: int = 2):
    """embed_simplex_types of the models with a learned type embedding (md17_cssmpnn.py:122-133): node_attr [S, K, D] and
    edge_attr [E, 2 K, D]. On the device one launch each way (csmpn_type_attr_*); the index tables are cached on the batch."""
    ei = batch.edge_index
    # the fused kernels sum the table gradient in 64 LDS bins (n_types * K values): larger tables take the composed path
    if table.weight.is_cuda and table.weight.dtype == torch.float32 and table.weight.numel() <= 64:
        from csmpn_hip import ops
        plan = batch.plan(max_dim)
        if "types_i32" not in plan:
            nt = batch.node_types
            if nt.numel() and not (0 <= int(nt.min()) and int(nt.max()) < table.num_embeddings):   # once per batch
                raise IndexError(f"node_types outside [0, {table.num_embeddings}) (nn.Embedding would raise too)")
            plan["types_i32"] = nt.to(torch.int32).contiguous()
            plan["ei_i32"] = (ei[0].to(torch.int32).contiguous(), ei[1].to(torch.int32).contiguous())
        return ops.type_attr_apply(table.weight, plan["types_i32"], plan["ei_i32"][0], plan["ei_i32"][1], algebra.dim)
    return type_attributes(algebra, type_embedding(table, batch.node_types), ei)


class HullsSimplicialMPNN(nn.Module):
    def __init__(self, in_features=1, hidden_features=28, out_features=1, num_layers=3, normalization_init=0,
                 residual=True, aggr="mean", condition=True, max_dim: int = 2):
        super().__init__()
        self.max_dim = max_dim
        self.algebra = CliffordAlgebra((1.0, 1.0, 1.0, 1.0, 1.0))
        self.hidden_features = hidden_features
        self.num_node_type = max_dim + 1 if condition else 0
        emb = SimplexEmbedding(self.algebra, in_features, hidden_features, max_dim)
        self.cl_feature_embedding = emb.cl_feature_embedding   # reference attribute name
        object.__setattr__(self, "_embed", emb)                # not a second registration
        self.layers = nn.Sequential(*[
            EGCL(self.algebra, hidden_features, hidden_features, hidden_features,
                 edge_attr_features=2 * self.num_node_type, node_attr_features=self.num_node_type,
                 residual=residual, normalization_init=normalization_init, aggr=aggr)
            for _ in range(num_layers)])
        self.projection = nn.Sequential(MVLinear(self.algebra, hidden_features, out_features))
        self.readout = nn.Linear(3, 1)   # present (unused) in the reference: kept for strict checkpoint loading

    def forward(self, batch, step=0, mode="train"):
        B = batch.num_graphs
        n = self.algebra.dim
        inp = batch.input
        plan = batch.plan(self.max_dim)
        vr = plan["vertex_rows"]
        # centre the vertices of every graph (hulls_cssmpnn.py:145-148)
        pos = inp.index_select(0, vr).reshape(B, -1, n)
        centred = (pos - pos.mean(dim=1, keepdim=True)).reshape(-1, n)
        inp = inp.index_copy(0, vr, centred)
        x = self._embed(batch, [(inp.unsqueeze(1), 1)])
        types = torch.nn.functional.one_hot(batch.node_types, self.num_node_type).float()
        node_attr, edge_attr = type_attributes(self.algebra, types, batch.edge_index)
        for layer in self.layers:
            x = layer(x, batch.edge_index, node_attr=node_attr, edge_attr=edge_attr)
        head = self.projection[0]
        if x.is_cuda and len(self.projection) == 1 and head.out_features == 1 and head.weight.dim() == 3:
            # fused readout + loss (csmpn_readout_mse_*): blade 0 of the projection, mean per graph, squared error
            from csmpn_hip import ops
            if "ptr_i32" not in plan:
                plan["ptr_i32"] = batch.x_ind_ptr.to(torch.int32)
            loss, _pred = ops.readout_mse(x, head.weight, getattr(head, "bias", None), plan["ptr_i32"], batch.target, n)
        else:
            pred = self.projection(x)[:, :, 0]
            pred = segment_mean(pred, batch.x_ind_batch, B)
            loss = (pred.squeeze(-1) - batch.target) ** 2
        return loss.mean(0), {"loss": loss}


class MD17SimplicialMPNN(nn.Module):
    def __init__(self, max_dim: int = 2, num_input: int = 30, num_hidden: int = 32, num_out: int = 10,
                 num_layers: int = 5, condition=True):
        super().__init__()
        self.algebra = CliffordAlgebra((1.0, 1.0, 1.0))
        self.max_dim = max_dim
        self.num_hidden = num_hidden
        self.num_node_type = max_dim + 1 if condition else 0
        self.feature_embedding = MVLinear(self.algebra, num_hidden + self.num_node_type, num_hidden, subspaces=False)
        emb = SimplexEmbedding(self.algebra, num_input, num_hidden, max_dim)
        self.cl_feature_embedding = emb.cl_feature_embedding
        object.__setattr__(self, "_embed", emb)
        self.sim_type_embedding = nn.Embedding(num_embeddings=max_dim + 1, embedding_dim=max_dim + 1)
        self.layers = nn.ModuleList([
            EGCL(self.algebra, num_hidden, num_hidden, num_hidden, edge_attr_features=2 * self.num_node_type,
                 node_attr_features=self.num_node_type, aggr="sum", normalization_init=0)
            for _ in range(num_layers)])
        self.projection = nn.Sequential(CEMLP(self.algebra, num_hidden, num_hidden, num_hidden, n_layers=1),
                                        MVLinear(self.algebra, num_hidden, num_out))

    def forward(self, batch, step=0, mode="train"):
        B = batch.num_graphs
        F_ = batch.loc.shape[1]                       # frames
        plan = batch.plan(self.max_dim)
        vr = plan["vertex_rows"]
        loc_node = batch.loc.index_select(0, vr)
        # mean position of a graph's vertices over vertices and frames (md17_cssmpnn.py:135-139)
        per_graph = segment_mean(loc_node.reshape(-1, F_ * 3), plan["graph_of_vertex"], B).reshape(B, F_, 3)
        per_graph = per_graph.mean(dim=1, keepdim=True).expand(B, F_, 3)
        pos = batch.loc - per_graph[batch.x_ind_batch]
        node_attr, edge_attr = embedded_type_attributes(self.algebra, self.sim_type_embedding, batch, self.max_dim)
        x = self._embed(batch, [(pos, 1), (batch.vel, 1), (batch.charges, 0)])
        x = self.feature_embedding(torch.cat((x, node_attr), dim=1))
        for layer in self.layers:
            x = layer(x, batch.edge_index, edge_attr, node_attr)
        return self.readout(batch, x)

    def readout(self, batch, x):
        """Head + loss behind the message passing (md17_cssmpnn.py:165-176): x [S, hidden, 8] -> (loss, parts)."""
        B = batch.num_graphs
        F_ = batch.loc.shape[1]
        plan = batch.plan(self.max_dim)
        vr = plan["vertex_rows"]
        loc_node = batch.loc.index_select(0, vr)
        head = self.projection[-1]
        if _fused_traj_readout(head, x):
            # fused head + loss (csmpn_readout_traj_*): the CEMLP of the head on the vertex rows, then MVLinear -> vector blades
            # -> + loc -> per-graph MSE / ADE / FDE in one launch each way
            from csmpn_hip import ops
            if "traj" not in plan:
                plan["traj"] = ops.readout_traj_tables(plan["graph_of_vertex"], B)
            z = self.projection[0](x.index_select(0, vr))
            per_graph, _pv, _pred = ops.readout_traj(z, head.weight, loc_node, batch.y, plan["traj"], 3)
            loss = per_graph[:, 0]
            return loss.mean() + _zero_grad_of(head), {"loss": loss, "ade_loss": per_graph[:, 1], "fde_loss": per_graph[:, 2]}
        pred = self.projection(x.index_select(0, vr))[..., 1:4]
        loc_pred = loc_node + pred
        tgt = batch.y
        sq = ((loc_pred.reshape(-1, 3) - tgt.reshape(-1, 3)) ** 2)
        ade = sq.sum(-1).sqrt().reshape(B, -1, F_).mean(-1).mean(-1)
        fde = ((loc_pred[:, -1, :] - tgt[:, -1, :]) ** 2).sum(-1).sqrt().reshape(B, -1).mean(-1)
        loss = sq.reshape(B, -1, 3).sum(-1).mean(-1)
        return loss.mean(), {"loss": loss, "ade_loss": ade, "fde_loss": fde}


class MotionSimplicialMPNN(nn.Module):
    """The motion-capture task model (csmpn/models/motion_cssmpnn.py:13-163): Cl(3,0), 16 hidden channels, 4 shared EGCL
    layers (aggr = mean) - S2's layer shape, served by the channel-MFMA / row-per-lane kernels. batch: pos, vel [S, 3]
    (vertex rows filled), y [V, 3]. Same parameter names as the reference (feature_embedding exists there and is unused)."""

    def __init__(self, max_dim: int = 2, num_input: int = 2, num_hidden: int = 16, num_out: int = 1, num_layers: int = 4,
                 condition=True):
        super().__init__()
        self.algebra = CliffordAlgebra((1.0, 1.0, 1.0))
        self.max_dim = max_dim
        self.num_hidden = num_hidden
        self.num_node_type = max_dim + 1 if condition else 0
        self.feature_embedding = MVLinear(self.algebra, num_input + self.num_node_type, num_hidden, subspaces=False)
        emb = SimplexEmbedding(self.algebra, num_input, num_hidden, max_dim)
        self.cl_feature_embedding = emb.cl_feature_embedding
        object.__setattr__(self, "_embed", emb)
        self.sim_type_embedding = nn.Embedding(num_embeddings=max_dim + 1, embedding_dim=max_dim + 1)
        self.layers = nn.ModuleList([
            EGCL(self.algebra, num_hidden, num_hidden, num_hidden, edge_attr_features=2 * self.num_node_type,
                 node_attr_features=self.num_node_type, aggr="mean", normalization_init=0)
            for _ in range(num_layers)])
        self.projection = nn.Sequential(MVLinear(self.algebra, num_hidden, num_out))

    def forward(self, batch, step=0, mode="train"):
        B = batch.num_graphs
        plan = batch.plan(self.max_dim)
        vr = plan["vertex_rows"]
        node_pos = batch.pos.index_select(0, vr)
        # positions relative to the mean vertex position of their graph (motion_cssmpnn.py:139-144)
        mean = segment_mean(node_pos, plan["graph_of_vertex"], B)
        pos = batch.pos.index_copy(0, vr, node_pos - mean[plan["graph_of_vertex"]])
        node_attr, edge_attr = embedded_type_attributes(self.algebra, self.sim_type_embedding, batch, self.max_dim)
        x = self._embed(batch, [(pos.unsqueeze(1), 1), (batch.vel.unsqueeze(1), 1)])
        for layer in self.layers:
            x = layer(x, batch.edge_index, edge_attr, node_attr)
        return self.readout(batch, x)

    def readout(self, batch, x):
        """Head + loss behind the message passing (motion_cssmpnn.py:150-163): x [S, hidden, 8] -> (loss, parts)."""
        B = batch.num_graphs
        plan = batch.plan(self.max_dim)
        vr = plan["vertex_rows"]
        node_pos = batch.pos.index_select(0, vr)
        head = self.projection[0]
        if head.out_features == 1 and _fused_traj_readout(head, x):
            from csmpn_hip import ops      # fused head + loss: vertex rows gathered inside the kernel
            if "traj" not in plan:
                plan["traj"] = ops.readout_traj_tables(plan["graph_of_vertex"], B, vertex_rows=vr, n_rows=x.shape[0])
            _pg, loss, _pred = ops.readout_traj(x, head.weight, node_pos, batch.y, plan["traj"], 3)
            return loss.mean() + _zero_grad_of(head), {"loss": loss}
        pred = node_pos + self.projection(x.index_select(0, vr))[..., 0, 1:4]
        loss = ((pred - batch.y.reshape(-1, 3)) ** 2).mean(dim=1)
        return loss.mean(), {"loss": loss}


class NBASimplicialMPNN(nn.Module):
    """The NBA trajectory task model (csmpn/models/nba_cssmpnn.py:12-190): Cl(2,0), 40 hidden channels, 4 shared EGCL layers
    (aggr = sum; general row-tile kernels). batch: pos, vel [S, F, 2] (vertex rows filled; num_input = 2 F channels per
    vertex), y [B * (agents - 1), num_out, 2]; every graph has `agents` vertices, the last one (the ball) is not scored."""

    def __init__(self, max_dim: int = 2, num_input: int = 20, num_hidden: int = 40, num_out: int = 40, num_layers: int = 4,
                 condition=True, agents: int = 6):
        super().__init__()
        self.algebra = CliffordAlgebra((1.0, 1.0))
        self.max_dim, self.num_input, self.num_hidden, self.num_out, self.agents = max_dim, num_input, num_hidden, num_out, agents
        self.num_node_type = max_dim + 1 if condition else 0
        self.feature_embedding = MVLinear(self.algebra, num_input + self.num_node_type, num_hidden, subspaces=False)
        A = self.algebra
        self.cl_feature_embedding = nn.Sequential(
            MVLinear(A, num_input, num_input, subspaces=False),
            CEMLP(A, 2 * num_input, num_hidden, num_input, n_layers=1, normalization_init=0),
            nn.Sequential(CEMLP(A, 3 * num_input, num_hidden, num_hidden, n_layers=1, normalization_init=0),
                          CEMLP(A, num_hidden, num_hidden, num_input, n_layers=1, normalization_init=0)))
        self.sim_type_embedding = nn.Embedding(num_embeddings=max_dim + 1, embedding_dim=max_dim + 1)
        self.layers = nn.ModuleList([
            EGCL(A, num_hidden, num_hidden, num_hidden, edge_attr_features=2 * self.num_node_type,
                 node_attr_features=self.num_node_type, aggr="sum", normalization_init=0)
            for _ in range(num_layers)])
        self.projection = MVLinear(A, num_hidden, num_out)

    def embed(self, batch, pos, vel):
        """nba_cssmpnn.py:125-158: per dimension, every vertex order of a simplex -> [pos of its vertices | vel of its
        vertices] as grade-1 channels -> the dimension's module -> sum over the orders."""
        plan = batch.plan(self.max_dim)
        out = torch.zeros(batch.x_ind.shape[0], self.num_input, 2 ** self.algebra.dim, device=pos.device, dtype=pos.dtype)
        for d in range(self.max_dim + 1):
            rows, pv, nperm = plan["rows"][d], plan["verts"][d], plan["nperm"][d]
            if rows.shape[0] == 0:
                continue
            feats = []
            for t in (pos, vel):
                g = t[pv]                                                # [rows * nperm, d + 1, F, 2]
                feats.append(self.algebra.embed_grade(g.reshape(g.shape[0], (d + 1) * t.shape[1], t.shape[2]), 1))
            e = self.cl_feature_embedding[d](torch.cat(feats, dim=1).contiguous())
            out = out.index_copy(0, rows, e.reshape(rows.shape[0], nperm, self.num_input, -1).sum(dim=1))
        return out

    def forward(self, batch, step=0, mode="train"):
        B = batch.num_graphs
        F_ = batch.pos.shape[1]
        plan = batch.plan(self.max_dim)
        vr = plan["vertex_rows"]
        node_attr, edge_attr = embedded_type_attributes(self.algebra, self.sim_type_embedding, batch, self.max_dim)
        x = self.embed(batch, batch.pos, batch.vel)
        x = self.feature_embedding(torch.cat((x, node_attr), dim=1))
        for layer in self.layers:
            x = layer(x, batch.edge_index, edge_attr, node_attr)
        return self.readout(batch, x)

    def readout(self, batch, x):
        """Head + loss behind the message passing (nba_cssmpnn.py:176-188): x [S, hidden, 4] -> (loss, parts)."""
        B = batch.num_graphs
        F_ = batch.pos.shape[1]
        plan = batch.plan(self.max_dim)
        vr = plan["vertex_rows"]
        if _fused_traj_readout(self.projection, x):
            from csmpn_hip import ops      # fused head + loss: the last agent of every graph (the ball) is not scored
            if "traj" not in plan:
                plan["traj"] = ops.readout_traj_tables(plan["graph_of_vertex"], B, vertex_rows=vr, n_rows=x.shape[0], unscored_last=1)
            per_graph, _pv, _pred = ops.readout_traj(x, self.projection.weight, None, batch.y, plan["traj"], 2)
            ade = per_graph[:, 1]
            return ade.mean() + _zero_grad_of(self.projection), {"loss": ade, "ade_loss": ade, "fde_loss": per_graph[:, 2]}
        pred = self.projection(x.index_select(0, vr))[..., 1:3]                      # [V, num_out, 2]
        loc_pred = pred.reshape(B, self.agents, self.num_out, -1)[:, :-1].reshape(-1, self.num_out, 2)
        tgt = batch.y
        d2 = ((loc_pred.reshape(-1, 2) - tgt.reshape(-1, 2)) ** 2).sum(-1)
        ade = d2.sqrt().reshape(B, -1, F_).mean(-1).mean(-1)
        fde = ((loc_pred[:, -1, :] - tgt[:, -1, :]) ** 2).sum(-1).sqrt().reshape(B, -1).mean(-1)
        return ade.mean(), {"loss": ade, "ade_loss": ade, "fde_loss": fde}
