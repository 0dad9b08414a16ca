"""Lifted-graph input format of the shared simplicial message-passing path, without
torch_geometric / gudhi.

The reference turns every simplicial complex into ONE "big graph" whose nodes are all simplices
(vertices first, then edges, then triangles) and whose `edge_index` is the union of the adjacency
types `adj_{s}_{t}` with per-dimension row offsets (csmpn/data/modules/simplicial_data.py:105-157),
then lets PyG collate graphs into a batch with `follow_batch=["node_types", "x_ind"]`
(csmpn/data/hulls.py:109-122). This module restates that format:

  lift(n_vertices, top_simplices, max_dim)   faces + the reference's adjacency rules
                                             (csmpn/data/modules/utils.py:63-103: boundaries,
                                             upper adjacencies through a shared coface, the
                                             fully connected 0-0 block with its duplicate
                                             directed edges, and the flipped k+1 -> k copies,
                                             simplicial_data.py:105-110)
  hull_complex(points) / rips_complex(...)   the two complex constructions the task data use
                                             (utils.py:210-247 convex hull faces; utils.py:106-137
                                             distance-thresholded cliques)
  collate(complexes)                          the batch the task models read: x_ind (vertex ids
                                             LOCAL to each graph, as in the reference: x_ind has no
                                             `__inc__`), node_types, edge_index (offset per graph),
                                             batch / ptr / x_ind_batch / x_ind_ptr

A batch caches the target-sorted CSR of its `edge_index` (complexes are static: built once in
`pre_transform`, hulls.py:68-78), so no layer or step rebuilds it.
"""
from __future__ import annotations

import itertools
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch


@dataclass
class SimplicialComplex:
    n_vertices: int
    x_ind: torch.Tensor          # [S, max_dim + 1] int64: vertex ids of every simplex, zero padded
    node_types: torch.Tensor     # [S] int64: dimension of every simplex
    edge_index: torch.Tensor     # [2, E] int64 over the S simplices (row 0 = source, row 1 = target)
    edge_types: torch.Tensor     # [E, 2] int64: (source dimension, target dimension)
    features: Dict[str, torch.Tensor] = field(default_factory=dict)   # per-simplex tensors [S, ...]
    labels: Dict[str, torch.Tensor] = field(default_factory=dict)     # per-graph tensors

    @property
    def n_simplices(self) -> int:
        return int(self.x_ind.shape[0])


def lift(n_vertices: int, top_simplices: Sequence[Sequence[int]], max_dim: int = 2):
    """All faces up to `max_dim` of the given simplices and the reference's adjacencies.

    Returns (x_dict, adj): x_dict[d] = int64 [n_d, d+1] (sorted vertex tuples, lexicographic order),
    adj["s_t"] = int64 [2, n] (source index in dimension s, target index in dimension t)."""
    faces: List[set] = [set((v,) for v in range(n_vertices))] + [set() for _ in range(max_dim)]
    for simplex in top_simplices:
        verts = tuple(sorted(int(v) for v in simplex))
        for k in range(1, max_dim + 1):
            for sub in itertools.combinations(verts, k + 1):
                faces[k].add(sub)
    ordered = [sorted(f) for f in faces]
    index = [{s: i for i, s in enumerate(f)} for f in ordered]
    adj: Dict[str, List] = {}

    def add(key, src, dst):
        adj.setdefault(key, []).append((src, dst))

    for k in range(max_dim + 1):
        for s in ordered[k]:
            si = index[k][s]
            # upper adjacency: the other boundaries of every coface of s (utils.py:72-83)
            if k + 1 <= max_dim:
                sset = set(s)
                for c in ordered[k + 1]:
                    if sset.issubset(c):
                        for b in itertools.combinations(c, k + 1):
                            if b != s:
                                add(f"{k}_{k}", index[k][b], si)
            # boundaries (utils.py:86-89)
            if k >= 1:
                for b in itertools.combinations(s, k):
                    add(f"{k - 1}_{k}", index[k - 1][b], si)
    # fully connected 0-0 block over the non-edges; the test `[i, j] not in edges_present` compares with
    # SORTED vertex lists, so (i, j) with i > j is added even when {i, j} is an edge (utils.py:91-97)
    if max_dim >= 1:
        present = set(ordered[1])
        for i in range(n_vertices):
            for j in range(n_vertices):
                if i != j and (i, j) not in present:
                    add("0_0", i, j)
    out = {k: torch.tensor(v, dtype=torch.int64).t().contiguous() for k, v in adj.items()}
    # downward communication: the flipped copies (simplicial_data.py:105-110)
    for k in range(max_dim):
        if f"{k}_{k + 1}" in out:
            out[f"{k + 1}_{k}"] = out[f"{k}_{k + 1}"][[1, 0]].clone()
    x_dict = {k: torch.tensor(ordered[k], dtype=torch.int64).reshape(-1, k + 1) for k in range(max_dim + 1)}
    return x_dict, out


def to_complex(n_vertices: int, x_dict, adj, max_dim: int = 2) -> SimplicialComplex:
    """The big graph of one complex (simplicial_data.py:112-157): simplices of all dimensions are the
    rows, adjacency types are concatenated in (source dimension, target dimension) order."""
    counts = [int(x_dict[d].shape[0]) if d in x_dict else 0 for d in range(max_dim + 1)]
    offs = np.concatenate([[0], np.cumsum(counts)])
    S = int(offs[-1])
    x_ind = torch.zeros(S, max_dim + 1, dtype=torch.int64)
    node_types = torch.zeros(S, dtype=torch.int64)
    for d in range(max_dim + 1):
        x_ind[offs[d]:offs[d + 1], :d + 1] = x_dict[d]
        node_types[offs[d]:offs[d + 1]] = d
    ei, et = [], []
    for s in range(max_dim + 1):
        for t in range(max_dim + 1):
            a = adj.get(f"{s}_{t}")
            if a is None or a.numel() == 0:
                continue
            e = a.clone()
            e[0] += int(offs[s])
            e[1] += int(offs[t])
            ei.append(e)
            et.append(torch.tensor([[s, t]], dtype=torch.int64).expand(a.shape[1], 2))
    edge_index = torch.cat(ei, dim=1) if ei else torch.zeros(2, 0, dtype=torch.int64)
    edge_types = torch.cat(et, dim=0) if et else torch.zeros(0, 2, dtype=torch.int64)
    return SimplicialComplex(n_vertices, x_ind, node_types, edge_index, edge_types)


def hull_complex(points: np.ndarray, max_dim: int = 2) -> SimplicialComplex:
    """Faces of the convex hull of `points` [V, n] (utils.py:210-247)."""
    from scipy.spatial import ConvexHull
    hull = ConvexHull(np.asarray(points, dtype=np.float64))
    x_dict, adj = lift(len(points), hull.simplices.tolist(), max_dim)
    return to_complex(len(points), x_dict, adj, max_dim)


def rips_complex(points: np.ndarray, dis: float, max_dim: int = 2) -> SimplicialComplex:
    """Vietoris-Rips complex: an edge for every pair closer than `dis`, higher simplices = cliques
    (utils.py:106-137)."""
    pts = np.asarray(points, dtype=np.float64)
    V = len(pts)
    close = np.linalg.norm(pts[:, None] - pts[None], axis=-1) <= dis
    tops = []
    for k in range(1, max_dim + 1):
        for sub in itertools.combinations(range(V), k + 1):
            if all(close[a, b] for a, b in itertools.combinations(sub, 2)):
                tops.append(sub)
    x_dict, adj = lift(V, tops, max_dim)
    return to_complex(V, x_dict, adj, max_dim)


class SimplicialBatch:
    """What the task models read from a collated batch (hulls.py:109-122 with
    follow_batch=["node_types", "x_ind"]). Per-simplex feature tensors and per-graph labels are
    attributes under their own names (`input`, `target`, `pos`, ...)."""

    def __init__(self, **tensors):
        self._names = list(tensors)
        for k, v in tensors.items():
            setattr(self, k, v)
        self._csr = None

    def to(self, device):
        out = SimplicialBatch(**{k: getattr(self, k).to(device) for k in self._names})
        return out

    @property
    def num_graphs(self) -> int:
        return int(self.ptr.shape[0]) - 1

    def plan(self, max_dim: int = 2):
        """Index tables of the embedding stage, computed once per batch (the only host round trips of
        a model step: everything downstream uses fixed-shape index_select / index_copy and can be
        captured in a HIP graph):
          rows[d]   rows of the d-simplices                                    [n_d]
          verts[d]  their vertices in all (d+1)! orders, as batch rows          [n_d * (d+1)!, d+1]
          vertex_rows = rows[0]; graph_of_vertex = graph id of every vertex row."""
        cached = getattr(self, "_plan", None)
        if cached is not None and cached["max_dim"] == max_dim:
            return cached
        import itertools
        start = self.x_ind_ptr[:-1][self.x_ind_batch]
        vrows = self.x_ind.long() + start.unsqueeze(-1)
        plan = {"max_dim": max_dim, "rows": [], "verts": [], "nperm": []}
        for d in range(max_dim + 1):
            rows = torch.nonzero(self.node_types == d, as_tuple=False).flatten()
            perms = torch.tensor(list(itertools.permutations(range(d + 1))), device=rows.device)
            plan["rows"].append(rows)
            vt = vrows[rows][:, : d + 1]
            # once per batch: the fused embedding kernels (csmpn_embed_cemlp_*) index the vertex features with these rows
            # UNCHECKED (include/csmpn_hip.h states the precondition) and the composed path's gather clamps silently -
            # check here, where PyTorch indexing would have asserted
            if vt.numel() and (int(vt.min()) < 0 or int(vt.max()) >= int(self.node_types.shape[0])):
                raise IndexError(f"x_ind of the {d}-simplices points outside the batch ({int(vt.min())}..{int(vt.max())} "
                                 f"of {int(self.node_types.shape[0])} rows)")
            plan["verts"].append(vt[:, perms].reshape(-1, d + 1).contiguous())
            plan["nperm"].append(int(perms.shape[0]))
        plan["vertex_rows"] = plan["rows"][0]
        plan["graph_of_vertex"] = self.batch[plan["rows"][0]]
        plan["vertices_per_graph"] = torch.bincount(plan["graph_of_vertex"], minlength=self.num_graphs)
        self._plan = plan
        return plan

    def csr(self):
        """Target-sorted adjacency of the batch, built once (device batches only)."""
        if self._csr is None:
            from csmpn_hip import ops
            self._csr = ops.Csr(self.edge_index, int(self.node_types.shape[0]))
            try:   # the layers look the CSR up on the tensor object (ops.get_csr)
                self.edge_index._csmpn_csr = (self.edge_index._version, self._csr)
            except Exception:
                pass
        return self._csr


def collate(complexes: Sequence[SimplicialComplex]) -> SimplicialBatch:
    sizes = [c.n_simplices for c in complexes]
    ptr = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int64)
    batch = torch.repeat_interleave(torch.arange(len(complexes)), torch.tensor(sizes))
    out = dict(
        x_ind=torch.cat([c.x_ind for c in complexes], dim=0),
        node_types=torch.cat([c.node_types for c in complexes], dim=0),
        edge_index=torch.cat([c.edge_index + int(ptr[i]) for i, c in enumerate(complexes)], dim=1),
        edge_types=torch.cat([c.edge_types for c in complexes], dim=0),
        batch=batch, ptr=ptr, x_ind_batch=batch.clone(), x_ind_ptr=ptr.clone(),
    )
    for name in complexes[0].features:
        out[name] = torch.cat([c.features[name] for c in complexes], dim=0)
    for name in complexes[0].labels:
        out[name] = torch.stack([c.labels[name] for c in complexes], dim=0)
    return SimplicialBatch(**out)


def hulls_example(points: np.ndarray, target: Optional[float] = None, max_dim: int = 2) -> SimplicialComplex:
    """One convex-hulls sample as the reference's transform lays it out (simplicial_data.py:177-196):
    `input` [S, n] holds the vertex coordinates in the vertex rows and zeros elsewhere."""
    c = hull_complex(points, max_dim)
    inp = torch.zeros(c.n_simplices, points.shape[1], dtype=torch.float32)
    inp[: c.n_vertices] = torch.as_tensor(points, dtype=torch.float32)
    c.features["input"] = inp
    if target is None:
        from scipy.spatial import ConvexHull
        target = float(ConvexHull(np.asarray(points, dtype=np.float64)).volume)
    c.labels["target"] = torch.tensor(target, dtype=torch.float32)
    return c
