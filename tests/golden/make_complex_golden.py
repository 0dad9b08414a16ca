"""Lifted-graph format fixtures from the IMPORTED reference data modules (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_complex_golden.py

Pins SURVEY.md §8(f)-3 - `generate_*_single` (csmpn/data/modules/utils.py:25-103) and `SimplicialTransform.add_missing_adj`
/ `get_num_simplicies` / `get_edge` / `gen_hulls_feat` (csmpn/data/modules/simplicial_data.py:105-217) - to the reference's
own code for hand-built complexes. The reference imports `gudhi`, `torch_geometric.data` / `.transforms` / `.typing` and
`torch_scatter`, none of which is installed or installable here; they are replaced by in-memory stand-ins that hold no
arithmetic of the path:

  gudhi.SimplexTree   a set of vertex tuples closed under faces.
                      get_simplices(): lexicographic order of the sorted vertex tuples (= the pre-order walk of gudhi's
                      simplex trie: [0], [0,1], [0,1,2], [0,2], [1], ...), filtration 0.0;
                      get_boundaries(s): the facets of s, removing the LAST vertex first ([0,1], [0,2], [1,2] for [0,1,2]);
                      get_cofaces(s, 1): the simplices with one more vertex that contain s, in get_simplices order.
                      The enumeration order decides the ORDER of the adjacency columns inside one type and nothing else
                      (indices per dimension are assigned in get_simplices order, which is fixed by the lexicographic
                      rule): the test compares adjacency types as sorted multisets and everything else exactly.
  torch_geometric.data.Data   an attribute bag with item access, `keys`, `to_dict` / `from_dict`.

Writes complexes_ref.npz: for every complex its top simplices and, from the reference: x_<d>, adj_<s>_<t> (incl. the flipped
copies), edge_index, edge_attr (type pairs), node_types, x_ind.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np
import torch

import pyg_standin

REF = os.environ.get("CSMPN_REFERENCE", "/root/reference")
if not os.path.isdir(REF):
    print("reference not present: nothing to do")
    sys.exit(0)


# ----------------------------------------------------------------------------- stand-ins (no arithmetic of the path)
class SimplexTree:
    def __init__(self):
        self._s = set()

    def insert(self, simplex, filtration=0.0):
        import itertools
        verts = tuple(sorted(int(v) for v in simplex))
        for k in range(1, len(verts) + 1):
            for sub in itertools.combinations(verts, k):
                self._s.add(sub)
        return True

    def get_simplices(self):
        for s in sorted(self._s):
            yield list(s), 0.0

    def get_boundaries(self, simplex):
        s = list(simplex)
        if len(s) == 1:
            return
        for drop in range(len(s) - 1, -1, -1):
            yield s[:drop] + s[drop + 1:], 0.0

    def get_cofaces(self, simplex, codim):
        base = set(simplex)
        return [(list(t), 0.0) for t in sorted(self._s) if len(t) == len(simplex) + codim and base.issubset(t)]


class Data:
    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    def __getitem__(self, k):
        return getattr(self, k)

    def __setitem__(self, k, v):
        setattr(self, k, v)

    @property
    def keys(self):
        return [k for k in self.__dict__ if not k.startswith("_")]

    def to_dict(self):
        return {k: getattr(self, k) for k in self.keys}

    def from_dict(self, d):
        out = type(self)()
        for k, v in d.items():
            setattr(out, k, v)
        return out


def install_standins():
    pyg_standin.install()
    g = types.ModuleType("gudhi")
    gst = types.ModuleType("gudhi.simplex_tree")
    g.SimplexTree = gst.SimplexTree = SimplexTree
    g.simplex_tree = gst
    sys.modules["gudhi"], sys.modules["gudhi.simplex_tree"] = g, gst
    tg = sys.modules["torch_geometric"]
    for name, attrs in (("torch_geometric.data", {"Data": Data}), ("torch_geometric.typing", {"Adj": object}),
                        ("torch_geometric.transforms", {"BaseTransform": object})):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        setattr(tg, name.split(".")[-1], m)
    ts = types.ModuleType("torch_scatter")
    ts.scatter = None
    sys.modules["torch_scatter"] = ts


install_standins()
sys.path.insert(0, REF)
from csmpn.data.modules import utils as RU  # noqa: E402
from csmpn.data.modules.simplicial_data import SimplicialTransform, SimplicialComplexData  # noqa: E402

# hand-built complexes: (name, number of vertices, top simplices)
COMPLEXES = [
    ("tetra_surface", 4, [[0, 1, 2], [0, 1, 3], [0, 2, 3], [1, 2, 3]]),
    ("strip_with_tail", 6, [[0, 1, 2], [1, 2, 3], [3, 4]]),               # vertex 5 isolated, edge 3-4 dangling
    ("octahedron", 6, [[0, 2, 4], [0, 2, 5], [0, 3, 4], [0, 3, 5], [1, 2, 4], [1, 2, 5], [1, 3, 4], [1, 3, 5]]),
]


def reference_lift(n_vertices, tops, dim=2):
    st = SimplexTree()
    for v in range(n_vertices):
        st.insert([v])
    for t in tops:
        st.insert(t)
    simplices = RU.generate_simplicies_single(st)
    indices = RU.generate_indices_single(st)
    adj = RU.generate_adjacencies_single(indices, st)
    x_dict = RU.generate_features_single(simplices, indices)
    tr = SimplicialTransform(dim=dim, label="hulls")
    graph = Data(input=torch.arange(n_vertices * 5, dtype=torch.float32).reshape(n_vertices, 5), y=torch.zeros(1))
    scd = SimplicialComplexData().from_dict(graph.to_dict())
    for k, v in x_dict.items():
        scd[f"x_{k}"] = v
    for k, v in adj.items():
        scd[f"adj_{k}"] = v
    num_per_dim = tr.get_num_simplicies(x_dict)
    scd = tr.add_missing_adj(scd)
    scd = tr.get_edge(x_dict, scd, num_per_dim)
    scd = tr.gen_hulls_feat(scd, num_per_dim)
    return x_dict, scd


if __name__ == "__main__":
    out = {}
    for name, nv, tops in COMPLEXES:
        x_dict, scd = reference_lift(nv, tops)
        out[f"{name}/n_vertices"] = np.array(nv)
        out[f"{name}/tops"] = np.array([t + [-1] * (3 - len(t)) for t in tops], dtype=np.int64)
        for k, v in x_dict.items():
            out[f"{name}/x_{k}"] = v.numpy()
        for k in scd.keys:
            if k.startswith("adj_"):
                out[f"{name}/{k}"] = scd[k].numpy()
        out[f"{name}/edge_index"] = scd.edge_index.numpy()
        out[f"{name}/edge_attr"] = scd.edge_attr.numpy()
        out[f"{name}/node_types"] = scd.node_types.numpy()
        out[f"{name}/x_ind"] = scd.x_ind.numpy().astype(np.int64)
        print(name, {k: tuple(v.shape) for k, v in out.items() if k.startswith(name + "/adj_")}, "edges", scd.edge_index.shape[1])
    np.savez_compressed(os.path.join(HERE, "complexes_ref.npz"), **out)
