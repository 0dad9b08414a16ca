"""Hand-computed known-answer case for the PyG boundary.

torch_geometric 2.3.0 / torch_scatter 2.1.1 are third-party and absent; no reference test pins
`MessagePassing.propagate`. This 3-node / 4-edge case is worked out BY HAND below and holds
(a) the stand-in used to generate the golden fixtures (tests/golden/pyg_standin.py),
(b) the oracle's scatter (oracle/ref_path.py:scatter_rows) to the semantics the reference relies on
(cegnn_utils.py:229,254-284): flow source_to_target, x_i = x[edge_index[1]], x_j = x[edge_index[0]],
aggregation over edge_index[1] with dim_size = N, mean = sum / clamp(count, 1), isolated node -> 0.

    nodes:  x0 = (1, 10), x1 = (2, 20), x2 = (4, 40)
    edges (source j -> target i):  0->1, 2->1, 1->0, 0->1 (duplicate)      node 2 has no incoming edge
    message m_e = x_i - x_j:
        e0: x1 - x0 = (1, 10)     e1: x1 - x2 = (-2, -20)    e2: x0 - x1 = (-1, -10)    e3: x1 - x0 = (1, 10)
    sum  over targets:  node0 = (-1, -10)      node1 = (1-2+1, 10-20+10) = (0, 0)      node2 = (0, 0)
    mean over targets:  node0 = (-1, -10) / 1  node1 = (0, 0) / 3                      node2 = 0 / max(0,1)
    with messages 2*m_e + 1 (to make node1 non-trivial):
        e0: (3, 21)  e1: (-3, -39)  e2: (-1, -19)  e3: (3, 21)
        sum:  node0 = (-1, -19)   node1 = (3, 3)     node2 = (0, 0)
        mean: node0 = (-1, -19)   node1 = (1, 1)     node2 = (0, 0)
"""
import os
import sys

import torch

from oracle import ref_path as O

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import pyg_standin  # noqa: E402

X = torch.tensor([[1.0, 10.0], [2.0, 20.0], [4.0, 40.0]])
EI = torch.tensor([[0, 2, 1, 0], [1, 1, 0, 1]])
MSG = torch.tensor([[3.0, 21.0], [-3.0, -39.0], [-1.0, -19.0], [3.0, 21.0]])
SUM = torch.tensor([[-1.0, -19.0], [3.0, 3.0], [0.0, 0.0]])
MEAN = torch.tensor([[-1.0, -19.0], [1.0, 1.0], [0.0, 0.0]])


class _Probe(pyg_standin.MessagePassing):
    def message(self, x_i, x_j, bias):
        return 2.0 * (x_i - x_j) + bias

    def update(self, aggr_out, x):
        return aggr_out, x


def test_standin_propagate_known_answer():
    for aggr, want in (("sum", SUM), ("add", SUM), ("mean", MEAN)):
        out, x_passed = _Probe(aggr=aggr).propagate(EI, x=X, bias=1.0)
        assert torch.equal(out, want), (aggr, out)
        assert x_passed is X


def test_standin_message_operands():
    """x_i is the TARGET row (edge_index[1]), x_j the SOURCE row (edge_index[0])."""
    class M(pyg_standin.MessagePassing):
        def message(self, x_i, x_j):
            M.seen = (x_i.clone(), x_j.clone())
            return x_i

        def update(self, aggr_out):
            return aggr_out
    M(aggr="sum").propagate(EI, x=X)
    assert torch.equal(M.seen[0], X[EI[1]]) and torch.equal(M.seen[1], X[EI[0]])


def test_oracle_scatter_known_answer():
    assert torch.equal(O.scatter_rows(MSG, EI[1], 3, "sum"), SUM)
    assert torch.equal(O.scatter_rows(MSG, EI[1], 3, "mean"), MEAN)


def test_global_mean_pool_known_answer():
    batch = torch.tensor([0, 0, 1])
    got = pyg_standin.global_mean_pool(X, batch)
    assert torch.equal(got, torch.tensor([[1.5, 15.0], [4.0, 40.0]]))
