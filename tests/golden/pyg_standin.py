"""In-memory stand-in for ``torch_geometric.nn`` used ONLY while generating
golden vectors from the imported reference (torch_geometric 2.3.0 /
torch_scatter 2.1.1 are pinned by the reference's environment.yml:13-15 but are
not installed and cannot be: no network).

Restates the semantics the reference relies on at its call sites
(csmpn/models/cegnn_utils.py:4,216,229,279; hulls_cssmpnn.py:7,158):
flow = source_to_target, ``x_i = x[edge_index[1]]``, ``x_j = x[edge_index[0]]``,
aggregation over ``edge_index[1]`` with ``dim_size = x.size(0)``;
``sum`` = scatter-add into zeros, ``mean`` = sum / clamp(count, min=1);
``update(aggr_out, **kwargs)``.
"""
import inspect
import sys
import types

import torch
from torch import nn


def _scatter(msg, index, dim_size, reduce):
    out = msg.new_zeros((dim_size,) + tuple(msg.shape[1:]))
    out.index_add_(0, index, msg)
    if reduce == "mean":
        cnt = msg.new_zeros(dim_size)
        cnt.index_add_(0, index, torch.ones_like(index, dtype=msg.dtype))
        out = out / cnt.clamp(min=1).reshape((-1,) + (1,) * (msg.dim() - 1))
    elif reduce not in ("sum", "add"):
        raise ValueError(reduce)
    return out


class MessagePassing(nn.Module):
    def __init__(self, aggr="add", flow="source_to_target", node_dim=-2):
        super().__init__()
        self.aggr = aggr
        assert flow == "source_to_target"

    def propagate(self, edge_index, size=None, **kwargs):
        j, i = edge_index[0], edge_index[1]
        msg_args = {}
        dim_size = None
        for name in inspect.signature(self.message).parameters:
            if name.endswith("_i") or name.endswith("_j"):
                base = kwargs[name[:-2]]
                dim_size = base.size(0)
                msg_args[name] = base.index_select(0, i if name.endswith("_i") else j)
            else:
                msg_args[name] = kwargs.get(name)
        msg = self.message(**msg_args)
        out = _scatter(msg, i, dim_size, self.aggr)
        upd_args = {n: kwargs.get(n) for n in list(inspect.signature(self.update).parameters)[1:]}
        return self.update(out, **upd_args)


def global_mean_pool(x, batch, size=None):
    size = int(batch.max()) + 1 if size is None else size
    return _scatter(x, batch, size, "mean")


def install():
    tg = types.ModuleType("torch_geometric")
    tgnn = types.ModuleType("torch_geometric.nn")
    tgnn.MessagePassing = MessagePassing
    tgnn.global_mean_pool = global_mean_pool
    tg.nn = tgnn
    tg.seed_everything = lambda seed: torch.manual_seed(seed)
    sys.modules.setdefault("torch_geometric", tg)
    sys.modules.setdefault("torch_geometric.nn", tgnn)
