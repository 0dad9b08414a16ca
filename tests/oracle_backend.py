"""Test-only compute backend for csmpn_hip.sharded: the four EGCL stages done by the
oracle on CPU tensors, so that the sharding / collective plumbing can be tested with
gloo where there is no GPU. Never used by the product path."""
import torch

from oracle import ref_path as O

KEYS = ["0.weight", "0.bias", "1.a", "1.b", "2.weight", "2.normalization.a", "2.linear_right.weight",
        "2.linear_left.weight", "2.linear_left.bias", "3.a"]


def _pdict(flat, prefix):
    return {f"{prefix}layers.{k // 10}.{KEYS[k % 10]}": p for k, p in enumerate(flat)}


class _Csr:
    def __init__(self, edge_index, n_nodes):
        self.edge_index = edge_index
        self.n_edges = edge_index.shape[1]
        self.n_nodes = n_nodes
        self.deg = torch.bincount(edge_index[1], minlength=n_nodes).to(torch.int32)


class OracleBackend:
    @staticmethod
    def _alg(spec):
        return O.Algebra(list(spec.edge.metric))

    @staticmethod
    def build_csr(edge_index, n_nodes):
        return _Csr(edge_index, n_nodes)

    @staticmethod
    def _messages(spec, csr, h, edge_attr, p):
        alg = OracleBackend._alg(spec)
        src, dst = csr.edge_index[0], csr.edge_index[1]
        x = h.index_select(0, dst) - h.index_select(0, src)
        if edge_attr is not None:
            x = torch.cat([x, edge_attr], dim=1)
        msg = O.cemlp(alg, x, p, "edge_model.")
        return O.scatter_rows(msg.reshape(msg.shape[0], spec.O * alg.D), dst, h.shape[0], "sum").reshape(h.shape[0], spec.O, alg.D)

    @staticmethod
    def edge_forward(spec, csr, h, edge_attr, pe):
        with torch.no_grad():
            return OracleBackend._messages(spec, csr, h, edge_attr, _pdict(pe, "edge_model.")), None

    @staticmethod
    def _node(spec, deg, h, agg, node_attr, p):
        alg = OracleBackend._alg(spec)
        if spec.mean:
            agg = agg / deg.clamp(min=1).to(agg.dtype)[:, None, None]
        parts = [h, agg] if node_attr is None else [h, agg, node_attr]
        out = O.cemlp(alg, torch.cat(parts, dim=1), p, "node_model.")
        return h + out if spec.residual else out

    @staticmethod
    def node_forward(spec, deg, h, agg, node_attr, pn):
        with torch.no_grad():
            return OracleBackend._node(spec, deg, h, agg, node_attr, _pdict(pn, "node_model.")), None

    @staticmethod
    def node_backward(spec, deg, h, agg, node_attr, pn, gout, want_gna, state=None):
        with torch.enable_grad():
            hh = h.detach().clone().requires_grad_(True)
            aa = agg.detach().clone().requires_grad_(True)
            na = None if node_attr is None else node_attr.detach().clone().requires_grad_(True)
            ps = [q.detach().clone().requires_grad_(True) for q in pn]
            out = OracleBackend._node(spec, deg, hh, aa, na, _pdict(ps, "node_model."))
            out.backward(gout)
        return hh.grad, aa.grad, (na.grad if (na is not None and want_gna) else None), [q.grad for q in ps]

    @staticmethod
    def edge_backward(spec, csr, h, edge_attr, pe, g_agg, gh, want_gea, state=None):
        with torch.enable_grad():
            hh = h.detach().clone().requires_grad_(True)
            ea = None if edge_attr is None else edge_attr.detach().clone().requires_grad_(True)
            ps = [q.detach().clone().requires_grad_(True) for q in pe]
            agg = OracleBackend._messages(spec, csr, hh, ea, _pdict(ps, "edge_model."))
            agg.backward(g_agg)
        gh.add_(hh.grad if hh.grad is not None else 0)
        return (ea.grad if (ea is not None and want_gea) else None), [q.grad if q.grad is not None else torch.zeros_like(q) for q in ps]
