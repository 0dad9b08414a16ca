"""Clifford-equivariant layers with the reference's nn.Module surface
(csmpn/models/cegnn_utils.py): MVLinear :287, MVSiLU :53, NormalizationLayer :34,
SteerableGeometricProductLayer :98, MVLayerNorm :86, CEMLP :160, EGCL :216 — same
constructor signatures, attribute names, initialisers and state_dict keys.

CEMLP.forward and EGCL.forward do not run their sub-modules: they hand the
sub-modules' parameters to the fused HIP kernels through the C-ABI
(csmpn_hip.ops). They require float32 GPU tensors and raise otherwise: there is
no CPU path in this package (the CPU restatement is oracle/, test-only).

The five small layers keep a standalone forward (none of the reference's models
calls MVSiLU / NormalizationLayer / SteerableGeometricProductLayer / MVLayerNorm
outside a CEMLP): float32 [rows, channels, D] device tensors go through the
standalone HIP entry points (csmpn_mvlinear_*, csmpn_mvsilu_*, csmpn_mvnorm_*,
csmpn_mvlayernorm_*, csmpn_wgp_*); anything else (CPU tensors of the module
construction / state_dict tests, extra middle dimensions) takes the host formulation.
"""
import math

import torch
import torch.nn as nn

from csmpn_hip import ops

EPS = 1e-6


def unsqueeze_like(tensor: torch.Tensor, like: torch.Tensor, dim=0):
    """Append singleton dims to `tensor` (after its first `dim` dims) until it has like.ndim dims."""
    extra = like.ndim - tensor.ndim
    if extra < 0:
        raise ValueError(f"tensor.ndim={tensor.ndim} > like.ndim={like.ndim}")
    if extra == 0:
        return tensor
    return tensor[(slice(None),) * dim + (None,) * extra]


def unsorted_segment_mean(data, segment_ids, num_segments):
    out = data.new_zeros(num_segments, data.size(1))
    cnt = data.new_zeros(num_segments, data.size(1))
    idx = segment_ids.unsqueeze(-1).expand(-1, data.size(1))
    out.scatter_add_(0, idx, data)
    cnt.scatter_add_(0, idx, torch.ones_like(data))
    return out / cnt.clamp(min=1)


def _blade_grades(algebra, device):
    return algebra.bbo.grades.to(device)


class MVLinear(nn.Module):
    def __init__(self, algebra, in_features, out_features, subspaces=True, bias=True):
        super().__init__()
        self.algebra = algebra
        self.in_features = in_features
        self.out_features = out_features
        self.subspaces = subspaces
        if subspaces:
            self.weight = nn.Parameter(torch.empty(out_features, in_features, algebra.n_subspaces))
        else:
            self.weight = nn.Parameter(torch.empty(out_features, in_features))
        if bias:
            self.bias = nn.Parameter(torch.empty(1, out_features, 1))
            self.b_dims = (0,)
        else:
            self.register_parameter("bias", None)
            self.b_dims = ()
        self.reset_parameters()

    def reset_parameters(self):
        torch.nn.init.normal_(self.weight, std=1 / math.sqrt(self.in_features))
        if self.bias is not None:
            torch.nn.init.zeros_(self.bias)

    def forward(self, input):
        # y[b,o,...,d] = sum_i W[o,i,grade(d)] x[b,i,...,d]  (+ bias on the scalar blade)
        if input.is_cuda and input.dim() == 3 and input.dtype == torch.float32 and self.algebra.dim <= 5:
            # the HIP entry points (include/csmpn_hip.h: csmpn_mvlinear_forward / _backward)
            return ops.mvlinear_apply(input, self.weight, self.bias, self.algebra.dim)
        # host-side formulation for CPU tensors and inputs with extra middle dimensions (module
        # construction / state_dict tests; no reference model calls MVLinear that way)
        w = self.weight
        if self.subspaces:
            w = w[..., _blade_grades(self.algebra, w.device)]          # [O, I, D]
            result = torch.einsum("bm...i,nmi->bn...i", input, w)
        else:
            result = torch.einsum("bm...i,nm->bn...i", input, w)
        if self.bias is not None:
            shift = self.algebra.embed(self.bias, self.b_dims)
            result = result + unsqueeze_like(shift, result, dim=2)
        return result


class MVSiLU(nn.Module):
    def __init__(self, algebra, channels, invariant="mag2", exclude_dual=False):
        super().__init__()
        self.algebra = algebra
        self.channels = channels
        self.exclude_dual = exclude_dual
        self.invariant = invariant
        self.a = nn.Parameter(torch.ones(1, channels, algebra.dim + 1))
        self.b = nn.Parameter(torch.zeros(1, channels, algebra.dim + 1))
        if invariant not in ("norm", "mag2"):
            raise ValueError(f"Invariant {invariant} not recognized.")

    def forward(self, input):
        alg = self.algebra
        if self.invariant == "mag2" and ops.small_layer_on_hip(alg, input, self.a, self.b):
            return ops.mvsilu_apply(alg, input, self.a, self.b)      # csmpn_mvsilu_forward / _backward
        higher = alg.grades[1:]
        inv = alg.norms(input, grades=higher) if self.invariant == "norm" else alg.qs(input, grades=higher)
        inv = torch.cat([input[..., :1], *inv], dim=-1)
        gate = torch.sigmoid(unsqueeze_like(self.a, inv, dim=2) * inv + unsqueeze_like(self.b, inv, dim=2))
        return gate[..., _blade_grades(alg, gate.device)] * input


class NormalizationLayer(nn.Module):
    def __init__(self, algebra, features, init: float = 0):
        super().__init__()
        self.algebra = algebra
        self.in_features = features
        self.a = nn.Parameter(torch.zeros(self.in_features, algebra.n_subspaces) + init)

    def forward(self, input):
        assert input.shape[1] == self.in_features
        if ops.small_layer_on_hip(self.algebra, input, self.a):
            return ops.mvnorm_apply(self.algebra, input, self.a)     # csmpn_mvnorm_forward / _backward
        nrm = torch.cat(self.algebra.norms(input), dim=-1)
        nrm = torch.sigmoid(self.a) * (nrm - 1) + 1
        return input / (nrm[..., _blade_grades(self.algebra, nrm.device)] + EPS)


class MVLayerNorm(nn.Module):
    def __init__(self, algebra, channels):
        super().__init__()
        self.algebra = algebra
        self.channels = channels
        self.a = nn.Parameter(torch.ones(1, channels))

    def forward(self, input):
        if ops.small_layer_on_hip(self.algebra, input, self.a):
            return ops.mvlayernorm_apply(self.algebra, input, self.a)   # csmpn_mvlayernorm_forward / _backward
        scale = self.algebra.norm(input)[..., :1].mean(dim=1, keepdim=True) + EPS
        return unsqueeze_like(self.a, scale, dim=2) * input / scale


class SteerableGeometricProductLayer(nn.Module):
    def __init__(self, algebra, features, include_first_order=True, normalization_init=0):
        super().__init__()
        self.algebra = algebra
        self.features = features
        self.include_first_order = include_first_order
        if normalization_init is not None:
            self.normalization = NormalizationLayer(algebra, features, normalization_init)
        else:
            self.normalization = nn.Identity()
        self.linear_right = MVLinear(algebra, features, features, bias=False)
        if include_first_order:
            self.linear_left = MVLinear(algebra, features, features, bias=True)
        self.product_paths = algebra.geometric_product_paths
        self.weight = nn.Parameter(torch.empty(features, int(self.product_paths.sum())))
        self.reset_parameters()

    def reset_parameters(self):
        torch.nn.init.normal_(self.weight, std=1 / (math.sqrt(self.algebra.dim + 1)))

    def _get_weight(self):
        """Dense [C, D, D, D] path-weighted Cayley tensor (standalone use only)."""
        alg = self.algebra
        G = alg.dim + 1
        per_path = self.weight.new_zeros(self.features, G, G, G)
        per_path[:, self.product_paths] = self.weight
        g = _blade_grades(alg, self.weight.device)
        return alg.cayley * per_path[:, g[:, None, None], g[None, :, None], g[None, None, :]]

    def forward(self, input):
        right = self.normalization(self.linear_right(input))
        if ops.small_layer_on_hip(self.algebra, input, self.weight):
            prod = ops.wgp_apply(self.algebra, input, right, self.weight)   # csmpn_wgp_forward / _backward
        else:
            prod = torch.einsum("bni,nijk,bnk->bnj", input, self._get_weight(), right)
        if self.include_first_order:
            return (self.linear_left(input) + prod) / math.sqrt(2)
        return prod


def _block_params(seq):
    lin, silu, sgp, ln = seq[0], seq[1], seq[2], seq[3]
    return [lin.weight, lin.bias, silu.a, silu.b, sgp.weight, sgp.normalization.a, sgp.linear_right.weight,
            sgp.linear_left.weight, sgp.linear_left.bias, ln.a]


class CEMLP(nn.Module):
    def __init__(self, algebra, in_features, hidden_features, out_features, n_layers=2, normalization_init=0):
        super().__init__()
        self.algebra = algebra
        self.in_features = in_features
        self.hidden_features = hidden_features
        self.out_features = out_features
        self.n_layers = n_layers
        widths = []
        cur = in_features
        for _ in range(n_layers - 1):
            widths.append((cur, hidden_features))
            cur = hidden_features
        widths.append((cur, out_features))
        self.layers = nn.Sequential(*[
            nn.Sequential(
                MVLinear(algebra, i_f, o_f),
                MVSiLU(algebra, o_f),
                SteerableGeometricProductLayer(algebra, o_f, normalization_init=normalization_init),
                MVLayerNorm(algebra, o_f),
            )
            for i_f, o_f in widths
        ])
        self._widths = widths
        self._binding = None

    def __getstate__(self):
        # the cached ctypes binding holds raw pointers: never copied / pickled (deepcopy, torch.save of the
        # module, EMA / best-model snapshots); it is rebuilt on the next forward
        state = self.__dict__.copy()
        state["_binding"] = None
        return state

    def binding(self) -> "ops.CemlpBinding":
        if self._binding is None:
            specs = [dict(in_features=i, out_features=o, lin_subspaces=True) for i, o in self._widths]
            self._binding = ops.CemlpBinding(self.algebra.metric_tuple, specs)
        return self._binding

    def flat_params(self):
        out = []
        for seq in self.layers:
            out.extend(_block_params(seq))
        return out

    def forward(self, x):
        return ops.cemlp_apply(x, self.binding(), self.flat_params())


class EGCL(nn.Module):
    """Shared simplicial message-passing layer: one edge model and one node model for
    every adjacency type, conditioned only through edge_attr / node_attr."""

    def __init__(self, algebra, in_features, hidden_features, out_features, edge_attr_features=0,
                 node_attr_features=0, residual=True, normalization_init=0, aggr="mean"):
        super().__init__()
        self.aggr = aggr
        self.residual = residual
        self.in_features = in_features
        self.hidden_features = hidden_features
        self.out_features = out_features
        self.edge_attr_features = edge_attr_features
        self.node_attr_features = node_attr_features
        self.edge_model = CEMLP(algebra, in_features + edge_attr_features, hidden_features, out_features,
                                normalization_init=normalization_init)
        self.node_model = CEMLP(algebra, in_features + out_features + node_attr_features, hidden_features,
                                out_features, normalization_init=normalization_init)
        self.algebra = algebra
        self._spec = None

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_spec"] = None   # holds ctypes bindings (see CEMLP.__getstate__)
        return state

    def spec(self) -> "ops.EgclSpec":
        if self._spec is None:
            self._spec = ops.EgclSpec(self.edge_model.binding(), self.node_model.binding(), self.in_features,
                                      self.out_features, self.edge_attr_features, self.node_attr_features,
                                      self.aggr, self.residual)
        return self._spec

    # PyG-style pieces, kept for callers that use them directly
    def message(self, h_i, h_j, edge_attr=None):
        h_i, h_j = self.algebra.split(h_i), self.algebra.split(h_j)
        x = h_i - h_j if edge_attr is None else torch.cat([h_i - h_j, edge_attr], dim=1)
        return self.algebra.flatten(self.edge_model(x))

    def update(self, h_agg, h, node_attr):
        h_agg, h = self.algebra.split(h_agg), self.algebra.split(h)
        parts = [h, h_agg] if node_attr is None else [h, h_agg, node_attr]
        out = self.node_model(torch.cat(parts, dim=1))
        if self.residual:
            out = h + out
        return self.algebra.flatten(out)

    def forward(self, h, edge_index, edge_attr=None, node_attr=None):
        csr = ops.get_csr(edge_index, h.shape[0])
        params = self.edge_model.flat_params() + self.node_model.flat_params()
        return ops.egcl_apply(h, edge_attr, node_attr, self.spec(), csr, params)
