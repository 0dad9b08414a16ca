"""torch.autograd Functions over the C-ABI (include/csmpn_hip.h).

PyTorch is plumbing here: it owns device memory and streams and records the
autograd graph; every forward and backward of the path runs in the HIP library.
There is no eager fallback: CPU tensors or a missing library raise.
"""
from __future__ import annotations

import ctypes as C
import os
import warnings
from typing import List, Optional, Sequence

import torch

from . import native
from .native import BlockGrads, BlockParams, PARAM_FIELDS, check

NP = len(PARAM_FIELDS)  # parameter slots per CEMLP block


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _on_device_of(arg_index):
    """Decorator: run the function with the device of its `arg_index`-th argument current (kernel
    attributes, launches and the stream belong to that device; the caller's current device may be
    another one)."""
    import functools

    def deco(fn):
        @functools.wraps(fn)
        def wrapped(*args, **kwargs):
            t = args[arg_index]
            if isinstance(t, torch.Tensor) and t.is_cuda:
                with torch.cuda.device(t.device):
                    return fn(*args, **kwargs)
            return fn(*args, **kwargs)
        return wrapped
    return deco


def _require_device(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(
            f"{what} is on {t.device}: the csmpn HIP path runs on MI355X only and has no CPU fallback "
            f"(the CPU restatement lives in oracle/ and is test infrastructure)."
        )
    if t.dtype != torch.float32:
        raise RuntimeError(f"{what} must be float32, got {t.dtype}")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


_FUSED_GRAD_ACCUM = False


def set_fused_grad_accumulation(flag: bool):
    """Opt-in: parameter gradients are added by the kernels straight into the existing `p.grad` tensors (the kernels
    accumulate anyway) and the autograd functions return None for the parameters, instead of handing views of a flat
    buffer to autograd's AccumulateGrad - one elementwise kernel per parameter tensor less (the convex-hulls model has
    ~150 of them per step). Needs `optimizer.zero_grad(set_to_none=False)`; falls back to the flat buffer whenever a
    parameter has no (contiguous float32, same-device) .grad yet. Parameter hooks (e.g. DistributedDataParallel's) do
    not see gradients delivered this way: leave it off under DDP."""
    global _FUSED_GRAD_ACCUM
    _FUSED_GRAD_ACCUM = bool(flag)


def _fusable(params, device) -> bool:
    if not _FUSED_GRAD_ACCUM:
        return False
    for p in params:
        if p is None:
            continue
        g = p.grad
        if g is None or not p.requires_grad or g.dtype != torch.float32 or g.device != device or not g.is_contiguous() \
                or g.shape != p.shape:
            return False
    return True


class CemlpBinding:
    """ctypes view of one CEMLP's parameters, cached across calls.

    `specs` is a list (one per block) of dicts with in_features, out_features,
    lin_subspaces. Parameters arrive per call as a flat list of NP tensors per
    block in PARAM_FIELDS order (None for an absent MVLinear bias).
    """

    def __init__(self, metric: Sequence[float], specs: List[dict]):
        self.metric = tuple(float(m) for m in metric)
        self.n = len(self.metric)
        self.D = 1 << self.n
        self.metric_arr = native.metric_array(self.metric)
        self.nblk = len(specs)
        if not 1 <= self.nblk <= native.MAX_BLOCKS:
            raise native.CsmpnError(f"CEMLP with {self.nblk} blocks not supported (1..{native.MAX_BLOCKS})")
        self.params = (BlockParams * self.nblk)()
        self.grads = (BlockGrads * self.nblk)()
        for k, s in enumerate(specs):
            self.params[k].in_features = int(s["in_features"])
            self.params[k].out_features = int(s["out_features"])
            self.params[k].lin_subspaces = 1 if s.get("lin_subspaces", True) else 0
        self.in_features = int(specs[0]["in_features"])
        self.out_features = int(specs[-1]["out_features"])
        self._key = None
        self._grad_layout = None
        self._ws_bytes = None
        self._saved_per_row = None

    def supported(self) -> bool:
        return bool(native.lib().csmpn_metric_supported(self.metric_arr, self.n))

    def bind(self, params: Sequence[Optional[torch.Tensor]]):
        assert len(params) == NP * self.nblk
        key = tuple(_ptr(p) for p in params)
        if key != self._key:
            for k in range(self.nblk):
                blk = self.params[k]
                for j, name in enumerate(PARAM_FIELDS):
                    p = params[k * NP + j]
                    if p is not None and not p.is_contiguous():
                        raise RuntimeError(f"parameter {name} of block {k} must be contiguous")
                    setattr(blk, name, key[k * NP + j])
            self._key = key
            self._grad_layout = None
        if self._ws_bytes is None:
            self._ws_bytes = int(native.lib().csmpn_cemlp_workspace_bytes(self.n, self.params, self.nblk))

    def new_saved(self, rows: int, device, save_state: bool = False) -> Optional[torch.Tensor]:
        """Buffer for the inputs of blocks 1.. (written by forward, read by backward). save_state: the forward / backward pair
        will be called with CSMPN_FLAG_SAVE_STATE - the buffer also holds the state regions (3-4x the block inputs on the
        D = 32 and 32-channel shapes)."""
        if self._saved_per_row is None:
            self._saved_per_row = int(native.lib().csmpn_cemlp_saved_floats_per_row(self.n, self.params, self.nblk))
        if self._saved_per_row == 0 or rows == 0:
            return None
        # the size of THIS launch: regions a launch of `rows` rows never touches are left out (csmpn_cemlp_saved_floats)
        return torch.empty(int(native.lib().csmpn_cemlp_saved_floats(self.n, self.params, self.nblk, rows,
                                                                     native.FLAG_SAVE_STATE if save_state else 0)),
                           dtype=torch.float32, device=device)

    def workspace(self, device) -> torch.Tensor:
        return torch.empty(max(self._ws_bytes, 16), dtype=torch.uint8, device=device)

    def grad_floats(self, params: Sequence[Optional[torch.Tensor]]) -> int:
        """Floats of the flat gradient buffer new_grads() lays out for these parameters."""
        self._layout(params)
        return max(self._grad_layout[1], 1)

    def _layout(self, params):
        if self._grad_layout is None:
            offs, total = [], 0
            for p in params:
                if p is None:
                    offs.append(None)
                else:
                    offs.append((total, p.numel(), tuple(p.shape)))
                    total += (p.numel() + 3) // 4 * 4
            self._grad_layout = (offs, total)

    def new_grads(self, params: Sequence[Optional[torch.Tensor]], device, flat=None, fused_into=None):
        """One zeroed flat buffer for all parameter gradients (a single memset; or the caller's
        already zeroed `flat` of grad_floats() elements); returns (flat, views) with views[i]
        shaped like params[i]."""
        self._layout(params)
        offs, total = self._grad_layout
        if fused_into is not None:
            # the kernels add into the parameters' own .grad tensors; nothing for autograd to accumulate
            for k in range(self.nblk):
                g = self.grads[k]
                for j, name in enumerate(PARAM_FIELDS):
                    p = fused_into[k * NP + j]
                    setattr(g, name, None if p is None else p.grad.data_ptr())
            return None, [None] * len(params)
        if flat is None:
            flat = torch.zeros(max(total, 1), dtype=torch.float32, device=device)
        base = flat.data_ptr()
        views = []
        for k in range(self.nblk):
            g = self.grads[k]
            for j, name in enumerate(PARAM_FIELDS):
                o = offs[k * NP + j]
                if o is None:
                    setattr(g, name, None)
                    views.append(None)
                else:
                    setattr(g, name, base + 4 * o[0])
                    views.append(flat[o[0]:o[0] + o[1]].view(o[2]))
        return flat, views


# --------------------------------------------------------------------------------- CEMLP


class _CemlpFn(torch.autograd.Function):
    @staticmethod
    @_on_device_of(1)
    def forward(ctx, x, binding: CemlpBinding, *params):
        _require_device(x, "CEMLP input")
        if x.dim() != 3 or x.shape[1] != binding.in_features or x.shape[2] != binding.D:
            raise RuntimeError(f"CEMLP input must be [rows, {binding.in_features}, {binding.D}], got {tuple(x.shape)}")
        x = x.contiguous()
        binding.bind(params)
        rows = x.shape[0]
        y = torch.empty(rows, binding.out_features, binding.D, dtype=torch.float32, device=x.device)
        ws = binding.workspace(x.device)
        # CSMPN_FLAG_SAVE_STATE: the library honours it for the standalone shapes whose saved buffer has state regions (the
        # 32-channel Cl(3,0) CEMLPs of the md17 model: csmpn_cemlp_saved_floats sizes them) and ignores it everywhere else
        # (asked for only where the standalone entry points honour it: other shapes would get state regions sized and never used)
        want_state = _SAVE_STATE and binding.n == 3 and binding.out_features == 32
        saved = binding.new_saved(rows, x.device, want_state) if any(ctx.needs_input_grad) else None
        ctx.flags = native.FLAG_SAVE_STATE if (want_state and saved is not None) else 0
        check(native.lib().csmpn_cemlp_forward(binding.metric_arr, binding.n, binding.params, binding.nblk,
                                               x.data_ptr(), rows, y.data_ptr(), _ptr(saved), ws.data_ptr(),
                                               ws.numel(), ctx.flags, _stream(x.device)))
        ctx.binding = binding
        ctx.param_refs = params          # the caller's parameter objects (their .grad, for fused accumulation)
        ctx.ws, ctx.saved = ws, saved   # packed weights / block inputs are reused by backward
        ctx.save_for_backward(x, *[p for p in params if p is not None])
        ctx.mask = [p is not None for p in params]
        return y

    @staticmethod
    @_on_device_of(1)
    def backward(ctx, gy):
        binding = ctx.binding
        x, *present = ctx.saved_tensors
        it = iter(present)
        params = [next(it) if m else None for m in ctx.mask]
        gy = gy.contiguous()
        binding.bind(params)
        fused = ctx.param_refs if _fusable(ctx.param_refs, x.device) else None
        flat, views = binding.new_grads(params, x.device, fused_into=fused)
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        ws = ctx.ws
        check(native.lib().csmpn_cemlp_backward(binding.metric_arr, binding.n, binding.params, binding.grads,
                                                binding.nblk, x.data_ptr(), gy.data_ptr(), x.shape[0], _ptr(gx),
                                                _ptr(ctx.saved), ws.data_ptr(), ws.numel(),
                                                native.FLAG_WEIGHTS_PACKED | ctx.flags,
                                                _stream(x.device)))
        return (gx, None, *views)


def cemlp_apply(x, binding: CemlpBinding, params):
    return _CemlpFn.apply(x, binding, *params)


_EMBED_LAUNCHES = 0


def embed_launches() -> int:
    """Forward launches of the fused embedding entry point so far (tests assert that the HIP path ran)."""
    return _EMBED_LAUNCHES


class _EmbedCemlpFn(torch.autograd.Function):
    """Fused simplex embedding (csmpn_embed_cemlp_*; hulls_cssmpnn.py:96-125): vertex features gathered in every vertex
    order -> CEMLP -> sum over the orders, without the [n_simplices * (d+1)!, ., D] input / output rows in memory.
    vertex_feat [S, K, D] (data: no gradient), verts [n_simplices * n_orders, d+1] int32."""

    @staticmethod
    @_on_device_of(1)
    def forward(ctx, vertex_feat, verts, n_orders, validated, binding: CemlpBinding, *params):
        _require_device(vertex_feat, "embedding vertex features")
        binding.bind(params)
        rows = int(verts.shape[0])
        out = torch.empty(rows // n_orders, binding.out_features, binding.D, dtype=torch.float32, device=vertex_feat.device)
        ws = binding.workspace(vertex_feat.device)
        need_grad = any(ctx.needs_input_grad[5:])
        saved = binding.new_saved(rows, vertex_feat.device, _SAVE_STATE and binding.nblk > 1) if need_grad else None
        # the layer's own saved buffer, sized for exactly these rows: CSMPN_FLAG_SAVE_STATE (two-block modules: y, R, s)
        st_flag = native.FLAG_SAVE_STATE if (_SAVE_STATE and saved is not None and binding.nblk > 1) else 0
        # validated: the caller has range-checked verts against vertex_feat's rows once (the batch plan); otherwise the entry
        # point does, with one host round trip (not capturable)
        st_flag |= native.FLAG_NO_VALIDATE if validated else 0
        check(native.lib().csmpn_embed_cemlp_forward(
            binding.metric_arr, binding.n, binding.params, binding.nblk, vertex_feat.data_ptr(), int(vertex_feat.shape[0]),
            int(vertex_feat.shape[1]),
            verts.data_ptr(), int(verts.shape[1]), int(n_orders), rows, out.data_ptr(), _ptr(saved), ws.data_ptr(), ws.numel(),
            st_flag, _stream(vertex_feat.device)))
        ctx.st_flag = st_flag
        global _EMBED_LAUNCHES
        _EMBED_LAUNCHES += 1
        ctx.binding, ctx.param_refs, ctx.ws, ctx.saved, ctx.n_orders = binding, params, ws, saved, int(n_orders)
        ctx.save_for_backward(vertex_feat, verts, *[p for p in params if p is not None])
        ctx.mask = [p is not None for p in params]
        return out

    @staticmethod
    @_on_device_of(1)
    def backward(ctx, gout):
        binding = ctx.binding
        vertex_feat, verts, *present = ctx.saved_tensors
        it = iter(present)
        params = [next(it) if m else None for m in ctx.mask]
        gout = gout.contiguous()
        binding.bind(params)
        fused = ctx.param_refs if _fusable(ctx.param_refs, vertex_feat.device) else None
        _flat, views = binding.new_grads(params, vertex_feat.device, fused_into=fused)
        check(native.lib().csmpn_embed_cemlp_backward(
            binding.metric_arr, binding.n, binding.params, binding.grads, binding.nblk, vertex_feat.data_ptr(),
            int(vertex_feat.shape[0]), int(vertex_feat.shape[1]), verts.data_ptr(), int(verts.shape[1]), ctx.n_orders,
            int(verts.shape[0]), gout.data_ptr(), _ptr(ctx.saved), ctx.ws.data_ptr(), ctx.ws.numel(),
            ctx.st_flag | native.FLAG_NO_VALIDATE,     # the forward has checked (or the caller vouched for) this table
            _stream(vertex_feat.device)))
        return (None, None, None, None, None, *views)


def embed_cemlp_supported(binding: CemlpBinding, verts_per_row: int, channels_per_vertex: int) -> bool:
    """Shapes the fused embedding serves: Cl(5,0) / Cl(4,1), 16 / 24 / 28 / 32 output channels, <= 8 input channels."""
    return (binding.n == 5 and binding.nblk in (1, 2) and binding.out_features in (16, 24, 28, 32)
            and verts_per_row * channels_per_vertex == binding.in_features <= 8)


def embed_cemlp_apply(vertex_feat, verts_i32, n_orders, binding: CemlpBinding, params, validated=False):
    """validated=True: every entry of verts_i32 is known to lie in [0, vertex_feat.shape[0]) (checked once per batch by the
    caller); False: the C-ABI checks on every forward (one synchronous round trip)."""
    return _EmbedCemlpFn.apply(vertex_feat, verts_i32, n_orders, bool(validated), binding, *params)


# --------------------------------------------------------------------------------- CSR


_DETERMINISTIC = None   # None: follow CSMPN_DETERMINISTIC, else torch.are_deterministic_algorithms_enabled()
_warned_soft_det = False


def set_deterministic(flag):
    """True / False: force the atomic-free aggregation on / off; None: follow the environment
    (CSMPN_DETERMINISTIC=0|1) and otherwise torch.use_deterministic_algorithms, which the reference
    switches on (engineer/utils/seed.py:30)."""
    global _DETERMINISTIC
    _DETERMINISTIC = None if flag is None else bool(flag)


def deterministic_request():
    """'hard' (explicitly requested: unsupported shapes raise), 'soft' (inherited from
    torch.use_deterministic_algorithms: unsupported shapes warn once and use float atomics) or None."""
    if _DETERMINISTIC is not None:
        return "hard" if _DETERMINISTIC else None
    env = os.environ.get("CSMPN_DETERMINISTIC")
    if env is not None and env != "":
        return "hard" if env != "0" else None
    return "soft" if torch.are_deterministic_algorithms_enabled() else None


def _soft_fallback(rc):
    """An unsupported shape under a 'soft' request: warn once, tell the caller to take the atomic path."""
    global _warned_soft_det
    if rc != native.ERR_UNSUPPORTED:
        return False
    if not _warned_soft_det:
        _warned_soft_det = True
        warnings.warn("csmpn_hip: deterministic aggregation is not available for this layer shape "
                      f"({native.lib().csmpn_last_error().decode()}); using float atomics")
    return True


def _warn_soft_slice():
    global _warned_soft_det
    if not _warned_soft_det:
        _warned_soft_det = True
        warnings.warn("csmpn_hip: torch.use_deterministic_algorithms(True) is set, but this launch covers a slice of "
                      "the adjacency (graph-segment step): using float atomics (ops.set_deterministic(True) makes "
                      "this an error)")


def segment_reduce(rows, out, add=None, sub=None, accumulate=True):
    """out[v] (+)= sum rows[add segment of v] - sum rows[sub segment of v] in a fixed order.
    add / sub: (row_ptr, order or None) int32 device tensors."""
    ap, ao = add if add is not None else (None, None)
    sp, so = sub if sub is not None else (None, None)
    check(native.lib().csmpn_segment_reduce(rows.data_ptr(), rows[0].numel(), out.shape[0], _ptr(ap), _ptr(ao),
                                            _ptr(sp), _ptr(so), out.data_ptr(), 1 if accumulate else 0,
                                            _stream(out.device)))
    return out


class Csr:
    """Target-sorted adjacency of one complex (built once, reused by every layer and step)."""

    __slots__ = ("perm", "src", "dst", "deg", "row_ptr", "n_edges", "n_nodes", "build_ms", "_src_order")

    def source_order(self):
        """(row_ptr_src, order): the sorted edge positions grouped by source (deterministic mode)."""
        if self._src_order is None:
            dev = self.src.device
            order = torch.empty(max(self.n_edges, 1), dtype=torch.int32, device=dev)
            rp = torch.empty(self.n_nodes + 1, dtype=torch.int32, device=dev)
            ws = torch.empty(int(native.lib().csmpn_csr_workspace_bytes(self.n_edges, self.n_nodes)),
                             dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                check(native.lib().csmpn_csr_source_order(self.src.data_ptr(), self.n_edges, self.n_nodes,
                                                          order.data_ptr(), rp.data_ptr(), ws.data_ptr(), ws.numel(),
                                                          _stream(dev)))
            self._src_order = (rp, order)
        return self._src_order

    def __init__(self, edge_index: torch.Tensor, n_nodes: int):
        if not edge_index.is_cuda:
            raise RuntimeError("edge_index must live on the GPU (no CPU fallback)")
        if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise RuntimeError("edge_index must be int64 [2, E]")
        ei = edge_index.contiguous()
        dev = ei.device
        E = ei.shape[1]
        i32 = dict(dtype=torch.int32, device=dev)
        self.n_edges, self.n_nodes = E, n_nodes
        self._src_order = None
        self.perm = torch.empty(max(E, 1), **i32)
        self.src = torch.empty(max(E, 1), **i32)
        self.dst = torch.empty(max(E, 1), **i32)
        self.deg = torch.empty(n_nodes, **i32)
        self.row_ptr = torch.empty(n_nodes + 1, **i32)
        ws = torch.empty(int(native.lib().csmpn_csr_workspace_bytes(E, n_nodes)), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            # validates edge_index against [0, n_nodes) (raises CsmpnError); one host sync per complex
            check(native.lib().csmpn_csr_build(ei.data_ptr(), E, n_nodes, self.perm.data_ptr(), self.src.data_ptr(),
                                               self.dst.data_ptr(), self.deg.data_ptr(), self.row_ptr.data_ptr(),
                                               ws.data_ptr(), ws.numel(), 0, _stream(dev)))
            t1.record()
            t1.synchronize()
            self.build_ms = t0.elapsed_time(t1)


class CsrSlice:
    """Rows [lo, hi) of a target-sorted adjacency (views; for launching the edge forward in pieces)."""

    __slots__ = ("perm", "src", "dst", "n_edges", "n_nodes")

    def __init__(self, csr: Csr, lo: int, hi: int):
        self.perm, self.src, self.dst = csr.perm[lo:hi], csr.src[lo:hi], csr.dst[lo:hi]
        self.n_edges, self.n_nodes = hi - lo, csr.n_nodes


def get_csr(edge_index: torch.Tensor, n_nodes: int) -> Csr:
    """CSR cached on the edge_index tensor object itself (complexes are static:
    every layer of a model and every epoch pass the same tensor)."""
    cached = getattr(edge_index, "_csmpn_csr", None)
    if cached is not None and cached[0] == edge_index._version and cached[1].n_nodes == n_nodes:
        return cached[1]
    csr = Csr(edge_index, n_nodes)
    try:
        edge_index._csmpn_csr = (edge_index._version, csr)
    except Exception:
        pass
    return csr


# --------------------------------------------------------------------------------- EGCL


class EgclSpec:
    """Static description of one EGCL layer for the autograd Function."""

    def __init__(self, edge: CemlpBinding, node: CemlpBinding, channels: int, out_channels: int,
                 edge_attr_channels: int, node_attr_channels: int, aggr: str, residual: bool):
        if aggr not in ("mean", "sum", "add"):
            raise native.CsmpnError(f"aggr={aggr!r} not supported by the HIP path (mean | sum | add)")
        self.edge, self.node = edge, node
        self.C, self.O = channels, out_channels
        self.A, self.T = edge_attr_channels, node_attr_channels
        self.mean = 1 if aggr == "mean" else 0
        self.residual = 1 if residual else 0


class HipBackend:
    """The four stages of one EGCL layer as C-ABI calls (device tensors in, device tensors out).
    The sharded layer (csmpn_hip.sharded) composes the same stages with collectives."""

    @staticmethod
    def build_csr(edge_index, n_nodes):
        return get_csr(edge_index, n_nodes)

    @staticmethod
    @_on_device_of(2)
    def edge_forward(spec, csr, h, edge_attr, pe, save=True, agg=None, saved=None):
        """Returns (agg, state); state = (workspace with packed weights, saved block inputs).
        agg: optional [N, O, D] buffer to accumulate into (zeroed by the caller); saved: optional
        buffer for the saved block inputs of these csr.n_edges rows."""
        e = spec.edge
        e.bind(pe)
        N, D = h.shape[0], e.D
        if agg is None:
            agg = torch.zeros(N, spec.O, D, dtype=torch.float32, device=h.device)
        ws = e.workspace(h.device)
        # CSMPN_FLAG_SAVE_STATE only for a buffer of our own, laid out for exactly these rows (a caller's buffer may be a
        # slice of a larger one: sharded split forward); the backward finds the flag on the tensor
        st_flag = 0
        if saved is None and save:
            saved = e.new_saved(csr.n_edges, h.device, _SAVE_STATE)
            if _SAVE_STATE and saved is not None:
                st_flag = native.FLAG_SAVE_STATE
                saved.csmpn_save_state = True
        det = deterministic_request()
        if det and not isinstance(csr, Csr):
            # a slice of the adjacency (graph-segment steps): only an explicit request is an error; the inherited
            # torch.use_deterministic_algorithms(True) of the reference's seeding warns once and keeps the atomics
            if det == "hard":
                raise native.CsmpnError("deterministic aggregation needs the whole adjacency (no sliced launches: "
                                        "set CSMPN_SPLIT_FWD=0)")
            _warn_soft_slice()
            det = None
        if det:
            rows = torch.empty(max(csr.n_edges, 1), spec.O, D, dtype=torch.float32, device=h.device)
            rc = native.lib().csmpn_egcl_edge_forward(
                e.metric_arr, e.n, e.params, e.nblk, h.data_ptr(), spec.C, _ptr(edge_attr), spec.A,
                csr.perm.data_ptr(), csr.src.data_ptr(), csr.dst.data_ptr(), csr.n_edges, N, rows.data_ptr(),
                _ptr(saved), ws.data_ptr(), ws.numel(), native.FLAG_DETERMINISTIC | st_flag, _stream(h.device))
            if not (det == "soft" and _soft_fallback(rc)):
                check(rc)
                segment_reduce(rows, agg, add=(csr.row_ptr, None))
                return agg, (ws, saved)
        check(native.lib().csmpn_egcl_edge_forward(
            e.metric_arr, e.n, e.params, e.nblk, h.data_ptr(), spec.C, _ptr(edge_attr), spec.A,
            csr.perm.data_ptr(), csr.src.data_ptr(), csr.dst.data_ptr(), csr.n_edges, N, agg.data_ptr(),
            _ptr(saved), ws.data_ptr(), ws.numel(), st_flag, _stream(h.device)))
        return agg, (ws, saved)

    @staticmethod
    @_on_device_of(2)
    def node_forward(spec, deg, h, agg, node_attr, pn, save=True):
        nd = spec.node
        nd.bind(pn)
        N, D = h.shape[0], nd.D
        out = torch.empty(N, nd.out_features, D, dtype=torch.float32, device=h.device)
        ws = nd.workspace(h.device)
        saved = nd.new_saved(N, h.device, _SAVE_STATE) if save else None
        st_flag = 0
        if _SAVE_STATE and saved is not None:
            st_flag = native.FLAG_SAVE_STATE
            saved.csmpn_save_state = True

        def call(flags):
            return native.lib().csmpn_egcl_node_forward(
                nd.metric_arr, nd.n, nd.params, nd.nblk, h.data_ptr(), spec.C, agg.data_ptr(), spec.O,
                _ptr(node_attr), spec.T, deg.data_ptr(), spec.mean, spec.residual, N, out.data_ptr(),
                _ptr(saved), ws.data_ptr(), ws.numel(), flags, _stream(h.device))

        det = deterministic_request()   # node stage: the flag only selects kernels with atomic-free parameter sums
        rc = call((native.FLAG_DETERMINISTIC if det else 0) | st_flag)
        if det == "soft" and _soft_fallback(rc):
            rc = call(st_flag)
        check(rc)
        return out, (ws, saved)

    @staticmethod
    @_on_device_of(2)
    def node_backward(spec, deg, h, agg, node_attr, pn, gout, want_gna, state=None, gflat=None, fused_into=None):
        """state: the workspace node_forward returned (its packed weights are reused);
        gflat: optional zeroed flat gradient buffer (CemlpBinding.grad_floats elements)."""
        nd = spec.node
        nd.bind(pn)
        N, D, dev = h.shape[0], nd.D, h.device
        _flat, views = nd.new_grads(pn, dev, gflat, fused_into=fused_into)
        gh = torch.empty_like(h)
        g_agg = torch.empty(N, spec.O, D, dtype=torch.float32, device=dev)
        g_na = torch.empty_like(node_attr) if (node_attr is not None and want_gna) else None
        ws, saved = state if state is not None else (nd.workspace(dev), None)
        flags = native.FLAG_WEIGHTS_PACKED if state is not None else 0
        if getattr(saved, "csmpn_save_state", False):
            flags |= native.FLAG_SAVE_STATE

        def call(fl):
            return native.lib().csmpn_egcl_node_backward(
                nd.metric_arr, nd.n, nd.params, nd.grads, nd.nblk, h.data_ptr(), spec.C, agg.data_ptr(), spec.O,
                _ptr(node_attr), spec.T, deg.data_ptr(), spec.mean, spec.residual, N, gout.data_ptr(),
                gh.data_ptr(), g_agg.data_ptr(), _ptr(g_na), _ptr(saved), ws.data_ptr(), ws.numel(), fl, _stream(dev))

        det = deterministic_request()
        rc = call(flags | (native.FLAG_DETERMINISTIC if det else 0))
        if det == "soft" and _soft_fallback(rc):
            rc = call(flags)
        check(rc)
        return gh, g_agg, g_na, views

    @staticmethod
    @_on_device_of(2)
    def edge_backward(spec, csr, h, edge_attr, pe, g_agg, gh, want_gea, state=None, gflat=None, fused_into=None):
        """gh is accumulated in place (+= scatter of +-d/d(h_i - h_j)); gflat as in node_backward."""
        e = spec.edge
        e.bind(pe)
        N, dev = h.shape[0], h.device
        _flat, views = e.new_grads(pe, dev, gflat, fused_into=fused_into)
        g_ea = torch.empty_like(edge_attr) if (edge_attr is not None and want_gea) else None
        ws, saved = state if state is not None else (e.workspace(dev), None)
        flags = native.FLAG_WEIGHTS_PACKED if state is not None else 0
        if getattr(saved, "csmpn_save_state", False):
            flags |= native.FLAG_SAVE_STATE
        det = deterministic_request()
        if det and not isinstance(csr, Csr):
            if det == "hard":
                raise native.CsmpnError("deterministic aggregation needs the whole adjacency (no sliced launches)")
            _warn_soft_slice()
            det = None
        if det:
            rows = torch.empty(max(csr.n_edges, 1), spec.C, e.D, dtype=torch.float32, device=dev)
            rc = native.lib().csmpn_egcl_edge_backward(
                e.metric_arr, e.n, e.params, e.grads, e.nblk, h.data_ptr(), spec.C, _ptr(edge_attr), spec.A,
                csr.perm.data_ptr(), csr.src.data_ptr(), csr.dst.data_ptr(), csr.n_edges, N, g_agg.data_ptr(),
                rows.data_ptr(), _ptr(g_ea), _ptr(saved), ws.data_ptr(), ws.numel(),
                flags | native.FLAG_DETERMINISTIC, _stream(dev))
            if not (det == "soft" and _soft_fallback(rc)):
                check(rc)
                segment_reduce(rows, gh, add=(csr.row_ptr, None), sub=csr.source_order())
                return g_ea, views
        check(native.lib().csmpn_egcl_edge_backward(
            e.metric_arr, e.n, e.params, e.grads, e.nblk, h.data_ptr(), spec.C, _ptr(edge_attr), spec.A,
            csr.perm.data_ptr(), csr.src.data_ptr(), csr.dst.data_ptr(), csr.n_edges, N, g_agg.data_ptr(),
            gh.data_ptr(), _ptr(g_ea), _ptr(saved), ws.data_ptr(), ws.numel(), flags, _stream(dev)))
        return g_ea, views


# CSMPN_FLAG_SAVE_STATE (round 4): the EGCL stage forwards also store every block's output in front of its layer norm and the
# backwards read it instead of recomputing linear_left + the geometric product. Honoured by the Cl(3,0) 8-channel kernels
# (S1: step 0.1915 -> 0.1866 ms), a no-op elsewhere; CSMPN_SAVE_STATE=0 turns it off (A/B runs).
_SAVE_STATE = os.environ.get("CSMPN_SAVE_STATE", "1") not in ("0", "")


def _check_egcl_inputs(spec, csr, h, edge_attr, node_attr, n_edges):
    D = spec.edge.D
    if h.dim() != 3 or h.shape[1] != spec.C or h.shape[2] != D:
        raise RuntimeError(f"h must be [N, {spec.C}, {D}], got {tuple(h.shape)}")
    if (edge_attr is None) != (spec.A == 0) or (node_attr is None) != (spec.T == 0):
        raise RuntimeError("edge_attr/node_attr presence does not match edge_attr_features/node_attr_features")
    if edge_attr is not None and tuple(edge_attr.shape) != (n_edges, spec.A, D):
        raise RuntimeError(f"edge_attr must be [{n_edges}, {spec.A}, {D}], got {tuple(edge_attr.shape)}")
    if node_attr is not None and tuple(node_attr.shape) != (h.shape[0], spec.T, D):
        raise RuntimeError(f"node_attr must be [{h.shape[0]}, {spec.T}, {D}], got {tuple(node_attr.shape)}")


class _EgclFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, edge_attr, node_attr, spec: EgclSpec, csr: Csr, *params):
        _require_device(h, "EGCL input h")
        _check_egcl_inputs(spec, csr, h, edge_attr, node_attr, csr.n_edges)
        h = h.contiguous()
        if edge_attr is not None:
            _require_device(edge_attr, "edge_attr")
            edge_attr = edge_attr.contiguous()
        if node_attr is not None:
            _require_device(node_attr, "node_attr")
            node_attr = node_attr.contiguous()
        ne = spec.edge.nblk * NP
        pe, pn = params[:ne], params[ne:]
        # one memset for everything that must start from zero: the aggregate and, when a backward
        # will follow, both models' flat gradient buffers
        spec.edge.bind(pe); spec.node.bind(pn)
        n_agg = h.shape[0] * spec.O * spec.edge.D
        ctx.param_refs = params
        fuse = _fusable(params, h.device)   # gradients go straight into p.grad: no flat buffers to zero
        n_ge, n_gn = (spec.edge.grad_floats(pe), spec.node.grad_floats(pn)) if (any(ctx.needs_input_grad) and not fuse) else (0, 0)
        zeros = torch.zeros(n_agg + n_ge + n_gn, dtype=torch.float32, device=h.device)
        agg = zeros[:n_agg].view(h.shape[0], spec.O, spec.edge.D)
        ctx.gflats = (zeros[n_agg:n_agg + n_ge], zeros[n_agg + n_ge:]) if n_ge else None
        agg, st_e = HipBackend.edge_forward(spec, csr, h, edge_attr, pe, agg=agg)
        out, st_n = HipBackend.node_forward(spec, csr.deg, h, agg, node_attr, pn)
        ctx.spec, ctx.csr, ctx.st_e, ctx.st_n = spec, csr, st_e, st_n
        ctx.has_ea, ctx.has_na = edge_attr is not None, node_attr is not None
        ctx.mask = [p is not None for p in params]
        saved = [h, agg]
        if ctx.has_ea:
            saved.append(edge_attr)
        if ctx.has_na:
            saved.append(node_attr)
        ctx.save_for_backward(*saved, *[p for p in params if p is not None])
        return out

    @staticmethod
    def backward(ctx, gout):
        spec, csr = ctx.spec, ctx.csr
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
        ne = spec.edge.nblk * NP
        pe, pn = params[:ne], params[ne:]
        gout = gout.contiguous()
        gfe, gfn = ctx.gflats if ctx.gflats is not None else (None, None)
        ctx.gflats = None   # zeroed once: a second backward through the same graph allocates afresh
        refs = ctx.param_refs
        fuse = _fusable(refs, h.device)
        gh, g_agg, g_na, views_n = HipBackend.node_backward(spec, csr.deg, h, agg, node_attr, pn, gout,
                                                            ctx.needs_input_grad[2], ctx.st_n, gfn,
                                                            fused_into=refs[ne:] if fuse else None)
        g_ea, views_e = HipBackend.edge_backward(spec, csr, h, edge_attr, pe, g_agg, gh, ctx.needs_input_grad[1],
                                                 ctx.st_e, gfe, fused_into=refs[:ne] if fuse else None)
        return (gh, g_ea, g_na, None, None, *views_e, *views_n)


def egcl_apply(h, edge_attr, node_attr, spec: EgclSpec, csr: Csr, params):
    return _EgclFn.apply(h, edge_attr, node_attr, spec, csr, *params)


# --------------------------------------------------------------------------------- geometric product


class _GpFn(torch.autograd.Function):
    @staticmethod
    @_on_device_of(1)
    def forward(ctx, a, b, metric):
        _require_device(a, "geometric_product operand")
        _require_device(b, "geometric_product operand")
        shape = torch.broadcast_shapes(a.shape, b.shape)
        a2 = a.expand(shape).contiguous()
        b2 = b.expand(shape).contiguous()
        n = len(metric)
        D = 1 << n
        rows = a2.numel() // D
        out = torch.empty_like(a2)
        check(native.lib().csmpn_geometric_product_forward(native.metric_array(metric), n, a2.data_ptr(),
                                                           b2.data_ptr(), out.data_ptr(), rows, _stream(a.device)))
        ctx.save_for_backward(a2, b2)
        ctx.metric, ctx.shapes = metric, (a.shape, b.shape)
        return out

    @staticmethod
    @_on_device_of(1)
    def backward(ctx, gout):
        a2, b2 = ctx.saved_tensors
        metric = ctx.metric
        n = len(metric)
        D = 1 << n
        gout = gout.contiguous()
        ga, gb = torch.zeros_like(a2), torch.zeros_like(b2)
        check(native.lib().csmpn_geometric_product_backward(native.metric_array(metric), n, a2.data_ptr(),
                                                            b2.data_ptr(), gout.data_ptr(), ga.data_ptr(),
                                                            gb.data_ptr(), a2.numel() // D, _stream(a2.device)))
        sa, sb = ctx.shapes
        return ga.sum_to_size(sa), gb.sum_to_size(sb), None


# --------------------------------------------------------------------------------- standalone MVLinear


class _MVLinearFn(torch.autograd.Function):
    """y[b,o,d] = sum_i W[o,i,grade(d)] x[b,i,d] (+ bias on blade 0) through csmpn_mvlinear_*
    (cegnn_utils.py:326-338), for the MVLinear calls outside a CEMLP."""

    @staticmethod
    @_on_device_of(1)
    def forward(ctx, x, weight, bias, n):
        _require_device(x, "MVLinear input")
        for name, t in (("weight", weight), ("bias", bias)):
            if t is not None and (t.device != x.device or t.dtype != torch.float32):
                raise RuntimeError(f"MVLinear {name} must be float32 on {x.device}, got {t.dtype} on {t.device}")
        x = x.contiguous()
        w = weight.contiguous()
        b = bias.contiguous() if bias is not None else None
        rows, I, D = x.shape
        O = w.shape[0]
        if D != (1 << n) or w.shape[1] != I:
            raise RuntimeError(f"MVLinear: input {tuple(x.shape)} does not match weight {tuple(w.shape)} (n={n})")
        y = torch.empty(rows, O, D, dtype=torch.float32, device=x.device)
        check(native.lib().csmpn_mvlinear_forward(n, x.data_ptr(), w.data_ptr(), _ptr(b), rows, I, O,
                                                  1 if w.dim() == 3 else 0, y.data_ptr(), _stream(x.device)))
        ctx.save_for_backward(x, w)
        ctx.n, ctx.has_bias = n, bias is not None
        # the caller's parameter objects: with fused gradient accumulation the backward kernel adds into their .grad
        ctx.param_refs = (weight if weight.is_contiguous() else None, bias if (bias is None or bias.is_contiguous()) else False)
        return y

    @staticmethod
    @_on_device_of(1)
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        rows, I, D = x.shape
        O = w.shape[0]
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        need_w, need_b = ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        pw, pb = ctx.param_refs
        if (pw is not None and pb is not False and (need_w or need_b)
                and _fusable([pw if need_w else None, pb if need_b else None], x.device)
                and (not need_b or pb.grad.numel() == O)):
            # fused accumulation (set_fused_grad_accumulation): the kernel's atomics land in the parameters' own .grad - no
            # zero fills, no AccumulateGrad adds (6 + 6 launches per md17 step)
            check(native.lib().csmpn_mvlinear_backward(ctx.n, x.data_ptr(), w.data_ptr(), gy.data_ptr(), rows, I, O,
                                                       1 if w.dim() == 3 else 0, _ptr(gx), pw.grad.data_ptr() if need_w else None,
                                                       pb.grad.data_ptr() if need_b else None, _stream(x.device)))
            return gx, None, None, None
        gw = torch.zeros_like(w) if need_w else None   # None: frozen weight (bias still gets its gradient)
        gb = torch.zeros(1, O, 1, dtype=torch.float32, device=x.device) if need_b else None
        check(native.lib().csmpn_mvlinear_backward(ctx.n, x.data_ptr(), w.data_ptr(), gy.data_ptr(), rows, I, O,
                                                   1 if w.dim() == 3 else 0, _ptr(gx), _ptr(gw), _ptr(gb),
                                                   _stream(x.device)))
        return gx, gw, gb, None


# --------------------------------------------------------------------------------- the small layers on their own

_SMALL_LAUNCHES = 0


def small_layer_launches() -> int:
    """Launches of the standalone small-layer entry points so far (tests assert that the HIP path ran)."""
    return _SMALL_LAUNCHES


def _small_args(x, what, params):
    _require_device(x, f"{what} input")
    if x.dim() != 3:
        raise RuntimeError(f"{what}: the HIP path takes [rows, channels, D] inputs, got {tuple(x.shape)}")
    for t in params:
        if t.device != x.device or t.dtype != torch.float32:
            raise RuntimeError(f"{what} parameters must be float32 on {x.device}, got {t.dtype} on {t.device}")
    return x.contiguous()


def _count():
    global _SMALL_LAUNCHES
    _SMALL_LAUNCHES += 1


class _MVSiLUFn(torch.autograd.Function):
    """csmpn_mvsilu_* (cegnn_utils.py:53-83, invariant "mag2"); a, b [1, C, G]."""

    @staticmethod
    @_on_device_of(1)
    def forward(ctx, x, a, b, metric):
        x = _small_args(x, "MVSiLU", (a, b))
        a2, b2 = a.contiguous(), b.contiguous()
        y = torch.empty_like(x)
        check(native.lib().csmpn_mvsilu_forward(native.metric_array(metric), len(metric), x.data_ptr(), a2.data_ptr(),
                                                b2.data_ptr(), x.shape[0], x.shape[1], y.data_ptr(), _stream(x.device)))
        _count()
        ctx.save_for_backward(x, a2, b2)
        ctx.metric = metric
        return y

    @staticmethod
    @_on_device_of(1)
    def backward(ctx, gy):
        x, a, b = ctx.saved_tensors
        gy = gy.contiguous()
        gx, ga, gb = torch.empty_like(x), torch.zeros_like(a), torch.zeros_like(b)
        check(native.lib().csmpn_mvsilu_backward(native.metric_array(ctx.metric), len(ctx.metric), x.data_ptr(),
                                                 a.data_ptr(), b.data_ptr(), gy.data_ptr(), x.shape[0], x.shape[1],
                                                 gx.data_ptr(), ga.data_ptr(), gb.data_ptr(), _stream(x.device)))
        _count()
        return gx, ga, gb, None


class _RowParamFn(torch.autograd.Function):
    """NormalizationLayer (csmpn_mvnorm_*, a [C, G]) and MVLayerNorm (csmpn_mvlayernorm_*, a [1, C]): one
    parameter tensor, same call shape."""

    @staticmethod
    @_on_device_of(1)
    def forward(ctx, x, a, metric, which):
        x = _small_args(x, which, (a,))
        a2 = a.contiguous()
        y = torch.empty_like(x)
        fwd = getattr(native.lib(), f"csmpn_{which}_forward")
        check(fwd(native.metric_array(metric), len(metric), x.data_ptr(), a2.data_ptr(), x.shape[0], x.shape[1],
                  y.data_ptr(), _stream(x.device)))
        _count()
        ctx.save_for_backward(x, a2)
        ctx.metric, ctx.which = metric, which
        return y

    @staticmethod
    @_on_device_of(1)
    def backward(ctx, gy):
        x, a = ctx.saved_tensors
        gy = gy.contiguous()
        gx, ga = torch.empty_like(x), torch.zeros_like(a)
        bwd = getattr(native.lib(), f"csmpn_{ctx.which}_backward")
        check(bwd(native.metric_array(ctx.metric), len(ctx.metric), x.data_ptr(), a.data_ptr(), gy.data_ptr(),
                  x.shape[0], x.shape[1], gx.data_ptr(), ga.data_ptr(), _stream(x.device)))
        _count()
        return gx, ga, None, None


class _WgpFn(torch.autograd.Function):
    """The path-weighted geometric product of SteerableGeometricProductLayer (csmpn_wgp_*,
    cegnn_utils.py:126-152): z, r [rows, C, D], weight [C, P]."""

    @staticmethod
    @_on_device_of(1)
    def forward(ctx, z, r, weight, metric):
        z = _small_args(z, "weighted geometric product", (r, weight))
        r2, w2 = r.contiguous(), weight.contiguous()
        if r2.shape != z.shape or w2.shape[0] != z.shape[1]:
            raise RuntimeError(f"weighted geometric product: shapes {tuple(z.shape)}, {tuple(r2.shape)}, {tuple(w2.shape)}")
        y = torch.empty_like(z)
        check(native.lib().csmpn_wgp_forward(native.metric_array(metric), len(metric), z.data_ptr(), r2.data_ptr(),
                                             w2.data_ptr(), z.shape[0], z.shape[1], y.data_ptr(), _stream(z.device)))
        _count()
        ctx.save_for_backward(z, r2, w2)
        ctx.metric = metric
        return y

    @staticmethod
    @_on_device_of(1)
    def backward(ctx, gy):
        z, r, w = ctx.saved_tensors
        gy = gy.contiguous()
        gz, gr, gw = torch.empty_like(z), torch.empty_like(r), torch.zeros_like(w)
        check(native.lib().csmpn_wgp_backward(native.metric_array(ctx.metric), len(ctx.metric), z.data_ptr(), r.data_ptr(),
                                              w.data_ptr(), gy.data_ptr(), z.shape[0], z.shape[1], gz.data_ptr(),
                                              gr.data_ptr(), gw.data_ptr(), _stream(z.device)))
        _count()
        return gz, gr, gw, None


def _metric_of(algebra):
    mt = getattr(algebra, "metric_tuple", None)     # host copy: no device synchronisation per call
    return tuple(float(m) for m in (mt if mt is not None else algebra.metric.tolist()))


def small_layer_on_hip(algebra, x, *params) -> bool:
    """True when a standalone small layer takes the HIP entry points: float32 [rows, C, D] device input, a
    signature the kernels are compiled for, at most 256 channels, parameters on the same device."""
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.shape[1] <= 256
            and getattr(algebra, "hip_supported", False) and all(p.is_cuda for p in params))


def mvsilu_apply(algebra, x, a, b):
    return _MVSiLUFn.apply(x, a, b, _metric_of(algebra))


def mvnorm_apply(algebra, x, a):
    return _RowParamFn.apply(x, a, _metric_of(algebra), "mvnorm")


def mvlayernorm_apply(algebra, x, a):
    return _RowParamFn.apply(x, a, _metric_of(algebra), "mvlayernorm")


def wgp_apply(algebra, z, r, weight):
    return _WgpFn.apply(z, r, weight, _metric_of(algebra))


def mvlinear_apply(x, weight, bias, n):
    return _MVLinearFn.apply(x, weight, bias, int(n))


def geometric_product_apply(a, b, metric):
    return _GpFn.apply(a, b, tuple(float(m) for m in metric))


# --------------------------------------------------------------------------------- callers either side (SURVEY §8(f)-1,2)


@_on_device_of(2)
def simplex_rows(n: int, blocks, verts: torch.Tensor) -> torch.Tensor:
    """Input rows of the simplex feature embedding (csmpn_simplex_rows). blocks: [(tensor [S, K, n_g], grade)],
    verts [rows, d+1] int64 batch rows of the vertices in one vertex order. The feature tensors are data
    (no gradient flows to them in the reference's models): tensors that require grad are refused."""
    if not verts.is_cuda:
        raise RuntimeError("simplex vertex table must live on the GPU (no CPU fallback)")
    if verts.dtype != torch.int64 or verts.dim() != 2:
        raise RuntimeError("verts must be int64 [rows, vertices]")
    verts = verts.contiguous()
    arr = (native.VertexBlock * len(blocks))()
    keep, ctot, S = [], 0, None
    for i, (t, grade) in enumerate(blocks):
        if t.requires_grad:
            raise RuntimeError("simplex_rows: feature tensors are inputs (no gradient); detach them")
        _require_device(t, "simplex feature block")
        t = t.contiguous()
        if t.dim() != 3 or t.dtype != torch.float32 or t.device != verts.device:
            raise RuntimeError(f"feature block {i} must be float32 [S, K, n_g] on {verts.device}")
        if t.shape[2] != _binom(n, int(grade)):
            raise RuntimeError(f"feature block {i}: last dimension {t.shape[2]} is not the size of grade {grade}")
        if S is None:
            S = t.shape[0]
        elif S != t.shape[0]:
            raise RuntimeError("feature blocks disagree on the number of rows")
        keep.append(t)
        arr[i].data, arr[i].channels, arr[i].grade = t.data_ptr(), t.shape[1], int(grade)
        ctot += verts.shape[1] * t.shape[1]
    out = torch.empty(verts.shape[0], ctot, 1 << n, dtype=torch.float32, device=verts.device)
    check(native.lib().csmpn_simplex_rows(n, arr, len(blocks), verts.data_ptr(), verts.shape[0], verts.shape[1], S,
                                          out.data_ptr(), _stream(verts.device)))
    return out


class _TypeAttrFn(torch.autograd.Function):
    """(node_attr [S, K, D], edge_attr [E, 2 K, D]) from a table of per-type features through csmpn_type_attr_* (one launch
    each way). table [T, K] float32; types [S], src / dst [E] int32 (constants of the batch)."""

    @staticmethod
    @_on_device_of(1)
    def forward(ctx, table, types, src, dst, n):
        _require_device(table, "type-attribute table")
        tab = table.contiguous()
        T, K = tab.shape
        S, E, D = int(types.shape[0]), int(src.shape[0]), 1 << n
        node_attr = torch.empty(S, K, D, dtype=torch.float32, device=tab.device)
        edge_attr = torch.empty(E, 2 * K, D, dtype=torch.float32, device=tab.device)
        check(native.lib().csmpn_type_attr_forward(n, tab.data_ptr(), T, K, types.data_ptr(), S, src.data_ptr(), dst.data_ptr(), E,
                                                   node_attr.data_ptr(), edge_attr.data_ptr(), _stream(tab.device)))
        ctx.save_for_backward(types, src, dst)
        ctx.dims = (n, T, K)
        ctx.table_ref = table if table.is_contiguous() else None
        return node_attr, edge_attr

    @staticmethod
    @_on_device_of(1)
    def backward(ctx, g_node, g_edge):
        types, src, dst = ctx.saved_tensors
        n, T, K = ctx.dims
        if not ctx.needs_input_grad[0] or (g_node is None and g_edge is None):
            return None, None, None, None, None
        g_node = g_node.contiguous() if g_node is not None else None
        g_edge = g_edge.contiguous() if g_edge is not None else None
        ref = ctx.table_ref
        fuse = ref is not None and _fusable([ref], types.device)
        g_tab = ref.grad if fuse else torch.zeros(T, K, dtype=torch.float32, device=types.device)
        check(native.lib().csmpn_type_attr_backward(n, T, K, types.data_ptr(), int(types.shape[0]), src.data_ptr(), dst.data_ptr(),
                                                    int(src.shape[0]), _ptr(g_node), _ptr(g_edge), g_tab.data_ptr(),
                                                    _stream(types.device)))
        return (None if fuse else g_tab), None, None, None, None


def type_attr_apply(table, types_i32, src_i32, dst_i32, n):
    return _TypeAttrFn.apply(table, types_i32, src_i32, dst_i32, int(n))


def _binom(n, k):
    import math
    return math.comb(n, k)


class _ReadoutMseFn(torch.autograd.Function):
    """loss_g = (mean_{s in g} MVLinear(x)[s, 0, blade 0] - target_g)^2 (hulls_cssmpnn.py:93,155-164)."""

    @staticmethod
    @_on_device_of(1)
    def forward(ctx, x, weight, bias, graph_ptr, target, n):
        _require_device(x, "readout input")
        x = x.contiguous()
        S, Cc, D = x.shape
        if weight.shape[0] != 1 or weight.shape[1] != Cc or D != (1 << n):
            raise RuntimeError(f"readout: weight {tuple(weight.shape)} does not fit input {tuple(x.shape)} with out_features = 1")
        w = weight.contiguous()
        stride = w.shape[2] if w.dim() == 3 else 1
        b = bias.contiguous() if bias is not None else None
        B = graph_ptr.shape[0] - 1
        tgt = target.contiguous().float()
        pred = torch.empty(B, dtype=torch.float32, device=x.device)
        loss = torch.empty(B, dtype=torch.float32, device=x.device)
        xs = torch.empty(B, Cc, dtype=torch.float32, device=x.device)
        check(native.lib().csmpn_readout_mse_forward(n, x.data_ptr(), w.data_ptr(), stride, _ptr(b), S, Cc,
                                                     graph_ptr.data_ptr(), B, tgt.data_ptr(), pred.data_ptr(),
                                                     loss.data_ptr(), xs.data_ptr(), _stream(x.device)))
        ctx.save_for_backward(w, graph_ptr, pred, tgt, xs)
        ctx.n, ctx.shape, ctx.has_bias = n, (S, Cc, D), bias is not None
        ctx.mark_non_differentiable(pred)
        return loss, pred

    @staticmethod
    @_on_device_of(1)
    def backward(ctx, g_loss, _g_pred):
        w, graph_ptr, pred, tgt, xs = ctx.saved_tensors
        S, Cc, D = ctx.shape
        B = graph_ptr.shape[0] - 1
        cnt = (graph_ptr[1:] - graph_ptr[:-1]).clamp(min=1).float()
        dpred = g_loss.contiguous() * 2.0 * (pred - tgt)            # d/d pred_g
        coef = (dpred / cnt).contiguous()
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty(S, Cc, D, dtype=torch.float32, device=w.device)
            stride = w.shape[2] if w.dim() == 3 else 1
            check(native.lib().csmpn_readout_mse_backward(ctx.n, w.data_ptr(), stride, S, Cc, graph_ptr.data_ptr(), B,
                                                          coef.data_ptr(), gx.data_ptr(), _stream(w.device)))
        gw = None
        if ctx.needs_input_grad[1]:
            gw = torch.zeros_like(w)
            gc = coef @ xs                                           # [C]
            if w.dim() == 3:
                gw[0, :, 0] = gc
            else:
                gw[0, :] = gc
        gb = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            nonempty = ((graph_ptr[1:] - graph_ptr[:-1]) > 0).float()
            gb = (dpred * nonempty).sum().reshape(1, 1, 1)
        return gx, gw, gb, None, None, None


class _ReadoutTrajFn(torch.autograd.Function):
    """Vector readout + loss of the trajectory models (csmpn_readout_traj_*; md17_cssmpnn.py:165-176,
    motion_cssmpnn.py:150-168, nba_cssmpnn.py:176-191). Returns (per_graph [B, 3] = MSE / ADE / FDE, per_vertex [V] MSE,
    pred [V, O, n]); gradients w.r.t. x and the MVLinear weight (its grade-1 entries)."""

    @staticmethod
    @_on_device_of(1)
    def forward(ctx, x, weight, loc, target, tables, n):
        _require_device(x, "readout input")
        x = x.contiguous()
        S, Cc, D = x.shape
        w = weight.contiguous()
        if w.dim() != 3 or w.shape[1] != Cc or w.shape[2] < 2 or D != (1 << n):
            raise RuntimeError(f"trajectory readout: weight {tuple(weight.shape)} does not fit input {tuple(x.shape)}")
        O = int(w.shape[0])
        vrows, vof, trow, gov, vptr = tables["vrows"], tables["vertex_of_row"], tables["trow"], tables["graph_of_vertex"], tables["vptr"]
        V, B = int(gov.shape[0]), int(vptr.shape[0]) - 1
        if vrows is None and S != V:
            raise RuntimeError("trajectory readout: identity vertex rows need one input row per vertex")
        tgt = target.contiguous().float().reshape(-1, O, n)
        lc = None if loc is None else loc.contiguous().float().reshape(V, O, n)
        pred = torch.empty(V, O, n, dtype=torch.float32, device=x.device)
        per_graph = torch.empty(B, 3, dtype=torch.float32, device=x.device)
        per_vertex = torch.empty(V, dtype=torch.float32, device=x.device)
        check(native.lib().csmpn_readout_traj_forward(n, x.data_ptr(), Cc, _ptr(vrows), V, w.data_ptr(), O, int(w.shape[2]), _ptr(lc),
                                                      tgt.data_ptr(), _ptr(trow), vptr.data_ptr(), B, pred.data_ptr(),
                                                      per_graph.data_ptr(), per_vertex.data_ptr(), _stream(x.device)))
        ctx.save_for_backward(x, w, pred, tgt)
        ctx.tables, ctx.n = tables, n
        ctx.mark_non_differentiable(pred)
        return per_graph, per_vertex, pred

    @staticmethod
    @_on_device_of(1)
    def backward(ctx, g_graph, g_vertex, _g_pred):
        x, w, pred, tgt = ctx.saved_tensors
        t = ctx.tables
        S, Cc, D = x.shape
        V, O, n = pred.shape
        gg = None if g_graph is None else g_graph.contiguous().float()
        gv = None if g_vertex is None else g_vertex.contiguous().float()
        gx = torch.empty_like(x)
        gw = torch.zeros_like(w)
        scratch = torch.empty(V, O, n, dtype=torch.float32, device=x.device)
        check(native.lib().csmpn_readout_traj_backward(
            ctx.n, x.data_ptr(), Cc, S, _ptr(t["vrows"]), _ptr(t["vertex_of_row"]), V, w.data_ptr(), O, int(w.shape[2]),
            pred.data_ptr(), tgt.data_ptr(), _ptr(t["trow"]), t["graph_of_vertex"].data_ptr(), t["vptr"].data_ptr(), _ptr(gg),
            _ptr(gv), scratch.data_ptr(), gx.data_ptr(), gw.data_ptr(), _stream(x.device)))
        return gx, gw, None, None, None, None


def readout_traj_tables(graph_of_vertex, n_graphs, vertex_rows=None, n_rows=None, unscored_last=0):
    """Index tables of csmpn_readout_traj_* (int32, on the device of graph_of_vertex), built once per batch:
    vptr [B + 1] (the vertices of a graph are contiguous in the vertex list), graph_of_vertex [V]; with `vertex_rows` (rows of
    the layer output that are vertices) also its inverse vertex_of_row [n_rows]; unscored_last = k: the last k vertices of
    every graph are not scored (NBA: the ball) and the target rows number the scored ones consecutively."""
    gov = graph_of_vertex.to(torch.int64)
    dev = gov.device
    V = int(gov.shape[0])
    if V > 1 and bool((gov[1:] < gov[:-1]).any()):
        raise RuntimeError("trajectory readout: the vertices of a graph must be contiguous in the vertex list")
    cnt = torch.bincount(gov, minlength=n_graphs)
    vptr = torch.zeros(n_graphs + 1, dtype=torch.int64, device=dev)
    vptr[1:] = torch.cumsum(cnt, 0)
    trow = None
    if unscored_last:
        pos = torch.arange(V, device=dev) - vptr[:-1][gov]                 # position inside the graph
        scored = pos < (cnt[gov] - unscored_last)
        first = torch.cumsum(torch.cat([cnt.new_zeros(1), (cnt - unscored_last).clamp(min=0)[:-1]]), 0)
        trow = torch.where(scored, first[gov] + pos, torch.full_like(pos, -1)).to(torch.int32).contiguous()
    vrows = vof = None
    if vertex_rows is not None:
        vrows = vertex_rows.to(torch.int32).contiguous()
        vof = torch.full((int(n_rows),), -1, dtype=torch.int32, device=dev)
        vof[vertex_rows.long()] = torch.arange(V, dtype=torch.int32, device=dev)
    return {"vrows": vrows, "vertex_of_row": vof, "trow": trow, "graph_of_vertex": gov.to(torch.int32).contiguous(),
            "vptr": vptr.to(torch.int32).contiguous()}


def readout_traj(x, weight, loc, target, tables, n):
    """(per_graph [B, 3] = (MSE, ADE, FDE), per_vertex [V] MSE, pred [V, O, n]); tables = readout_traj_tables(...)."""
    return _ReadoutTrajFn.apply(x, weight, loc, target, tables, int(n))


def readout_traj_supported(head, x) -> bool:
    """MVLinear(subspaces=True) heads on device float32 rows within the kernel's limits."""
    w = getattr(head, "weight", None)
    return (x.is_cuda and x.dtype == torch.float32 and w is not None and w.dim() == 3 and w.shape[1] <= 64
            and w.shape[0] <= 1024 and w.shape[0] * w.shape[1] <= 4096)


def readout_mse(x, weight, bias, graph_ptr_i32, target, n):
    """(loss per graph, prediction per graph); graph_ptr_i32 [B+1] int32 device tensor (rows of a graph are contiguous)."""
    return _ReadoutMseFn.apply(x, weight, bias, graph_ptr_i32, target, int(n))
