"""ctypes binding of the C++ CPU twin (include/csmpn_cpu.h, oracle/cpu_twin/csmpn_cpu.cpp).
TEST INFRASTRUCTURE: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "_build", "libcsmpn_cpu.so")
LIB64_PATH = os.path.join(os.path.dirname(_HERE), "_build", "libcsmpn_cpu64.so")   # -DCSMPN_CPU_REAL64: float64 throughout
FIELDS = ("lin_w", "lin_b", "silu_a", "silu_b", "gp_w", "norm_a", "right_w", "left_w", "left_b", "ln_a")
KEYS = ("0.weight", "0.bias", "1.a", "1.b", "2.weight", "2.normalization.a", "2.linear_right.weight",
        "2.linear_left.weight", "2.linear_left.bias", "3.a")


class BlockParams(C.Structure):
    _fields_ = [("in_features", C.c_int32), ("out_features", C.c_int32), ("lin_subspaces", C.c_int32),
                ("reserved", C.c_int32)] + [(n, C.c_void_p) for n in FIELDS]


class BlockGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in FIELDS]


_libs = {}
_REAL = np.float32   # dtype of the library in use (set per call by egcl_layer / cemlp)


def lib(real64=False):
    path = LIB64_PATH if real64 else LIB_PATH
    if path not in _libs:
        if not os.path.exists(path):
            raise RuntimeError(f"{path} not found: make -C oracle")
        l = C.CDLL(path)
        l.csmpn_cpu_last_error.restype = C.c_char_p
        _libs[path] = l
    return _libs[path]


def _f32(a):
    """contiguous array in the dtype of the library in use (float32, or float64 for the REAL64 build)"""
    return np.ascontiguousarray(np.asarray(a.detach().cpu().numpy() if hasattr(a, "detach") else a, dtype=_REAL))


def _blocks(p, prefix, want_grads):
    """state_dict-style dict (numpy / torch tensors) -> (params array, grads array, kept arrays, grad dict)."""
    n = 0
    while f"{prefix}layers.{n}.0.weight" in p:
        n += 1
    bp, bg = (BlockParams * n)(), (BlockGrads * n)()
    keep, grads = [], {}
    for k in range(n):
        w = _f32(p[f"{prefix}layers.{k}.0.weight"])
        bp[k].out_features, bp[k].in_features = w.shape[0], w.shape[1]
        bp[k].lin_subspaces = 1 if w.ndim == 3 else 0
        for field, key in zip(FIELDS, KEYS):
            name = f"{prefix}layers.{k}.{key}"
            if name not in p or p[name] is None:
                continue
            a = _f32(p[name])
            keep.append(a)
            setattr(bp[k], field, a.ctypes.data)
            if want_grads:
                g = np.zeros_like(a)
                grads[name] = g
                setattr(bg[k], field, g.ctypes.data)
    return bp, bg, n, keep, grads


def egcl_layer(metric, params, h, edge_index, edge_attr=None, node_attr=None, aggr="mean", residual=True, gout=None,
               want_attr_grads=False, threads=0, real64=False):
    """One EGCL layer on the CPU twin. params: dict with the reference's state_dict keys
    (edge_model.layers.k.*, node_model.layers.k.*). Returns dict(out, gh, g_edge_attr, g_node_attr, grads).
    real64: the float64 build of the same source (inputs converted, outputs float64)."""
    global _REAL
    _REAL = np.float64 if real64 else np.float32
    L = lib(real64)
    m = _f32(metric)
    h = _f32(h)
    ei = np.ascontiguousarray(np.asarray(edge_index, dtype=np.int64))
    N, Cc, D = h.shape
    E = ei.shape[1]
    ea = None if edge_attr is None else _f32(edge_attr)
    na = None if node_attr is None else _f32(node_attr)
    bwd = gout is not None
    ebp, ebg, ne, k1, ge = _blocks(params, "edge_model.", bwd)
    nbp, nbg, nn, k2, gn = _blocks(params, "node_model.", bwd)
    O = int(nbp[nn - 1].out_features)
    out = np.empty((N, O, D), _REAL)
    go = None if not bwd else _f32(gout)
    gh = np.empty_like(h) if bwd else None
    gea = np.empty_like(ea) if (bwd and want_attr_grads and ea is not None) else None
    gna = np.empty_like(na) if (bwd and want_attr_grads and na is not None) else None
    ptr = lambda a: None if a is None else C.c_void_p(a.ctypes.data)
    rc = L.csmpn_egcl_layer_cpu(
        ptr(m), C.c_int(len(m)), ebp, ebg, C.c_int(ne), nbp, nbg, C.c_int(nn), ptr(h), C.c_int32(Cc), ptr(ei),
        C.c_int64(E), C.c_int64(N), ptr(ea), C.c_int32(0 if ea is None else ea.shape[1]), ptr(na),
        C.c_int32(0 if na is None else na.shape[1]), C.c_int32(1 if aggr == "mean" else 0), C.c_int32(1 if residual else 0),
        ptr(go), ptr(out), ptr(gh), ptr(gea), ptr(gna), C.c_int32(threads))
    if rc != 0:
        raise RuntimeError(f"csmpn_cpu error {rc}: {L.csmpn_cpu_last_error().decode()}")
    grads = {**ge, **gn}
    return dict(out=out, gh=gh, g_edge_attr=gea, g_node_attr=gna, grads=grads)


def cemlp(metric, params, x, gy=None, prefix="", threads=0, real64=False):
    global _REAL
    _REAL = np.float64 if real64 else np.float32
    L = lib(real64)
    m = _f32(metric)
    x = _f32(x)
    bwd = gy is not None
    bp, bg, n, keep, grads = _blocks(params, prefix, bwd)
    rows, D = x.shape[0], x.shape[2]
    y = np.empty((rows, int(bp[n - 1].out_features), D), _REAL)
    g = None if not bwd else _f32(gy)
    gx = np.empty_like(x) if bwd else None
    ptr = lambda a: None if a is None else C.c_void_p(a.ctypes.data)
    rc = L.csmpn_cemlp_cpu(ptr(m), C.c_int(len(m)), bp, bg, C.c_int(n), ptr(x), C.c_int64(rows), ptr(g), ptr(y),
                               ptr(gx), C.c_int32(threads))
    if rc != 0:
        raise RuntimeError(f"csmpn_cpu error {rc}: {L.csmpn_cpu_last_error().decode()}")
    return y, gx, grads
