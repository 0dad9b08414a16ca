"""ctypes binding of libcsmpn_hip.so (C-ABI declared in include/csmpn_hip.h).

The library is the product: there is no CPU or PyTorch-eager fallback. If it is
missing the import fails loudly with the build command.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CSMPN_LIB") or os.path.join(_HERE, "libcsmpn_hip.so")

ABI_VERSION = 2     # include/csmpn_hip.h: csmpn_abi_version()
MAX_BLOCKS = 4
FLAG_WEIGHTS_PACKED = 1
FLAG_NO_VALIDATE = 2
FLAG_DETERMINISTIC = 4
FLAG_SAVE_STATE = 8
ERR_UNSUPPORTED, ERR_INVALID, ERR_HIP = 1, 2, 3

# every symbol include/csmpn_hip.h declares
EXPORTS = (
    "csmpn_metric_supported",
    "csmpn_algebra_tables",
    "csmpn_geometric_product_forward",
    "csmpn_geometric_product_backward",
    "csmpn_cemlp_workspace_bytes",
    "csmpn_cemlp_saved_floats_per_row",
    "csmpn_cemlp_saved_floats",
    "csmpn_cemlp_forward",
    "csmpn_cemlp_backward",
    "csmpn_mvlinear_forward",
    "csmpn_mvlinear_backward",
    "csmpn_mvsilu_forward",
    "csmpn_mvsilu_backward",
    "csmpn_mvnorm_forward",
    "csmpn_mvnorm_backward",
    "csmpn_mvlayernorm_forward",
    "csmpn_mvlayernorm_backward",
    "csmpn_wgp_forward",
    "csmpn_wgp_backward",
    "csmpn_csr_workspace_bytes",
    "csmpn_csr_build",
    "csmpn_csr_source_order",
    "csmpn_segment_reduce",
    "csmpn_egcl_edge_forward",
    "csmpn_egcl_edge_backward",
    "csmpn_egcl_node_forward",
    "csmpn_egcl_node_backward",
    "csmpn_simplex_rows",
    "csmpn_embed_cemlp_forward",
    "csmpn_embed_cemlp_backward",
    "csmpn_type_attr_forward",
    "csmpn_type_attr_backward",
    "csmpn_readout_mse_forward",
    "csmpn_readout_mse_backward",
    "csmpn_readout_traj_forward",
    "csmpn_readout_traj_backward",
    "csmpn_last_error",
    "csmpn_last_kernel",
    "csmpn_abi_version",
    "csmpn_build_target",
)

PARAM_FIELDS = ("lin_w", "lin_b", "silu_a", "silu_b", "gp_w", "norm_a", "right_w", "left_w", "left_b", "ln_a")


class BlockParams(C.Structure):
    _fields_ = [
        ("in_features", C.c_int32),
        ("out_features", C.c_int32),
        ("lin_subspaces", C.c_int32),
        ("reserved", C.c_int32),
    ] + [(name, C.c_void_p) for name in PARAM_FIELDS]


class BlockGrads(C.Structure):
    _fields_ = [(name, C.c_void_p) for name in PARAM_FIELDS]


class VertexBlock(C.Structure):
    _fields_ = [("data", C.c_void_p), ("channels", C.c_int32), ("grade", C.c_int32)]


class NativeLibraryMissing(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryMissing(
            f"{LIB_PATH} not found. This package has no CPU/eager fallback: build the HIP "
            f"library first (python -c 'import __graft_entry__ as g; g.build()' or "
            f"make -C {os.path.join(os.path.dirname(_HERE), 'csrc')})."
        )
    lib = C.CDLL(LIB_PATH)
    lib.csmpn_abi_version.restype = C.c_int
    if lib.csmpn_abi_version() != ABI_VERSION:
        raise NativeLibraryMissing(f"{LIB_PATH} has ABI version {lib.csmpn_abi_version()}, this package binds version "
                                   f"{ABI_VERSION}: rebuild it (make -C {os.path.join(os.path.dirname(_HERE), 'csrc')})")
    vp, i32, i64, sz, u32 = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t, C.c_uint32
    fp = C.POINTER(C.c_float)
    bp, bg = C.POINTER(BlockParams), C.POINTER(BlockGrads)

    def sig(name, res, args):
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args

    sig("csmpn_metric_supported", C.c_int, [fp, C.c_int])
    sig("csmpn_algebra_tables", C.c_int, [fp, C.c_int, vp, vp, vp, vp, vp, vp])
    sig("csmpn_geometric_product_forward", C.c_int, [fp, C.c_int, vp, vp, vp, i64, vp])
    sig("csmpn_geometric_product_backward", C.c_int, [fp, C.c_int, vp, vp, vp, vp, vp, i64, vp])
    sig("csmpn_cemlp_workspace_bytes", sz, [C.c_int, bp, C.c_int])
    sig("csmpn_cemlp_saved_floats_per_row", sz, [C.c_int, bp, C.c_int])
    sig("csmpn_cemlp_saved_floats", sz, [C.c_int, bp, C.c_int, i64, u32])
    sig("csmpn_cemlp_forward", C.c_int, [fp, C.c_int, bp, C.c_int, vp, i64, vp, vp, vp, sz, u32, vp])
    sig("csmpn_cemlp_backward", C.c_int, [fp, C.c_int, bp, bg, C.c_int, vp, vp, i64, vp, vp, vp, sz, u32, vp])
    sig("csmpn_mvlinear_forward", C.c_int, [C.c_int, vp, vp, vp, i64, i32, i32, i32, vp, vp])
    sig("csmpn_mvlinear_backward", C.c_int, [C.c_int, vp, vp, vp, i64, i32, i32, i32, vp, vp, vp, vp])
    sig("csmpn_mvsilu_forward", C.c_int, [fp, C.c_int, vp, vp, vp, i64, i32, vp, vp])
    sig("csmpn_mvsilu_backward", C.c_int, [fp, C.c_int, vp, vp, vp, vp, i64, i32, vp, vp, vp, vp])
    sig("csmpn_mvnorm_forward", C.c_int, [fp, C.c_int, vp, vp, i64, i32, vp, vp])
    sig("csmpn_mvnorm_backward", C.c_int, [fp, C.c_int, vp, vp, vp, i64, i32, vp, vp, vp])
    sig("csmpn_mvlayernorm_forward", C.c_int, [fp, C.c_int, vp, vp, i64, i32, vp, vp])
    sig("csmpn_mvlayernorm_backward", C.c_int, [fp, C.c_int, vp, vp, vp, i64, i32, vp, vp, vp])
    sig("csmpn_wgp_forward", C.c_int, [fp, C.c_int, vp, vp, vp, i64, i32, vp, vp])
    sig("csmpn_wgp_backward", C.c_int, [fp, C.c_int, vp, vp, vp, vp, i64, i32, vp, vp, vp, vp])
    sig("csmpn_csr_workspace_bytes", sz, [i64, i64])
    sig("csmpn_csr_build", C.c_int, [vp, i64, i64, vp, vp, vp, vp, vp, vp, sz, u32, vp])
    sig("csmpn_csr_source_order", C.c_int, [vp, i64, i64, vp, vp, vp, sz, vp])
    sig("csmpn_segment_reduce", C.c_int, [vp, i64, i64, vp, vp, vp, vp, vp, i32, vp])
    sig("csmpn_egcl_edge_forward", C.c_int,
        [fp, C.c_int, bp, C.c_int, vp, i32, vp, i32, vp, vp, vp, i64, i64, vp, vp, vp, sz, u32, vp])
    sig("csmpn_egcl_edge_backward", C.c_int,
        [fp, C.c_int, bp, bg, C.c_int, vp, i32, vp, i32, vp, vp, vp, i64, i64, vp, vp, vp, vp, vp, sz, u32, vp])
    sig("csmpn_egcl_node_forward", C.c_int,
        [fp, C.c_int, bp, C.c_int, vp, i32, vp, i32, vp, i32, vp, i32, i32, i64, vp, vp, vp, sz, u32, vp])
    sig("csmpn_egcl_node_backward", C.c_int,
        [fp, C.c_int, bp, bg, C.c_int, vp, i32, vp, i32, vp, i32, vp, i32, i32, i64, vp, vp, vp, vp, vp, vp, sz, u32, vp])
    sig("csmpn_simplex_rows", C.c_int, [C.c_int, C.POINTER(VertexBlock), C.c_int, vp, i64, i32, i64, vp, vp])
    sig("csmpn_embed_cemlp_forward", C.c_int, [fp, C.c_int, bp, C.c_int, vp, i64, i32, vp, i32, i32, i64, vp, vp, vp, sz, u32, vp])
    sig("csmpn_embed_cemlp_backward", C.c_int, [fp, C.c_int, bp, bg, C.c_int, vp, i64, i32, vp, i32, i32, i64, vp, vp, vp, sz, u32, vp])
    sig("csmpn_type_attr_forward", C.c_int, [C.c_int, vp, i32, i32, vp, i64, vp, vp, i64, vp, vp, vp])
    sig("csmpn_type_attr_backward", C.c_int, [C.c_int, i32, i32, vp, i64, vp, vp, i64, vp, vp, vp, vp])
    sig("csmpn_readout_mse_forward", C.c_int, [C.c_int, vp, vp, i32, vp, i64, i32, vp, i64, vp, vp, vp, vp, vp])
    sig("csmpn_readout_mse_backward", C.c_int, [C.c_int, vp, i32, i64, i32, vp, i64, vp, vp, vp])
    sig("csmpn_readout_traj_forward", C.c_int, [C.c_int, vp, i32, vp, i64, vp, i32, i32, vp, vp, vp, vp, i64, vp, vp, vp, vp])
    sig("csmpn_readout_traj_backward", C.c_int,
        [C.c_int, vp, i32, i64, vp, vp, i64, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp])
    sig("csmpn_last_error", C.c_char_p, [])
    sig("csmpn_last_kernel", C.c_char_p, [])
    sig("csmpn_abi_version", C.c_int, [])
    sig("csmpn_build_target", C.c_char_p, [])
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


class CsmpnError(RuntimeError):
    pass


def check(rc: int):
    if rc != 0:
        raise CsmpnError(f"csmpn_hip error {rc}: {lib().csmpn_last_error().decode()}")


def metric_array(metric):
    arr = (C.c_float * len(metric))(*[float(m) for m in metric])
    return arr
