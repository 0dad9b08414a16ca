"""csmpn_hip: MI355X-native Clifford geometric-product / CEMLP / EGCL hot path.

`native` is the ctypes binding of the C-ABI library, `ops` the autograd
Functions over it, `sharded` the edge-sharded multi-GPU layer.
"""
from . import native  # noqa: F401


def library_available() -> bool:
    import os
    return os.path.exists(native.LIB_PATH)
