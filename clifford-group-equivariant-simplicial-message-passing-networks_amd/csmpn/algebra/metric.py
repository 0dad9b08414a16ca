"""Blade order and Cayley-table helpers, same public names as the reference's
csmpn/algebra/metric.py (ShortLexBasisBladeOrder :18-29, canonical_reordering_sign*
:50-79, gmt_element :82-89, construct_gmt :92-120).

The tables themselves come from the native library (csmpn_algebra_tables, a host
function of the C-ABI); nothing here loops over D^2 entries in Python.
"""
import ctypes as C

import numpy as np
import torch

from csmpn_hip import native


def native_tables(metric):
    """All algebra tables for a diagonal metric, as numpy arrays (host)."""
    metric = [float(m) for m in metric]
    n = len(metric)
    D, G = 1 << n, n + 1
    out = {
        "cayley": np.zeros((D, D, D), dtype=np.float32),
        "index_to_bitmap": np.zeros(D, dtype=np.int64),
        "bitmap_to_index": np.zeros(D, dtype=np.int64),
        "grades": np.zeros(D, dtype=np.int64),
        "subspaces": np.zeros(G, dtype=np.int64),
        "paths": np.zeros((G, G, G), dtype=np.uint8),
    }
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    native.check(native.lib().csmpn_algebra_tables(
        native.metric_array(metric), n, ptr(out["cayley"]), ptr(out["index_to_bitmap"]),
        ptr(out["bitmap_to_index"]), ptr(out["grades"]), ptr(out["subspaces"]), ptr(out["paths"])))
    return out


class ShortLexBasisBladeOrder:
    def __init__(self, n_vectors):
        t = native_tables([1.0] * n_vectors)
        self.index_to_bitmap = torch.from_numpy(t["index_to_bitmap"])
        self.grades = torch.from_numpy(t["grades"])
        self.bitmap_to_index = torch.from_numpy(t["bitmap_to_index"])


def set_bit_indices(x: int):
    n = 0
    while x > 0:
        if x & 1:
            yield n
        x >>= 1
        n += 1


def count_set_bits(bitmap: int) -> int:
    return bin(int(bitmap)).count("1")


def canonical_reordering_sign_euclidean(bitmap_a, bitmap_b):
    a, b = int(bitmap_a) >> 1, int(bitmap_b)
    swaps = 0
    while a:
        swaps += count_set_bits(a & b)
        a >>= 1
    return -1 if swaps & 1 else 1


def canonical_reordering_sign(bitmap_a, bitmap_b, metric):
    sign = canonical_reordering_sign_euclidean(bitmap_a, bitmap_b)
    for i in set_bit_indices(int(bitmap_a) & int(bitmap_b)):
        sign = sign * metric[i]
    return sign


def gmt_element(bitmap_a, bitmap_b, sig_array):
    return bitmap_a ^ bitmap_b, canonical_reordering_sign(bitmap_a, bitmap_b, sig_array)


def construct_gmt(index_to_bitmap, bitmap_to_index, signature):
    """Sparse [D, D, D] Cayley tensor, layout (left, out, right)."""
    t = native_tables([float(s) for s in signature])
    dense = torch.from_numpy(t["cayley"])
    n = dense.shape[0]
    left = torch.arange(n).repeat_interleave(n)
    right = torch.arange(n).repeat(n)
    out = torch.from_numpy(t["bitmap_to_index"])[
        torch.as_tensor(index_to_bitmap)[left] ^ torch.as_tensor(index_to_bitmap)[right]]
    vals = dense[left, out, right]
    return torch.sparse_coo_tensor(indices=torch.stack([left, out, right]), values=vals, size=(n, n, n))
