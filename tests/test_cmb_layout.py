"""CPU: the index arithmetic of the round-4 channel-MFMA backward (csrc/cemlp_cmb.hpp), restated and checked exhaustively -
the bank rules are the ones of MI355X_MICROARCH.md (ds_*_b32: banks = dword address mod 32 over the two 32-lane halves).

  * slot layout: element (blade d, row r, q, v) at 512 v + 64 d + 16 q + (r ^ (v << 2) ^ ((q >> 1) << 1)) - a bijection onto
    the 2 048 floats of a slot; the owner's / per-channel accesses and the transposed reads of the weight-gradient MFMAs
    (lane (i, k), step s -> row 4 s + k, column i) are bank-conflict free;
  * ONE table set: with lane (i, k)'s vector stored at unit (i & 3) 16 + (i >> 3) 8 + 2 k + ((i >> 2) & 1), the transposed
    operand T[(i', k')][v'] = W[4 v' + k'][orow(i')] is the float at 32 (k' >> 1) + 8 (i' >> 2) + 4 (k' & 1) + (i' & 3)
    + 64 v' - consecutive addresses across the wave;
  * the transposing butterfly: lane j of a 16-lane row ends with the sum over the row's lanes of value j (16 / 8 / 4 values),
    for either rotation direction of the two rotating steps."""
import numpy as np
import pytest

orow = lambda i: 4 * (i & 3) + (i >> 2)
unit = lambda i, k: (i & 3) * 16 + (i >> 3) * 8 + k * 2 + ((i >> 2) & 1)
slot = lambda d, r, q, v: 512 * v + 64 * d + 16 * q + (r ^ (v << 2) ^ ((q >> 1) << 1))


def test_forward_table_serves_the_transposed_mix():
    rng = np.random.default_rng(0)
    W = rng.normal(size=(16, 16))
    tab = np.zeros(256)
    assert sorted(unit(i, k) for i in range(16) for k in range(4)) == list(range(64))
    for i in range(16):
        for k in range(4):
            for v in range(4):
                tab[4 * unit(i, k) + v] = W[orow(i)][4 * v + k]     # forward entry: A[i][k] of step v
    addr = []
    for lane in range(64):
        i, k = lane & 15, lane >> 4
        tofs = 32 * (k >> 1) + 8 * (i >> 2) + 4 * (k & 1) + (i & 3)
        addr.append(tofs)
        for v in range(4):
            assert tab[tofs + 64 * v] == W[4 * v + k][orow(i)]
    assert sorted(addr) == list(range(64))                          # consecutive dwords: conflict-free ds_read_b32


def test_slot_layout_is_a_conflict_free_bijection():
    cells = {slot(d, r, q, v) for d in range(8) for r in range(16) for q in range(4) for v in range(4)}
    assert cells == set(range(2048))
    for half in (0, 1):
        lanes = range(32 * half, 32 * half + 32)
        for v in range(4):      # owner / per-channel access of channel slot v: lane (r, q)
            assert len({slot(0, l & 15, l >> 4, v) % 32 for l in lanes}) == 32
        for s in range(4):      # transposed read: lane (i, k) takes row 4 s + k, column i = (q = i >> 2, v = i & 3)
            assert len({slot(0, 4 * s + (l >> 4), (l & 15) >> 2, (l & 15) & 3) % 32 for l in lanes}) == 32
    # the two-register form of the kernel: wr(v) = (w0 ^ 4 v) + 512 v, mr(s) = m0 ^ 4 s
    for lane in range(64):
        r, q = lane & 15, lane >> 4
        w0 = 16 * q + (r ^ ((q >> 1) << 1))
        i, k = lane & 15, lane >> 4
        m0 = 512 * (i & 3) + 16 * (i >> 2) + (k ^ ((i & 3) << 2) ^ ((i >> 3) << 1))
        for v in range(4):
            assert (w0 ^ (v << 2)) + 512 * v == slot(0, r, q, v)
        for s in range(4):
            assert m0 ^ (s << 2) == slot(0, 4 * s + k, i >> 2, i & 3)


def _rows_sum(X, direction):
    nv = X.shape[1]
    lanes = np.arange(16)

    def dpp(a, ctrl):
        src = {"x1": lanes ^ 1, "x2": lanes ^ 2, "r4": (lanes - 4 * direction) % 16, "r8": (lanes - 8 * direction) % 16}[ctrl]
        return a[src]

    cur = [X[:, j].copy() for j in range(nv)]
    for b, ctrl in enumerate(["x1", "x2", "r4", "r8"]):
        bit = (lanes >> b) & 1
        if len(cur) > 1:
            cur = [np.where(bit, cur[2 * j + 1], cur[2 * j]) + dpp(np.where(bit, cur[2 * j], cur[2 * j + 1]), ctrl)
                   for j in range(len(cur) // 2)]
        else:
            cur = [cur[0] + dpp(cur[0], ctrl)]
    return cur[0]


@pytest.mark.parametrize("nv", [16, 8, 4])
@pytest.mark.parametrize("direction", [1, -1])
def test_transposing_butterfly(nv, direction):
    X = np.random.default_rng(nv).normal(size=(16, nv))
    got = _rows_sum(X, direction)
    want = np.array([X[:, l % nv].sum() for l in range(16)])
    assert np.allclose(got, want)
