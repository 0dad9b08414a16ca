// "pg" kernels: D = 32 layers (Cl(5,0) / Cl(4,1)) of 17 .. 32 channels on 16-ROW TILES with the dense channel mixing on
// v_mfma_f32_16x16x4_f32 at full M / K (round 5; the convex-hulls width is 28 channels, hulls_cssmpnn.py:16-28).
//
// Same arithmetic as every other family (csmpn/models/cegnn_utils.py:34-155,287-338). The wide
// parity-lane kernels (cemlp_plw.hpp) keep (row, channel, blade parity) in a lane and mix channels with DPP rotations at half
// the VALU rate on 4-row tiles, one wave per SIMD: 5-6 % of the HBM roofline. Here a workgroup of 8 waves (two per SIMD) owns a
// 16-row tile and every tensor of the tile changes hands between two lane layouts through LDS:
//
//   MIX layout (MFMA phases)      wave w owns the blades 4w .. 4w+3 of all rows and channels. out[r, o, d] = sum_c W[o][c][grade d]
//                                 x[r, c, d] is, per blade, a [32 x K] x [K x 16 rows] product: B operand = the tile itself
//                                 (lane (row n, k) reads channel 4s + k of ITS 4 blades with ONE ds_read_b128 per k-step), A
//                                 operand = weight fragments packed once per launch into the workspace (L2 / L1 resident, f4
//                                 per lane = 4 k-steps), result D[o][row] lands in lane (row, o / 4), register o % 4 and is
//                                 written back as 16-byte pieces. No padding: 16 rows are the 16 MFMA columns of one blade, 28 / 32
//                                 channels fill M and K; 64 MFMAs per wave and 32 x 32 matrix, no VALU work.
//   ROW layout (VALU phases)      lane = (row = lane & 15, channel = 4 wave + lane >> 4) with ALL 32 blades of its multivector in
//                                 registers (float t[32]): gates, normalisation, the geometric product (1 024 sign-table terms,
//                                 in-lane, no exchange) and the layer norm's per-channel part need no other lane; the mean over
//                                 the channels of a row goes through a 2 KB LDS array.
//
// LDS tensor buffer: element (channel c, row r, blade d) at c * 516 + r * 32 + 4 * ((d >> 2) ^ (r & 7)) + (d & 3) floats -
// 16-byte pieces stay whole (every access is a b128), the XOR spreads the rows over the banks for the MIX reads, the channel
// stride 516 = 4 mod 64 spreads the channels for the ROW reads; two buffers of 66 KB + path weights + parameters = 159 KB,
// one workgroup per CU.
// Forward of a tile: [inputs -> A (+ B: attributes / aggregate)] -> W1 on the MFMA -> A -> gates -> z -> A -> linear_right -> B,
// linear_left -> A (one pass over z) -> normalisation, geometric product, layer norm -> block output -> B -> block 1 likewise ->
// rows leave through A with coalesced stores / one atomic per 256 bytes and equal consecutive targets summed first.
// State for the backward (CSMPN_FLAG_SAVE_STATE): y, R, s of both blocks in ROW-layout lane order, whole 1 KB pieces.
#pragma once
#include "cemlp_device.hpp"

namespace csmpn {

constexpr int kPgRows = 16;                      // rows per tile
constexpr int kPgWaves = 8, kPgThreads = 64 * kPgWaves;
constexpr int kPgCS = kPgRows * 32 + 4;          // channel stride of a tensor buffer (floats)
constexpr int kPgSlots = 32;                     // channel slots per buffer
constexpr int kPgBuf = kPgSlots * kPgCS;         // floats per buffer
constexpr int kPgMaxGroups = 256;                // one workgroup per CU

// diagnostic build only (-DCSMPN_STAMPS, never shipped, never timed): shader-clock cycles per phase, summed per wave
struct PgStamp {
#ifdef CSMPN_STAMPS
    static constexpr int kSlots = 24;
    unsigned long long t0, acc[kSlots];
    CSMPN_DEV explicit PgStamp(int) {
#pragma unroll
        for (int i = 0; i < kSlots; ++i) acc[i] = 0;
        t0 = __builtin_amdgcn_s_memtime();
    }
    CSMPN_DEV void operator()(int id) {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        acc[id] += t1 - t0;
        t0 = t1;
        __builtin_amdgcn_sched_barrier(0);
    }
    CSMPN_DEV void flush(unsigned long long* out, int lane) {
        if (out && lane == 0) {
#pragma unroll
            for (int i = 0; i < kSlots; ++i) atomicAdd(out + i, acc[i]);
            atomicAdd(out + kSlots, 1ull);
        }
    }
#else
    CSMPN_DEV explicit PgStamp(int) {}
    CSMPN_DEV void operator()(int) {}
    CSMPN_DEV void flush(unsigned long long*, int) {}
#endif
};

CSMPN_DEV f4 pg_ld4(const float* p) { return *reinterpret_cast<const f4*>(p); }
CSMPN_DEV void pg_st4(float* p, f4 v) { *reinterpret_cast<f4*>(p) = v; }
// float offset of 16-byte piece j (blades 4j .. 4j+3) of (channel slot c, row r)
CSMPN_DEV int pg_off(int c, int r, int j) { return c * kPgCS + r * 32 + 4 * (j ^ (r & 7)); }
// grade of blade d in the grade-sorted order of 5 generators (1, 5, 10, 10, 5, 1)
CSMPN_DEV int pg_grade(int d) { return (d >= 1) + (d >= 6) + (d >= 16) + (d >= 26) + (d >= 31); }

// ---------------------------------------------------------------------------------
// compile-time description: input chunks of block 0, weight-fragment tables
template <class ALG, int C_, int MODE_, int NA_>
struct PgCfg {
    static_assert(ALG::n == 5, "32 blades");
    static_assert(MODE_ == MODE_EDGE || MODE_ == MODE_NODE, "edge or node program");
    static constexpr int C = C_, MODE = MODE_, NA = NA_, D = ALG::D, G = ALG::G, P = ALG::P, NBLK = 2;
    static_assert(C > 16 && C <= 32 && NA > 0 && NA <= 8, "17 .. 32 channels, one attribute chunk");
    static constexpr int ROW = C * D;
    static constexpr int NST = (C + 3) / 4;           // k-steps of a C-channel operand
    static constexpr int NSTA = (NA + 3) / 4;         // ... of the attribute chunk
    static constexpr int par_stride = 24;             // b1 bL la 0 | sa[6] | sb[6] | sigmoid(an)[6]
    // matrix-chunks ("mats") of a block, in table order. Block 0: the W1 column blocks of the input chunks, then WR, WL;
    // block 1: W1, WR, WL.  EDGE chunks: [h_dst - h_src (C) in A][edge_attr (NA) in B];  NODE: [h (C) in A][agg (C) in B][node_attr (NA) in E]
    static constexpr int NCH0 = MODE == MODE_EDGE ? 2 : 3;
    static constexpr int I0 = MODE == MODE_EDGE ? C + NA : 2 * C + NA;
    static constexpr int nmat(int K) { return K == 0 ? NCH0 + 2 : 3; }
    static constexpr int nst(int K, int m) { return (K == 0 && m == NCH0 - 1) ? NSTA : NST; }     // k-steps
    static constexpr int nch(int K, int m) { return (K == 0 && m == NCH0 - 1) ? NA : C; }        // valid input channels
    static constexpr int cbase(int K, int m) { return (K == 0 && m < NCH0) ? (m == NCH0 - 1 ? (NCH0 - 1) * C : m * C) : 0; }
    static constexpr int which(int K, int m) { return K == 0 ? (m < NCH0 ? 0 : m - NCH0 + 1) : m; }   // 0 W1, 1 WR, 2 WL
    static constexpr int ks4(int K, int m) { return (nst(K, m) + 3) / 4; }
    // f4 entries of one mat: [grade][o-tile][s4][lane]
    static constexpr int mat_f4(int K, int m) { return G * 2 * ks4(K, m) * 64; }
    static constexpr int toff(int K, int m) {   // f4 offset of mat m of block K (forward tables)
        int o = 0;
        for (int k = 0; k < K; ++k) for (int q = 0; q < nmat(k); ++q) o += mat_f4(k, q);
        for (int q = 0; q < m; ++q) o += mat_f4(K, q);
        return o;
    }
    static constexpr int fwd_f4 = toff(1, 2) + mat_f4(1, 2);
    // transposed tables (backward: d/d(operand) = W^T g): mat m of block K as [grade][c-tile][s4 over the OUT channels][lane]
    static constexpr int nct(int K, int m) { return (nch(K, m) + 15) / 16; }        // 16-channel tiles of the operand
    static constexpr int tmat_f4(int K, int m) { return G * nct(K, m) * ((NST + 3) / 4) * 64; }
    static constexpr int ttoff(int K, int m) {
        int o = fwd_f4;
        for (int k = 0; k < K; ++k) for (int q = 0; q < nmat(k); ++q) o += tmat_f4(k, q);
        for (int q = 0; q < m; ++q) o += tmat_f4(K, q);
        return o;
    }
    static constexpr int all_f4 = ttoff(1, 2) + tmat_f4(1, 2);
    static constexpr int tab_floats = 4 * all_f4;
    // backward: slice of weight-gradient tiles of block K: mat m at woff(K, m): [grade][o-tile][c-tile][lane][4]; then the
    // per-channel sums [32 channels][80]
    static constexpr int kSmall = 80;      // w[P = 56] | an[6] | 2 pad | (sa, sb)[6] | b1 | la | bL | pad
    static constexpr int wmat_floats(int K, int m) { return G * 2 * nct(K, m) * 256; }
    static constexpr int woff(int K, int m) { int o = 0; for (int q = 0; q < m; ++q) o += wmat_floats(K, q); return o; }
    static constexpr int slice_w(int K) { return woff(K, nmat(K)); }
    static constexpr int slice_floats(int K) { return slice_w(K) + 32 * kSmall; }
    static constexpr int slice_max = slice_floats(0) > slice_floats(1) ? slice_floats(0) : slice_floats(1);
    // backward LDS (floats): A | B | E (8 slots: attribute chunk) | two row-sum arrays | path weights and parameters of ONE block | indices
    static constexpr int b_E = 2 * kPgBuf, b_ln = b_E + 8 * kPgCS, b_w = b_ln + 2 * kPgRows * 32, b_par = b_w + 32 * P,
                         b_idx = b_par + 32 * par_stride, bwd_lds_floats = b_idx + 128;
    static_assert(bwd_lds_floats * 4 <= 160 * 1024, "LDS footprint of the backward");
    // LDS (floats)
    static constexpr int o_A = 0, o_B = kPgBuf, o_E = 2 * kPgBuf;                 // E: 4 slots (node attributes)
    static constexpr int o_ln = o_E + (MODE == MODE_NODE ? 4 * kPgCS : 0);        // [16 rows][32 channels]
    static constexpr int o_w = o_ln + kPgRows * 32;                               // path weights [2 blocks][32 channels][P]
    static constexpr int o_par = o_w + 2 * 32 * P;
    static constexpr int o_idx = o_par + 2 * 32 * par_stride;                    // 3 x 16 ints (targets, sources, attribute rows) + 16 floats
    static constexpr int lds_floats = o_idx + 128;
    static_assert(lds_floats * 4 <= 160 * 1024, "LDS footprint");
};

// weight fragments -> workspace, one thread per float. Entry (K, m, g, ot, s4, lane)[j]: A operand of k-step s = 4 s4 + j for
// the lane (i = lane & 15, k = lane >> 4): W[16 ot + i][cbase + 4 s + k][g] (zero outside the matrix).
template <class CF, class ALG>
__global__ void pg_pack_kernel(const DevCemlp Cd, float* tabs) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= CF::tab_floats) return;
    int e = (int)(t >> 2);
    const int j = (int)(t & 3);
    int K = 0, m = 0;
    bool found = false, tr = false;
    static_for<0, CF::NBLK>([&](auto kk) {
        static_for<0, CF::nmat(decltype(kk)::value)>([&](auto mm) {
            constexpr int K_ = decltype(kk)::value, m_ = decltype(mm)::value;
            constexpr int lo = CF::toff(K_, m_), n = CF::mat_f4(K_, m_);
            if (!found && e >= lo && e < lo + n) { K = K_; m = m_; e -= lo; found = true; }
            constexpr int tlo = CF::ttoff(K_, m_), tn = CF::tmat_f4(K_, m_);
            if (!found && e >= tlo && e < tlo + tn) { K = K_; m = m_; e -= tlo; found = true; tr = true; }
        });
    });
    const DevBlock& B = Cd.b[K];
    const int which = CF::which(K, m), nchv = CF::nch(K, m), cb = CF::cbase(K, m);
    const float* W = which == 0 ? B.W1 : (which == 1 ? B.WR : B.WL);
    const int I = which == 0 ? B.I : CF::C;
    const int lane = e & 63, i = lane & 15, k = lane >> 4;
    int rest = e >> 6;
    int o, cl, g;
    if (!tr) {
        const int ks4 = CF::ks4(K, m);
        const int s4 = rest % ks4; rest /= ks4;
        const int ot = rest & 1;
        g = rest >> 1;
        o = 16 * ot + i; cl = 4 * (4 * s4 + j) + k;
    } else {   // A[i = operand channel 16 ct + i][k = out channel 4 s + k]
        constexpr int ks4 = (CF::NST + 3) / 4;
        const int nct = CF::nct(K, m);
        const int s4 = rest % ks4; rest /= ks4;
        const int ct = rest % nct;
        g = rest / nct;
        cl = 16 * ct + i; o = 4 * (4 * s4 + j) + k;
    }
    tabs[t] = (o < CF::C && cl < nchv) ? W[((size_t)o * I + cb + cl) * CF::G + g] : 0.f;
}

// ---------------------------------------------------------------------------------
// MIX phase: acc[bl][ot] += W (mat at `tab`) x (operand tile in `buf`), this wave's 4 blades. NSTEP k-steps.
template <int NSTEP, int NT = 2>
CSMPN_DEV void pg_mix_acc(f4 (&acc)[4][NT], const float* buf, const f4* tab, int lane, int wave) {
    // the fragment loads start HERE: left to the scheduler they are hoisted over the barriers to the top of the tile (the
    // tables are read-only memory) and spilled until their phase - 400+ scratch accesses per tile
    int fence_ = 0;
    asm volatile("" : "+s"(fence_));
    tab += fence_;
    constexpr int KS4 = (NSTEP + 3) / 4;
    const int n = lane & 15, k = lane >> 4;
    const float* bp = buf + pg_off(k, n, wave);
    f4 b[NSTEP];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) b[s] = pg_ld4(bp + 4 * s * kPgCS);
    const int g_lo = pg_grade(4 * wave), g_hi = pg_grade(4 * wave + 3);   // at most two (consecutive) grades per 4 blades
    f4 alo[NT][KS4], ahi[NT][KS4];
#pragma unroll
    for (int ot = 0; ot < NT; ++ot)
#pragma unroll
        for (int s4 = 0; s4 < KS4; ++s4) {
            alo[ot][s4] = tab[((g_lo * NT + ot) * KS4 + s4) * 64 + lane];
            ahi[ot][s4] = tab[((g_hi * NT + ot) * KS4 + s4) * 64 + lane];
        }
    bool hi[4];
#pragma unroll
    for (int bl = 0; bl < 4; ++bl) hi[bl] = pg_grade(4 * wave + bl) != g_lo;
#pragma unroll
    for (int s = 0; s < NSTEP; ++s)
#pragma unroll
        for (int bl = 0; bl < 4; ++bl)
#pragma unroll
            for (int ot = 0; ot < NT; ++ot) {
                const float a = hi[bl] ? ahi[ot][s / 4][s % 4] : alo[ot][s / 4][s % 4];
                acc[bl][ot] = mfma16(a, b[s][bl], acc[bl][ot]);
            }
}
// two matrices on the same operand (linear_right and linear_left of z): one pass over the B fragments
template <int NSTEP>
CSMPN_DEV void pg_mix_acc2(f4 (&accR)[4][2], f4 (&accL)[4][2], const float* buf, const f4* tabR, const f4* tabL, int lane, int wave) {
    int fence_ = 0;
    asm volatile("" : "+s"(fence_));   // as pg_mix_acc
    tabR += fence_; tabL += fence_;
    constexpr int KS4 = (NSTEP + 3) / 4;
    const int n = lane & 15, k = lane >> 4;
    const float* bp = buf + pg_off(k, n, wave);
    f4 b[NSTEP];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) b[s] = pg_ld4(bp + 4 * s * kPgCS);
    const int g_lo = pg_grade(4 * wave), g_hi = pg_grade(4 * wave + 3);
    bool hi[4];
#pragma unroll
    for (int bl = 0; bl < 4; ++bl) hi[bl] = pg_grade(4 * wave + bl) != g_lo;
    static_for<0, 2>([&](auto mm) {
        const f4* tab = decltype(mm)::value == 0 ? tabR : tabL;
        f4 alo[2][KS4], ahi[2][KS4];
#pragma unroll
        for (int ot = 0; ot < 2; ++ot)
#pragma unroll
            for (int s4 = 0; s4 < KS4; ++s4) {
                alo[ot][s4] = tab[((g_lo * 2 + ot) * KS4 + s4) * 64 + lane];
                ahi[ot][s4] = tab[((g_hi * 2 + ot) * KS4 + s4) * 64 + lane];
            }
#pragma unroll
        for (int s = 0; s < NSTEP; ++s)
#pragma unroll
            for (int bl = 0; bl < 4; ++bl)
#pragma unroll
                for (int ot = 0; ot < 2; ++ot) {
                    const float a = hi[bl] ? ahi[ot][s / 4][s % 4] : alo[ot][s / 4][s % 4];
                    if constexpr (decltype(mm)::value == 0) accR[bl][ot] = mfma16(a, b[s][bl], accR[bl][ot]);
                    else accL[bl][ot] = mfma16(a, b[s][bl], accL[bl][ot]);
                }
    });
}
template <int NT>
CSMPN_DEV void pg_zero(f4 (&acc)[4][NT]) {
#pragma unroll
    for (int bl = 0; bl < 4; ++bl)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[bl][t] = f4{0.f, 0.f, 0.f, 0.f};
}
// result D[o][row]: lane (row n, q) holds the channels 16 ot + 4 q + v in register v -> piece `wave` of (channel, row n)
// (SLOTS: channel slots of the target; results for further channels - zero rows of the table - are not stored)
template <int NT, int SLOTS = 16 * NT>
CSMPN_DEV void pg_write_d(float* buf, const f4 (&acc)[4][NT], int lane, int wave) {
    const int n = lane & 15, q = lane >> 4;
#pragma unroll
    for (int ot = 0; ot < NT; ++ot)
#pragma unroll
        for (int v = 0; v < 4; ++v)
            if (SLOTS >= 16 * NT || 16 * ot + 4 * q + v < SLOTS)
                pg_st4(buf + pg_off(16 * ot + 4 * q + v, n, wave), f4{acc[0][ot][v], acc[1][ot][v], acc[2][ot][v], acc[3][ot][v]});
}

// empty asm statements that take a tensor's registers: every value is final at this point (without them the compiler sinks
// accumulations of the geometric product's backward across the next phases and keeps their operands in scratch)
CSMPN_DEV void pg_pin32(float (&t)[32]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
        asm volatile("" : "+v"(t[8 * j]), "+v"(t[8 * j + 1]), "+v"(t[8 * j + 2]), "+v"(t[8 * j + 3]), "+v"(t[8 * j + 4]), "+v"(t[8 * j + 5]),
                          "+v"(t[8 * j + 6]), "+v"(t[8 * j + 7]));
}
// ROW layout: the 32 blades of (row r, channel slot c)
CSMPN_DEV void pg_ld32(float (&t)[32], const float* buf, int r, int c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const f4 v = pg_ld4(buf + pg_off(c, r, j));
        t[4 * j] = v.x; t[4 * j + 1] = v.y; t[4 * j + 2] = v.z; t[4 * j + 3] = v.w;
    }
}
CSMPN_DEV void pg_st32(float* buf, int r, int c, const float (&t)[32]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) pg_st4(buf + pg_off(c, r, j), f4{t[4 * j], t[4 * j + 1], t[4 * j + 2], t[4 * j + 3]});
}
// state rows (CSMPN_FLAG_SAVE_STATE): piece j of lane l of wave w of tile t at (((t * 8 + w) * 8 + j) * 64 + l) * 4 floats of the
// tensor's region - one store / load instruction of a wave covers 1 KB
CSMPN_DEV size_t pg_state_off(long tile, int wave, int lane) { return ((size_t)(tile * kPgWaves + wave) * 8) * 256 + 4 * lane; }
CSMPN_DEV void pg_store_state(float* p, const float (&t)[32]) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
        __builtin_nontemporal_store(f4{t[4 * j], t[4 * j + 1], t[4 * j + 2], t[4 * j + 3]}, reinterpret_cast<f4*>(p + 256 * j));
}
CSMPN_DEV void pg_load_state(float (&t)[32], const float* p) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p + 256 * j));
        t[4 * j] = v.x; t[4 * j + 1] = v.y; t[4 * j + 2] = v.z; t[4 * j + 3] = v.w;
    }
}

// out[j] += sum_p w[p] sum_{(i,k) -> j in path p} sign(i,k) z[i] r[k]   (cegnn_utils.py:126-152), all 32 blades in the lane;
// wrow: this channel's P path weights (LDS, 16-byte aligned)
template <class ALG>
CSMPN_DEV void pg_weighted_gp(float (&out)[32], const float (&z)[32], const float (&r)[32], const float* wrow) {
    constexpr int P = ALG::P;
    static_assert(P % 4 == 0, "whole 16-byte pieces of path weights");
    static_for<0, P / 4>([&](auto qq) {
        const f4 wv = pg_ld4(wrow + 4 * decltype(qq)::value);
        static_for<0, 4>([&](auto pp) {
            constexpr int p = 4 * decltype(qq)::value + decltype(pp)::value;
            constexpr int gi = ALG::t.path_g[p][0], gj = ALG::t.path_g[p][1], gk = ALG::t.path_g[p][2];
            constexpr int i0 = ALG::gstart(gi), ni = ALG::gsize(gi);
            constexpr int j0 = ALG::gstart(gj), nj = ALG::gsize(gj);
            constexpr int k0 = ALG::gstart(gk), nk = ALG::gsize(gk);
            const float w = wv[decltype(pp)::value];
            float tmp[nj];
#pragma unroll
            for (int t = 0; t < nj; ++t) tmp[t] = 0.f;
            static_for<0, ni>([&](auto ii) {
                static_for<0, nk>([&](auto kk) {
                    constexpr int i = i0 + ii, k = k0 + kk;
                    constexpr int j = ALG::t.out[i][k];
                    if constexpr (j >= j0 && j < j0 + nj) {
                        constexpr float sg = float(ALG::t.sign[i][k]);
                        tmp[j - j0] = __builtin_fmaf(sg * z[i], r[k], tmp[j - j0]);
                    }
                });
            });
#pragma unroll
            for (int t = 0; t < nj; ++t) out[j0 + t] = __builtin_fmaf(w, tmp[t], out[j0 + t]);
        });
    });
}

// ---------------------------------------------------------------------------------
// forward kernel: two blocks of C channels, EGCL edge / node program
template <class ALG, class CF>
__global__ void __launch_bounds__(kPgThreads, 2) cemlp_pg_fwd_kernel(const DevCemlp C_arg, const RowIO io_arg) {
    typedef const char __attribute__((address_space(4))) * KArgPtr;
    const KArgPtr ka = (KArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr size_t kIoOffset = (sizeof(DevCemlp) + alignof(RowIO) - 1) / alignof(RowIO) * alignof(RowIO);
    const DevCemlp& Cd = *(const DevCemlp*)(const char*)ka;
    const RowIO& io = *(const RowIO*)(const char*)(ka + kIoOffset);
    (void)C_arg; (void)io_arg;
    constexpr int C = CF::C, MODE = CF::MODE, NA = CF::NA, D = 32, G = 6, P = CF::P, ROW = CF::ROW, NST = CF::NST;
    constexpr int ROWP = 32 * D;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const bufA = smem + CF::o_A;
    float* const bufB = smem + CF::o_B;
    float* const bufE = smem + CF::o_E;
    float* const lnx = smem + CF::o_ln;
    int* sidx = reinterpret_cast<int*>(smem + CF::o_idx);            // this tile's [0..15] target / row, [16..31] source, [32..47] attribute row,
    int* sidx_n = sidx + 64;                                          // [48..63] node program: 1 / max(deg, 1) (float); the next tile's in the other half
    constexpr int PPR = C * 8, PPA = 4 * CF::NSTA * 8;                // 16-byte pieces per row: a C-channel segment, the attribute chunk padded to whole k-steps
    constexpr int NPRE = PPR * kPgRows / kPgThreads, NPA = (PPA * kPgRows + kPgThreads - 1) / kPgThreads;
    static_assert(NPRE * kPgThreads == PPR * kPgRows, "whole pieces per thread");
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 15, c = 4 * wave + (lane >> 4);             // ROW layout
    const bool cvalid = c < C;
    const int cc = cvalid ? c : C - 1;
    const f4* tabs = reinterpret_cast<const f4*>(io.plw_tabs);   // (re-read per tile: re-read per tile, see the asm statement in the tile loop)
    PgStamp stamp(0);

    // per-channel parameters and path weights of both blocks -> LDS (zero beyond C)
    static_for<0, 2>([&](auto kk) {
        constexpr int K = decltype(kk)::value;
        const DevBlock& B = Cd.b[K];
        for (int e = tid; e < 32 * CF::par_stride; e += kPgThreads) {
            const int ch = e / CF::par_stride, s = e % CF::par_stride;
            float v = 0.f;
            if (ch < C) {
                if (s == 0) v = B.has_b1 ? B.b1[ch] : 0.f;
                else if (s == 1) v = B.bL[ch];
                else if (s == 2) v = B.la[ch];
                else if (s >= 4 && s < 10) v = B.sa[ch * G + (s - 4)];
                else if (s >= 10 && s < 16) v = B.sb[ch * G + (s - 10)];
                else if (s >= 16 && s < 22) v = sigmoidf(B.an[ch * G + (s - 16)]);
            }
            smem[CF::o_par + K * 32 * CF::par_stride + e] = v;
        }
        for (int e = tid; e < 32 * P; e += kPgThreads) smem[CF::o_w + K * 32 * P + e] = e < C * P ? B.w[e] : 0.f;
    });
    // channel slots that no tile writes but a k-step reads: zero once
    if constexpr (C % 4 != 0) {
        for (int e = tid; e < (4 * NST - C) * kPgCS; e += kPgThreads) { bufA[C * kPgCS + e] = 0.f; bufB[C * kPgCS + e] = 0.f; }
    }
    __syncthreads();

    const bool save_state = io.save_state != 0 && io.save != nullptr;
    const long ntiles = (io.rows + kPgRows - 1) / kPgRows;
    // Software pipeline over the workgroup's tiles: the indices of tile t + 1 are fetched while tile t computes; its input rows
    // are requested in front of tile t's row stores / atomics (few registers are alive there; vmcnt counts in issue order, a load
    // behind an atomic would wait for its acknowledgement) and written to LDS at the top of tile t + 1.
    auto load_idx = [&](int* dst, long tile_, int t) {     // threads 0 .. 15
        const long row = tile_ * kPgRows + t;
        const bool valid = tile_ < ntiles && row < io.rows;
        if constexpr (MODE == MODE_EDGE) {
            dst[t] = valid ? io.seg[0].ia[row] : -1;
            dst[16 + t] = valid ? io.seg[0].ib[row] : 0;
            dst[32 + t] = valid ? io.seg[1].ia[row] : 0;
        } else {
            dst[t] = valid ? t : -1;
            float sc = 1.0f;
            if (valid && io.seg[1].deg) { const int dg = io.seg[1].deg[row]; sc = 1.0f / float(dg > 1 ? dg : 1); }
            reinterpret_cast<float*>(dst)[48 + t] = sc;
        }
    };
    f4 pre_a[NPRE], pre_b[NPRE], pre_x[NPA];
    auto issue_rows = [&](const int* idx, long tile_, int t) {
        const float* sc_ = reinterpret_cast<const float*>(idx) + 48;
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int p = t + i * kPgThreads, rr = p / PPR, e = p % PPR;
            pre_a[i] = pre_b[i] = f4{0.f, 0.f, 0.f, 0.f};
            if (idx[rr] >= 0) {
                if constexpr (MODE == MODE_EDGE) {
                    pre_a[i] = pg_ld4(io.seg[0].a + (size_t)idx[rr] * ROW + 4 * e);
                    pre_b[i] = pg_ld4(io.seg[0].b + (size_t)idx[16 + rr] * ROW + 4 * e);
                } else {
                    pre_a[i] = pg_ld4(io.seg[0].a + (size_t)(tile_ * kPgRows + rr) * ROW + 4 * e);
                    pre_b[i] = pg_ld4(io.seg[1].a + (size_t)(tile_ * kPgRows + rr) * ROW + 4 * e) * sc_[rr];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            const int p = t + i * kPgThreads, rr = (p / PPA) & 15, e = p % PPA;
            pre_x[i] = f4{0.f, 0.f, 0.f, 0.f};
            if (p < kPgRows * PPA && idx[rr] >= 0 && e < NA * 8) {
                if constexpr (MODE == MODE_EDGE) pre_x[i] = pg_ld4(io.seg[1].a + (size_t)idx[32 + rr] * (NA * D) + 4 * e);
                else pre_x[i] = pg_ld4(io.seg[2].a + (size_t)(tile_ * kPgRows + rr) * (NA * D) + 4 * e);
            }
        }
    };
    if (tid < kPgRows) load_idx(sidx, blockIdx.x, tid);
    __syncthreads();
    issue_rows(sidx, blockIdx.x, tid);
    stamp(0);
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long row0 = tile * kPgRows;
        // ---- block-0 input chunks (requested during the previous tile) -> LDS: 16-byte pieces, consecutive threads =
        // consecutive pieces of a row (coalesced)
        {
#pragma unroll
            for (int i = 0; i < NPRE; ++i) {
                const int p = tid + i * kPgThreads, rr = p / PPR, e = p % PPR;
                if constexpr (MODE == MODE_EDGE) {
                    pg_st4(bufA + pg_off(e >> 3, rr, e & 7), pre_a[i] - pre_b[i]);
                } else {
                    pg_st4(bufA + pg_off(e >> 3, rr, e & 7), pre_a[i]);
                    pg_st4(bufB + pg_off(e >> 3, rr, e & 7), pre_b[i]);
                }
            }
            float* const bufX = MODE == MODE_EDGE ? bufB : bufE;
#pragma unroll
            for (int i = 0; i < NPA; ++i) {
                const int p = tid + i * kPgThreads, rr = (p / PPA) & 15, e = p % PPA;
                if (p < kPgRows * PPA) pg_st4(bufX + pg_off(e >> 3, rr, e & 7), pre_x[i]);
            }
            if (tid < kPgRows) load_idx(sidx_n, tile + gridDim.x, tid);
        }
        __syncthreads();
        stamp(1);

        float out[32];
        static_for<0, 2>([&](auto kk) {
            constexpr int K = decltype(kk)::value;
            const float* par = smem + CF::o_par + K * 32 * CF::par_stride + c * CF::par_stride;
            const float* wrow = smem + CF::o_w + K * 32 * P + c * P;
            // ---- MIX: y = W1 x -> A
            {
                f4 acc[4][2];
                pg_zero(acc);
                if constexpr (K == 0) {
                    pg_mix_acc<NST>(acc, bufA, tabs + CF::toff(0, 0), lane, wave);
                    if constexpr (MODE == MODE_EDGE) {
                        pg_mix_acc<CF::NSTA>(acc, bufB, tabs + CF::toff(0, 1), lane, wave);
                    } else {
                        pg_mix_acc<NST>(acc, bufB, tabs + CF::toff(0, 1), lane, wave);
                        pg_mix_acc<CF::NSTA>(acc, bufE, tabs + CF::toff(0, 2), lane, wave);
                    }
                } else {
                    pg_mix_acc<NST>(acc, bufB, tabs + CF::toff(1, 0), lane, wave);
                    // the block-1 input rows leave for the backward (coalesced, while the MFMAs run)
                    if (io.save) {
                        for (int p = tid; p < kPgRows * PPR; p += kPgThreads) {
                            const int rr = p / PPR, e = p % PPR;
                            if (row0 + rr < io.rows) pg_st4(io.save + (size_t)(row0 + rr) * ROW + 4 * e, pg_ld4(bufB + pg_off(e >> 3, rr, e & 7)));
                        }
                    }
                }
                pg_write_d(bufA, acc, lane, wave);   // in place where A is an operand: this wave has read its 4 blades of A
            }
            stamp(2 + 6 * K);
            __syncthreads();
            stamp(3 + 6 * K);
            // ---- ROW: bias, gates, z -> A
            float z[32];
            {
                float y[32];
                pg_ld32(y, bufA, r, c);
                y[0] += par[0];
                if (save_state && cvalid && row0 + r < io.rows)
                    pg_store_state(io.save + state_region<ROW, ROWP>(io.rows, 1, K) + pg_state_off(tile, wave, lane), y);
                static_for<0, G>([&](auto g) {
                    constexpr int d0 = ALG::gstart(g), nd = ALG::gsize(g);
                    float u;
                    if constexpr (g == 0) {
                        u = y[0];
                    } else {
                        u = 0.f;
                        static_for<0, nd>([&](auto t) {
                            constexpr int d = d0 + decltype(t)::value;
                            u = __builtin_fmaf(qsf<ALG, d> * y[d], y[d], u);
                        });
                    }
                    const float gate = sigmoidf(__builtin_fmaf(par[4 + g], u, par[10 + g]));
#pragma unroll
                    for (int t = 0; t < nd; ++t) z[d0 + t] = gate * y[d0 + t];
                });
                pg_st32(bufA, r, c, z);
            }
            stamp(4 + 6 * K);
            __syncthreads();
            // ---- MIX: R = WR z -> B, L = WL z -> A (in place)
            {
                f4 accR[4][2], accL[4][2];
                pg_zero(accR);
                pg_zero(accL);
                constexpr int mR = CF::nmat(K) - 2;
                pg_mix_acc2<NST>(accR, accL, bufA, tabs + CF::toff(K, mR), tabs + CF::toff(K, mR + 1), lane, wave);
                pg_write_d(bufB, accR, lane, wave);
                pg_write_d(bufA, accL, lane, wave);
            }
            stamp(5 + 6 * K);
            __syncthreads();
            stamp(3 + 6 * K);
            // ---- ROW: normalisation, geometric product, layer norm
            float s[32];
            {
                float R[32];
                pg_ld32(R, bufB, r, c);
                pg_ld32(s, bufA, r, c);     // s accumulates: linear_left output + product
                s[0] += par[1];
                if (save_state && cvalid && row0 + r < io.rows)
                    pg_store_state(io.save + state_region<ROW, ROWP>(io.rows, 2, K) + pg_state_off(tile, wave, lane), R);
                static_for<0, G>([&](auto g) {
                    constexpr int d0 = ALG::gstart(g), nd = ALG::gsize(g);
                    float qq = 0.f;
                    static_for<0, nd>([&](auto t) {
                        constexpr int d = d0 + decltype(t)::value;
                        qq = __builtin_fmaf(qsf<ALG, d> * R[d], R[d], qq);
                    });
                    const float m = __builtin_fmaf(par[16 + g], sqrt_pos(sqrt_pos(__builtin_fmaf(qq, qq, kSmooth))) - 1.0f, 1.0f);
                    const float inv = fast_rcp(m + kEps);
#pragma unroll
                    for (int t = 0; t < nd; ++t) R[d0 + t] *= inv;
                });
                pg_weighted_gp<ALG>(s, z, R, wrow);
            }
            float qs = 0.f;
            static_for<0, 32>([&](auto dd) {
                constexpr int d = decltype(dd)::value;
                s[d] = cvalid ? s[d] * kInvSqrt2 : 0.f;
                qs = __builtin_fmaf(qsf<ALG, d> * s[d], s[d], qs);
            });
            if (save_state && cvalid && row0 + r < io.rows)
                pg_store_state(io.save + state_region<ROW, ROWP>(io.rows, 0, K) + pg_state_off(tile, wave, lane), s);
            const float nl = sqrt_pos(sqrt_pos(__builtin_fmaf(qs, qs, kSmooth)));
            lnx[r * 32 + c] = cvalid ? nl : 0.f;
            stamp(6 + 6 * K);
            __syncthreads();
            float tot = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f4 v = pg_ld4(lnx + r * 32 + 4 * j);
                tot += (v.x + v.y) + (v.z + v.w);
            }
            const float kf = par[2] * fast_rcp(__builtin_fmaf(tot, 1.0f / float(C), kEps));
#pragma unroll
            for (int d = 0; d < 32; ++d) out[d] = kf * s[d];
            // block 0: the block-1 input -> B (R has been read by its own lane only: (r, c) -> (r, c)); block 1: rows -> A
            pg_st32(K == 0 ? bufB : bufA, r, c, out);
            __syncthreads();
            stamp(7 + 6 * K);
        });

        // ---- rows leave through A; the next tile's input rows are requested first
        issue_rows(sidx_n, tile + gridDim.x, tid);
        if constexpr (MODE == MODE_EDGE) {
            if (io.row_store) {   // deterministic mode: message rows to the [E, C, D] table in sorted edge order
                for (int p = tid; p < kPgRows * PPR; p += kPgThreads) {
                    const int rr = p / PPR, e = p % PPR;
                    if (row0 + rr < io.rows) pg_st4(io.agg + (size_t)(row0 + rr) * ROW + 4 * e, pg_ld4(bufA + pg_off(e >> 3, rr, e & 7)));
                }
            } else {
                // one atomic per 256 bytes of a target row; equal consecutive targets (the rows are sorted by target) are summed first
                for (int col = tid; col < ROW; col += kPgThreads) {
                    const int ch = col >> 5, d = col & 31;
                    float acc = 0.f;
                    int cur = sidx[0];
#pragma unroll
                    for (int rr = 0; rr < kPgRows; ++rr) {
                        const int t_ = sidx[rr];
                        if (t_ != cur) {
                            if (cur >= 0) atomicAdd(io.agg + (size_t)cur * ROW + col, acc);
                            cur = t_;
                            acc = 0.f;
                        }
                        acc += bufA[pg_off(ch, rr, d >> 2) + (d & 3)];
                    }
                    if (cur >= 0) atomicAdd(io.agg + (size_t)cur * ROW + col, acc);
                }
            }
        } else {
            for (int p = tid; p < kPgRows * PPR; p += kPgThreads) {
                const int rr = p / PPR, e = p % PPR;
                if (row0 + rr < io.rows) {
                    f4 v = pg_ld4(bufA + pg_off(e >> 3, rr, e & 7));
                    if (io.resid) v += pg_ld4(io.resid + (size_t)(row0 + rr) * ROW + 4 * e);
                    pg_st4(io.y + (size_t)(row0 + rr) * ROW + 4 * e, v);
                }
            }
        }
        __syncthreads();   // A / B / the index arrays are free for the next tile
        { int* t_ = sidx; sidx = sidx_n; sidx_n = t_; }
        stamp(14);
    }
    stamp.flush(io.stamps, lane);
}


// =================================================================================
// backward
//
// One launch per block (K = 1, then K = 0; d/d(block-1 input) travels as rows through io.plw_g1), always on the state the
// forward saved (CSMPN_FLAG_SAVE_STATE: y, R, s in ROW-layout lane order): no channel mix and no geometric product is
// recomputed. Per 16-row tile:
//   d/d(out) -> A | ROW: z = gate(y) y -> A, layer-norm backward -> ggp -> B | MIX: gz = WL^T ggp (own 4 blades), d/dWL += ggp^T z
//   -> gz over ggp in B | ROW: geometric product + normalisation backward (in-lane; two passes) -> gR -> B | MIX: WR^T gR,
//   d/dWR += gR^T z | ROW: MVSiLU backward -> gy -> B, block input -> A (+ E) | MIX: d/dW1 += gy^T x, gx = W1^T gy -> rows out.
// Weight gradients: contraction over the 16 rows - A operand = the gradient tile, B operand = the operand tile, both read from
// the LDS buffers with the rows as the k index (ds_read_b128: 4 blades per read). A wave owns ONE (o-tile, c-tile) tile of the
// matrix and one half of the blades (three grades: 12 accumulator registers per matrix, persistent over the launch), so
// every element of the slice has one owner and no sums cross waves. Per-channel parameter gradients: summed over the 16
// rows of a tile by the transposing butterfly of cemlp_cmb.hpp (a DPP row = the 16 rows of one channel), 5 registers per lane.

// 16 values in, lane j of the 16-lane DPP row keeps the sum over the row's lanes of value j (cb_rows_sum of cemlp_cmb.hpp)
CSMPN_DEV float pg_rows_sum(float (&x)[16], int l16) {
    const bool b0 = l16 & 1, b1 = l16 & 2, b2 = l16 & 4, b3 = l16 & 8;
    float y[8], z[4], u[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float keep = b0 ? x[2 * j + 1] : x[2 * j], send = b0 ? x[2 * j] : x[2 * j + 1];
        y[j] = keep + dpp_mov<0xB1>(send);   // quad_perm [1,0,3,2]
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float keep = b1 ? y[2 * j + 1] : y[2 * j], send = b1 ? y[2 * j] : y[2 * j + 1];
        z[j] = keep + dpp_mov<0x4E>(send);   // quad_perm [2,3,0,1]
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float keep = b2 ? z[2 * j + 1] : z[2 * j], send = b2 ? z[2 * j] : z[2 * j + 1];
        u[j] = keep + dpp_mov<0x124>(send);  // row_ror 4
    }
    const float keep = b3 ? u[1] : u[0], send = b3 ? u[0] : u[1];
    return keep + dpp_mov<0x128>(send);      // row_ror 8
}
// collects the per-channel gradients of a tile in slot order and runs a butterfly whenever 16 are there
struct PgCollect {
    float buf[16];
    template <int IDX>
    CSMPN_DEV void add(float v, float (&small)[5], int l16) {
        buf[IDX % 16] = v;
        if constexpr (IDX % 16 == 15) small[IDX / 16] += pg_rows_sum(buf, l16);
    }
};

// geometric product backward, all 32 blades in the lane, two passes (each keeps four tensors live):
//   Z: gz[i] += w_p U[i], gw_p = sum_i z[i] U[i],  U[i] = sum sign ggp[j] r[k];   R: gr[k] = sum_p w_p sum sign ggp[j] z[i]
template <class ALG>
CSMPN_DEV void pg_gp_bwd_z(const float (&ggp)[32], const float (&z)[32], const float (&rf)[32], float (&gz)[32], const float* wrow,
                           PgCollect& col, float (&small)[5], int l16) {
    constexpr int P = ALG::P;
    static_for<0, P / 4>([&](auto qq) {
        const f4 wv = pg_ld4(wrow + 4 * decltype(qq)::value);
        static_for<0, 4>([&](auto pp) {
            constexpr int p = 4 * decltype(qq)::value + decltype(pp)::value;
            constexpr int gi = ALG::t.path_g[p][0], gj = ALG::t.path_g[p][1], gk = ALG::t.path_g[p][2];
            constexpr int i0 = ALG::gstart(gi), ni = ALG::gsize(gi);
            constexpr int j0 = ALG::gstart(gj), nj = ALG::gsize(gj);
            constexpr int k0 = ALG::gstart(gk), nk = ALG::gsize(gk);
            const float w = wv[decltype(pp)::value];
            float U[ni];
#pragma unroll
            for (int t = 0; t < ni; ++t) U[t] = 0.f;
            static_for<0, ni>([&](auto ii) {
                static_for<0, nk>([&](auto kk) {
                    constexpr int i = i0 + ii, k = k0 + kk;
                    constexpr int j = ALG::t.out[i][k];
                    if constexpr (j >= j0 && j < j0 + nj) {
                        constexpr float sg = float(ALG::t.sign[i][k]);
                        U[ii] = __builtin_fmaf(sg * ggp[j], rf[k], U[ii]);
                    }
                });
            });
            float gwv = 0.f;
#pragma unroll
            for (int t = 0; t < ni; ++t) { gz[i0 + t] = __builtin_fmaf(w, U[t], gz[i0 + t]); gwv = __builtin_fmaf(z[i0 + t], U[t], gwv); }
            col.template add<p>(gwv, small, l16);
        });
    });
}
template <class ALG>
CSMPN_DEV void pg_gp_bwd_r(const float (&ggp)[32], const float (&z)[32], float (&gr)[32], const float* wrow) {
    constexpr int P = ALG::P;
    static_for<0, P / 4>([&](auto qq) {
        const f4 wv = pg_ld4(wrow + 4 * decltype(qq)::value);
        static_for<0, 4>([&](auto pp) {
            constexpr int p = 4 * decltype(qq)::value + decltype(pp)::value;
            constexpr int gi = ALG::t.path_g[p][0], gj = ALG::t.path_g[p][1], gk = ALG::t.path_g[p][2];
            constexpr int i0 = ALG::gstart(gi), ni = ALG::gsize(gi);
            constexpr int j0 = ALG::gstart(gj), nj = ALG::gsize(gj);
            constexpr int k0 = ALG::gstart(gk), nk = ALG::gsize(gk);
            const float w = wv[decltype(pp)::value];
            float V[nk];
#pragma unroll
            for (int t = 0; t < nk; ++t) V[t] = 0.f;
            static_for<0, ni>([&](auto ii) {
                static_for<0, nk>([&](auto kk) {
                    constexpr int i = i0 + ii, k = k0 + kk;
                    constexpr int j = ALG::t.out[i][k];
                    if constexpr (j >= j0 && j < j0 + nj) {
                        constexpr float sg = float(ALG::t.sign[i][k]);
                        V[kk] = __builtin_fmaf(sg * ggp[j], z[i], V[kk]);
                    }
                });
            });
#pragma unroll
            for (int t = 0; t < nk; ++t) gr[k0 + t] = __builtin_fmaf(w, V[t], gr[k0 + t]);
        });
    });
}

// d/dW tile (ot, ct) += G^T X over the 16 rows, the 16 blades of one half: acc[gs] = the half's three grades.
// G: gradient tile (its channel slots are the rows of the matrix), X: operand tile, both in LDS.
template <int HALF>
CSMPN_DEV void pg_wgrad_half(f4 (&acc)[3], const float* bufG, const float* bufX, int ot, int ct, int lane, int xmask = 15) {
    const int i = lane & 15, k = lane >> 4;
    static_for<0, 4>([&](auto jj) {
        constexpr int j = 4 * HALF + decltype(jj)::value;      // 16-byte piece = blades 4 j .. 4 j + 3
        f4 a[4], b[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            a[s] = pg_ld4(bufG + pg_off(16 * ot + i, 4 * s + k, j));
            b[s] = pg_ld4(bufX + pg_off(16 * ct + (i & xmask), 4 * s + k, j));
        }
        static_for<0, 4>([&](auto bb) {
            constexpr int d = 4 * j + decltype(bb)::value;
            constexpr int g = (d >= 1) + (d >= 6) + (d >= 16) + (d >= 26) + (d >= 31);
            constexpr int gs = g - 3 * HALF;
            static_assert(gs >= 0 && gs < 3, "three grades per blade half");
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[gs] = mfma16(a[s][decltype(bb)::value], b[s][decltype(bb)::value], acc[gs]);
        });
    });
}
CSMPN_DEV void pg_wgrad(f4 (&acc)[3], const float* bufG, const float* bufX, int unit, int lane) {
    const int half = unit & 1, tile = unit >> 1;
    if (half) pg_wgrad_half<1>(acc, bufG, bufX, tile >> 1, tile & 1, lane);
    else pg_wgrad_half<0>(acc, bufG, bufX, tile >> 1, tile & 1, lane);
}
// ... a one-c-tile operand (the attribute chunk): units 0..3 = (ot, half)
CSMPN_DEV void pg_wgrad1(f4 (&acc)[3], const float* bufG, const float* bufX, int unit, int lane) {
    const int half = unit & 1, ot = unit >> 1;
    // the operand has 8 channel slots (E): columns 8 .. 15 of the tile repeat 0 .. 7 and are dropped by the reduction
    if (half) pg_wgrad_half<1>(acc, bufG, bufX, ot, 0, lane, 7);
    else pg_wgrad_half<0>(acc, bufG, bufX, ot, 0, lane, 7);
}
// slice store: unit (tile, half) of mat at `base` ([grade][ot][ct][lane][4], NCT c-tiles)
template <int NCT>
CSMPN_DEV void pg_store_unit(float* base, const f4 (&acc)[3], int ot, int ct, int half, int lane) {
#pragma unroll
    for (int gs = 0; gs < 3; ++gs) pg_st4(base + ((((3 * half + gs) * 2 + ot) * NCT + ct) * 64 + lane) * 4, acc[gs]);
}

// lane-derived indices of ONE phase. The thread id passes through an empty asm statement: derived from the same value in
// every phase, the LDS addresses of all phases (hundreds: the XOR swizzle makes every (row, piece) pair its own value) are
// loop invariants that the compiler computes once and keeps alive across the tile loop - 370 spills per tile.
CSMPN_DEV int pg_tid() { int t = threadIdx.x; asm volatile("" : "+v"(t)); return t; }
#define PG_PHASE_IDS()                                                                                              \
    const int tid = pg_tid();                                                                                       \
    const int wave = tid >> 6, lane = tid & 63, r = lane & 15, c = 4 * wave + (lane >> 4), l16 = lane & 15;         \
    const bool cvalid = c < C, live = cvalid && row0 + r < io.rows;                                                 \
    const float* par = smem + CF::b_par + c * CF::par_stride;                                                       \
    const float* wrow = smem + CF::b_w + c * P;                                                                     \
    const size_t soff = pg_state_off(tile, wave, lane);                                                             \
    (void)l16; (void)live; (void)par; (void)wrow; (void)soff; (void)cvalid; (void)r

template <class ALG, class CF, int K>
__global__ void __launch_bounds__(kPgThreads, 2) cemlp_pg_bwd_kernel(const DevCemlp C_arg, const RowIO io_arg) {
    typedef const char __attribute__((address_space(4))) * KArgPtr;
    const KArgPtr ka = (KArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr size_t kIoOffset = (sizeof(DevCemlp) + alignof(RowIO) - 1) / alignof(RowIO) * alignof(RowIO);
    const DevCemlp& Cd = *(const DevCemlp*)(const char*)ka;
    const RowIO& io = *(const RowIO*)(const char*)(ka + kIoOffset);
    (void)C_arg; (void)io_arg;
    constexpr int C = CF::C, MODE = CF::MODE, NA = CF::NA, D = 32, G = 6, P = CF::P, ROW = CF::ROW, NST = CF::NST;
    constexpr int ROWP = 32 * D, PPR = C * 8;
    constexpr int NM = CF::nmat(K), mR = NM - 2, mL = NM - 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const bufA = smem;
    float* const bufB = smem + kPgBuf;
    float* const bufE = smem + CF::b_E;
    float* const ln1 = smem + CF::b_ln;
    float* const ln2 = ln1 + kPgRows * 32;
    int* sidx = reinterpret_cast<int*>(smem + CF::b_idx);          // this tile's [0..15] targets, [16..31] sources, [32..47] attribute rows,
    int* sidx_n = sidx + 64;                                        // [48..63] 1 / max(deg, 1) as float; the next tile's in the other half
    constexpr int NPRE = PPR * kPgRows / kPgThreads;               // 16-byte pieces of a C-channel tile per thread
    static_assert(NPRE * kPgThreads == PPR * kPgRows, "whole pieces per thread");
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 15, c = 4 * wave + (lane >> 4), l16 = lane & 15;
    const bool cvalid = c < C;
    const f4* tabs = reinterpret_cast<const f4*>(io.plw_tabs);   // (re-read per tile: re-read per tile, see the asm statement in the tile loop)
    PgStamp stamp(0);
    {
        const DevBlock& B = Cd.b[K];
        for (int e = tid; e < 32 * CF::par_stride; e += kPgThreads) {
            const int ch = e / CF::par_stride, s_ = e % CF::par_stride;
            float v = 0.f;
            if (ch < C) {
                if (s_ == 0) v = B.has_b1 ? B.b1[ch] : 0.f;
                else if (s_ == 1) v = B.bL[ch];
                else if (s_ == 2) v = B.la[ch];
                else if (s_ >= 4 && s_ < 10) v = B.sa[ch * G + (s_ - 4)];
                else if (s_ >= 10 && s_ < 16) v = B.sb[ch * G + (s_ - 10)];
                else if (s_ >= 16 && s_ < 22) v = sigmoidf(B.an[ch * G + (s_ - 16)]);
            }
            smem[CF::b_par + e] = v;
        }
        for (int e = tid; e < 32 * P; e += kPgThreads) smem[CF::b_w + e] = e < C * P ? B.w[e] : 0.f;
    }
    // persistent sums
    f4 accL[3], accR[3], accW0[3], accW1[3], accW2[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) accL[g] = accR[g] = accW0[g] = accW1[g] = accW2[g] = f4{0.f, 0.f, 0.f, 0.f};
    float small[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const long ntiles = (io.rows + kPgRows - 1) / kPgRows;
    // Software pipeline over the workgroup's tiles: the indices of tile t + 1 are fetched while tile t computes, its d/d(out)
    // rows are requested in front of tile t's row stores / atomics (few registers are alive there) and written to LDS at the
    // top of tile t + 1; the state rows of a phase are requested one phase ahead.
    auto load_idx = [&](int* dst, long tile_, int t) {     // threads 0 .. 15
        const long row = tile_ * kPgRows + t;
        const bool valid = tile_ < ntiles && row < io.rows;
        if constexpr (MODE == MODE_EDGE) {
            dst[t] = valid ? io.seg[0].ia[row] : -1;
            dst[16 + t] = valid ? io.seg[0].ib[row] : 0;
            dst[32 + t] = valid ? io.seg[1].ia[row] : 0;
        } else {
            dst[t] = valid ? t : -1;
            float sc = 1.0f;
            if (valid && io.seg[1].deg) { const int dg = io.seg[1].deg[row]; sc = 1.0f / float(dg > 1 ? dg : 1); }
            reinterpret_cast<float*>(dst)[48 + t] = sc;
        }
    };
    f4 pre[NPRE];
    auto issue_gout = [&](const int* idx, long tile_, int t) {
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int p = t + i * kPgThreads, rr = p / PPR, e = p % PPR;
            pre[i] = f4{0.f, 0.f, 0.f, 0.f};
            if (idx[rr] >= 0) {
                if constexpr (K == 1) {
                    const size_t grow = MODE == MODE_EDGE ? (size_t)idx[rr] : (size_t)(tile_ * kPgRows + rr);
                    pre[i] = pg_ld4(io.gy + grow * ROW + 4 * e);
                } else {
                    pre[i] = pg_ld4(io.plw_g1 + (size_t)(tile_ * kPgRows + rr) * ROW + 4 * e);
                }
            }
        }
    };
    if (threadIdx.x < kPgRows) load_idx(sidx, blockIdx.x, threadIdx.x);
    __syncthreads();
    issue_gout(sidx, blockIdx.x, threadIdx.x);
    stamp(0);

    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long row0 = tile * kPgRows;
        const float* const sscale = reinterpret_cast<const float*>(sidx) + 48;
        float s_st[32], y_st[32];     // state rows of the first ROW phase, requested in front of the staging barrier
        {
        PG_PHASE_IDS();
#pragma unroll
        for (int d = 0; d < 32; ++d) { s_st[d] = 0.f; y_st[d] = 0.f; }
        if (live) {
            pg_load_state(s_st, io.saved + state_region<ROW, ROWP>(io.rows, 0, K) + soff);
            pg_load_state(y_st, io.saved + state_region<ROW, ROWP>(io.rows, 1, K) + soff);
        }
        // ---- d/d(block output) rows (requested during the previous tile) -> A
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int p = tid + i * kPgThreads, rr = p / PPR, e = p % PPR;
            pg_st4(bufA + pg_off(e >> 3, rr, e & 7), pre[i]);
        }
        if (tid < kPgRows) load_idx(sidx_n, tile + gridDim.x, tid);
        }
        __syncthreads();
        stamp(1);
        // ---- ROW: z -> A, layer-norm backward -> ggp -> B
        float ggp[32];
        float g_la, g_bL;
        {
            PG_PHASE_IDS();
            float (&s)[32] = s_st;
            float (&y)[32] = y_st;
            pg_ld32(ggp, bufA, r, c);     // d/d(out)
            static_for<0, G>([&](auto g) {
                constexpr int d0 = ALG::gstart(g), nd = ALG::gsize(g);
                float u;
                if constexpr (g == 0) {
                    u = y[0];
                } else {
                    u = 0.f;
                    static_for<0, nd>([&](auto t) {
                        constexpr int d = d0 + decltype(t)::value;
                        u = __builtin_fmaf(qsf<ALG, d> * y[d], y[d], u);
                    });
                }
                const float gate = sigmoidf(__builtin_fmaf(par[4 + g], u, par[10 + g]));
#pragma unroll
                for (int t = 0; t < nd; ++t) y[d0 + t] *= gate;
            });
            pg_st32(bufA, r, c, y);       // z (zero rows / channels beyond the tile)
            float qs = 0.f, dot = 0.f;
            static_for<0, 32>([&](auto dd) {
                constexpr int d = decltype(dd)::value;
                qs = __builtin_fmaf(qsf<ALG, d> * s[d], s[d], qs);
                dot = __builtin_fmaf(ggp[d], s[d], dot);
            });
            const float nl = sqrt_pos(sqrt_pos(__builtin_fmaf(qs, qs, kSmooth)));
            const float la = par[2];
            ln1[r * 32 + c] = cvalid ? nl : 0.f;
            ln2[r * 32 + c] = live ? la * dot : 0.f;
            __syncthreads();
            float tot = 0.f, totd = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f4 v1 = pg_ld4(ln1 + r * 32 + 4 * j), v2 = pg_ld4(ln2 + r * 32 + 4 * j);
                tot += (v1.x + v1.y) + (v1.z + v1.w);
                totd += (v2.x + v2.y) + (v2.z + v2.w);
            }
            const float invMn = fast_rcp(__builtin_fmaf(tot, 1.0f / float(C), kEps));
            const float gMn = -totd * invMn * invMn * (1.0f / float(C));
            const float inl = fast_rcp(nl);
            const float gqs = gMn * (0.5f * qs) * (inl * inl * inl);
            const float k0 = la * invMn;
            static_for<0, 32>([&](auto dd) {
                constexpr int d = decltype(dd)::value;
                const float gs = __builtin_fmaf(k0, ggp[d], gqs * (2.0f * qsf<ALG, d>) * s[d]);
                ggp[d] = live ? gs * kInvSqrt2 : 0.f;
            });
            g_la = live ? dot * invMn : 0.f;
            g_bL = ggp[0];
            pg_st32(bufB, r, c, ggp);
        }
        __syncthreads();
        stamp(2);
        // ---- MIX: gz = WL^T ggp (this wave's 4 blades), d/dWL tile += ggp^T z; then gz over ggp in B
        float R_st[32];               // the next ROW phase's state rows travel under the MFMAs
        {
            PG_PHASE_IDS();
#pragma unroll
            for (int d = 0; d < 32; ++d) R_st[d] = 0.f;
            if (live) pg_load_state(R_st, io.saved + state_region<ROW, ROWP>(io.rows, 2, K) + soff);
            f4 acc[4][2];
            pg_zero(acc);
            // the two waves of a SIMD share its matrix pipe: waves 4-7 take the two MFMA sections in the other order, so that one
            // wave's LDS reads run under its partner's MFMAs
            if (wave < 4) {
                pg_mix_acc<NST>(acc, bufB, tabs + CF::ttoff(K, mL), lane, wave);
                pg_wgrad(accL, bufB, bufA, wave, lane);
            } else {
                pg_wgrad(accL, bufB, bufA, wave, lane);
                pg_mix_acc<NST>(acc, bufB, tabs + CF::ttoff(K, mL), lane, wave);
            }
            __syncthreads();
            pg_write_d(bufB, acc, lane, wave);
        }
        __syncthreads();
        stamp(3);
        // ---- ROW: geometric product + normalisation backward -> gR -> B; gz stays in registers
        float gz[32];
        PgCollect col;
        {
            PG_PHASE_IDS();
            float z[32];
            float (&R)[32] = R_st;
            pg_ld32(gz, bufB, r, c);
            pg_ld32(z, bufA, r, c);
            float invden[G], den[G], nu[G], qR[G];
            static_for<0, G>([&](auto g) {
                constexpr int d0 = ALG::gstart(g), nd = ALG::gsize(g);
                float qq = 0.f;
                static_for<0, nd>([&](auto t) {
                    constexpr int d = d0 + decltype(t)::value;
                    qq = __builtin_fmaf(qsf<ALG, d> * R[d], R[d], qq);
                });
                qR[g] = qq;
                nu[g] = sqrt_pos(sqrt_pos(__builtin_fmaf(qq, qq, kSmooth)));
                den[g] = __builtin_fmaf(par[16 + g], nu[g] - 1.0f, 1.0f) + kEps;
                invden[g] = fast_rcp(den[g]);
#pragma unroll
                for (int t = 0; t < nd; ++t) R[d0 + t] *= invden[g];    // R holds r = R / den from here on (R = r den where it is needed again)
            });
            float (&rf)[32] = R;
            pg_gp_bwd_z<ALG>(ggp, z, rf, gz, wrow, col, small, l16);
            pg_pin32(gz);
            CSMPN_PHASE();
            float gr[32];
#pragma unroll
            for (int d = 0; d < 32; ++d) gr[d] = 0.f;
            pg_gp_bwd_r<ALG>(ggp, z, gr, wrow);
            pg_pin32(gr);
            CSMPN_PHASE();
            // NormalizationLayer backward: gR (into gr), d/d(an)
            static_for<0, G>([&](auto g) {
                constexpr int d0 = ALG::gstart(g), nd = ALG::gsize(g);
                float gden = 0.f;
                static_for<0, nd>([&](auto t) {
                    constexpr int d = d0 + decltype(t)::value;
                    gden = __builtin_fmaf(-gr[d], rf[d], gden);
                });
                gden *= invden[g];          // -sum gr R / den^2 with R = r den
                const float sg = par[16 + g];
                col.template add<56 + g>(gden * (nu[g] - 1.0f) * sg * (1.0f - sg), small, l16);
                const float inu = fast_rcp(nu[g]);
                const float gq = (gden * sg) * (0.5f * qR[g]) * (inu * inu * inu);
                static_for<0, nd>([&](auto t) {
                    constexpr int d = d0 + decltype(t)::value;
                    gr[d] = __builtin_fmaf(gr[d], invden[g], (gq * den[g]) * (2.0f * qsf<ALG, d>) * rf[d]);
                });
            });
            col.template add<62>(0.f, small, l16);
            col.template add<63>(0.f, small, l16);
            pg_st32(bufB, r, c, gr);      // gR
        }
        __syncthreads();
        stamp(4);
        // ---- MIX: WR^T gR, d/dWR tile += gR^T z; the result over gR in B
        float y2_st[32];
        {
            PG_PHASE_IDS();
#pragma unroll
            for (int d = 0; d < 32; ++d) y2_st[d] = 0.f;
            if (live) pg_load_state(y2_st, io.saved + state_region<ROW, ROWP>(io.rows, 1, K) + soff);
            f4 acc[4][2];
            pg_zero(acc);
            // the two waves of a SIMD share its matrix pipe: waves 4-7 take the two MFMA sections in the other order, so that one
            // wave's LDS reads run under its partner's MFMAs
            if (wave < 4) {
                pg_mix_acc<NST>(acc, bufB, tabs + CF::ttoff(K, mR), lane, wave);
                pg_wgrad(accR, bufB, bufA, wave, lane);
            } else {
                pg_wgrad(accR, bufB, bufA, wave, lane);
                pg_mix_acc<NST>(acc, bufB, tabs + CF::ttoff(K, mR), lane, wave);
            }
            __syncthreads();
            pg_write_d(bufB, acc, lane, wave);
        }
        __syncthreads();
        stamp(5);
        // ---- block input -> A (+ E): z has been read for the last time. Coalesced pieces, all threads; the rows are
        // requested here and written behind the MVSiLU backward
        f4 xa[NPRE], xb[K == 0 && MODE == MODE_EDGE ? NPRE : 1];
        {
        PG_PHASE_IDS();
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int p = tid + i * kPgThreads, rr = p / PPR, e = p % PPR;
            xa[i] = f4{0.f, 0.f, 0.f, 0.f};
            if constexpr (K == 0 && MODE == MODE_EDGE) xb[i] = f4{0.f, 0.f, 0.f, 0.f};
            if (sidx[rr] >= 0) {
                if constexpr (K == 1) xa[i] = pg_ld4(io.saved + (size_t)(row0 + rr) * ROW + 4 * e);
                else if constexpr (MODE == MODE_EDGE) {
                    xa[i] = pg_ld4(io.seg[0].a + (size_t)sidx[rr] * ROW + 4 * e);
                    xb[i] = pg_ld4(io.seg[0].b + (size_t)sidx[16 + rr] * ROW + 4 * e);
                } else xa[i] = pg_ld4(io.seg[0].a + (size_t)(row0 + rr) * ROW + 4 * e);
            }
        }
        if constexpr (K == 0) {
            constexpr int PPA = 16 * 8;   // one 16-channel tile of attribute slots (the rows-contracting MFMA reads all of them)
            for (int p = tid; p < kPgRows * PPA; p += kPgThreads) {
                const int rr = p / PPA, e = p % PPA;
                f4 v = f4{0.f, 0.f, 0.f, 0.f};
                if (sidx[rr] >= 0 && e < NA * 8) {
                    if constexpr (MODE == MODE_EDGE) v = pg_ld4(io.seg[1].a + (size_t)sidx[32 + rr] * (NA * D) + 4 * e);
                    else v = pg_ld4(io.seg[2].a + (size_t)(row0 + rr) * (NA * D) + 4 * e);
                }
                if (e < 8 * 8) pg_st4(bufE + pg_off(e >> 3, rr, e & 7), v);
            }
        }
        }
        // ---- ROW: MVSiLU backward -> gy -> B
        {
            PG_PHASE_IDS();
            float t_[32];
            float (&y)[32] = y2_st;
            pg_ld32(t_, bufB, r, c);
#pragma unroll
            for (int d = 0; d < 32; ++d) gz[d] += t_[d];
            static_for<0, G>([&](auto g) {
                constexpr int d0 = ALG::gstart(g), nd = ALG::gsize(g);
                float u, ggate = 0.f;
                if constexpr (g == 0) {
                    u = y[0];
                } else {
                    u = 0.f;
                    static_for<0, nd>([&](auto t) {
                        constexpr int d = d0 + decltype(t)::value;
                        u = __builtin_fmaf(qsf<ALG, d> * y[d], y[d], u);
                    });
                }
                const float gate = sigmoidf(__builtin_fmaf(par[4 + g], u, par[10 + g]));
#pragma unroll
                for (int t = 0; t < nd; ++t) ggate = __builtin_fmaf(gz[d0 + t], y[d0 + t], ggate);
                const float gpre = ggate * gate * (1.0f - gate);
                col.template add<64 + 2 * g>(gpre * u, small, l16);
                col.template add<64 + 2 * g + 1>(gpre, small, l16);
                const float gu = gpre * par[4 + g];
                static_for<0, nd>([&](auto t) {
                    constexpr int d = d0 + decltype(t)::value;
                    float v = gz[d] * gate;
                    if constexpr (g == 0) v += gu;
                    else v = __builtin_fmaf(gu * (2.0f * qsf<ALG, d>), y[d], v);
                    gz[d] = live ? v : 0.f;     // gy
                });
            });
            col.template add<76>(gz[0], small, l16);
            col.template add<77>(g_la, small, l16);
            col.template add<78>(g_bL, small, l16);
            col.template add<79>(0.f, small, l16);
            pg_st32(bufB, r, c, gz);
#pragma unroll
            for (int i = 0; i < NPRE; ++i) {
                const int p = tid + i * kPgThreads, rr = p / PPR, e = p % PPR;
                if constexpr (K == 0 && MODE == MODE_EDGE) xa[i] -= xb[i];
                pg_st4(bufA + pg_off(e >> 3, rr, e & 7), xa[i]);
            }
        }
        __syncthreads();
        stamp(6);
        // ---- MIX: d/dW1 += gy^T x, gx = W1^T gy
        if constexpr (K == 1 || MODE == MODE_EDGE) {
            PG_PHASE_IDS();
            f4 acc[4][2];
            pg_zero(acc);
            pg_mix_acc<NST>(acc, bufB, tabs + CF::ttoff(K, 0), lane, wave);
            pg_wgrad(accW0, bufB, bufA, wave, lane);
            f4 acca[4][1];
            bool want_a = false;
            if constexpr (K == 0) {
                if (wave < 4) pg_wgrad1(accW1, bufB, bufE, wave, lane);
                want_a = io.gx[1] != nullptr;
                if (want_a) {
                    pg_zero(acca);
                    pg_mix_acc<NST, 1>(acca, bufB, tabs + CF::ttoff(0, 1), lane, wave);
                }
            }
            __syncthreads();
            pg_write_d(bufB, acc, lane, wave);
            if constexpr (K == 0) { if (want_a) pg_write_d<1, 8>(bufE, acca, lane, wave); }
        } else {
            // node program, block 0: h in A now; the aggregate follows through A, the attributes sit in E
            PG_PHASE_IDS();
            f4 acch[4][2], accg[4][2];
            pg_zero(acch);
            pg_mix_acc<NST>(acch, bufB, tabs + CF::ttoff(0, 0), lane, wave);
            pg_wgrad(accW0, bufB, bufA, wave, lane);
            if (wave < 4) pg_wgrad1(accW2, bufB, bufE, wave, lane);
            f4 acca[4][1];
            const bool want_a = io.gx[2] != nullptr;
            if (want_a) {
                pg_zero(acca);
                pg_mix_acc<NST, 1>(acca, bufB, tabs + CF::ttoff(0, 2), lane, wave);
            }
            __syncthreads();             // every wave is done with h in A
            for (int p = tid; p < kPgRows * PPR; p += kPgThreads) {
                const int rr = p / PPR, e = p % PPR;
                f4 v = f4{0.f, 0.f, 0.f, 0.f};
                if (sidx[rr] >= 0) v = pg_ld4(io.seg[1].a + (size_t)(row0 + rr) * ROW + 4 * e) * sscale[rr];
                pg_st4(bufA + pg_off(e >> 3, rr, e & 7), v);
            }
            __syncthreads();
            pg_zero(accg);
            pg_mix_acc<NST>(accg, bufB, tabs + CF::ttoff(0, 1), lane, wave);
            pg_wgrad(accW1, bufB, bufA, wave, lane);
            __syncthreads();             // gy in B and the aggregate in A have been read for the last time
            pg_write_d(bufA, acch, lane, wave);      // d/dh
            pg_write_d(bufB, accg, lane, wave);      // d/d(scaled aggregate)
            if (want_a) pg_write_d<1, 8>(bufE, acca, lane, wave);
        }
        __syncthreads();
        stamp(7);
        // ---- rows out; the next tile's d/d(out) rows are requested first (vmcnt counts in order: behind the atomics
        // they would wait for their acknowledgement)
        {
        PG_PHASE_IDS();
        issue_gout(sidx_n, tile + gridDim.x, tid);
        if constexpr (K == 1) {
            for (int p = tid; p < kPgRows * PPR; p += kPgThreads) {
                const int rr = p / PPR, e = p % PPR;
                if (row0 + rr < io.rows) pg_st4(io.plw_g1 + (size_t)(row0 + rr) * ROW + 4 * e, pg_ld4(bufB + pg_off(e >> 3, rr, e & 7)));
            }
        } else if constexpr (MODE == MODE_EDGE) {
            if (io.gx[0]) {
                if (io.row_store) {
                    for (int p = tid; p < kPgRows * PPR; p += kPgThreads) {
                        const int rr = p / PPR, e = p % PPR;
                        if (row0 + rr < io.rows) pg_st4(io.gx[0] + (size_t)(row0 + rr) * ROW + 4 * e, pg_ld4(bufB + pg_off(e >> 3, rr, e & 7)));
                    }
                } else {
                    for (int col_ = tid; col_ < ROW; col_ += kPgThreads) {
                        const int ch = col_ >> 5, d = col_ & 31;
                        float acc = 0.f;
                        int cur = sidx[0];
#pragma unroll
                        for (int rr = 0; rr < kPgRows; ++rr) {
                            const int t_ = sidx[rr];
                            if (t_ != cur) {
                                if (cur >= 0) atomicAdd(io.gx[0] + (size_t)cur * ROW + col_, acc);
                                cur = t_;
                                acc = 0.f;
                            }
                            const float v = bufB[pg_off(ch, rr, d >> 2) + (d & 3)];
                            acc += v;
                            if (t_ >= 0) atomicAdd(io.gx[0] + (size_t)sidx[16 + rr] * ROW + col_, -v);
                        }
                        if (cur >= 0) atomicAdd(io.gx[0] + (size_t)cur * ROW + col_, acc);
                    }
                }
            }
            if (io.gx[1]) {
                for (int p = tid; p < kPgRows * NA * 8; p += kPgThreads) {
                    const int rr = p / (NA * 8), e = p % (NA * 8);
                    if (row0 + rr < io.rows) pg_st4(io.gx[1] + (size_t)sidx[32 + rr] * (NA * D) + 4 * e, pg_ld4(bufE + pg_off(e >> 3, rr, e & 7)));
                }
            }
        } else {
            for (int p = tid; p < kPgRows * PPR; p += kPgThreads) {
                const int rr = p / PPR, e = p % PPR;
                if (row0 + rr < io.rows) {
                    if (io.gx[0]) {
                        f4 v = pg_ld4(bufA + pg_off(e >> 3, rr, e & 7));
                        if (io.resid_bwd) v += pg_ld4(io.gy + (size_t)(row0 + rr) * ROW + 4 * e);
                        pg_st4(io.gx[0] + (size_t)(row0 + rr) * ROW + 4 * e, v);
                    }
                    if (io.gx[1]) pg_st4(io.gx[1] + (size_t)(row0 + rr) * ROW + 4 * e, pg_ld4(bufB + pg_off(e >> 3, rr, e & 7)) * sscale[rr]);
                }
            }
            if (io.gx[2]) {
                for (int p = tid; p < kPgRows * NA * 8; p += kPgThreads) {
                    const int rr = p / (NA * 8), e = p % (NA * 8);
                    if (row0 + rr < io.rows) pg_st4(io.gx[2] + (size_t)(row0 + rr) * (NA * D) + 4 * e, pg_ld4(bufE + pg_off(e >> 3, rr, e & 7)));
                }
            }
        }
        }
        __syncthreads();
        { int* t_ = sidx; sidx = sidx_n; sidx_n = t_; }
        stamp(8);
    }
    // ---- this workgroup's slice: every element has one owner
    float* slice = io.plw_part + (size_t)blockIdx.x * CF::slice_floats(K);
    {
        const int half = wave & 1, tl = wave >> 1, ot = tl >> 1, ct = tl & 1;
        pg_store_unit<2>(slice + CF::woff(K, mL), accL, ot, ct, half, lane);
        pg_store_unit<2>(slice + CF::woff(K, mR), accR, ot, ct, half, lane);
        pg_store_unit<2>(slice + CF::woff(K, 0), accW0, ot, ct, half, lane);
        if constexpr (K == 0) {
            if constexpr (MODE == MODE_EDGE) {
                if (wave < 4) pg_store_unit<1>(slice + CF::woff(0, 1), accW1, wave >> 1, 0, half, lane);
            } else {
                pg_store_unit<2>(slice + CF::woff(0, 1), accW1, ot, ct, half, lane);
                if (wave < 4) pg_store_unit<1>(slice + CF::woff(0, 2), accW2, wave >> 1, 0, half, lane);
            }
        }
#pragma unroll
        for (int g = 0; g < 5; ++g) slice[CF::slice_w(K) + c * CF::kSmall + 16 * g + l16] = cvalid ? small[g] : 0.f;
    }
    stamp(9);
    stamp.flush(io.stamps, lane);
}

// grads += sum over the workgroups' slices, fixed order (one thread per slice element)
// A workgroup takes 64 consecutive elements; thread (j = tid & 63, q = tid >> 6) sums the slices q, q + 4, ... (16 loads in
// flight, compensated), the four partial sums of an element meet in LDS and are added in order.
template <class ALG, class CF, int K>
__global__ void __launch_bounds__(256) pg_reduce_kernel(const DevCemlp Cd, const float* part, int nslices) {
    constexpr int C = CF::C, G = 6, P = CF::P, SF = CF::slice_floats(K), SW = CF::slice_w(K);
    __shared__ float red[4][64];
    const int j = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + j;
    // compensated (Kahan) sums: the slices of a large launch cancel heavily on indefinite metrics
    float s = 0.f, comp = 0.f;
    auto add = [&](float v) {
        const float yk = v - comp, t = s + yk;
        comp = (t - s) - yk;
        s = t;
    };
    if (e < SF) {
        for (int sl = q; sl < nslices; sl += 64) {
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = sl + 4 * i < nslices ? part[(size_t)(sl + 4 * i) * SF + e] : 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) add(v[i]);
        }
    }
    red[q][j] = s;
    __syncthreads();
    if (q != 0 || e >= SF) return;
    s = ((red[0][j] + red[1][j]) + red[2][j]) + red[3][j];
    const DevBlock& B = Cd.b[K];
    if (e < SW) {
        int m = 0, f = e;
        bool found = false;
        static_for<0, CF::nmat(K)>([&](auto mm) {
            constexpr int m_ = decltype(mm)::value;
            if (!found && e >= CF::woff(K, m_) && e < CF::woff(K, m_) + CF::wmat_floats(K, m_)) { m = m_; f = e - CF::woff(K, m_); found = true; }
        });
        const int nct = CF::nct(K, m), which = CF::which(K, m);
        const int v = f & 3, lane = (f >> 2) & 63;
        int rest = f >> 8;
        const int ct = rest % nct; rest /= nct;
        const int ot = rest & 1, g = rest >> 1;
        const int o = 16 * ot + 4 * (lane >> 4) + v, cl = 16 * ct + (lane & 15);
        if (o < C && cl < CF::nch(K, m)) {
            float* gW = which == 0 ? B.gW1 : (which == 1 ? B.gWR : B.gWL);
            const int I = which == 0 ? B.I : C;
            gW[((size_t)o * I + CF::cbase(K, m) + cl) * G + g] += s;
        }
    } else {
        const int f = e - SW, ch = f / CF::kSmall, idx = f % CF::kSmall;
        if (ch < C) {
            if (idx < P) B.gw[ch * P + idx] += s;
            else if (idx < P + G) B.gan[ch * G + (idx - P)] += s;
            else if (idx >= 64 && idx < 76) { const int g = (idx - 64) >> 1; if (idx & 1) B.gsb[ch * G + g] += s; else B.gsa[ch * G + g] += s; }
            else if (idx == 76) { if (B.has_b1) B.gb1[ch] += s; }
            else if (idx == 77) B.gla[ch] += s;
            else if (idx == 78) B.gbL[ch] += s;
        }
    }
}

}  // namespace csmpn
