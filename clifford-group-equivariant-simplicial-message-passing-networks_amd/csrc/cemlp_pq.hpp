// "pq" kernels: Cl(3,0) layers of 32 channels (md17's width, md17_cssmpnn.py:12-14) on 16-ROW TILES with the dense channel
// mixing on v_mfma_f32_16x16x4_f32, four-wave workgroups, THREE workgroups per CU (round 5).
//
// Same arithmetic as every other family (csmpn/models/cegnn_utils.py:34-155,287-338) and the same two
// lane layouts as the D = 32 kernels of cemlp_pg.hpp, scaled to 8 blades: a tile is 16 rows x 32 channels x 8 blades = 16 KB,
// so a workgroup is 4 waves and a CU holds three of them - the workgroups run out of phase and one's MFMA phase fills the
// matrix pipe while its neighbours' VALU phases run (the one-workgroup-per-CU kernels alternate between the two pipes).
//
//   MIX layout (MFMA phases)   wave w = (piece p = w & 1: blades 4p .. 4p+3, output tile ot = w >> 1: channels 16 ot .. 16 ot + 15).
//                              out[r, o, d] = sum_c W[o][c][grade d] x[r, c, d]: per blade a [16 x 32] x [32 x 16 rows] product,
//                              B operand = the tile itself (lane (row n, k) reads channel 4s + k of its 4 blades with ONE
//                              ds_read_b128 per k-step), A operand = weight fragments packed once per launch into the workspace
//                              (pg_pack_kernel), result D[o][row] in lane (row, o / 4), register o % 4, written back as 16-byte
//                              pieces: 32 MFMAs per wave and 32 x 32 matrix, no padding.
//   ROW layout (VALU phases)   thread t = (row n = t & 15, cq = t >> 4) holds the multivectors of the channels cq and cq + 16,
//                              all 8 blades in registers: gates, normalisation, geometric product (64 sign-table terms) and
//                              the per-channel part of the layer norm in-lane; the mean over a row's channels: two lane
//                              exchanges inside the wave + one float per (wave, row) through LDS.
//
// LDS tensor buffer: element (channel c, row r, blade d) at 136 c + 8 r + 4 ((d >> 2) ^ (r & 1)) + (d & 3) floats: every access
// is a b128, and with the slot stride 136 = 8 (mod 64) and the XOR all of them - MIX reads, ROW reads, the rows-contracting
// weight-gradient reads (lane = (channel, row)), the coalesced row I/O and the dword reads of the scatter - are free of bank
// conflicts (searched exhaustively: tools/pq_layout.py); two buffers of 17 KB + an 8-slot attribute region + parameters = 46-52 KB.
// Backward: one launch per block on the state the forward saved (y, R, s in ROW-layout lane order), phases as cemlp_pg.hpp;
// a wave owns ONE (o-tile, c-tile) tile of every weight gradient (all four grades: 16 registers per matrix).
#pragma once
#include "cemlp_device.hpp"
#include "cemlp_pg.hpp"   // pg_pack_kernel, pg_ld4 / pg_st4, pg_rows_sum, pg_tid, PgStamp

namespace csmpn {

constexpr int kPqRows = 16;
constexpr int kPqWaves = 4, kPqThreads = 64 * kPqWaves;
constexpr int kPqCS = 136;                       // channel-slot stride of a tensor buffer (floats)
constexpr int kPqBuf = 32 * kPqCS;               // floats per buffer
constexpr int kPqMaxGroups = 768;                // three workgroups per CU

// float offset of 16-byte piece p (blades 4p .. 4p+3) of (channel slot c, row r)
CSMPN_DEV int pq_off(int c, int r, int p) { return c * kPqCS + 8 * r + 4 * (p ^ (r & 1)); }

// MODE_EDGE / MODE_NODE: the EGCL programs (NA_ attribute channels, two blocks). MODE_PLAIN (standalone CEMLP of NBLK_ = 1 or 2 blocks:
// the simplex embeddings and the head of the md17 model, md17_cssmpnn.py:85-120,165-176): NA_ = the input channels I0 <= 96, taken
// as chunks of 32 (the last one may be narrower, more than 16 channels wide).
template <class ALG, int C_, int MODE_, int NA_, int NBLK_ = 2>
struct PqCfg {
    static_assert(ALG::n == 3, "8 blades");
    static constexpr int C = C_, MODE = MODE_, NA = NA_, NBLK = NBLK_, D = ALG::D, G = ALG::G, P = ALG::P;
    static constexpr bool PLAIN = MODE_ == MODE_PLAIN;
    static_assert(C == 32 && G == 4 && D == 8 && P % 4 == 0 && (NBLK == 1 || NBLK == 2), "32 channels");
    static_assert(PLAIN ? (NA > 0 && NA <= 96 && (NA % 32 == 0 || NA % 32 > 16)) : (NA > 0 && NA <= 8 && NBLK == 2),
                  "EGCL: one attribute chunk, two blocks; plain: up to three chunks of 17 .. 32 channels");
    static constexpr int ROW = C * D;
    static constexpr int NST = (C + 3) / 4;           // k-steps of a C-channel operand
    static constexpr int NSTA = (NA + 3) / 4;         // ... of the attribute chunk
    static constexpr int par_stride = 16;             // b1 bL la 0 | sa[4] | sb[4] | sigmoid(an)[4]
    // matrix-chunks ("mats") of a block, in table order (as PgCfg). Block 0: the W1 column blocks of the input chunks, then WR, WL;
    // block 1: W1, WR, WL.  EDGE chunks: [h_dst - h_src (C) in A][edge_attr (NA) in B];  NODE: [h (C) in A][agg (C) in B][node_attr (NA) in E]
    static constexpr int NCH0 = PLAIN ? (NA + 31) / 32 : (MODE == MODE_EDGE ? 2 : 3);
    static constexpr int I0 = PLAIN ? NA : (MODE == MODE_EDGE ? C + NA : 2 * C + NA);
    static constexpr int nmat(int K) { return K == 0 ? NCH0 + 2 : 3; }
    static constexpr int nch(int K, int m) {
        if (K == 0 && m < NCH0) return PLAIN ? (I0 - 32 * m < 32 ? I0 - 32 * m : 32) : (m == NCH0 - 1 ? NA : C);
        return C;
    }
    static constexpr int nst(int K, int m) { return (nch(K, m) + 3) / 4; }
    static constexpr int cbase(int K, int m) { return (K == 0 && m < NCH0) ? m * C : 0; }
    static constexpr int which(int K, int m) { return K == 0 ? (m < NCH0 ? 0 : m - NCH0 + 1) : m; }   // 0 W1, 1 WR, 2 WL
    static constexpr int ks4(int K, int m) { return (nst(K, m) + 3) / 4; }
    static constexpr int mat_f4(int K, int m) { return G * 2 * ks4(K, m) * 64; }     // [grade][o-tile][s4][lane]
    static constexpr int toff(int K, int m) {
        int o = 0;
        for (int k = 0; k < K; ++k) for (int q = 0; q < nmat(k); ++q) o += mat_f4(k, q);
        for (int q = 0; q < m; ++q) o += mat_f4(K, q);
        return o;
    }
    static constexpr int fwd_f4 = toff(NBLK, 0);
    static constexpr int nct(int K, int m) { return (nch(K, m) + 15) / 16; }        // 16-channel tiles of the operand
    static constexpr int tmat_f4(int K, int m) { return G * nct(K, m) * ((NST + 3) / 4) * 64; }   // [grade][c-tile][s4 over the OUT channels][lane]
    static constexpr int ttoff(int K, int m) {
        int o = fwd_f4;
        for (int k = 0; k < K; ++k) for (int q = 0; q < nmat(k); ++q) o += tmat_f4(k, q);
        for (int q = 0; q < m; ++q) o += tmat_f4(K, q);
        return o;
    }
    static constexpr int all_f4 = ttoff(NBLK, 0);
    static constexpr int tab_floats = 4 * all_f4;
    // backward: slice of block K: mat m at woff(K, m): [grade][o-tile][c-tile][lane][4]; then the per-channel sums [32 channels][kSmall]
    static constexpr int kSmall = 48;      // w[P = 20] | an[4] | 8 pad | (sa, sb)[4] | b1 | la | bL | 5 pad
    static constexpr int s_an = P, s_gate = 32, s_b1 = 40, s_la = 41, s_bL = 42;
    static_assert(P + G <= 32, "slot order");
    static constexpr int wmat_floats(int K, int m) { return G * 2 * nct(K, m) * 256; }
    static constexpr int woff(int K, int m) { int o = 0; for (int q = 0; q < m; ++q) o += wmat_floats(K, q); return o; }
    static constexpr int slice_w(int K) { return woff(K, nmat(K)); }
    static constexpr int slice_floats(int K) { return slice_w(K) + 32 * kSmall; }
    static constexpr int slice_both = slice_floats(0) + (NBLK > 1 ? slice_floats(1) : 0);   // each block's launch has its own region
    // LDS (floats). backward: A | B | E (8 slots) | two row-sum arrays [4 waves][16 rows] | path weights and parameters of ONE block | indices
    static constexpr int b_E = 2 * kPqBuf, b_ln = b_E + 8 * kPqCS, b_w = b_ln + 2 * kPqWaves * kPqRows, b_par = b_w + 32 * P,
                         b_idx = b_par + 32 * par_stride, bwd_lds_floats = b_idx + 128;
    // forward: A | B | E | one row-sum array | path weights and parameters of BOTH blocks | indices
    static constexpr int o_A = 0, o_B = kPqBuf, o_E = 2 * kPqBuf, o_ln = o_E + 8 * kPqCS, o_w = o_ln + kPqWaves * kPqRows,
                         o_par = o_w + 2 * 32 * P, o_idx = o_par + 2 * 32 * par_stride, lds_floats = o_idx + 128;
    static_assert(lds_floats * 4 * 3 <= 160 * 1024 && bwd_lds_floats * 4 * 3 <= 160 * 1024, "three workgroups per CU");
};

// ---------------------------------------------------------------------------------
// MIX phase: acc[bl] += W (o-tile `ot` of the mat at `tab`, NT o-tiles in the table) x (operand tile in `buf`), piece p. NSTEP k-steps.
// The A fragments (grades of piece 0: 0 1 1 1, of piece 1: 2 2 2 3) are requested by pq_load_a - one ROW phase ahead, IN FRONT of
// that phase's stores / state loads: vmcnt counts in issue order, a fragment load issued behind them would wait for them.
template <int NSTEP>
struct PqA {
    static constexpr int KS4 = (NSTEP + 3) / 4;
    f4 lo[KS4], hi[KS4];
};
template <int NSTEP>
CSMPN_DEV void pq_load_a(PqA<NSTEP>& a, const f4* tab, int lane, int p, int ot, int NT) {
    int fence_ = 0;
    asm volatile("" : "+s"(fence_));   // the loads start here, not at the top of the tile loop (cemlp_pg.hpp: pg_mix_acc)
    tab += fence_;
    constexpr int KS4 = PqA<NSTEP>::KS4;
#pragma unroll
    for (int s4 = 0; s4 < KS4; ++s4) {
        a.lo[s4] = tab[(((2 * p) * NT + ot) * KS4 + s4) * 64 + lane];
        a.hi[s4] = tab[(((2 * p + 1) * NT + ot) * KS4 + s4) * 64 + lane];
    }
}
template <int NSTEP>
CSMPN_DEV void pq_mix_run(f4 (&acc)[4], const float* buf, const PqA<NSTEP>& a, int lane, int p) {
    const int n = lane & 15, k = lane >> 4;
    const float* bp = buf + pq_off(k, n, p);
    f4 b[NSTEP];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) b[s] = pg_ld4(bp + 4 * s * kPqCS);
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        const float a0 = a.lo[s / 4][s % 4], a3 = a.hi[s / 4][s % 4];
        const float am = p ? a0 : a3;       // blades 1, 2 of the piece
        acc[0] = mfma16(a0, b[s][0], acc[0]);
        acc[1] = mfma16(am, b[s][1], acc[1]);
        acc[2] = mfma16(am, b[s][2], acc[2]);
        acc[3] = mfma16(a3, b[s][3], acc[3]);
    }
}
template <int NSTEP>
CSMPN_DEV void pq_mix_acc(f4 (&acc)[4], const float* buf, const f4* tab, int lane, int p, int ot, int NT) {
    PqA<NSTEP> a;
    pq_load_a<NSTEP>(a, tab, lane, p, ot, NT);
    pq_mix_run<NSTEP>(acc, buf, a, lane, p);
}
// two matrices on the same operand (linear_right and linear_left of z): one pass over the B fragments
template <int NSTEP>
CSMPN_DEV void pq_mix_run2(f4 (&accR)[4], f4 (&accL)[4], const float* buf, const PqA<NSTEP>& aR, const PqA<NSTEP>& aL, int lane, int p) {
    const int n = lane & 15, k = lane >> 4;
    const float* bp = buf + pq_off(k, n, p);
    f4 b[NSTEP];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) b[s] = pg_ld4(bp + 4 * s * kPqCS);
    static_for<0, 2>([&](auto mm) {
        const PqA<NSTEP>& a = decltype(mm)::value == 0 ? aR : aL;
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const float a0 = a.lo[s / 4][s % 4], a3 = a.hi[s / 4][s % 4];
            const float am = p ? a0 : a3;
            if constexpr (decltype(mm)::value == 0) {
                accR[0] = mfma16(a0, b[s][0], accR[0]);
                accR[1] = mfma16(am, b[s][1], accR[1]);
                accR[2] = mfma16(am, b[s][2], accR[2]);
                accR[3] = mfma16(a3, b[s][3], accR[3]);
            } else {
                accL[0] = mfma16(a0, b[s][0], accL[0]);
                accL[1] = mfma16(am, b[s][1], accL[1]);
                accL[2] = mfma16(am, b[s][2], accL[2]);
                accL[3] = mfma16(a3, b[s][3], accL[3]);
            }
        }
    });
}
CSMPN_DEV void pq_zero(f4 (&acc)[4]) {
#pragma unroll
    for (int bl = 0; bl < 4; ++bl) acc[bl] = f4{0.f, 0.f, 0.f, 0.f};
}
// result D[o][row]: lane (row n, q) holds the channels 16 ot + 4 q + v in register v -> piece p of (channel, row n); SLOTS: channel
// slots of the target (results for further channels - zero rows of the table - are not stored)
template <int SLOTS = 32>
CSMPN_DEV void pq_write_d(float* buf, const f4 (&acc)[4], int lane, int p, int ot) {
    const int n = lane & 15, q = lane >> 4;
#pragma unroll
    for (int v = 0; v < 4; ++v)
        if (SLOTS >= 32 || 16 * ot + 4 * q + v < SLOTS)
            pg_st4(buf + pq_off(16 * ot + 4 * q + v, n, p), f4{acc[0][v], acc[1][v], acc[2][v], acc[3][v]});
}

// ROW layout: the 8 blades of (row r, channel slot c)
CSMPN_DEV void pq_ld8(float (&t)[8], const float* buf, int r, int c) {
    const f4 a = pg_ld4(buf + pq_off(c, r, 0)), b = pg_ld4(buf + pq_off(c, r, 1));
    t[0] = a.x; t[1] = a.y; t[2] = a.z; t[3] = a.w; t[4] = b.x; t[5] = b.y; t[6] = b.z; t[7] = b.w;
}
CSMPN_DEV void pq_st8(float* buf, int r, int c, const float (&t)[8]) {
    pg_st4(buf + pq_off(c, r, 0), f4{t[0], t[1], t[2], t[3]});
    pg_st4(buf + pq_off(c, r, 1), f4{t[4], t[5], t[6], t[7]});
}
CSMPN_DEV void pq_pin8(float (&t)[8]) {
    asm volatile("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]));
}
// state rows (CSMPN_FLAG_SAVE_STATE): piece (mv, p) of lane l of wave w of tile t at (((t * 4 + w) * 4 + 2 mv + p) * 64 + l) * 4 floats of the
// tensor's region - one store / load instruction of a wave covers 1 KB; a tile is 16 rows x 256 floats
CSMPN_DEV size_t pq_state_off(long tile, int wave, int lane, int mv) { return ((size_t)(tile * kPqWaves + wave) * 4 + 2 * mv) * 256 + 4 * lane; }
CSMPN_DEV void pq_store_state(float* p, const float (&t)[8]) {
    __builtin_nontemporal_store(f4{t[0], t[1], t[2], t[3]}, reinterpret_cast<f4*>(p));
    __builtin_nontemporal_store(f4{t[4], t[5], t[6], t[7]}, reinterpret_cast<f4*>(p + 256));
}
CSMPN_DEV void pq_load_state(float (&t)[8], const float* p) {
    const f4 a = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p)), b = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p + 256));
    t[0] = a.x; t[1] = a.y; t[2] = a.z; t[3] = a.w; t[4] = b.x; t[5] = b.y; t[6] = b.z; t[7] = b.w;
}
// piece i of thread t of a C-channel row tile: 64 pieces per row - the row (t >> 6) + 4 i is the same for a whole wave (a scalar),
// the piece is the lane; index-array entries of such a row are scalars too (row pointers in SGPRs, scalar validity branches)
CSMPN_DEV int pq_row_of(int t, int i) { return __builtin_amdgcn_readfirstlane(t >> 6) + 4 * i; }
CSMPN_DEV int pq_sidx(const int* a, int k) { return __builtin_amdgcn_readfirstlane(a[k]); }
// state region of tensor t (0 s, 1 y, 2 R) and block K behind the saved block inputs + hand-over rows (two-block programs: as
// cemlp_device.hpp::state_region; one block: the regions alone)
template <int ROW, int NBLK>
CSMPN_DEV size_t pq_state_region(long rows, int t, int K) {
    return (NBLK == 2 ? (size_t)2 * rows * ROW : 0) + (size_t)(NBLK * t + K) * state_rows(rows) * ROW;
}
// sum over the four lanes n, n + 16, n + 32, n + 48 of a wave (the four channel groups of a row)
CSMPN_DEV float pq_sum_q(float v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// out[j] += sum_p w[p] sum_{(i,k) -> j in path p} sign(i,k) z[i] r[k]   (cegnn_utils.py:126-152); wrow: this channel's P path weights (LDS)
template <class ALG>
CSMPN_DEV void pq_weighted_gp(float (&out)[8], const float (&z)[8], const float (&r)[8], const float* wrow) {
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
__global__ void __launch_bounds__(kPqThreads, 3) cemlp_pq_fwd_kernel(const DevCemlp C_arg, const RowIO io_arg) {
    typedef const char __attribute__((address_space(4))) * KArgPtr;
    const KArgPtr ka = (KArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr size_t kIoOffset = (sizeof(DevCemlp) + alignof(RowIO) - 1) / alignof(RowIO) * alignof(RowIO);
    const DevCemlp& Cd = *(const DevCemlp*)(const char*)ka;
    const RowIO& io = *(const RowIO*)(const char*)(ka + kIoOffset);
    (void)C_arg; (void)io_arg;
    constexpr int C = CF::C, MODE = CF::MODE, NA = CF::NA, D = 8, G = 4, P = CF::P, ROW = CF::ROW, NST = CF::NST;
    constexpr bool PLAIN = CF::PLAIN;
    constexpr int NBLK = CF::NBLK, I0 = CF::I0, NCH0 = CF::NCH0;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const bufA = smem + CF::o_A;
    float* const bufB = smem + CF::o_B;
    float* const bufE = smem + CF::o_E;
    float* const lnx = smem + CF::o_ln;
    int* sidx = reinterpret_cast<int*>(smem + CF::o_idx);            // this tile's [0..15] target / row, [16..31] source, [32..47] attribute row,
    int* sidx_n = sidx + 64;                                          // [48..63] node program: 1 / max(deg, 1) (float); the next tile's in the other half
    constexpr int PPR = C * 2, PPA = PLAIN ? 1 : 4 * CF::NSTA * 2;   // 16-byte pieces per row: a C-channel segment, the attribute chunk padded to whole k-steps
    constexpr int NPRE = PPR * kPqRows / kPqThreads, NPA = PLAIN ? 1 : (PPA * kPqRows + kPqThreads - 1) / kPqThreads;
    static_assert(NPRE * kPqThreads == PPR * kPqRows && PPR == 64, "whole pieces per thread, one row per wave and piece index (pq_row_of)");
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = tid & 15, cq = tid >> 4;                            // ROW layout: channels cq, cq + 16
    const int mp = wave & 1, mot = wave >> 1;                         // MIX layout: piece, output tile
    const f4* tabs = reinterpret_cast<const f4*>(io.plw_tabs);
    PgStamp stamp(0);

    // per-channel parameters and path weights of the blocks -> LDS
    static_for<0, NBLK>([&](auto kk) {
        constexpr int K = decltype(kk)::value;
        const DevBlock& B = Cd.b[K];
        for (int e = tid; e < 32 * CF::par_stride; e += kPqThreads) {
            const int ch = e / CF::par_stride, s = e % CF::par_stride;
            float v = 0.f;
            if (s == 0) v = B.has_b1 ? B.b1[ch] : 0.f;
            else if (s == 1) v = B.bL[ch];
            else if (s == 2) v = B.la[ch];
            else if (s >= 4 && s < 8) v = B.sa[ch * G + (s - 4)];
            else if (s >= 8 && s < 12) v = B.sb[ch * G + (s - 8)];
            else if (s >= 12 && s < 16) v = sigmoidf(B.an[ch * G + (s - 12)]);
            smem[CF::o_par + K * 32 * CF::par_stride + e] = v;
        }
        for (int e = tid; e < 32 * P; e += kPqThreads) smem[CF::o_w + K * 32 * P + e] = B.w[e];
    });
    __syncthreads();

    const bool save_state = io.save_state != 0 && io.save != nullptr;
    const long ntiles = (io.rows + kPqRows - 1) / kPqRows;
    // Software pipeline over the workgroup's tiles (as cemlp_pg.hpp): indices of tile t + 1 while tile t computes, its input rows
    // requested in front of tile t's row stores / atomics and written to LDS at the top of tile t + 1.
    auto load_idx = [&](int* dst, long tile_, int t) {     // threads 0 .. 15
        const long row = tile_ * kPqRows + t;
        const bool valid = tile_ < ntiles && row < io.rows;
        if constexpr (MODE == MODE_EDGE) {
            dst[t] = valid ? io.seg[0].ia[row] : -1;
            dst[16 + t] = valid ? io.seg[0].ib[row] : 0;
            dst[32 + t] = valid ? io.seg[1].ia[row] : 0;
        } else {
            dst[t] = valid ? t : -1;
            float sc = 1.0f;
            if constexpr (MODE == MODE_NODE) {
                if (valid && io.seg[1].deg) { const int dg = io.seg[1].deg[row]; sc = 1.0f / float(dg > 1 ? dg : 1); }
            }
            reinterpret_cast<float*>(dst)[48 + t] = sc;
        }
    };
    // plain program: input chunk m of row `row` (pieces 64 m .. 64 m + 63 of the I0-channel row; zero beyond it)
    auto plain_piece = [&](long row, int m, int e) {
        return 64 * m + e < 2 * I0 ? pg_ld4(io.seg[0].a + (size_t)row * (I0 * D) + 256 * m + 4 * e) : f4{0.f, 0.f, 0.f, 0.f};
    };
    f4 pre_a[NPRE], pre_b[NPRE], pre_x[PLAIN && NCH0 == 3 ? NPRE : NPA];
    auto issue_rows = [&](const int* idx, long tile_, int t) {
        const float* sc_ = reinterpret_cast<const float*>(idx) + 48;
        // the index reads of all rows first (one LDS round trip), then the scalar row pointers and the loads
        int ia[NPRE], ib[NPRE];
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int rr = pq_row_of(t, i);
            ia[i] = idx[rr];
            ib[i] = MODE == MODE_EDGE ? idx[16 + rr] : 0;
        }
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            ia[i] = __builtin_amdgcn_readfirstlane(ia[i]);
            ib[i] = __builtin_amdgcn_readfirstlane(ib[i]);
        }
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int rr = pq_row_of(t, i), e = t & 63;
            pre_a[i] = pre_b[i] = f4{0.f, 0.f, 0.f, 0.f};
            if constexpr (PLAIN && NCH0 == 3) pre_x[i] = f4{0.f, 0.f, 0.f, 0.f};
            if (ia[i] >= 0) {
                if constexpr (MODE == MODE_EDGE) {
                    pre_a[i] = pg_ld4(io.seg[0].a + (size_t)ia[i] * ROW + 4 * e);
                    pre_b[i] = pg_ld4(io.seg[0].b + (size_t)ib[i] * ROW + 4 * e);
                } else if constexpr (PLAIN) {
                    pre_a[i] = plain_piece(tile_ * kPqRows + rr, 0, e);
                    if constexpr (NCH0 >= 2) pre_b[i] = plain_piece(tile_ * kPqRows + rr, 1, e);
                    if constexpr (NCH0 == 3) pre_x[i] = plain_piece(tile_ * kPqRows + rr, 2, e);
                } else {
                    pre_a[i] = pg_ld4(io.seg[0].a + (size_t)(tile_ * kPqRows + rr) * ROW + 4 * e);
                    pre_b[i] = pg_ld4(io.seg[1].a + (size_t)(tile_ * kPqRows + rr) * ROW + 4 * e) * sc_[rr];
                }
            }
        }
        if constexpr (!PLAIN) {
#pragma unroll
            for (int i = 0; i < NPA; ++i) {
                const int p = t + i * kPqThreads, rr = (p / PPA) & 15, e = p % PPA;
                pre_x[i] = f4{0.f, 0.f, 0.f, 0.f};
                if (p < kPqRows * PPA && idx[rr] >= 0 && e < NA * 2) {
                    if constexpr (MODE == MODE_EDGE) pre_x[i] = pg_ld4(io.seg[1].a + (size_t)idx[32 + rr] * (NA * D) + 4 * e);
                    else pre_x[i] = pg_ld4(io.seg[2].a + (size_t)(tile_ * kPqRows + rr) * (NA * D) + 4 * e);
                }
            }
        }
    };
    if (tid < kPqRows) load_idx(sidx, blockIdx.x, tid);
    __syncthreads();
    issue_rows(sidx, blockIdx.x, tid);
    // weight fragments of the first MIX phase of a tile (requested at the end of the previous tile, in front of its stores / atomics)
    // (chunk m of the plain program: CF::nst(0, m) k-steps - the last chunk may be narrower)
    constexpr int NST1 = PLAIN ? (NCH0 >= 2 ? CF::nst(0, 1) : 1) : (MODE == MODE_NODE ? NST : 1);
    constexpr int NSTX = PLAIN ? (NCH0 == 3 ? CF::nst(0, 2) : 1) : CF::NSTA;
    PqA<PLAIN ? CF::nst(0, 0) : NST> aW0;
    PqA<NST1> aW0g;      // node program: the aggregate's chunk; plain: chunk 1
    PqA<NSTX> aW0x;      // attribute chunk; plain: chunk 2
    auto load_w0 = [&]() {
        pq_load_a(aW0, tabs + CF::toff(0, 0), lane, mp, mot, 2);
        if constexpr (MODE == MODE_NODE || (PLAIN && NCH0 >= 2)) pq_load_a(aW0g, tabs + CF::toff(0, 1), lane, mp, mot, 2);
        if constexpr (!PLAIN || NCH0 == 3) pq_load_a(aW0x, tabs + CF::toff(0, CF::NCH0 - 1), lane, mp, mot, 2);
    };
    load_w0();
    stamp(0);
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long row0 = tile * kPqRows;
        const bool rvalid = row0 + r < io.rows;
        // ---- block-0 input chunks (requested during the previous tile) -> LDS
        {
#pragma unroll
            for (int i = 0; i < NPRE; ++i) {
                const int rr = pq_row_of(tid, i), e = tid & 63;
                if constexpr (MODE == MODE_EDGE) {
                    pg_st4(bufA + pq_off(e >> 1, rr, e & 1), pre_a[i] - pre_b[i]);
                } else {
                    pg_st4(bufA + pq_off(e >> 1, rr, e & 1), pre_a[i]);
                    if constexpr (!PLAIN || NCH0 >= 2) pg_st4(bufB + pq_off(e >> 1, rr, e & 1), pre_b[i]);
                }
            }
            if constexpr (!PLAIN) {
                float* const bufX = MODE == MODE_EDGE ? bufB : bufE;
#pragma unroll
                for (int i = 0; i < NPA; ++i) {
                    const int p = tid + i * kPqThreads, rr = (p / PPA) & 15, e = p % PPA;
                    if (p < kPqRows * PPA) pg_st4(bufX + pq_off(e >> 1, rr, e & 1), pre_x[i]);
                }
            }
            if (tid < kPqRows) load_idx(sidx_n, tile + gridDim.x, tid);
        }
        __syncthreads();
        stamp(1);

        PqA<NST> aW1;     // block 1's W1 fragments: requested in front of block 0's last ROW phase
        static_for<0, NBLK>([&](auto kk) {
            constexpr int K = decltype(kk)::value;
            const float* parb = smem + CF::o_par + K * 32 * CF::par_stride;
            const float* wb = smem + CF::o_w + K * 32 * P;
            // ---- MIX: y = W1 x -> A
            {
                f4 acc[4];
                pq_zero(acc);
                if constexpr (K == 0) {
                    pq_mix_run(acc, bufA, aW0, lane, mp);
                    if constexpr (MODE == MODE_EDGE) {
                        pq_mix_run(acc, bufB, aW0x, lane, mp);
                    } else if constexpr (PLAIN) {
                        if constexpr (NCH0 >= 2) pq_mix_run(acc, bufB, aW0g, lane, mp);
                        if constexpr (NCH0 == 3) {   // the third chunk follows through B
                            __syncthreads();
#pragma unroll
                            for (int i = 0; i < NPRE; ++i) {
                                const int rr = pq_row_of(tid, i), e = tid & 63;
                                pg_st4(bufB + pq_off(e >> 1, rr, e & 1), pre_x[i]);
                            }
                            __syncthreads();
                            pq_mix_run(acc, bufB, aW0x, lane, mp);
                        }
                    } else {
                        pq_mix_run(acc, bufB, aW0g, lane, mp);
                        pq_mix_run(acc, bufE, aW0x, lane, mp);
                    }
                    __syncthreads();          // the other output tile's wave reads the same piece of A
                } else {
                    pq_mix_run<NST>(acc, bufB, aW1, lane, mp);
                    // the block-1 input rows leave for the backward (coalesced, while the MFMAs run)
                    if (io.save) {
#pragma unroll
                        for (int i = 0; i < NPRE; ++i) {
                            const int rr = pq_row_of(tid, i), e = tid & 63;
                            if (row0 + rr < io.rows) pg_st4(io.save + (size_t)(row0 + rr) * ROW + 4 * e, pg_ld4(bufB + pq_off(e >> 1, rr, e & 1)));
                        }
                    }
                }
                pq_write_d(bufA, acc, lane, mp, mot);
            }
            stamp(2 + 6 * K);
            __syncthreads();
            stamp(3 + 6 * K);
            // ---- ROW: bias, gates, z -> A (the next MIX phase's fragments first: in front of the state stores)
            constexpr int mR = CF::nmat(K) - 2;
            PqA<NST> aR, aL;
            pq_load_a<NST>(aR, tabs + CF::toff(K, mR), lane, mp, mot, 2);
            pq_load_a<NST>(aL, tabs + CF::toff(K, mR + 1), lane, mp, mot, 2);
            float z[2][8];
            static_for<0, 2>([&](auto mm) {
                constexpr int mv = decltype(mm)::value;
                const int c = cq + 16 * mv;
                const float* par = parb + c * CF::par_stride;
                float y[8];
                pq_ld8(y, bufA, r, c);
                y[0] += par[0];
                if (save_state && rvalid)
                    pq_store_state(io.save + pq_state_region<ROW, CF::NBLK>(io.rows, 1, K) + pq_state_off(tile, wave, lane, mv), y);
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
                    const float gate = sigmoidf(__builtin_fmaf(par[4 + g], u, par[8 + g]));
#pragma unroll
                    for (int t = 0; t < nd; ++t) z[mv][d0 + t] = gate * y[d0 + t];
                });
                pq_st8(bufA, r, c, z[mv]);
            });
            stamp(4 + 6 * K);
            __syncthreads();
            // ---- MIX: R = WR z -> B, L = WL z -> A (in place: behind a barrier)
            {
                f4 accR[4], accL[4];
                pq_zero(accR);
                pq_zero(accL);
                pq_mix_run2<NST>(accR, accL, bufA, aR, aL, lane, mp);
                pq_write_d(bufB, accR, lane, mp, mot);
                __syncthreads();
                pq_write_d(bufA, accL, lane, mp, mot);
            }
            stamp(5 + 6 * K);
            __syncthreads();
            stamp(3 + 6 * K);
            // ---- ROW: normalisation, geometric product, layer norm (block 0: block 1's W1 fragments first)
            if constexpr (K == 0 && NBLK > 1) pq_load_a<NST>(aW1, tabs + CF::toff(1, 0), lane, mp, mot, 2);
            float s[2][8];
            float nl[2];
            static_for<0, 2>([&](auto mm) {
                constexpr int mv = decltype(mm)::value;
                const int c = cq + 16 * mv;
                const float* par = parb + c * CF::par_stride;
                float R[8];
                pq_ld8(R, bufB, r, c);
                pq_ld8(s[mv], bufA, r, c);     // s accumulates: linear_left output + product
                s[mv][0] += par[1];
                if (save_state && rvalid)
                    pq_store_state(io.save + pq_state_region<ROW, CF::NBLK>(io.rows, 2, K) + pq_state_off(tile, wave, lane, mv), R);
                static_for<0, G>([&](auto g) {
                    constexpr int d0 = ALG::gstart(g), nd = ALG::gsize(g);
                    float qq = 0.f;
                    static_for<0, nd>([&](auto t) {
                        constexpr int d = d0 + decltype(t)::value;
                        qq = __builtin_fmaf(qsf<ALG, d> * R[d], R[d], qq);
                    });
                    const float m = __builtin_fmaf(par[12 + g], sqrt_pos(sqrt_pos(__builtin_fmaf(qq, qq, kSmooth))) - 1.0f, 1.0f);
                    const float inv = fast_rcp(m + kEps);
#pragma unroll
                    for (int t = 0; t < nd; ++t) R[d0 + t] *= inv;
                });
                pq_weighted_gp<ALG>(s[mv], z[mv], R, wb + c * P);
                float qs = 0.f;
                static_for<0, 8>([&](auto dd) {
                    constexpr int d = decltype(dd)::value;
                    s[mv][d] *= kInvSqrt2;
                    qs = __builtin_fmaf(qsf<ALG, d> * s[mv][d], s[mv][d], qs);
                });
                if (save_state && rvalid)
                    pq_store_state(io.save + pq_state_region<ROW, CF::NBLK>(io.rows, 0, K) + pq_state_off(tile, wave, lane, mv), s[mv]);
                nl[mv] = sqrt_pos(sqrt_pos(__builtin_fmaf(qs, qs, kSmooth)));
            });
            {
                const float wsum = pq_sum_q(nl[0] + nl[1]);
                if (lane < 16) lnx[wave * 16 + lane] = wsum;
            }
            stamp(6 + 6 * K);
            __syncthreads();
            const float tot = (lnx[r] + lnx[16 + r]) + (lnx[32 + r] + lnx[48 + r]);
            const float invM = fast_rcp(__builtin_fmaf(tot, 1.0f / float(C), kEps));
            static_for<0, 2>([&](auto mm) {
                constexpr int mv = decltype(mm)::value;
                const int c = cq + 16 * mv;
                const float kf = parb[c * CF::par_stride + 2] * invM;
                float out[8];
#pragma unroll
                for (int d = 0; d < 8; ++d) out[d] = kf * s[mv][d];
                // the last block: rows -> A; block 0 of two: the block-1 input -> B (R has been read by its own lane only)
                pq_st8(K == NBLK - 1 ? bufA : bufB, r, c, out);
            });
            __syncthreads();
            stamp(7 + 6 * K);
        });

        // ---- rows leave through A; the next tile's input rows and first weight fragments are requested first
        issue_rows(sidx_n, tile + gridDim.x, tid);
        load_w0();
        if constexpr (MODE == MODE_EDGE) {
            if (io.row_store) {   // deterministic mode: message rows to the [E, C, D] table in sorted edge order
#pragma unroll
                for (int i = 0; i < NPRE; ++i) {
                    const int rr = pq_row_of(tid, i), e = tid & 63;
                    if (row0 + rr < io.rows) pg_st4(io.agg + (size_t)(row0 + rr) * ROW + 4 * e, pg_ld4(bufA + pq_off(e >> 1, rr, e & 1)));
                }
            } else {
                // one atomic per 256 bytes of a target row; equal consecutive targets (the rows are sorted by target) are summed first
                static_assert(ROW == kPqThreads, "one column per thread");
                const int ch = tid >> 3, d = tid & 7;
                float acc = 0.f;
                int tg[kPqRows];
#pragma unroll
                for (int rr = 0; rr < kPqRows; ++rr) tg[rr] = sidx[rr];
#pragma unroll
                for (int rr = 0; rr < kPqRows; ++rr) tg[rr] = __builtin_amdgcn_readfirstlane(tg[rr]);
                int cur = tg[0];
#pragma unroll
                for (int rr = 0; rr < kPqRows; ++rr) {
                    const int t_ = tg[rr];
                    if (t_ != cur) {
                        if (cur >= 0) atomicAdd(io.agg + (size_t)cur * ROW + tid, acc);
                        cur = t_;
                        acc = 0.f;
                    }
                    acc += bufA[pq_off(ch, rr, d >> 2) + (d & 3)];
                }
                if (cur >= 0) atomicAdd(io.agg + (size_t)cur * ROW + tid, acc);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NPRE; ++i) {
                const int rr = pq_row_of(tid, i), e = tid & 63;
                if (row0 + rr < io.rows) {
                    f4 v = pg_ld4(bufA + pq_off(e >> 1, rr, e & 1));
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
// backward: one launch per block (K = 1, then K = 0; d/d(block-1 input) travels as rows through io.plw_g1), on the state the
// forward saved. Per 16-row tile (phases as cemlp_pg.hpp):
//   d/d(out) -> A | ROW: z = gate(y) y -> A, layer-norm backward -> ggp -> B | MIX: gz = WL^T ggp, d/dWL += ggp^T z -> gz over ggp in B
//   | ROW: geometric product + normalisation backward -> gR -> B | MIX: WR^T gR, d/dWR += gR^T z | ROW: MVSiLU backward -> gy -> B,
//   block input -> A (+ E) | MIX: d/dW1 += gy^T x, gx = W1^T gy -> rows out.
// Weight gradients: contraction over the 16 rows, wave w owns tile (ot = w >> 1, ct = w & 1) of every 32 x 32 matrix, all
// grades (16 accumulator registers per matrix, persistent over the launch); the attribute chunk (one c-tile): wave w owns
// (ot = w >> 1, piece w & 1). Per-channel parameter gradients: summed over the 16 rows by the transposing butterfly
// (pg_rows_sum: a DPP row = the 16 rows of one channel), 3 registers per lane and channel.

struct PqCollect {
    float buf[16];
    template <int IDX>
    CSMPN_DEV void add(float v, float (&small)[3], int l16) {
        buf[IDX % 16] = v;
        if constexpr (IDX % 16 == 15) small[IDX / 16] += pg_rows_sum(buf, l16);
    }
};

// geometric product backward, 8 blades in the lane, two passes (as pg_gp_bwd_z / _r)
template <class ALG>
CSMPN_DEV void pq_gp_bwd_z(const float (&ggp)[8], const float (&z)[8], const float (&rf)[8], float (&gz)[8], const float* wrow,
                           PqCollect& col, float (&small)[3], int l16) {
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
CSMPN_DEV void pq_gp_bwd_r(const float (&ggp)[8], const float (&z)[8], float (&gr)[8], const float* wrow) {
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

// d/dW tile (ot, ct) += G^T X over the 16 rows, piece PIECE: acc[grade]. G: gradient tile (its channel slots are the rows of the
// matrix), X: operand tile, both in LDS, lane = (channel i, rows 4 s + k).
template <int PIECE>
CSMPN_DEV void pq_wgrad_piece(f4 (&acc)[4], const float* bufG, const float* bufX, int ot, int ct, int lane, int xmask) {
    const int i = lane & 15, k = lane >> 4;
    f4 a[4], b[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        a[s] = pg_ld4(bufG + pq_off(16 * ot + i, 4 * s + k, PIECE));
        b[s] = pg_ld4(bufX + pq_off(16 * ct + (i & xmask), 4 * s + k, PIECE));
    }
    static_for<0, 4>([&](auto bb) {
        constexpr int d = 4 * PIECE + decltype(bb)::value;
        constexpr int g = (d >= 1) + (d >= 4) + (d >= 7);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[g] = mfma16(a[s][decltype(bb)::value], b[s][decltype(bb)::value], acc[g]);
    });
}
CSMPN_DEV void pq_wgrad(f4 (&acc)[4], const float* bufG, const float* bufX, int wave, int lane) {
    pq_wgrad_piece<0>(acc, bufG, bufX, wave >> 1, wave & 1, lane, 15);
    pq_wgrad_piece<1>(acc, bufG, bufX, wave >> 1, wave & 1, lane, 15);
}
// ... a one-c-tile operand (the attribute chunk, 8 channel slots in E: columns 8 .. 15 of the tile repeat 0 .. 7 and are dropped by
// the reduction): wave = (ot, piece)
CSMPN_DEV void pq_wgrad1(f4 (&acc)[4], const float* bufG, const float* bufX, int wave, int lane) {
    if (wave & 1) pq_wgrad_piece<1>(acc, bufG, bufX, wave >> 1, 0, lane, 7);
    else pq_wgrad_piece<0>(acc, bufG, bufX, wave >> 1, 0, lane, 7);
}
// slice store: tile (ot, ct) of the mat at `base` ([grade][ot][ct][lane][4], NCT c-tiles), grades G0 .. G0 + NG - 1
template <int NCT>
CSMPN_DEV void pq_store_unit(float* base, const f4 (&acc)[4], int ot, int ct, int lane, int g0, int ng) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
        if (g >= g0 && g < g0 + ng) pg_st4(base + (((g * 2 + ot) * NCT + ct) * 64 + lane) * 4, acc[g]);
}

#define PQ_PHASE_IDS()                                                                                              \
    const int tid = pg_tid();                                                                                       \
    const int wave = tid >> 6, lane = tid & 63, r = tid & 15, cq = tid >> 4, l16 = tid & 15;                        \
    const int mp = wave & 1, mot = wave >> 1;                                                                       \
    const bool live = row0 + r < io.rows;                                                                           \
    (void)l16; (void)live; (void)r; (void)cq; (void)mp; (void)mot; (void)lane

// grads += sum over the workgroups' slices, fixed order (as pg_reduce_kernel: 64 elements per workgroup, four slice groups, compensated)
template <class ALG, class CF, int K>
CSMPN_DEV void pq_reduce_body(const DevCemlp& Cd, const float* part, int nslices, int block) {
    constexpr int C = CF::C, G = CF::G, P = CF::P, SF = CF::slice_floats(K), SW = CF::slice_w(K);
    __shared__ float red[4][64];
    const int j = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int e = block * 64 + j;
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
            else if (idx >= CF::s_gate && idx < CF::s_gate + 2 * G) { const int g = (idx - CF::s_gate) >> 1; if (idx & 1) B.gsb[ch * G + g] += s; else B.gsa[ch * G + g] += s; }
            else if (idx == CF::s_b1) { if (B.has_b1) B.gb1[ch] += s; }
            else if (idx == CF::s_la) B.gla[ch] += s;
            else if (idx == CF::s_bL) B.gbL[ch] += s;
        }
    }
}
template <class ALG, class CF, int K>
__global__ void __launch_bounds__(256) pq_reduce_kernel(const DevCemlp Cd, const float* part, int nslices) {
    pq_reduce_body<ALG, CF, K>(Cd, part, nslices, blockIdx.x);
}

// Block 1's slices are summed by EXTRA workgroups of the block-0 launch (the first gridDim - aux.groups: the block-1 launch has finished -
// stream order - and its slices live in their own region): the memory-bound sums run under the block-0 tiles instead of in a
// launch of their own between the two (md17-sized launches: 15 us per edge stage).
struct PqAux {
    const float* part1;   // block 1's slices
    int groups;           // workgroups of this launch that run tiles
    int nslices1;         // slices behind part1
};
#ifndef PQ_BWD_WPE
#define PQ_BWD_WPE 3      // waves per SIMD the backward is compiled for (3: 168 registers; 2: 256)
#endif
template <class ALG, class CF, int K>
__global__ void __launch_bounds__(kPqThreads, PQ_BWD_WPE) cemlp_pq_bwd_kernel(const DevCemlp C_arg, const RowIO io_arg, const PqAux aux) {
    typedef const char __attribute__((address_space(4))) * KArgPtr;
    const KArgPtr ka = (KArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr size_t kIoOffset = (sizeof(DevCemlp) + alignof(RowIO) - 1) / alignof(RowIO) * alignof(RowIO);
    const DevCemlp& Cd = *(const DevCemlp*)(const char*)ka;
    const RowIO& io = *(const RowIO*)(const char*)(ka + kIoOffset);
    (void)C_arg; (void)io_arg;
    // (the summing workgroups come FIRST in the grid: dispatched in index order, they start with the launch)
    const int nred = (int)gridDim.x - aux.groups;
    if constexpr (K == 0) {
        if ((int)blockIdx.x < nred) {
            pq_reduce_body<ALG, CF, 1>(Cd, aux.part1, aux.nslices1, (int)blockIdx.x);
            return;
        }
    }
    const int ngroups = aux.groups;                 // workgroups that walk the tiles
    const int group = (int)blockIdx.x - nred;       // ... and this one's index among them
    constexpr int C = CF::C, MODE = CF::MODE, NA = CF::NA, D = 8, G = 4, P = CF::P, ROW = CF::ROW, NST = CF::NST;
    constexpr bool PLAIN = CF::PLAIN, LAST = K == CF::NBLK - 1;     // LAST: d/d(out) comes from the caller, not from the hand-over rows
    constexpr int I0 = CF::I0, NCH0 = CF::NCH0;
    constexpr int PPR = C * 2;
    constexpr int NM = CF::nmat(K), mR = NM - 2, mL = NM - 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const bufA = smem;
    float* const bufB = smem + kPqBuf;
    float* const bufE = smem + CF::b_E;
    float* const ln1 = smem + CF::b_ln;
    float* const ln2 = ln1 + kPqWaves * kPqRows;
    int* sidx = reinterpret_cast<int*>(smem + CF::b_idx);
    int* sidx_n = sidx + 64;
    constexpr int NPRE = PPR * kPqRows / kPqThreads;               // 16-byte pieces of a C-channel tile per thread
    static_assert(NPRE * kPqThreads == PPR * kPqRows && PPR == 64, "whole pieces per thread, one row per wave and piece index (pq_row_of)");
    const f4* tabs = reinterpret_cast<const f4*>(io.plw_tabs);
    PgStamp stamp(0);
    {
        const DevBlock& B = Cd.b[K];
        for (int e = threadIdx.x; e < 32 * CF::par_stride; e += kPqThreads) {
            const int ch = e / CF::par_stride, s_ = e % CF::par_stride;
            float v = 0.f;
            if (s_ == 0) v = B.has_b1 ? B.b1[ch] : 0.f;
            else if (s_ == 1) v = B.bL[ch];
            else if (s_ == 2) v = B.la[ch];
            else if (s_ >= 4 && s_ < 8) v = B.sa[ch * G + (s_ - 4)];
            else if (s_ >= 8 && s_ < 12) v = B.sb[ch * G + (s_ - 8)];
            else if (s_ >= 12 && s_ < 16) v = sigmoidf(B.an[ch * G + (s_ - 12)]);
            smem[CF::b_par + e] = v;
        }
        for (int e = threadIdx.x; e < 32 * P; e += kPqThreads) smem[CF::b_w + e] = B.w[e];
    }
    const float* const parb = smem + CF::b_par;
    const float* const wb = smem + CF::b_w;
    // persistent sums
    f4 accL[4], accR[4], accW0[4], accW1[4], accW2[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) accL[g] = accR[g] = accW0[g] = accW1[g] = accW2[g] = f4{0.f, 0.f, 0.f, 0.f};
    float small[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    const long ntiles = (io.rows + kPqRows - 1) / kPqRows;
    auto load_idx = [&](int* dst, long tile_, int t) {     // threads 0 .. 15
        const long row = tile_ * kPqRows + t;
        const bool valid = tile_ < ntiles && row < io.rows;
        if constexpr (MODE == MODE_EDGE) {
            dst[t] = valid ? io.seg[0].ia[row] : -1;
            dst[16 + t] = valid ? io.seg[0].ib[row] : 0;
            dst[32 + t] = valid ? io.seg[1].ia[row] : 0;
        } else {
            dst[t] = valid ? t : -1;
            float sc = 1.0f;
            if constexpr (MODE == MODE_NODE) {
                if (valid && io.seg[1].deg) { const int dg = io.seg[1].deg[row]; sc = 1.0f / float(dg > 1 ? dg : 1); }
            }
            reinterpret_cast<float*>(dst)[48 + t] = sc;
        }
    };
    // plain program: input chunk m of row `row` (pieces 64 m .. 64 m + 63 of the I0-channel row; zero beyond it)
    auto plain_piece = [&](long row, int m, int e) {
        return 64 * m + e < 2 * I0 ? pg_ld4(io.seg[0].a + (size_t)row * (I0 * D) + 256 * m + 4 * e) : f4{0.f, 0.f, 0.f, 0.f};
    };
    f4 pre[NPRE];
    auto issue_gout = [&](const int* idx, long tile_, int t) {
        int ia[NPRE];
#pragma unroll
        for (int i = 0; i < NPRE; ++i) ia[i] = idx[pq_row_of(t, i)];
#pragma unroll
        for (int i = 0; i < NPRE; ++i) ia[i] = __builtin_amdgcn_readfirstlane(ia[i]);
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int rr = pq_row_of(t, i), e = t & 63;
            pre[i] = f4{0.f, 0.f, 0.f, 0.f};
            if (ia[i] >= 0) {
                if constexpr (LAST) {
                    const size_t grow = MODE == MODE_EDGE ? (size_t)ia[i] : (size_t)(tile_ * kPqRows + rr);
                    pre[i] = pg_ld4(io.gy + grow * ROW + 4 * e);
                } else {
                    pre[i] = pg_ld4(io.plw_g1 + (size_t)(tile_ * kPqRows + rr) * ROW + 4 * e);
                }
            }
        }
    };
    if (threadIdx.x < kPqRows) load_idx(sidx, group, threadIdx.x);
    __syncthreads();
    issue_gout(sidx, group, threadIdx.x);
    stamp(0);

    for (long tile = group; tile < ntiles; tile += ngroups) {
        const long row0 = tile * kPqRows;
        const float* const sscale = reinterpret_cast<const float*>(sidx) + 48;
        float s_st[2][8], y_st[2][8];     // state rows of the first ROW phase, requested in front of the staging barrier
        {
            PQ_PHASE_IDS();
#pragma unroll
            for (int mv = 0; mv < 2; ++mv) {
#pragma unroll
                for (int d = 0; d < 8; ++d) { s_st[mv][d] = 0.f; y_st[mv][d] = 0.f; }
                if (live) {
                    pq_load_state(s_st[mv], io.saved + pq_state_region<ROW, CF::NBLK>(io.rows, 0, K) + pq_state_off(tile, wave, lane, mv));
                    pq_load_state(y_st[mv], io.saved + pq_state_region<ROW, CF::NBLK>(io.rows, 1, K) + pq_state_off(tile, wave, lane, mv));
                }
            }
            // ---- d/d(block output) rows (requested during the previous tile) -> A
#pragma unroll
            for (int i = 0; i < NPRE; ++i) {
                const int rr = pq_row_of(tid, i), e = tid & 63;
                pg_st4(bufA + pq_off(e >> 1, rr, e & 1), pre[i]);
            }
            if (tid < kPqRows) load_idx(sidx_n, tile + ngroups, tid);
        }
        __syncthreads();
        stamp(1);
        // ---- ROW: z -> A, layer-norm backward -> ggp -> B
        float ggp[2][8];
        float g_la[2], g_bL[2];
        PqA<NST> aT;                    // weight fragments of the next MIX phase, requested one ROW phase ahead
        {
            PQ_PHASE_IDS();
            // (fragments loaded at their MIX phase: the prefetch cost registers at three waves per SIMD and bought nothing)
            float qs[2], dot[2], nl[2];
            static_for<0, 2>([&](auto mm) {
                constexpr int mv = decltype(mm)::value;
                const int c = cq + 16 * mv;
                const float* par = parb + c * CF::par_stride;
                float (&y)[8] = y_st[mv];
                float (&s)[8] = s_st[mv];
                pq_ld8(ggp[mv], bufA, r, c);     // d/d(out)
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
                    const float gate = sigmoidf(__builtin_fmaf(par[4 + g], u, par[8 + g]));
#pragma unroll
                    for (int t = 0; t < nd; ++t) y[d0 + t] *= gate;
                });
                pq_st8(bufA, r, c, y);       // z (zero beyond the tile's rows)
                float q_ = 0.f, d_ = 0.f;
                static_for<0, 8>([&](auto dd) {
                    constexpr int d = decltype(dd)::value;
                    q_ = __builtin_fmaf(qsf<ALG, d> * s[d], s[d], q_);
                    d_ = __builtin_fmaf(ggp[mv][d], s[d], d_);
                });
                qs[mv] = q_; dot[mv] = d_;
                nl[mv] = sqrt_pos(sqrt_pos(__builtin_fmaf(q_, q_, kSmooth)));
            });
            {
                const float la0 = parb[cq * CF::par_stride + 2], la1 = parb[(cq + 16) * CF::par_stride + 2];
                const float w1 = pq_sum_q(nl[0] + nl[1]);
                const float w2 = pq_sum_q(live ? __builtin_fmaf(la0, dot[0], la1 * dot[1]) : 0.f);
                if (lane < 16) { ln1[wave * 16 + lane] = w1; ln2[wave * 16 + lane] = w2; }
            }
            __syncthreads();
            const float tot = (ln1[r] + ln1[16 + r]) + (ln1[32 + r] + ln1[48 + r]);
            const float totd = (ln2[r] + ln2[16 + r]) + (ln2[32 + r] + ln2[48 + r]);
            const float invMn = fast_rcp(__builtin_fmaf(tot, 1.0f / float(C), kEps));
            const float gMn = -totd * invMn * invMn * (1.0f / float(C));
            static_for<0, 2>([&](auto mm) {
                constexpr int mv = decltype(mm)::value;
                const int c = cq + 16 * mv;
                const float la = parb[c * CF::par_stride + 2];
                float (&s)[8] = s_st[mv];
                const float inl = fast_rcp(nl[mv]);
                const float gqs = gMn * (0.5f * qs[mv]) * (inl * inl * inl);
                const float k0 = la * invMn;
                static_for<0, 8>([&](auto dd) {
                    constexpr int d = decltype(dd)::value;
                    const float gs = __builtin_fmaf(k0, ggp[mv][d], gqs * (2.0f * qsf<ALG, d>) * s[d]);
                    ggp[mv][d] = live ? gs * kInvSqrt2 : 0.f;
                });
                g_la[mv] = live ? dot[mv] * invMn : 0.f;
                g_bL[mv] = ggp[mv][0];
                pq_st8(bufB, r, c, ggp[mv]);
            });
        }
        __syncthreads();
        stamp(2);
        // ---- MIX: gz = WL^T ggp (piece mp, c-tile mot), d/dWL tile += ggp^T z; then gz over ggp in B
        float R_st[2][8];               // the next ROW phase's state rows travel under the MFMAs
        {
            PQ_PHASE_IDS();
#pragma unroll
            for (int mv = 0; mv < 2; ++mv) {
#pragma unroll
                for (int d = 0; d < 8; ++d) R_st[mv][d] = 0.f;
                if (live) pq_load_state(R_st[mv], io.saved + pq_state_region<ROW, CF::NBLK>(io.rows, 2, K) + pq_state_off(tile, wave, lane, mv));
            }
            f4 acc[4];
            pq_zero(acc);
            pq_load_a<NST>(aT, tabs + CF::ttoff(K, mL), lane, mp, mot, 2);
            pq_mix_run<NST>(acc, bufB, aT, lane, mp);
            pq_wgrad(accL, bufB, bufA, wave, lane);
            __syncthreads();
            pq_write_d(bufB, acc, lane, mp, mot);
        }
        __syncthreads();
        stamp(3);
        // ---- ROW: geometric product + normalisation backward -> gR -> B; gz stays in registers
        float gz[2][8];
        PqCollect col;      // one buffer: a multivector's slots are flushed (16 at a time) before the next one starts
        {
            PQ_PHASE_IDS();
            // (fragments loaded at their MIX phase: the prefetch cost registers at three waves per SIMD and bought nothing)
            static_for<0, 2>([&](auto mm) {
                constexpr int mv = decltype(mm)::value;
                const int c = cq + 16 * mv;
                const float* par = parb + c * CF::par_stride;
                const float* wrow = wb + c * P;
                float z[8];
                float (&R)[8] = R_st[mv];
                pq_ld8(gz[mv], bufB, r, c);
                pq_ld8(z, bufA, r, c);
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
                    den[g] = __builtin_fmaf(par[12 + g], nu[g] - 1.0f, 1.0f) + kEps;
                    invden[g] = fast_rcp(den[g]);
#pragma unroll
                    for (int t = 0; t < nd; ++t) R[d0 + t] *= invden[g];    // R holds r = R / den from here on
                });
                float (&rf)[8] = R;
                pq_gp_bwd_z<ALG>(ggp[mv], z, rf, gz[mv], wrow, col, small[mv], l16);
                float gr[8];
#pragma unroll
                for (int d = 0; d < 8; ++d) gr[d] = 0.f;
                pq_gp_bwd_r<ALG>(ggp[mv], z, gr, wrow);
                // NormalizationLayer backward: gR (into gr), d/d(an)
                static_for<0, G>([&](auto g) {
                    constexpr int d0 = ALG::gstart(g), nd = ALG::gsize(g);
                    float gden = 0.f;
                    static_for<0, nd>([&](auto t) {
                        constexpr int d = d0 + decltype(t)::value;
                        gden = __builtin_fmaf(-gr[d], rf[d], gden);
                    });
                    gden *= invden[g];          // -sum gr R / den^2 with R = r den
                    const float sg = par[12 + g];
                    col.template add<CF::s_an + g>(gden * (nu[g] - 1.0f) * sg * (1.0f - sg), small[mv], l16);
                    const float inu = fast_rcp(nu[g]);
                    const float gq = (gden * sg) * (0.5f * qR[g]) * (inu * inu * inu);
                    static_for<0, nd>([&](auto t) {
                        constexpr int d = d0 + decltype(t)::value;
                        gr[d] = __builtin_fmaf(gr[d], invden[g], (gq * den[g]) * (2.0f * qsf<ALG, d>) * rf[d]);
                    });
                });
                static_for<CF::s_an + G, 32>([&](auto ii) { col.template add<decltype(ii)::value>(0.f, small[mv], l16); });
                pq_st8(bufB, r, c, gr);      // gR
                pq_pin8(gz[mv]);
            });
        }
        __syncthreads();
        stamp(4);
        // ---- MIX: WR^T gR, d/dWR tile += gR^T z; the result over gR in B
        float y2_st[2][8];
        {
            PQ_PHASE_IDS();
#pragma unroll
            for (int mv = 0; mv < 2; ++mv) {
#pragma unroll
                for (int d = 0; d < 8; ++d) y2_st[mv][d] = 0.f;
                if (live) pq_load_state(y2_st[mv], io.saved + pq_state_region<ROW, CF::NBLK>(io.rows, 1, K) + pq_state_off(tile, wave, lane, mv));
            }
            f4 acc[4];
            pq_zero(acc);
            pq_load_a<NST>(aT, tabs + CF::ttoff(K, mR), lane, mp, mot, 2);
            pq_mix_run<NST>(acc, bufB, aT, lane, mp);
            pq_wgrad(accR, bufB, bufA, wave, lane);
            __syncthreads();
            pq_write_d(bufB, acc, lane, mp, mot);
        }
        __syncthreads();
        stamp(5);
        // ---- block input -> A (+ E): z has been read for the last time. The rows are requested here and written behind the
        // MVSiLU backward
        f4 xa[NPRE], xb[K == 0 && MODE == MODE_EDGE ? NPRE : 1];
        {
            PQ_PHASE_IDS();
            // (fragments loaded at their MIX phase: the prefetch cost registers at three waves per SIMD and bought nothing)
            int ia[NPRE], ib[NPRE];
#pragma unroll
            for (int i = 0; i < NPRE; ++i) {
                const int rr = pq_row_of(tid, i);
                ia[i] = sidx[rr];
                ib[i] = (K == 0 && MODE == MODE_EDGE) ? sidx[16 + rr] : 0;
            }
#pragma unroll
            for (int i = 0; i < NPRE; ++i) {
                ia[i] = __builtin_amdgcn_readfirstlane(ia[i]);
                ib[i] = __builtin_amdgcn_readfirstlane(ib[i]);
            }
#pragma unroll
            for (int i = 0; i < NPRE; ++i) {
                const int rr = pq_row_of(tid, i), e = tid & 63;
                xa[i] = f4{0.f, 0.f, 0.f, 0.f};
                if constexpr (K == 0 && MODE == MODE_EDGE) xb[i] = f4{0.f, 0.f, 0.f, 0.f};
                if (ia[i] >= 0) {
                    if constexpr (K == 1) xa[i] = pg_ld4(io.saved + (size_t)(row0 + rr) * ROW + 4 * e);
                    else if constexpr (MODE == MODE_EDGE) {
                        xa[i] = pg_ld4(io.seg[0].a + (size_t)ia[i] * ROW + 4 * e);
                        xb[i] = pg_ld4(io.seg[0].b + (size_t)ib[i] * ROW + 4 * e);
                    } else if constexpr (PLAIN) xa[i] = plain_piece(row0 + rr, 0, e);
                    else xa[i] = pg_ld4(io.seg[0].a + (size_t)(row0 + rr) * ROW + 4 * e);
                }
            }
            if constexpr (K == 0 && !PLAIN) {
                constexpr int PPA = 8 * 2;   // the 8 attribute slots of E (the rows-contracting MFMA reads all of them)
                static_assert(kPqRows * PPA == kPqThreads, "one attribute piece per thread");
                const int rr = tid / PPA, e = tid % PPA;
                f4 v = f4{0.f, 0.f, 0.f, 0.f};
                if (sidx[rr] >= 0 && e < NA * 2) {
                    if constexpr (MODE == MODE_EDGE) v = pg_ld4(io.seg[1].a + (size_t)sidx[32 + rr] * (NA * D) + 4 * e);
                    else v = pg_ld4(io.seg[2].a + (size_t)(row0 + rr) * (NA * D) + 4 * e);
                }
                pg_st4(bufE + pq_off(e >> 1, rr, e & 1), v);
            }
        }
        // ---- ROW: MVSiLU backward -> gy -> B
        {
            PQ_PHASE_IDS();
            static_for<0, 2>([&](auto mm) {
                constexpr int mv = decltype(mm)::value;
                const int c = cq + 16 * mv;
                const float* par = parb + c * CF::par_stride;
                float t_[8];
                float (&y)[8] = y2_st[mv];
                float (&g_)[8] = gz[mv];
                pq_ld8(t_, bufB, r, c);
#pragma unroll
                for (int d = 0; d < 8; ++d) g_[d] += t_[d];
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
                    const float gate = sigmoidf(__builtin_fmaf(par[4 + g], u, par[8 + g]));
#pragma unroll
                    for (int t = 0; t < nd; ++t) ggate = __builtin_fmaf(g_[d0 + t], y[d0 + t], ggate);
                    const float gpre = ggate * gate * (1.0f - gate);
                    col.template add<CF::s_gate + 2 * g>(gpre * u, small[mv], l16);
                    col.template add<CF::s_gate + 2 * g + 1>(gpre, small[mv], l16);
                    const float gu = gpre * par[4 + g];
                    static_for<0, nd>([&](auto t) {
                        constexpr int d = d0 + decltype(t)::value;
                        float v = g_[d] * gate;
                        if constexpr (g == 0) v += gu;
                        else v = __builtin_fmaf(gu * (2.0f * qsf<ALG, d>), y[d], v);
                        g_[d] = live ? v : 0.f;     // gy
                    });
                });
                col.template add<CF::s_b1>(g_[0], small[mv], l16);
                col.template add<CF::s_la>(g_la[mv], small[mv], l16);
                col.template add<CF::s_bL>(g_bL[mv], small[mv], l16);
                static_for<CF::s_bL + 1, 48>([&](auto ii) { col.template add<decltype(ii)::value>(0.f, small[mv], l16); });
                pq_st8(bufB, r, c, g_);
            });
#pragma unroll
            for (int i = 0; i < NPRE; ++i) {
                const int rr = pq_row_of(tid, i), e = tid & 63;
                if constexpr (K == 0 && MODE == MODE_EDGE) xa[i] -= xb[i];
                pg_st4(bufA + pq_off(e >> 1, rr, e & 1), xa[i]);
            }
        }
        __syncthreads();
        stamp(6);
        // ---- MIX: d/dW1 += gy^T x, gx = W1^T gy
        if constexpr (K >= 1 || MODE == MODE_EDGE) {
            PQ_PHASE_IDS();
            f4 acc[4];
            pq_zero(acc);
            pq_load_a<NST>(aT, tabs + CF::ttoff(K, 0), lane, mp, mot, 2);
            pq_mix_run<NST>(acc, bufB, aT, lane, mp);
            pq_wgrad(accW0, bufB, bufA, wave, lane);
            f4 acca[4];
            bool want_a = false;
            if constexpr (K == 0) {
                pq_wgrad1(accW1, bufB, bufE, wave, lane);
                want_a = io.gx[1] != nullptr && mot == 0;
                if (want_a) {
                    pq_zero(acca);
                    pq_mix_acc<NST>(acca, bufB, tabs + CF::ttoff(0, 1), lane, mp, 0, 1);
                }
            }
            __syncthreads();
            pq_write_d(bufB, acc, lane, mp, mot);
            if constexpr (K == 0) { if (want_a) pq_write_d<8>(bufE, acca, lane, mp, 0); }
        } else if constexpr (PLAIN) {
            // plain program, block 0: input chunk 0 in A now, the others follow through A; every chunk's d/dx leaves as soon as it is there
            PQ_PHASE_IDS();
            static_for<0, NCH0>([&](auto mm) {
                constexpr int m = decltype(mm)::value;
                constexpr int NSTm = CF::nst(0, m);
                f4 accx[4];
                const bool want_x = io.gx[0] != nullptr;
                f4 nxt[NPRE];          // the next chunk's rows travel under the MFMAs
                if constexpr (m + 1 < NCH0) {
#pragma unroll
                    for (int i = 0; i < NPRE; ++i) {
                        const int rr = pq_row_of(tid, i), e = tid & 63;
                        nxt[i] = row0 + rr < io.rows ? plain_piece(row0 + rr, m + 1, e) : f4{0.f, 0.f, 0.f, 0.f};
                    }
                }
                if (want_x) {
                    pq_zero(accx);
                    PqA<NST> aX;
                    pq_load_a<NST>(aX, tabs + CF::ttoff(0, m), lane, mp, mot, 2);
                    pq_mix_run<NST>(accx, bufB, aX, lane, mp);
                }
                pq_wgrad(m == 0 ? accW0 : (m == 1 ? accW1 : accW2), bufB, bufA, wave, lane);
                __syncthreads();             // every wave is done with this chunk in A
                if (want_x) {
                    pq_write_d(bufA, accx, lane, mp, mot);
                    __syncthreads();
#pragma unroll
                    for (int i = 0; i < NPRE; ++i) {
                        const int rr = pq_row_of(tid, i), e = tid & 63;
                        if (row0 + rr < io.rows && 64 * m + e < 2 * I0)
                            pg_st4(io.gx[0] + (size_t)(row0 + rr) * (I0 * D) + 256 * m + 4 * e, pg_ld4(bufA + pq_off(e >> 1, rr, e & 1)));
                    }
                    if constexpr (m + 1 < NCH0) __syncthreads();
                }
                if constexpr (m + 1 < NCH0) {
#pragma unroll
                    for (int i = 0; i < NPRE; ++i) {
                        const int rr = pq_row_of(tid, i), e = tid & 63;
                        pg_st4(bufA + pq_off(e >> 1, rr, e & 1), nxt[i]);
                    }
                    __syncthreads();
                }
            });
        } else {
            // node program, block 0: h in A now; the aggregate follows through A, the attributes sit in E
            PQ_PHASE_IDS();
            f4 acch[4], accg[4];
            pq_zero(acch);
            pq_load_a<NST>(aT, tabs + CF::ttoff(K, 0), lane, mp, mot, 2);
            pq_mix_run<NST>(acch, bufB, aT, lane, mp);
            pq_wgrad(accW0, bufB, bufA, wave, lane);
            pq_wgrad1(accW2, bufB, bufE, wave, lane);
            f4 acca[4];
            const bool want_a = io.gx[2] != nullptr && mot == 0;
            if (want_a) {
                pq_zero(acca);
                pq_mix_acc<NST>(acca, bufB, tabs + CF::ttoff(0, 2), lane, mp, 0, 1);
            }
            __syncthreads();             // every wave is done with h in A
#pragma unroll
            for (int i = 0; i < NPRE; ++i) {
                const int rr = pq_row_of(tid, i), e = tid & 63;
                f4 v = f4{0.f, 0.f, 0.f, 0.f};
                if (pq_sidx(sidx, rr) >= 0) v = pg_ld4(io.seg[1].a + (size_t)(row0 + rr) * ROW + 4 * e) * sscale[rr];
                pg_st4(bufA + pq_off(e >> 1, rr, e & 1), v);
            }
            __syncthreads();
            pq_zero(accg);
            pq_mix_acc<NST>(accg, bufB, tabs + CF::ttoff(0, 1), lane, mp, mot, 2);
            pq_wgrad(accW1, bufB, bufA, wave, lane);
            __syncthreads();             // gy in B and the aggregate in A have been read for the last time
            pq_write_d(bufA, acch, lane, mp, mot);      // d/dh
            pq_write_d(bufB, accg, lane, mp, mot);      // d/d(scaled aggregate)
            if (want_a) pq_write_d<8>(bufE, acca, lane, mp, 0);
        }
        __syncthreads();
        stamp(7);
        // ---- rows out; the next tile's d/d(out) rows are requested first
        {
            PQ_PHASE_IDS();
            issue_gout(sidx_n, tile + ngroups, tid);
            if constexpr (K == 1) {
#pragma unroll
                for (int i = 0; i < NPRE; ++i) {
                    const int rr = pq_row_of(tid, i), e = tid & 63;
                    if (row0 + rr < io.rows) pg_st4(io.plw_g1 + (size_t)(row0 + rr) * ROW + 4 * e, pg_ld4(bufB + pq_off(e >> 1, rr, e & 1)));
                }
            } else if constexpr (PLAIN) {
                // (every chunk's d/dx rows left in the MIX phase above)
            } else if constexpr (MODE == MODE_EDGE) {
                if (io.gx[0]) {
                    if (io.row_store) {
#pragma unroll
                        for (int i = 0; i < NPRE; ++i) {
                            const int rr = pq_row_of(tid, i), e = tid & 63;
                            if (row0 + rr < io.rows) pg_st4(io.gx[0] + (size_t)(row0 + rr) * ROW + 4 * e, pg_ld4(bufB + pq_off(e >> 1, rr, e & 1)));
                        }
                    } else {
                        static_assert(ROW == kPqThreads, "one column per thread");
                        const int ch = tid >> 3, d = tid & 7;
                        float acc = 0.f;
                        int tg[kPqRows], ts[kPqRows];
#pragma unroll
                        for (int rr = 0; rr < kPqRows; ++rr) { tg[rr] = sidx[rr]; ts[rr] = sidx[16 + rr]; }
#pragma unroll
                        for (int rr = 0; rr < kPqRows; ++rr) { tg[rr] = __builtin_amdgcn_readfirstlane(tg[rr]); ts[rr] = __builtin_amdgcn_readfirstlane(ts[rr]); }
                        int cur = tg[0];
#pragma unroll
                        for (int rr = 0; rr < kPqRows; ++rr) {
                            const int t_ = tg[rr];
                            if (t_ != cur) {
                                if (cur >= 0) atomicAdd(io.gx[0] + (size_t)cur * ROW + tid, acc);
                                cur = t_;
                                acc = 0.f;
                            }
                            const float v = bufB[pq_off(ch, rr, d >> 2) + (d & 3)];
                            acc += v;
                            if (t_ >= 0) atomicAdd(io.gx[0] + (size_t)ts[rr] * ROW + tid, -v);
                        }
                        if (cur >= 0) atomicAdd(io.gx[0] + (size_t)cur * ROW + tid, acc);
                    }
                }
                if (io.gx[1]) {
                    const int rr = tid / (NA * 2), e = tid % (NA * 2);
                    if (tid < kPqRows * NA * 2 && row0 + rr < io.rows)
                        pg_st4(io.gx[1] + (size_t)sidx[32 + rr] * (NA * D) + 4 * e, pg_ld4(bufE + pq_off(e >> 1, rr, e & 1)));
                }
            } else {
#pragma unroll
                for (int i = 0; i < NPRE; ++i) {
                    const int rr = pq_row_of(tid, i), e = tid & 63;
                    if (row0 + rr < io.rows) {
                        if (io.gx[0]) {
                            f4 v = pg_ld4(bufA + pq_off(e >> 1, rr, e & 1));
                            if (io.resid_bwd) v += pg_ld4(io.gy + (size_t)(row0 + rr) * ROW + 4 * e);
                            pg_st4(io.gx[0] + (size_t)(row0 + rr) * ROW + 4 * e, v);
                        }
                        if (io.gx[1]) pg_st4(io.gx[1] + (size_t)(row0 + rr) * ROW + 4 * e, pg_ld4(bufB + pq_off(e >> 1, rr, e & 1)) * sscale[rr]);
                    }
                }
                if (io.gx[2]) {
                    const int rr = tid / (NA * 2), e = tid % (NA * 2);
                    if (tid < kPqRows * NA * 2 && row0 + rr < io.rows)
                        pg_st4(io.gx[2] + (size_t)(row0 + rr) * (NA * D) + 4 * e, pg_ld4(bufE + pq_off(e >> 1, rr, e & 1)));
                }
            }
        }
        __syncthreads();
        { int* t_ = sidx; sidx = sidx_n; sidx_n = t_; }
        stamp(8);
    }
    // ---- this workgroup's slice: every element has one owner
    float* slice = io.plw_part + (size_t)group * CF::slice_floats(K);
    {
        const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, cq = tid >> 4, l16 = tid & 15;
        const int ot = wave >> 1, ct = wave & 1;
        pq_store_unit<2>(slice + CF::woff(K, mL), accL, ot, ct, lane, 0, 4);
        pq_store_unit<2>(slice + CF::woff(K, mR), accR, ot, ct, lane, 0, 4);
        pq_store_unit<2>(slice + CF::woff(K, 0), accW0, ot, ct, lane, 0, 4);
        if constexpr (K == 0) {
            if constexpr (MODE == MODE_EDGE) {
                pq_store_unit<1>(slice + CF::woff(0, 1), accW1, ot, 0, lane, 2 * ct, 2);
            } else if constexpr (PLAIN) {
                if constexpr (NCH0 >= 2) pq_store_unit<2>(slice + CF::woff(0, 1), accW1, ot, ct, lane, 0, 4);
                if constexpr (NCH0 == 3) pq_store_unit<2>(slice + CF::woff(0, 2), accW2, ot, ct, lane, 0, 4);
            } else {
                pq_store_unit<2>(slice + CF::woff(0, 1), accW1, ot, ct, lane, 0, 4);
                pq_store_unit<1>(slice + CF::woff(0, 2), accW2, ot, 0, lane, 2 * ct, 2);
            }
        }
#pragma unroll
        for (int mv = 0; mv < 2; ++mv)
#pragma unroll
            for (int g = 0; g < 3; ++g) slice[CF::slice_w(K) + (cq + 16 * mv) * CF::kSmall + 16 * g + l16] = small[mv][g];
    }
    stamp(9);
    stamp.flush(io.stamps, threadIdx.x & 63);
}

}  // namespace csmpn
