// Row-per-lane-group ("rl") CEMLP kernels for narrow layers: every block has C = 4*NOG output
// channels (NOG = 2: 8 channels).
//
// Same arithmetic as cemlp_device.hpp (csmpn/models/cegnn_utils.py:34-155,287-338; SURVEY.md
// Appendix A), different mapping onto the hardware. Measured on MI355X (tools/mfma_probe*.hip,
// profiles/r02_probe*.log): fp32 MFMA and VALU FMA do NOT overlap (two waves of one SIMD running
// one each take the sum of both times), a wave64 VALU op issues every ~2.7-3.4 cycles, and
// v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 outer products per instruction, 256 MACs in ~9
// cycles) takes its B operand from the lane that receives the result. So:
//
//  * lane = (row, channel group og): a lane holds the 4 channels 4og..4og+3 of ONE row, all D
//    blades:  f4 t[D]  (t[d][i] = channel 4og+i). A wave covers R = 64/NOG rows. Gates, norms and
//    the sign-table geometric product are in-lane VALU work on 4-channel vectors with per-channel
//    parameters read from LDS as one f4; the only cross-lane traffic of the forward is the
//    exchange of the 4-channel pieces between the NOG lanes of a row in front of a dense mixing
//    (one DPP row rotate per value) and one DPP add for the LayerNorm channel mean.
//  * dense channel mixing: one 4x4x1 MFMA per (input channel, blade):
//    D_b[i][j] += A_b[i] * B_b[j] with block b = 4 rows of one channel group, B = x[c][d] of the
//    lane's row, A = W[4og + (lane & 3)][c][grade(d)] (one ds_read_b128 from the LDS copy of the
//    weight in its reference layout serves the 4 grades = all D blades). No padding of the channel
//    dimensions (the 16x16x4 tile wastes half of itself on 8 channels).
//  * sums over ROWS (all parameter gradients) are the remaining cross-lane work, and they stay in
//    REGISTERS for the whole launch (LDS float atomics measured ~8 cycles per active lane: the first
//    version of this file spent half of the backward in ds_add_f32): weight gradients go through a
//    per-wave LDS transposition ([channel][row] slices per blade) into 16x16x4 MFMAs with K = rows
//    whose accumulators persist across the wave's tiles; the small per-channel parameter gradients
//    are summed over the wave's rows by an LDS transposition of 8 f4 values at a time (write
//    [value][lane], read [lane & 7][8 source lanes], second hop over the 4 DPP rows) into 5 persistent
//    f4 accumulators per block. Every wave writes its sums once, at the end, to its own slice of a
//    global partial buffer; a second small kernel sums the slices in a fixed order (deterministic).
//  * every LDS address is (one of a few per-lane bases, RlGeo) + a compile-time immediate: the
//    kernels are fully unrolled and a loop-invariant address per access would otherwise be hoisted
//    out of the tile loop and spilled (round-2 first version: 200 spilled address registers).
//  * one wave per SIMD in the backward means every wait that directly follows its load is dead time; where the
//    scheduler (under ~500 live registers) produced such waits, the order is stated in the source: parameter
//    pointers as scalar values + batched value loads (rl_stage_store), four weight vectors in flight under the
//    previous group's MFMAs (rl_weight_pipeline), all input rows of a tile requested up front (PIN), the re-gather
//    for the weight gradient issued under the MVSiLU backward (early_loads). DESIGN.md 4.1.
//  * gathers are per-lane 16-byte loads of the lane's own rows (the NOG lanes of a row issue the same
//    addresses in the same instruction); scatters go through a per-wave LDS tile so that one atomic
//    instruction covers whole rows (segment-merged by target for the target-sorted edge list).
#pragma once
#include "cemlp_device.hpp"

namespace csmpn {

CSMPN_DEV f4 mfma4(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }
// Compiler barrier between the 16-byte vector stores into the staging tile and the float reads of the same LDS bytes (the
// two access types carry different alias information: cemlp_cm.hpp met the reordering this allows). Costs no instruction.
#define RL_LDS_ORDER() asm volatile("" ::: "memory")
CSMPN_DEV f4 ld4(const float* p) { return *reinterpret_cast<const f4*>(p); }

constexpr int kRlWaves = 4;        // waves per workgroup
constexpr int kRlMaxGroups = 256;  // workgroups of a backward launch (slices of the partial buffer)

template <int CTRL>
CSMPN_DEV int dpp_movi(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }

// ---------------------------------------------------------------------------------
// compile-time layout of the LDS (floats) for one kernel: NBLK blocks, block 0 has I0 inputs
template <class ALG, int NOG, int NBLK, int I0>
struct RlLay {
    static constexpr int C = 4 * NOG, D = ALG::D, G = ALG::G, P = ALG::P, R = 64 / NOG;
    static constexpr int IM = I0 > C ? I0 : C;
    // weight row strides (floats): an odd number of 16-byte slots (conflict-free broadcast reads);
    // WS1 for block 0's MVLinear [C][I0], WSC for every C x C matrix
    static constexpr int WS1 = 4 * (I0 | 1), WSC = 4 * (C | 1);
    static constexpr int wstride(int k) { return k == 0 ? WS1 : WSC; }
    // parameter store of block k: W1 [C][wstride(k)], WR, WL [C][WSC] in the reference's [o][c][g]
    // order (g padded to 4), then the per-channel parameters channel-fastest ([g][C]; w [p][C])
    static constexpr int store_total(int k) { return C * wstride(k) + 2 * C * WSC + C * (3 + 3 * G + P); }
    static constexpr int st_off(int k) { return k == 0 ? 0 : store_total(0); }
    static constexpr int store_all = store_total(0) + (NBLK > 1 ? store_total(1) : 0);
    static constexpr int o_W1(int k) { return st_off(k); }
    static constexpr int o_WR(int k) { return o_W1(k) + C * wstride(k); }
    static constexpr int o_WL(int k) { return o_WR(k) + C * WSC; }
    static constexpr int o_b1(int k) { return o_WL(k) + C * WSC; }
    static constexpr int o_bL(int k) { return o_b1(k) + C; }
    static constexpr int o_la(int k) { return o_bL(k) + C; }
    static constexpr int o_sa(int k) { return o_la(k) + C; }
    static constexpr int o_sb(int k) { return o_sa(k) + C * G; }
    static constexpr int o_sg(int k) { return o_sb(k) + C * G; }
    static constexpr int o_w(int k) { return o_sg(k) + C * G; }
    static constexpr int Iof(int k) { return k == 0 ? I0 : C; }
    // partial-buffer slice of one wave: per block [W1 [C][I][G] | WR [C][C][G] | WL | small], reference
    // layouts; small = b1 [C], sa [C][G], sb [C][G], w [C][P], an [C][G], bL [C], la [C]
    static constexpr int m_small = C * (3 + 3 * G + P);
    static constexpr int pWR(int k) { return C * Iof(k) * G; }
    static constexpr int pWL(int k) { return pWR(k) + C * C * G; }
    static constexpr int pS(int k) { return pWL(k) + C * C * G; }
    static constexpr int qb1 = 0, qsa = qb1 + C, qsb = qsa + C * G, qw = qsb + C * G, qan = qw + C * P, qbL = qan + C * G,
                         qla = qbL + C;
    static constexpr int part_blk(int k) { return pS(k) + m_small; }
    static constexpr int part_off(int k) { return k == 0 ? 0 : part_blk(0); }
    static constexpr int part_total = part_blk(0) + (NBLK > 1 ? part_blk(1) : 0);
    // small-parameter gradients in reduction order: la, bL, w[0..P), an[0..G), (sa[g], sb[g])..., b1
    static constexpr int n_red = 3 + 3 * G + P, n_chunk = (n_red + 7) / 8;
    // per-wave scratch: the scatter staging tile [R][C*D + 4] or the transposition slices
    static constexpr int RS = R + 4;                 // row stride of a transposition slice
    static constexpr int SS = C * D + 4;             // row stride of the staging tile
    static constexpr int stage_floats = R * SS;
    static constexpr int MA_RL = C / 8;              // A tiles (16 rows) of the linear_right | linear_left gradient
    static constexpr int b_off = 16 * MA_RL * RS;    // B slice behind the A rows
    static constexpr int slice_floats = b_off + (IM + 3) / 4 * 4 * RS;
    static constexpr int red_floats = (8 * 65 + 64) * 4;   // reduction: 8 values x 65 f4 slots, then 64 f4
    // the reduction slots hold pending values while the weight-gradient slices are in use: separate
    // regions; the staging tile (end of the tile pass) may alias both
    static constexpr int red_off = slice_floats;
    static constexpr int cmax(int a, int b) { return a > b ? a : b; }
    // ... and the image of one block's slice of the partial buffer at the end of the launch
    static constexpr int scratch1 = cmax(cmax(stage_floats, slice_floats + red_floats), cmax(part_blk(0), NBLK > 1 ? part_blk(1) : 0));
    // running totals of the small-parameter gradients: [block][chunk][lane] f4, read-modify-written by
    // their own lane only (registers are the scarce resource of the backward)
    static constexpr int tot_off = scratch1;
    static constexpr int tot_floats = NBLK * n_chunk * 64 * 4;
    static constexpr int scratch = scratch1 + tot_floats;
    static constexpr int sc_fwd = store_all, sc_bwd = store_all;
    static constexpr int fwd_total(bool edge) { return sc_fwd + (edge ? kRlWaves * stage_floats : 0); }
    static constexpr int bwd_total = sc_bwd + kRlWaves * scratch;
};

// per-lane geometry and the few per-lane LDS base offsets (floats) everything is addressed from
template <class ALG, int NOG, int NBLK, int I0>
struct RlGeo {
    static_assert(NOG == 2 || NOG == 4, "2 or 4 lanes per row");
    using LY = RlLay<ALG, NOG, NBLK, I0>;
    static constexpr int R = LY::R, C = LY::C, WS1 = LY::WS1, WSC = LY::WSC, RS = LY::RS, G = ALG::G, P = ALG::P;
    static constexpr int NGL0 = ((I0 + 3) / 4 + NOG - 1) / NOG;   // input channel groups of block 0 per lane
    int lane, l3, og, r;
    int a_x;            // block 0's MVLinear: weight row of this lane, (4og + l3) * WS1
    int a_f[NOG];       // C x C matrices: row (4og + l3) * WSC + the 4 input channels of piece k
    int a_t[NOG];       // transposed: rows of piece k's 4 output channels, column of this lane's input channel
    int a_ts[NOG];      // transposed to the gathered input: rows of piece k
    int c_t[NGL0];      // ... column 4 * min(4 (og + NOG t) + l3, I0 - 1)
    int p_og;           // 4 og: this lane's channel group inside a per-channel array
    int r_w, r_r, r_r2, r_t;  // row-sum transposition: write slot, stage-1 / stage-2 read bases, totals slot
    int s_w;            // transposition slice write: (4og) * RS + r
    int s_r;            // slice read: (lane & 15) * RS + 4 (lane >> 4)
    int s_rbC, s_rbX[(I0 + 15) / 16];   // B-operand reads, row clamped to the slice's valid channels
    int g_st;           // staging tile: r * SS + og * 4D

#ifdef CSMPN_STAMPS
    // diagnostic build only: shader-clock cycles per phase, summed per wave
    static constexpr int kStampSlots = 24;
    mutable unsigned long long t0, acc[kStampSlots];
    CSMPN_DEV void stamp_init() const { for (int i = 0; i < kStampSlots; ++i) acc[i] = 0; t0 = __builtin_amdgcn_s_memtime(); }
    CSMPN_DEV void stamp(int id) const {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        acc[id] += t1 - t0;
        t0 = t1;
        __builtin_amdgcn_sched_barrier(0);
    }
#else
    CSMPN_DEV void stamp_init() const {}
    CSMPN_DEV void stamp(int) const {}
#endif
    CSMPN_DEV explicit RlGeo(int lane_) : lane(lane_), l3(lane_ & 3) {
        int rg, src_og[NOG];
        if constexpr (NOG == 2) { og = (lane >> 3) & 1; rg = ((lane >> 4) << 1) | ((lane >> 2) & 1); }
        else { og = (lane >> 2) & 3; rg = lane >> 4; }
        r = 4 * rg + l3;
        src_og[0] = og;
        if constexpr (NOG == 2) {
            src_og[1] = dpp_movi<0x128>(og);
        } else {
            src_og[1] = dpp_movi<0x124>(og); src_og[2] = dpp_movi<0x128>(og); src_og[3] = dpp_movi<0x12C>(og);
        }
        a_x = (4 * og + l3) * WS1;
#pragma unroll
        for (int k = 0; k < NOG; ++k) {
            a_f[k] = (4 * og + l3) * WSC + 16 * src_og[k];
            a_ts[k] = 4 * src_og[k] * WS1;
            a_t[k] = 4 * src_og[k] * WSC + 4 * (4 * og + l3);
        }
#pragma unroll
        for (int t = 0; t < NGL0; ++t) {
            const int c = 4 * (og + NOG * t) + l3;
            c_t[t] = 4 * (c < I0 ? c : I0 - 1);
        }
        p_og = 4 * og;
        r_w = LY::red_off + 4 * lane;
        if constexpr (NOG == 2) {   // lane L sums value L & 7 over the 8 lanes (L & ~7) + s: one channel group, one DPP row
            r_r = LY::red_off + 4 * ((lane & 7) * 65 + (lane & ~7));
            r_r2 = LY::red_off + 4 * (8 * 65 + (lane & 15));
        } else {                    // value L & 7, channel group (L >> 3) & 3, rows of the DPP-row pair L >> 5
            r_r = LY::red_off + 4 * ((lane & 7) * 65 + 32 * (lane >> 5) + 4 * ((lane >> 3) & 3));
            r_r2 = LY::red_off + 4 * (8 * 65 + (lane & 31));
        }
        r_t = 4 * lane;
        s_w = 4 * og * RS + r;
        const int m = lane & 15, q = lane >> 4;
        s_r = m * RS + 4 * q;
        s_rbC = (m < C ? m : C - 1) * RS + 4 * q;
#pragma unroll
        for (int nt = 0; nt < (I0 + 15) / 16; ++nt) {
            const int n = 16 * nt + m;
            s_rbX[nt] = (n < I0 ? n : I0 - 1) * RS + 4 * q;
        }
        g_st = r * LY::SS + og * 4 * ALG::D;
    }
    // lane (of channel group 0) that holds row r
    static constexpr int lane_of_row(int r) {
        const int rg = r >> 2, j = r & 3;
        return NOG == 2 ? (((rg >> 1) << 4) | ((rg & 1) << 2) | j) : ((rg << 4) | j);
    }
};

// piece k of a distributed tensor: the 4 channels held by the k-th partner lane of the row
// row rotate: every lane has a source, so the destination needs no prior value (dpp_mov<> would
// zero it first: one more VALU instruction per exchanged value)
template <int CTRL>
CSMPN_DEV float dpp_rot(float v) {
    const int i = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, CTRL, 0xF, 0xF, true));
}
template <int NOG, int K>
CSMPN_DEV float rl_piece(float v) {
    if constexpr (K == 0) return v;
    else if constexpr (NOG == 2) return dpp_rot<0x128>(v);                       // row_ror:8
    else return dpp_rot<(K == 1 ? 0x124 : (K == 2 ? 0x128 : 0x12C))>(v);          // row_ror:4k
}
// sum over the NOG lanes of a row
template <int NOG>
CSMPN_DEV float rl_row_sum(float v) {
    v += dpp_mov<0x128>(v);
    if constexpr (NOG == 4) v += dpp_mov<0x124>(v);
    return v;
}

// ---------------------------------------------------------------------------------
// parameters -> LDS store (once per workgroup; all global loads of a thread in flight together)
template <class LY, int K>
__device__ void rl_stage_store(const DevBlock& B, float* lds, int tid) {
    constexpr int C = LY::C, G = LY::G, P = LY::P, I = LY::Iof(K), WSa = LY::wstride(K), WSC = LY::WSC;
    constexpr int total = LY::store_total(K), nW1 = C * WSa, nW = nW1 + 2 * C * WSC;
    constexpr int NIT = (total + 64 * kRlWaves - 1) / (64 * kRlWaves);
    float* st = lds + LY::st_off(K);
    // the parameter pointers first, as uniform values (scalar loads of the kernel arguments): selected per
    // thread by ADDRESS inside the branches below, the compiler fetched them with vector loads, one
    // dependent round trip in front of every value (2 NIT serialised memory latencies per block)
    const float *pW1 = B.W1, *pWR = B.WR, *pWL = B.WL, *pb1 = B.b1, *pbL = B.bL, *pla = B.la, *psa = B.sa, *psb = B.sb,
                *pan = B.an, *pw = B.w;
    const bool has_b1 = B.has_b1 != 0;
    float v[NIT];
    const float* src[NIT];
    bool sig[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int e = tid + it * 64 * kRlWaves;
        src[it] = nullptr;
        sig[it] = false;
        if (e < nW1) {
            const int o = e / WSa, r = e - o * WSa, c = r >> 2, g = r & 3;
            if (c < I && g < G) src[it] = pW1 + (o * I + c) * G + g;
        } else if (e < nW) {
            const int f0 = e - nW1, which = f0 / (C * WSC), f = f0 - which * (C * WSC);
            const int o = f / WSC, r = f - o * WSC, c = r >> 2, g = r & 3;
            if (c < C && g < G) src[it] = (which == 0 ? pWR : pWL) + (o * C + c) * G + g;
        } else if (e < total) {
            int f = e - nW;
            if (f < C) { if (has_b1) src[it] = pb1 + f; }
            else if ((f -= C) < C) src[it] = pbL + f;
            else if ((f -= C) < C) src[it] = pla + f;
            else if ((f -= C) < 3 * C * G) {
                const int which = f / (C * G), r = f - which * C * G, g = r / C, o = r - g * C;
                src[it] = (which == 0 ? psa : (which == 1 ? psb : pan)) + o * G + g;
                sig[it] = which == 2;
            } else {
                f -= 3 * C * G;
                const int p = f / C, o = f - p * C;
                src[it] = pw + o * P + p;
            }
        }
    }
    // all loads of the thread in flight together
#pragma unroll
    for (int it = 0; it < NIT; ++it) v[it] = src[it] ? *src[it] : 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int e = tid + it * 64 * kRlWaves;
        if (sig[it]) v[it] = sigmoidf(v[it]);
        if (e < total) st[e] = v[it];
    }
}

// second kernel of a backward: grads += sum over the workgroups' slices, fixed order (deterministic).
// A workgroup takes 16 consecutive elements; thread (j = tid & 15, w0 = tid >> 4) adds the slices
// w0, w0 + 16, ... of element 16 b + j, the 16 partial sums meet in LDS and are added in order.
template <class ALG, int NOG, int NBLK, int I0>
__global__ void __launch_bounds__(256) rl_reduce_kernel(const DevCemlp C_arg, const float* part, int nslices) {
    using LY = RlLay<ALG, NOG, NBLK, I0>;
    constexpr int C = LY::C, G = LY::G;
    __shared__ float red[16][17];
    const int j = threadIdx.x & 15, w0 = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + j;
    float s = 0.f;
    if (e < LY::part_total) {
        const float* p = part + e;
        int w = w0;
        for (; w + 48 < nslices; w += 64) {   // four independent loads in flight
            const float a = p[(size_t)w * LY::part_total], b = p[(size_t)(w + 16) * LY::part_total];
            const float c = p[(size_t)(w + 32) * LY::part_total], d = p[(size_t)(w + 48) * LY::part_total];
            s += (a + b) + (c + d);
        }
        for (; w < nslices; w += 16) s += p[(size_t)w * LY::part_total];
    }
    red[w0][j] = s;
    __syncthreads();
    if (w0 != 0 || e >= LY::part_total) return;
#pragma unroll
    for (int w = 1; w < 16; ++w) s += red[w][j];
    const int k = (NBLK > 1 && e >= LY::part_blk(0)) ? 1 : 0;
    const DevBlock& B = C_arg.b[k];
    const int I = k == 0 ? I0 : C;
    int f = e - (k == 0 ? 0 : LY::part_blk(0));
    float* dst = nullptr;
    const int nW1 = C * I * G, nWC = C * C * G;
    if (f < nW1) dst = B.gW1 + f;
    else if ((f -= nW1) < nWC) dst = B.gWR + f;
    else if ((f -= nWC) < nWC) dst = B.gWL + f;
    else {
        f -= nWC;
        if (f < LY::qsa) dst = B.has_b1 ? B.gb1 + f : nullptr;
        else if (f < LY::qsb) dst = B.gsa + (f - LY::qsa);
        else if (f < LY::qw) dst = B.gsb + (f - LY::qsb);
        else if (f < LY::qan) dst = B.gw + (f - LY::qw);
        else if (f < LY::qbL) dst = B.gan + (f - LY::qan);
        else if (f < LY::qla) dst = B.gbL + (f - LY::qbL);
        else dst = B.gla + (f - LY::qla);
    }
    if (dst) *dst += s;
}

// ---------------------------------------------------------------------------------
// Explicitly ordered LDS reads (asm): under the register pressure of the backward kernels the scheduler
// issues an LDS read right in front of its first use and waits for it with nothing in between; these
// helpers put four 16-byte reads in flight and let the caller decide where to wait.
CSMPN_DEV unsigned rl_lds_addr(const float* p) { return (unsigned)(unsigned long long)p; }   // LDS byte address
template <int O0, int O1, int O2, int O3>
// (No "memory" clobber: the only data read this way is the weight store, written once in front of the workgroup barrier
// that opens the tile loop. The registers in flight between this statement and rl_lds_wait are audited at build time:
// `make check-asm` / tools/check_asm_waits.py fails the build if a compiler instruction touches them.)
CSMPN_DEV void rl_lds_read4(unsigned a, f4 (&w)[4]) {   // byte offsets from a; read-only data (the weight store)
    asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\t"
                 "ds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8"
                 : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3])
                 : "v"(a), "n"(O0), "n"(O1), "n"(O2), "n"(O3));
}
CSMPN_DEV void rl_lds_wait(f4 (&w)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
}
// groups of four weight vectors, software-pipelined by hand: group g + 1 is read while the MFMAs of group g
// run. read(g, w) issues the reads of group g, use(g, w) consumes them.
template <int NG, class Read, class Use>
CSMPN_DEV void rl_weight_pipeline(Read&& read, Use&& use) {
    f4 w[4];
    read(IC<0>{}, w);
    rl_lds_wait(w);
    static_for<0, NG>([&](auto g) {
        f4 wn[4];
        if constexpr (g + 1 < NG) read(IC<g + 1>{}, wn);
        use(g, w);
        if constexpr (g + 1 < NG) {
            rl_lds_wait(wn);
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = wn[i];
        }
    });
}

// ---------------------------------------------------------------------------------
// dense channel mixing on the 4x4x1 MFMA. acc = the lane's 4 output channels. WOFF = LDS float
// offset of the weight matrix (compile time: an instruction immediate).

// input: I channels all present in the lane (the row's gathered input)
//   acc[d] += sum_c W[4og + i][c][grade(d)] * X[c][d]
// (a chunk of them: X = input channels C0 .. C0 + NCH). Weight vectors four at a time (rl_weight_pipeline).
template <class ALG, int WOFF, int C0, int NCH, class GE>
CSMPN_DEV void rl_linear_x(f4 (&acc)[ALG::D], const float (&X)[NCH][ALG::D], const float* lds, const GE& ge) {
    constexpr int D = ALG::D, NG = (NCH + 3) / 4;
    const unsigned a = rl_lds_addr(lds + ge.a_x);
    // a short last group reads its last vector again (a valid address; the value is not used)
    auto off = [](int g, int i) { const int c = 4 * g + i; return 4 * (WOFF + 4 * (C0 + (c < NCH ? c : NCH - 1))); };
    rl_weight_pipeline<NG>(
        [&](auto g, f4 (&w)[4]) { rl_lds_read4<off(g, 0), off(g, 1), off(g, 2), off(g, 3)>(a, w); },
        [&](auto g, const f4 (&w)[4]) {
            static_for<0, 4>([&](auto i) {
                constexpr int c = 4 * g + i;
                if constexpr (c < NCH) static_for<0, D>([&](auto d) { acc[d] = mfma4(w[i][ALG::grade(d)], X[c][d], acc[d]); });
            });
        });
}

// input: C channels distributed over the NOG lanes of the row (T = this lane's 4)
template <class ALG, int NOG, int WOFF, class GE>
CSMPN_DEV void rl_linear_d(f4 (&acc)[ALG::D], const f4 (&T)[ALG::D], const float* lds, const GE& ge) {
    constexpr int D = ALG::D;
    rl_weight_pipeline<NOG>(
        [&](auto k, f4 (&w)[4]) { rl_lds_read4<4 * WOFF, 4 * WOFF + 16, 4 * WOFF + 32, 4 * WOFF + 48>(rl_lds_addr(lds + ge.a_f[k]), w); },
        [&](auto k, const f4 (&w)[4]) {
            static_for<0, 4>([&](auto cl) {
                static_for<0, D>([&](auto d) {
                    acc[d] = mfma4(w[cl][ALG::grade(d)], rl_piece<NOG, k>(T[d][int(cl)]), acc[d]);
                });
            });
        });
}

// transposed, C -> C:  gx[d] (the lane's 4 INPUT channels) += sum_o W[o][4og + i][grade(d)] * Gin[o][d]
template <class ALG, int NOG, int WOFF, class GE>
CSMPN_DEV void rl_linear_dt(f4 (&gx)[ALG::D], const f4 (&Gin)[ALG::D], const float* lds, const GE& ge) {
    constexpr int D = ALG::D, WS = GE::WSC;
    rl_weight_pipeline<NOG>(
        [&](auto k, f4 (&w)[4]) { rl_lds_read4<4 * WOFF, 4 * (WOFF + WS), 4 * (WOFF + 2 * WS), 4 * (WOFF + 3 * WS)>(rl_lds_addr(lds + ge.a_t[k]), w); },
        [&](auto k, const f4 (&w)[4]) {
            static_for<0, 4>([&](auto ol) {
                static_for<0, D>([&](auto d) {
                    gx[d] = mfma4(w[ol][ALG::grade(d)], rl_piece<NOG, k>(Gin[d][int(ol)]), gx[d]);
                });
            });
        });
}

// transposed, C -> I0 (the gathered input of block 0): lane (row, og) produces the input channel
// groups cg = og + NOG*t, t < NGL0. The last group may reach past I0: those lanes read a clamped
// (valid, finite) weight column and produce values the caller ignores.
template <class ALG, int NOG, int WOFF, class GE>
CSMPN_DEV void rl_linear_xt(f4 (&gx)[GE::NGL0][ALG::D], const f4 (&Gin)[ALG::D], const float* lds, const GE& ge) {
    constexpr int D = ALG::D, WS = GE::WS1;
    rl_weight_pipeline<GE::NGL0 * NOG>(
        [&](auto g, f4 (&w)[4]) {
            constexpr int t = g / NOG, k = g % NOG;
            rl_lds_read4<4 * WOFF, 4 * (WOFF + WS), 4 * (WOFF + 2 * WS), 4 * (WOFF + 3 * WS)>(rl_lds_addr(lds + (ge.a_ts[k] + ge.c_t[t])), w);
        },
        [&](auto g, const f4 (&w)[4]) {
            constexpr int t = g / NOG, k = g % NOG;
            static_for<0, 4>([&](auto ol) {
                static_for<0, D>([&](auto d) {
                    gx[t][d] = mfma4(w[ol][ALG::grade(d)], rl_piece<NOG, k>(Gin[d][int(ol)]), gx[t][d]);
                });
            });
        });
}

// ---------------------------------------------------------------------------------
// sign-table geometric product with per-path weights, the lane's 4 channels
//   out[j] += sum_{(i,k)->j} sign(i,k) * w[path(g_i, g_j, g_k)] * z[i] * r[k]
// wl + C p: path p's weights of this lane's channel group
template <class ALG, int C>
CSMPN_DEV void rl_weighted_gp(f4 (&out)[ALG::D], const f4 (&z)[ALG::D], const f4 (&r)[ALG::D], const float* wl) {
    constexpr int P = ALG::P;
    static_for<0, P>([&](auto p) {
        constexpr int gi = ALG::t.path_g[p][0], gj = ALG::t.path_g[p][1], gk = ALG::t.path_g[p][2];
        constexpr int i0 = ALG::gstart(gi), ni = ALG::gsize(gi);
        constexpr int j0 = ALG::gstart(gj), nj = ALG::gsize(gj);
        constexpr int k0 = ALG::gstart(gk), nk = ALG::gsize(gk);
        const f4 w = ld4(wl + C * p);
        f4 tmp[nj];
#pragma unroll
        for (int t = 0; t < nj; ++t) tmp[t] = splat(0.f);
        static_for<0, ni>([&](auto ii) {
            static_for<0, nk>([&](auto kk) {
                constexpr int i = i0 + ii, k = k0 + kk;
                constexpr int j = ALG::t.out[i][k];
                if constexpr (j >= j0 && j < j0 + nj) {
                    constexpr float sg = float(ALG::t.sign[i][k]);
                    tmp[j - j0] += (sg * z[i]) * r[k];
                }
            });
        });
#pragma unroll
        for (int t = 0; t < nj; ++t) out[j0 + t] += w * tmp[t];
    });
}

// ---------------------------------------------------------------------------------
// Sums over the rows of a wave for the small per-channel parameter gradients. The block backward
// produces n_red f4 values per lane (the lane's 4 channels) in a fixed order; 8 at a time they go
// through the wave's LDS scratch: write [value v][lane] (f4 slots, value stride 65: conflict-free),
// lane L then adds the 8 lanes (L & ~7) + s of value L & 7 (8 rows of ITS channel group), a second
// hop adds the 4 DPP rows. Every lane ends with the wave total of (value L & 7, channel group of L),
// replicated 4x, and adds it to the persistent accumulator of the chunk.
template <class ALG, class GE>
struct RlRed {
    using LY = typename GE::LY;
    float* sc;
    const GE& ge;
    float* tot;   // this block's running totals in the wave's scratch: [chunk][lane] f4
    CSMPN_DEV RlRed(float* sc_, const GE& ge_, float* tot_) : sc(sc_), ge(ge_), tot(tot_) {}
    // the value goes to its LDS slot at once (no pending registers); the LDS executes a wave's
    // accesses in order, so the next chunk's writes cannot overtake this chunk's reads
    template <int IDX>
    CSMPN_DEV void push(f4 v) {
        *reinterpret_cast<f4*>(sc + ge.r_w + (IDX % 8) * 260) = v;
        if constexpr (IDX % 8 == 7 || IDX == LY::n_red - 1) flush<IDX / 8>();
    }
    // LDS byte address of a pointer into the dynamic LDS (the low half of the flat address)
    static CSMPN_DEV unsigned lds_addr(const float* p) { return (unsigned)(unsigned long long)p; }
    // four 16-byte LDS reads (byte offsets O0..O3 from address a), ALL in flight before the first use;
    // the caller waits (s_waitcnt lgkmcnt(0) in its own asm statement, tied to the values). Written as asm:
    // left to the scheduler, the register pressure of the backward makes it read, wait and add one value
    // at a time (8 + 4 serial LDS round trips per flush, 10 flushes per tile).
    template <int O0, int O1, int O2, int O3>
    static CSMPN_DEV void lds_read4(unsigned a, f4& v0, f4& v1, f4& v2, f4& v3) {
        asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\t"
                     "ds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8"
                     : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
                     : "v"(a), "n"(O0), "n"(O1), "n"(O2), "n"(O3)
                     : "memory");
    }
    template <int CH>
    CSMPN_DEV void flush() {
        constexpr int NOG = GE::C / 4;
        float* w = sc + ge.r_w;
        float* a = tot + ge.r_t + CH * 256;
        if constexpr (NOG == 2) {
            // lane L sums value L & 7 over the 8 consecutive source lanes of its channel group, then over the 4
            // DPP rows (256 floats apart), and adds the result to its running total. Batched reads (measured:
            // S1 edge backward -2.5 %; the 16-channel kernels have no registers to spare for them, below)
            const unsigned ar = lds_addr(sc + ge.r_r);
            f4 v[8];
            lds_read4<0, 16, 32, 48>(ar, v[0], v[1], v[2], v[3]);
            lds_read4<64, 80, 96, 112>(ar, v[4], v[5], v[6], v[7]);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) :: "memory");
            const f4 s = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            *reinterpret_cast<f4*>(w + 8 * 260) = s;
            f4 u[4], old;
            lds_read4<0, 256, 512, 768>(lds_addr(sc + ge.r_r2), u[0], u[1], u[2], u[3]);
            asm volatile("ds_read_b128 %0, %5\n\ts_waitcnt lgkmcnt(0)" : "=&v"(old), "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]) : "v"(lds_addr(a)) : "memory");
            *reinterpret_cast<f4*>(a) = old + ((u[0] + u[1]) + (u[2] + u[3]));
        } else {
            // 4 rows of each of two DPP rows, then the two DPP-row pairs (128 floats apart)
            const float* rd = sc + ge.r_r;
            f4 s = ld4(rd);
#pragma unroll
            for (int k = 1; k < 8; ++k) s += ld4(rd + 4 * ((k & 3) + 16 * (k >> 2)));
            *reinterpret_cast<f4*>(w + 8 * 260) = s;
            const float* r2 = sc + ge.r_r2;
            f4 t = ld4(r2);
#pragma unroll
            for (int k = 1; k < 8 / NOG; ++k) t += ld4(r2 + 32 * NOG * k);
            *reinterpret_cast<f4*>(a) = ld4(a) + t;
        }
    }
};
// reduction-order index -> (offset of the parameter's first element inside the small block, stride
// between channels)
template <class LY>
struct RlRedMap {
    static constexpr int G = LY::G, P = LY::P;
    static constexpr int off(int idx) {
        if (idx == 0) return LY::qla;
        if (idx == 1) return LY::qbL;
        if (idx < 2 + P) return LY::qw + (idx - 2);
        if (idx < 2 + P + G) return LY::qan + (idx - 2 - P);
        if (idx < 2 + P + 3 * G) return ((idx - 2 - P - G) & 1 ? LY::qsb : LY::qsa) + (idx - 2 - P - G) / 2;
        return LY::qb1;
    }
    static constexpr int stride(int idx) {
        if (idx == 0 || idx == 1) return 1;
        if (idx < 2 + P) return P;
        if (idx < 2 + P + 3 * G) return G;
        return 1;
    }
    static constexpr int i_la = 0, i_bL = 1, i_w = 2, i_an = 2 + P, i_sa = 2 + P + G, i_b1 = 2 + P + 3 * G;
};

// ---------------------------------------------------------------------------------
// forward state of one block kept for its backward (the lane's 4 channels)
template <class ALG>
struct RlFwd {
    f4 y[ALG::D];        // MVLinear output
    f4 gate[ALG::G];     // MVSiLU gates
    f4 R[ALG::D];        // linear_right output
    f4 invden[ALG::G];   // 1 / (interpolated norm + eps)
    f4 s[ALG::D];        // (left + gp) / sqrt2
    f4 qs, nl;
    float invMn;
};

// the part of a block forward after its MVLinear: S.y holds the MVLinear output (without bias).
// K = block index (its parameter store: RlLay::o_*(K)).
template <class ALG, int NOG, int K, class GE>
CSMPN_DEV void rl_block_tail(const float* lds, const GE& ge, RlFwd<ALG>& S, f4 (&out)[ALG::D]) {
    using LY = typename GE::LY;
    constexpr int D = ALG::D, G = ALG::G, C = 4 * NOG;
    const float* sp = lds + ge.p_og;   // this lane's channel group inside every per-channel array
    S.y[0] += ld4(sp + (LY::o_b1(K)));
    // 2. MVSiLU, invariant "mag2" (cegnn_utils.py:76-83)
    f4 z[D];
    static_for<0, G>([&](auto g) {
        constexpr int d0 = ALG::gstart(g), nd = ALG::gsize(g);
        f4 u;
        if constexpr (g == 0) {
            u = S.y[0];
        } else {
            u = splat(0.f);
            static_for<0, nd>([&](auto t) {
                constexpr int d = d0 + decltype(t)::value;
                u += qsf<ALG, d> * S.y[d] * S.y[d];
            });
        }
        S.gate[g] = sigmoid4(ld4(sp + (LY::o_sa(K) + C * g)) * u + ld4(sp + (LY::o_sb(K) + C * g)));
#pragma unroll
        for (int t = 0; t < nd; ++t) z[d0 + t] = S.gate[g] * S.y[d0 + t];
    });
    ge.stamp(3);
    CSMPN_PHASE();
    // 3. linear_right / linear_left (cegnn_utils.py:143-148)
    f4 L[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { S.R[d] = splat(0.f); L[d] = splat(0.f); }
    rl_linear_d<ALG, NOG, LY::o_WR(K)>(S.R, z, lds, ge);
    rl_linear_d<ALG, NOG, LY::o_WL(K)>(L, z, lds, ge);
    ge.stamp(4);
    CSMPN_PHASE();
    L[0] += ld4(sp + (LY::o_bL(K)));
    // 4. NormalizationLayer on the right operand (cegnn_utils.py:42-51)
    f4 r[D];
    static_for<0, G>([&](auto g) {
        constexpr int d0 = ALG::gstart(g), nd = ALG::gsize(g);
        f4 qq = splat(0.f);
        static_for<0, nd>([&](auto t) {
            constexpr int d = d0 + decltype(t)::value;
            qq += qsf<ALG, d> * S.R[d] * S.R[d];
        });
        const f4 m = ld4(sp + (LY::o_sg(K) + C * g)) * (smooth_abs_sqrt4(qq) - 1.0f) + 1.0f;
        S.invden[g] = rcp4(m + kEps);
#pragma unroll
        for (int t = 0; t < nd; ++t) r[d0 + t] = S.R[d0 + t] * S.invden[g];
    });
    ge.stamp(5);
    // 5. steerable geometric product + first-order term (cegnn_utils.py:126-152)
    rl_weighted_gp<ALG, C>(L, z, r, sp + (LY::o_w(K)));
    ge.stamp(6);
#pragma unroll
    for (int d = 0; d < D; ++d) S.s[d] = L[d] * kInvSqrt2;
    // 6. MVLayerNorm (cegnn_utils.py:93-96): mean over the C channels of the row
    f4 qs = splat(0.f);
    static_for<0, D>([&](auto dd) {
        constexpr int d = decltype(dd)::value;
        qs += qsf<ALG, d> * S.s[d] * S.s[d];
    });
    S.qs = qs;
    S.nl = smooth_abs_sqrt4(qs);
    S.invMn = fast_rcp(rl_row_sum<NOG>(hsum(S.nl)) * (1.0f / float(C)) + kEps);
    const f4 la = ld4(sp + (LY::o_la(K)));
#pragma unroll
    for (int d = 0; d < D; ++d) out[d] = la * S.s[d] * S.invMn;
    ge.stamp(7);
}

// ---------------------------------------------------------------------------------
// weight gradients through the per-wave LDS transposition. For every blade d the writers store
// the slices  A[m][row] (m < 16: gradient channels)  and  B[c][row] (c < NB: input channels) at
// sc / sc + b_off; lane (m = lane & 15, q = lane >> 4) then feeds the 16x16x4 MFMAs with
// A[16 ma + m][16t + 4q + v], B[16 nt + m][16t + 4q + v] (one ds_read_b128 each = 4 k-steps; rb[nt] =
// the lane's clamped B read base). acc[ma][nt][g][v] = sum over the wave's rows of
// A[16 ma + 4q + v] * B[16 nt + n] for the blades of grade g.
template <class ALG, int MA, int NT, class GE, class WriteA, class WriteB>
CSMPN_DEV void rl_wgrad(float* sc, const GE& ge, const int (&rb)[NT], f4 (&acc)[MA][NT][ALG::G], WriteA&& write_a, WriteB&& write_b) {
    using LY = typename GE::LY;
    constexpr int D = ALG::D, RS = GE::RS, KT = GE::R / 16;
    static_for<0, D>([&](auto d) {
        constexpr int g = ALG::grade(d);
        write_a(d, sc + ge.s_w);
        write_b(d, sc + LY::b_off);
        static_for<0, KT>([&](auto t) {
            f4 b[NT];
            static_for<0, NT>([&](auto nt) { b[nt] = ld4(sc + rb[nt] + (LY::b_off + 16 * t)); });
            static_for<0, MA>([&](auto ma) {
                const f4 a = ld4(sc + ge.s_r + (16 * ma * RS + 16 * t));
                static_for<0, NT>([&](auto nt) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) acc[ma][nt][g] = mfma16(a[v], b[nt][v], acc[ma][nt][g]);
                });
            });
        });
    });
}

// ---------------------------------------------------------------------------------
// block backward. S: the block's recomputed forward state; gout: d/d(out) (the lane's 4 channels).
// Adds the parameter gradients of the wave's rows into the persistent accumulators (tot: small
// parameters, accRL: linear_right | linear_left weight tiles), leaves d/d(MVLinear output) in gy.
// The MVLinear weight gradient and the transposed MVLinear are the caller's (they need the
// block's input again).
// early_loads(): called in front of the last phase (MVSiLU backward); the caller issues there the global loads
// its next step needs (the block's input for the MVLinear weight gradient), so that their latency passes
// under that phase instead of in front of the gradient loop.
template <class ALG, int NOG, int K, class GE, class Early>
CSMPN_DEV void rl_block_backward(float* lds, float* sc, const GE& ge, const RlFwd<ALG>& S, const f4 (&gout)[ALG::D],
                                 f4 (&gy)[ALG::D], float* tot, f4 (&accRL)[GE::LY::MA_RL][1][ALG::G], Early&& early_loads) {
    using LY = typename GE::LY;
    constexpr int D = ALG::D, G = ALG::G, P = ALG::P, C = 4 * NOG, RS = GE::RS;
    using RM = RlRedMap<LY>;
    const float* sp = lds + ge.p_og;
    RlRed<ALG, GE> red(sc, ge, tot);

    // ---- MVLayerNorm backward
    const f4 la = ld4(sp + (LY::o_la(K)));
    f4 dot = splat(0.f);
#pragma unroll
    for (int d = 0; d < D; ++d) dot += gout[d] * S.s[d];
    red.template push<RM::i_la>(dot * S.invMn);
    const float gMn = -rl_row_sum<NOG>(hsum(la * dot)) * S.invMn * S.invMn * (1.0f / float(C));   // d/d(mean norm) / C
    f4 ggp[D];   // = d/d(left) = d/d(gp)
    {
        const f4 inl = rcp4(S.nl);
        const f4 gqs = gMn * (0.5f * S.qs) * (inl * inl * inl);   // d nl/d qs = 0.5 qs / nl^3
        static_for<0, D>([&](auto dd) {
            constexpr int d = decltype(dd)::value;
            const f4 gs = (la * gout[d]) * S.invMn + gqs * (2.0f * qsf<ALG, d>) * S.s[d];
            ggp[d] = gs * kInvSqrt2;
        });
    }
    red.template push<RM::i_bL>(ggp[0]);
    ge.stamp(8);
    CSMPN_PHASE();
    // ---- d/dz from linear_left
    f4 gz[D];
#pragma unroll
    for (int d = 0; d < D; ++d) gz[d] = splat(0.f);
    rl_linear_dt<ALG, NOG, LY::o_WL(K)>(gz, ggp, lds, ge);
    ge.stamp(9);
    CSMPN_PHASE();
    // ---- geometric product backward (gz, gr accumulate; d/dw per path reduced at once)
    f4 gr[D], zf[D], rf[D];
#pragma unroll
    for (int d = 0; d < D; ++d) gr[d] = splat(0.f);
    static_for<0, D>([&](auto d) {
        zf[d] = S.gate[ALG::grade(d)] * S.y[d];
        rf[d] = S.R[d] * S.invden[ALG::grade(d)];
    });
    static_for<0, P>([&](auto p) {
        constexpr int gi = ALG::t.path_g[p][0], gj = ALG::t.path_g[p][1], gk = ALG::t.path_g[p][2];
        constexpr int i0 = ALG::gstart(gi), ni = ALG::gsize(gi);
        constexpr int j0 = ALG::gstart(gj), nj = ALG::gsize(gj);
        constexpr int k0 = ALG::gstart(gk), nk = ALG::gsize(gk);
        const f4 w = ld4(sp + (LY::o_w(K) + C * p));
        // U[i] = sum sign ggp[j] r[k] (unweighted d/dz), V[k] = sum sign ggp[j] z[i] (unweighted d/dr)
        f4 U[ni], V[nk];
#pragma unroll
        for (int t = 0; t < ni; ++t) U[t] = splat(0.f);
#pragma unroll
        for (int t = 0; t < nk; ++t) V[t] = splat(0.f);
        static_for<0, ni>([&](auto ii) {
            static_for<0, nk>([&](auto kk) {
                constexpr int i = i0 + ii, k = k0 + kk;
                constexpr int j = ALG::t.out[i][k];
                if constexpr (j >= j0 && j < j0 + nj) {
                    constexpr float sg = float(ALG::t.sign[i][k]);
                    U[ii] += (sg * ggp[j]) * rf[k];
                    V[kk] += (sg * ggp[j]) * zf[i];
                }
            });
        });
        f4 gwv = splat(0.f);
#pragma unroll
        for (int t = 0; t < ni; ++t) { gz[i0 + t] += w * U[t]; gwv += zf[i0 + t] * U[t]; }
#pragma unroll
        for (int t = 0; t < nk; ++t) gr[k0 + t] += w * V[t];
        red.template push<RM::i_w + p>(gwv);
    });
    ge.stamp(10);
    CSMPN_PHASE();
    // ---- NormalizationLayer backward -> gR
    f4 gR[D];
    static_for<0, G>([&](auto g) {
        constexpr int d0 = ALG::gstart(g), nd = ALG::gsize(g);
        f4 gden = splat(0.f), qR = splat(0.f);
        static_for<0, nd>([&](auto t) {
            constexpr int d = d0 + decltype(t)::value;
            gden -= gr[d] * S.R[d];
            qR += qsf<ALG, d> * S.R[d] * S.R[d];
        });
        gden *= S.invden[g] * S.invden[g];      // d/d(den): -sum gr * R / den^2
        const f4 nu = smooth_abs_sqrt4(qR);
        const f4 sg = ld4(sp + (LY::o_sg(K) + C * g));
        red.template push<RM::i_an + g>(gden * (nu - 1.0f) * sg * (1.0f - sg));
        const f4 inu = rcp4(nu);
        const f4 gq = (gden * sg) * (0.5f * qR) * (inu * inu * inu);
        static_for<0, nd>([&](auto t) {
            constexpr int d = d0 + decltype(t)::value;
            gR[d] = gr[d] * S.invden[g] + gq * (2.0f * qsf<ALG, d>) * S.R[d];
        });
    });
    ge.stamp(11);
    CSMPN_PHASE();
    rl_linear_dt<ALG, NOG, LY::o_WR(K)>(gz, gR, lds, ge);
    ge.stamp(12);
    CSMPN_PHASE();
    // ---- weight gradients of linear_right and linear_left; B = z = gate * y
    {
        const int rb[1] = {ge.s_rbC};
        // A rows 0..C-1: gR (-> WR), rows C..2C-1: ggp (-> WL)
        rl_wgrad<ALG, LY::MA_RL, 1>(sc, ge, rb, accRL,
            [&](auto d, float* p) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { p[i * RS] = gR[d][i]; p[(C + i) * RS] = ggp[d][i]; }
            },
            [&](auto d, float* sB) {
                float* p = sB + ge.s_w;
                const f4 zz = S.gate[ALG::grade(d)] * S.y[d];
#pragma unroll
                for (int i = 0; i < 4; ++i) p[i * RS] = zz[i];
            });
    }
    ge.stamp(13);
    CSMPN_PHASE();
    early_loads();
    // ---- MVSiLU backward -> gy
    static_for<0, G>([&](auto g) {
        constexpr int d0 = ALG::gstart(g), nd = ALG::gsize(g);
        f4 ggate = splat(0.f);
#pragma unroll
        for (int t = 0; t < nd; ++t) ggate += gz[d0 + t] * S.y[d0 + t];
        const f4 gpre = ggate * S.gate[g] * (1.0f - S.gate[g]);
        f4 u;
        if constexpr (g == 0) {
            u = S.y[0];
        } else {
            u = splat(0.f);
            static_for<0, nd>([&](auto t) {
                constexpr int d = d0 + decltype(t)::value;
                u += qsf<ALG, d> * S.y[d] * S.y[d];
            });
        }
        red.template push<RM::i_sa + 2 * g>(gpre * u);
        red.template push<RM::i_sa + 2 * g + 1>(gpre);
        const f4 gu = gpre * ld4(sp + (LY::o_sa(K) + C * g));
        static_for<0, nd>([&](auto t) {
            constexpr int d = d0 + decltype(t)::value;
            f4 v = gz[d] * S.gate[g];
            if constexpr (g == 0) v += gu;
            else v += gu * (2.0f * qsf<ALG, d>) * S.y[d];
            gy[d] = v;
        });
    });
    red.template push<RM::i_b1>(gy[0]);
    ge.stamp(14);
}

// MVLinear weight gradient of block K: A = gy (the lane's 4 channels), B = the block's input
// (write_b stores blade d of the input channels into the B slice); acc persists across tiles
template <class ALG, int NOG, int K, class GE, class WriteB>
CSMPN_DEV void rl_w1_grad(float* sc, const GE& ge, const f4 (&gy)[ALG::D], f4 (&acc)[1][(GE::LY::Iof(K) + 15) / 16][ALG::G],
                          WriteB&& write_b) {
    using LY = typename GE::LY;
    constexpr int RS = GE::RS, I = LY::Iof(K), NT = (I + 15) / 16;
    int rb[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) rb[nt] = K == 0 ? ge.s_rbX[nt] : ge.s_rbC;
    rl_wgrad<ALG, 1, NT>(sc, ge, rb, acc,
        [&](auto d, float* p) {
#pragma unroll
            for (int i = 0; i < 4; ++i) p[i * RS] = gy[d][i];
        },
        write_b);
}

// end of the launch: this wave's parameter-gradient sums of block K -> LDS image of its slice of the
// partial buffer (img: >= part_blk(K) floats of the wave's scratch); the caller copies the image out
// with coalesced 16-byte stores
template <class ALG, int NOG, int K, class GE>
CSMPN_DEV void rl_partials_image(float* img, const GE& ge, const f4 (&accW1)[1][(GE::LY::Iof(K) + 15) / 16][ALG::G],
                                 const f4 (&accRL)[GE::LY::MA_RL][1][ALG::G], const float* tot) {
    using LY = typename GE::LY;
    using RM = RlRedMap<LY>;
    constexpr int G = ALG::G, C = LY::C, I = LY::Iof(K), NT = (I + 15) / 16;
    const int n = ge.lane & 15, q = ge.lane >> 4;
    // MFMA tile element (i = 4q + v, j = n): W1[o = i][c = 16 nt + n] for i < C
    if (4 * q < C) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int c = 16 * nt + n;
            if (c < I) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    float* p = img + ((4 * q + v) * I + c) * G;
#pragma unroll
                    for (int g = 0; g < G; ++g) p[g] = accW1[0][nt][g][v];
                }
            }
        }
    }
    // A rows 0..C-1: linear_right, rows C..2C-1: linear_left
    if (n < C) {
#pragma unroll
        for (int ma = 0; ma < LY::MA_RL; ++ma) {
            const int i0 = 16 * ma + 4 * q;                       // first of this lane's 4 A rows
            float* base = img + (i0 < C ? LY::pWR(K) + (i0 * C + n) * G : LY::pWL(K) + ((i0 - C) * C + n) * G);
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int g = 0; g < G; ++g) base[v * C * G + g] = accRL[ma][0][g][v];
        }
    }
    // small parameters: lane L < 8 NOG holds (value L & 7 of every chunk, channel group L >> 3)
    if (ge.lane < 8 * NOG) {
        float* sm = img + LY::pS(K);
        const int og_img = ge.lane >> 3;
        static_for<0, LY::n_red>([&](auto idx) {
            if ((ge.lane & 7) == idx % 8) {
                float* p = sm + RM::off(idx) + 4 * og_img * RM::stride(idx);
                const f4 t = ld4(tot + ge.r_t + (idx / 8) * 256);
                p[0] = t.x; p[RM::stride(idx)] = t.y; p[2 * RM::stride(idx)] = t.z; p[3 * RM::stride(idx)] = t.w;
            }
        });
    }
}

// ---------------------------------------------------------------------------------
// per-lane row access (16-byte pieces)

// X[c0 .. c0+NCH) <- NCH*D contiguous floats at p
// PIN: the loads are issued HERE (an empty asm statement takes their results): the scheduler otherwise
// moves them down between the consumers of earlier loads and each one is waited for on its own
template <class ALG, int NCH, int I, bool PIN = false>
CSMPN_DEV void rl_load_channels(float (&X)[I][ALG::D], int c0, const float* p) {
    constexpr int D = ALG::D, NQ = NCH * D / 4;
    f4 v[NQ];
#pragma unroll
    for (int e = 0; e < NQ; ++e) v[e] = ld4(p + 4 * e);
    if constexpr (PIN) {
#pragma unroll
        for (int e = 0; e < NQ; ++e) asm volatile("" : "+v"(v[e]));
    }
    static_for<0, NQ>([&](auto e) {
        static_for<0, 4>([&](auto k) {
            constexpr int f = 4 * e + k;
            X[c0 + f / D][f % D] = v[e][int(k)];
        });
    });
}
// T (the lane's 4 channels) <- 4*D contiguous floats [c][d] at p
template <class ALG>
CSMPN_DEV void rl_load_t(f4 (&T)[ALG::D], const float* p) {
    constexpr int D = ALG::D;
    f4 v[D];
#pragma unroll
    for (int e = 0; e < D; ++e) v[e] = ld4(p + 4 * e);
    static_for<0, D>([&](auto e) {
        static_for<0, 4>([&](auto k) {
            constexpr int f = 4 * e + k, c = f / D, d = f % D;
            T[d][c] = v[e][int(k)];
        });
    });
}
// 4*D contiguous floats [c][d] at p <- T * scale
template <class ALG>
CSMPN_DEV void rl_store_t(const f4 (&T)[ALG::D], float* p, float scale) {
    constexpr int D = ALG::D;
    static_for<0, D>([&](auto e) {
        f4 v;
        static_for<0, 4>([&](auto k) {
            constexpr int f = 4 * e + k, c = f / D, d = f % D;
            v[int(k)] = T[d][c] * scale;
        });
        *reinterpret_cast<f4*>(p + 4 * e) = v;
    });
}
// store the channels [c0, c0 + 4) n [0, nch) of a row-major [nch][D] destination row
template <class ALG>
CSMPN_DEV void rl_store_group(const f4 (&T)[ALG::D], float* base, int c0, int nch) {
    constexpr int D = ALG::D;
    static_for<0, 4>([&](auto i) {
        if (c0 + i < nch) {
            static_for<0, D / 4>([&](auto q4) {
                *reinterpret_cast<f4*>(base + (c0 + i) * D + 4 * q4) =
                    f4{T[4 * q4][int(i)], T[4 * q4 + 1][int(i)], T[4 * q4 + 2][int(i)], T[4 * q4 + 3][int(i)]};
            });
        }
    });
}

// Rows of a staged tile [R][ROWLEN + 4] -> atomic adds into table rows of ROWLEN floats; lane =
// column (ROWLEN / 64 columns per lane). The row targets travel through SGPRs (v_readlane of the
// lane that holds the row). Adds the rows to table[t_add[row]] (rows sorted by that index: equal
// consecutive targets are summed first) and, when SUB, subtracts them from table[t_sub[row]]
// (unsorted). Negative targets are skipped.
template <class GE, int ROWLEN, bool SUB>
CSMPN_DEV void rl_scatter(const float* sc, int t_add, int t_sub, float* table, int lane) {
    constexpr int R = GE::R, SS = ROWLEN + 4, NC = (ROWLEN + 63) / 64;
    static_for<0, NC>([&](auto cc) {
        const int colx = 64 * cc + lane;
        const bool in = colx < ROWLEN;
        const float* col = sc + (in ? colx : 0);
        auto flush = [&](int target, float a) {
            if (target >= 0 && in) atomicAdd(table + (long)target * ROWLEN + colx, a);
        };
        float acc = 0.f;
        int cur = __builtin_amdgcn_readlane(t_add, GE::lane_of_row(0));
        static_for<0, R / 16>([&](auto gg) {
            constexpr int r0 = 16 * decltype(gg)::value;
            float val[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) val[i] = col[(r0 + i) * SS];
            static_for<0, 16>([&](auto rr) {
                constexpr int row = r0 + decltype(rr)::value;
                const int t = __builtin_amdgcn_readlane(t_add, GE::lane_of_row(row));
                if (t != cur) {
                    flush(cur, acc);
                    cur = t;
                    acc = 0.f;
                }
                acc += val[decltype(rr)::value];
            });
            if constexpr (SUB) {
                static_for<0, 16>([&](auto rr) {
                    constexpr int row = r0 + decltype(rr)::value;
                    flush(__builtin_amdgcn_readlane(t_sub, GE::lane_of_row(row)), -val[decltype(rr)::value]);
                });
            }
        });
        flush(cur, acc);
    });
}

// ---------------------------------------------------------------------------------
// the kernel. NBLK blocks (1 or 2): block 0 has I0 input channels, block 1 has C; all have C
// output channels. MODE_EDGE: I0 = C + A; MODE_NODE: I0 = 2C + T; MODE_PLAIN: I0 = in_features.
// Tile t of R rows belongs to wave (t mod 4) of workgroup ((t / 4) mod gridDim).
template <class ALG, int NOG, int MODE, int NBLK, int I0, bool BWD>
__global__ void __launch_bounds__(64 * kRlWaves, BWD ? 1 : 2) cemlp_rl_kernel(const DevCemlp C_arg, const RowIO io_arg) {
    typedef const char __attribute__((address_space(4))) * KArgPtr;
    const KArgPtr ka = (KArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr size_t kIoOffset = (sizeof(DevCemlp) + alignof(RowIO) - 1) / alignof(RowIO) * alignof(RowIO);
    const DevCemlp& Cd = *(const DevCemlp*)(const char*)ka;
    const RowIO& io = *(const RowIO*)(const char*)(ka + kIoOffset);
    (void)C_arg; (void)io_arg;
    using LY = RlLay<ALG, NOG, NBLK, I0>;
    using GE = RlGeo<ALG, NOG, NBLK, I0>;
    constexpr int D = ALG::D, C = 4 * NOG, R = GE::R, RS = GE::RS, ROW = C * D, PIECE = 4 * D;
    constexpr int NA = MODE == MODE_EDGE ? I0 - C : (MODE == MODE_NODE ? I0 - 2 * C : 0);   // attribute channels
    static_assert(NBLK == 1 || NBLK == 2, "one or two blocks");
    static_assert(NA >= 0, "bad input width");
    static_assert(LY::bwd_total * 4 <= 160 * 1024, "backward LDS footprint");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* lds = smem;
    const int wave = threadIdx.x >> 6;
    const GE ge(threadIdx.x & 63);
    ge.stamp_init();
    float* sc = lds + (BWD ? LY::sc_bwd + wave * LY::scratch : LY::sc_fwd + wave * LY::stage_floats);

    constexpr int PIECE_ = 4 * ALG::D, ROW_ = C * ALG::D;
    // the first tile's indices and first loads travel while the parameters are staged
    const long ntiles = (io.rows + R - 1) / R;
    const long tstride = (long)gridDim.x * kRlWaves;
    struct TileRows { long row, lrow; bool valid; int i_dst, i_src, i_perm; float scale; };
    auto tile_rows = [&](long t) {
        TileRows T;
        T.row = t * R + ge.r;
        T.valid = t < ntiles && T.row < io.rows;
        T.lrow = T.valid ? T.row : 0;   // invalid lanes compute on row 0 and contribute nothing
        T.i_dst = T.i_src = T.i_perm = 0;
        T.scale = 1.0f;
        if constexpr (MODE == MODE_EDGE) {
            T.i_dst = io.seg[0].ia[T.lrow];
            T.i_src = io.seg[0].ib[T.lrow];
            if constexpr (NA > 0) T.i_perm = io.seg[1].ia[T.lrow];
        }
        if constexpr (MODE == MODE_NODE) {   // mean aggregation
            if (io.seg[1].deg) { const int dg = io.seg[1].deg[T.lrow]; T.scale = 1.0f / float(dg > 1 ? dg : 1); }
        }
        return T;
    };
    // backward: the tile's first loads
    auto first_loads = [&](const TileRows& T, f4 (&gout)[ALG::D], f4 (&in1)[ALG::D]) {
        const long srow = MODE == MODE_EDGE ? (long)T.i_dst : T.lrow;
        rl_load_t<ALG>(gout, io.gy + (size_t)srow * ROW_ + ge.og * PIECE_);
        if constexpr (NBLK > 1) rl_load_t<ALG>(in1, io.saved + (size_t)T.lrow * ROW_ + ge.og * PIECE_);
    };
    long tile = (long)blockIdx.x * kRlWaves + wave;
    TileRows Tn = tile_rows(tile);
    f4 gout_n[ALG::D], in1_n[ALG::D];
    if constexpr (BWD) first_loads(Tn, gout_n, in1_n);

    rl_stage_store<LY, 0>(Cd.b[0], lds, threadIdx.x);
    if constexpr (NBLK > 1) rl_stage_store<LY, 1>(Cd.b[1], lds, threadIdx.x);
    __syncthreads();
    ge.stamp(0);
    // parameter-gradient sums of this wave, persistent across its tiles (backward only)
    constexpr int NT0 = (I0 + 15) / 16, G = ALG::G;
    f4 accW1_0[1][NT0][G], accRL_0[LY::MA_RL][1][G];
    f4 accW1_1[1][1][G], accRL_1[LY::MA_RL][1][G];
    float* tot_0 = sc + LY::tot_off;
    float* tot_1 = tot_0 + LY::n_chunk * 256;
    if constexpr (BWD) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
#pragma unroll
            for (int nt = 0; nt < NT0; ++nt) accW1_0[0][nt][g] = splat(0.f);
            accW1_1[0][0][g] = splat(0.f);
#pragma unroll
            for (int ma = 0; ma < LY::MA_RL; ++ma) { accRL_0[ma][0][g] = splat(0.f); accRL_1[ma][0][g] = splat(0.f); }
        }
#pragma unroll
        for (int c = 0; c < NBLK * LY::n_chunk; ++c) *reinterpret_cast<f4*>(tot_0 + ge.r_t + c * 256) = splat(0.f);
    }

    // Software pipeline over the wave's tiles (one wave per SIMD in the backward: nothing else hides a
    // memory round trip). The row indices of tile t+1 are loaded while tile t computes; in the
    // backward the first loads of tile t+1 (incoming gradient, saved block input) are issued BEFORE
    // the scatter atomics of tile t, so that waiting for them does not wait for the atomics.
    for (; tile < ntiles; tile += tstride) {
        const TileRows Tc = Tn;
        const long row = Tc.row, lrow = Tc.lrow;
        const bool valid = Tc.valid;
        const int i_dst = Tc.i_dst, i_src = Tc.i_src, i_perm = Tc.i_perm;
        const float scale = Tc.scale;
        Tn = tile_rows(tile + tstride);
        // ---- MVLinear of block 0 straight from the gathered rows, 8 input channels at a time (every
        // lane of a row loads all of them: the NOG lanes issue the same addresses in one instruction)
        auto mvlinear0 = [&](f4 (&y)[D]) {
            constexpr int W1 = LY::o_W1(0);
            if constexpr (NOG == 2 && MODE != MODE_PLAIN) {
                // 8-channel layers: ALL of the row's inputs are requested before the first one is used (they
                // fit in registers; the attribute loads are pinned to the request point). Chunk by chunk, the
                // attribute loads were issued between the MFMAs of the first chunk and waited for one at a time.
                constexpr int NAA = NA > 0 ? NA : 1;
                float X[C][D], Y[C][D], A[NAA][D];
                if constexpr (MODE == MODE_EDGE) {
                    rl_load_channels<ALG, C>(X, 0, io.seg[0].a + (size_t)i_dst * ROW);
                    rl_load_channels<ALG, C>(Y, 0, io.seg[0].b + (size_t)i_src * ROW);
                    if constexpr (NA > 0) rl_load_channels<ALG, NA, NAA, true>(A, 0, io.seg[1].a + (size_t)i_perm * (NA * D));
#pragma unroll
                    for (int c = 0; c < C; ++c)
#pragma unroll
                        for (int d = 0; d < D; ++d) X[c][d] -= Y[c][d];
                    rl_linear_x<ALG, W1, 0, C>(y, X, lds, ge);
                    if constexpr (NA > 0) rl_linear_x<ALG, W1, C, NA>(y, A, lds, ge);
                } else {
                    rl_load_channels<ALG, C>(X, 0, io.seg[0].a + (size_t)lrow * ROW);
                    rl_load_channels<ALG, C>(Y, 0, io.seg[1].a + (size_t)lrow * ROW);
                    if constexpr (NA > 0) rl_load_channels<ALG, NA, NAA, true>(A, 0, io.seg[2].a + (size_t)lrow * (NA * D));
#pragma unroll
                    for (int c = 0; c < C; ++c)
#pragma unroll
                        for (int d = 0; d < D; ++d) Y[c][d] *= scale;
                    rl_linear_x<ALG, W1, 0, C>(y, X, lds, ge);
                    rl_linear_x<ALG, W1, C, C>(y, Y, lds, ge);
                    if constexpr (NA > 0) rl_linear_x<ALG, W1, 2 * C, NA>(y, A, lds, ge);
                }
                return;
            }
            auto chunks = [&](auto c0, auto n_total, const float* pa, const float* pb, float mul) {
                // input channels c0 .. c0 + n_total from the contiguous row at pa (minus the row at pb)
                static_for<0, (decltype(n_total)::value + 7) / 8>([&](auto q8) {
                    constexpr int off = 8 * q8, NCH = decltype(n_total)::value - off < 8 ? decltype(n_total)::value - off : 8;
                    float X[NCH][D];
                    // edge kernels: the loads of the last (attribute) chunk are pinned to this point (16 channels, stage timers:
                    // edge backward -4 % in one A/B, within noise in the bench line; the node kernels measured +2 % with it)
                    if constexpr (NCH < 8 && MODE == MODE_EDGE) rl_load_channels<ALG, NCH, NCH, true>(X, 0, pa + off * D);
                    else rl_load_channels<ALG, NCH>(X, 0, pa + off * D);
                    if (pb) {
                        float Y[NCH][D];
                        rl_load_channels<ALG, NCH>(Y, 0, pb + off * D);
#pragma unroll
                        for (int c = 0; c < NCH; ++c)
#pragma unroll
                            for (int d = 0; d < D; ++d) X[c][d] -= Y[c][d];
                    }
                    if constexpr (MODE == MODE_NODE) {
#pragma unroll
                        for (int c = 0; c < NCH; ++c)
#pragma unroll
                            for (int d = 0; d < D; ++d) X[c][d] *= mul;
                    }
                    rl_linear_x<ALG, W1, decltype(c0)::value + off, NCH>(y, X, lds, ge);
                });
            };
            if constexpr (MODE == MODE_EDGE) {
                chunks(IC<0>{}, IC<C>{}, io.seg[0].a + (size_t)i_dst * ROW, io.seg[0].b + (size_t)i_src * ROW, 1.0f);
                if constexpr (NA > 0) chunks(IC<C>{}, IC<NA>{}, io.seg[1].a + (size_t)i_perm * (NA * D), nullptr, 1.0f);
            } else if constexpr (MODE == MODE_NODE) {
                chunks(IC<0>{}, IC<C>{}, io.seg[0].a + (size_t)lrow * ROW, nullptr, 1.0f);
                chunks(IC<C>{}, IC<C>{}, io.seg[1].a + (size_t)lrow * ROW, nullptr, scale);
                if constexpr (NA > 0) chunks(IC<2 * C>{}, IC<NA>{}, io.seg[2].a + (size_t)lrow * (NA * D), nullptr, 1.0f);
            } else {
                chunks(IC<0>{}, IC<I0>{}, io.seg[0].a + (size_t)lrow * (I0 * D), nullptr, 1.0f);
            }
        };
        // The B slices of block 0's MVLinear weight gradient: blade d of the input channels og, og + NOG, ...
        // of this lane's row. The row is gathered again (cache hits: the forward recompute has just read it),
        // ALL pieces in flight at once: issued blade by blade inside the gradient loop, the compiler waited
        // for seven groups of loads one after the other.
        constexpr int NT_IN = (I0 + NOG - 1) / NOG, DQ = D / 4;
        // 8-channel kernels: the re-gather is issued in front of the last phase of the block backward. (Not for
        // 16 channels: the register allocation of that variant crashes clang 22's AGPR-copy rewrite pass.)
        constexpr bool kEarly = NOG == 2;
        auto load_input_pieces = [&](f4 (&xin)[NT_IN][DQ]) {
            constexpr int NTC = MODE == MODE_EDGE ? C / NOG : 1;
            f4 ysrc[NTC][DQ];
            static_for<0, NT_IN>([&](auto t) {
                constexpr int cbase = NOG * t;          // segments start at multiples of NOG: one segment per t
                const float* pa;
                if constexpr (MODE == MODE_EDGE) {
                    if constexpr (cbase < C) {
                        pa = io.seg[0].a + (size_t)i_dst * ROW + (cbase + ge.og) * D;
                        const float* pb = io.seg[0].b + (size_t)i_src * ROW + (cbase + ge.og) * D;
#pragma unroll
                        for (int q = 0; q < DQ; ++q) ysrc[t][q] = ld4(pb + 4 * q);
                    } else {
                        const int ca = cbase - C + ge.og;
                        pa = io.seg[1].a + (size_t)i_perm * (NA * D) + (ca < NA ? ca : NA - 1) * D;
                    }
                } else if constexpr (MODE == MODE_NODE) {
                    if constexpr (cbase < C) pa = io.seg[0].a + (size_t)lrow * ROW + (cbase + ge.og) * D;
                    else if constexpr (cbase < 2 * C) pa = io.seg[1].a + (size_t)lrow * ROW + (cbase - C + ge.og) * D;
                    else {
                        const int ca = cbase - 2 * C + ge.og;
                        pa = io.seg[2].a + (size_t)lrow * (NA * D) + (ca < NA ? ca : NA - 1) * D;
                    }
                } else {
                    const int ca = cbase + ge.og;
                    pa = io.seg[0].a + (size_t)lrow * (I0 * D) + (ca < I0 ? ca : I0 - 1) * D;
                }
#pragma unroll
                for (int q = 0; q < DQ; ++q) xin[t][q] = ld4(pa + 4 * q);
            });
            static_for<0, NT_IN>([&](auto t) {
                constexpr int cbase = NOG * t;
                if constexpr (MODE == MODE_EDGE && cbase < C) {
#pragma unroll
                    for (int q = 0; q < DQ; ++q) xin[t][q] -= ysrc[t][q];
                }
                if constexpr (MODE == MODE_NODE && cbase >= C && cbase < 2 * C) {
#pragma unroll
                    for (int q = 0; q < DQ; ++q) xin[t][q] *= scale;
                }
            });
        };
        auto write_input_slice = [&](const f4 (&xin)[NT_IN][DQ], auto d, float* sB) {
            float* p = sB + (ge.og * RS + ge.r);
            static_for<0, NT_IN>([&](auto t) {
                constexpr int cbase = NOG * t;
                const float v = xin[t][int(d) / 4][int(d) % 4];
                if constexpr (cbase + NOG <= I0) p[cbase * RS] = v;
                else if (cbase + ge.og < I0) p[cbase * RS] = v;
            });
        };
        // block forward from the gathered input / from a distributed C-channel input
        auto forward0 = [&](RlFwd<ALG>& S, f4 (&out)[D]) {
#pragma unroll
            for (int d = 0; d < D; ++d) S.y[d] = splat(0.f);
            mvlinear0(S.y);
            ge.stamp(2);
            CSMPN_PHASE();
            rl_block_tail<ALG, NOG, 0>(lds, ge, S, out);
        };
        auto forward1 = [&](const f4 (&in)[D], RlFwd<ALG>& S, f4 (&out)[D]) {
#pragma unroll
            for (int d = 0; d < D; ++d) S.y[d] = splat(0.f);
            rl_linear_d<ALG, NOG, LY::o_W1(1)>(S.y, in, lds, ge);
            ge.stamp(2);
            CSMPN_PHASE();
            rl_block_tail<ALG, NOG, 1>(lds, ge, S, out);
        };

        if constexpr (!BWD) {
            // ------------------------------------------------------------ forward
            f4 out[D];
            {
                RlFwd<ALG> S;
                forward0(S, out);
            }
            if constexpr (NBLK > 1) {
                if (io.save && valid) rl_store_t<ALG>(out, io.save + (size_t)row * ROW + ge.og * PIECE, 1.0f);
                f4 in1[D];
#pragma unroll
                for (int d = 0; d < D; ++d) in1[d] = out[d];
                RlFwd<ALG> S;
                forward1(in1, S, out);
            }
            if constexpr (MODE == MODE_EDGE) {
                if (io.row_store) {
                    if (valid) rl_store_t<ALG>(out, io.agg + (size_t)lrow * ROW + ge.og * PIECE, 1.0f);
                } else {
                    RL_LDS_ORDER();
                    rl_store_t<ALG>(out, sc + ge.g_st, 1.0f);
                    RL_LDS_ORDER();
                    rl_scatter<GE, ROW, false>(sc, valid ? i_dst : -1, -1, io.agg, ge.lane);
                }
                ge.stamp(18);
            } else {
                if (valid) {
                    if (MODE == MODE_NODE && io.resid) {
                        f4 res[D];
                        rl_load_t<ALG>(res, io.resid + (size_t)row * ROW + ge.og * PIECE);
#pragma unroll
                        for (int d = 0; d < D; ++d) out[d] += res[d];
                    }
                    rl_store_t<ALG>(out, io.y + (size_t)row * ROW + ge.og * PIECE, 1.0f);
                }
                ge.stamp(18);
            }
        } else {
            // ------------------------------------------------------------ backward
            f4 gout[D];
#pragma unroll
            for (int d = 0; d < D; ++d) gout[d] = valid ? gout_n[d] : splat(0.f);
            if constexpr (NBLK > 1) {
                // last block first; its input was saved by the forward
                f4 in1[D];
#pragma unroll
                for (int d = 0; d < D; ++d) in1[d] = in1_n[d];
                ge.stamp(1);
                f4 gy[D];
                f4 in1b[D];   // the block input, loaded again: 4*D registers less across most of the block backward
                {
                    RlFwd<ALG> S;
                    f4 unused[D];
                    forward1(in1, S, unused);
                    rl_block_backward<ALG, NOG, 1>(lds, sc, ge, S, gout, gy, tot_1, accRL_1, [&] {
                        if constexpr (kEarly) {
                            asm volatile("" ::: "memory");
                            rl_load_t<ALG>(in1b, io.saved + (size_t)lrow * ROW + ge.og * PIECE);
                        }
                    });
                }
                CSMPN_PHASE();
                {
                    if constexpr (!kEarly) {
                        asm volatile("" ::: "memory");
                        rl_load_t<ALG>(in1b, io.saved + (size_t)lrow * ROW + ge.og * PIECE);
                    }
                    rl_w1_grad<ALG, NOG, 1>(sc, ge, gy, accW1_1, [&](auto d, float* sB) {
                        float* p = sB + ge.s_w;
#pragma unroll
                        for (int i = 0; i < 4; ++i) p[i * RS] = in1b[d][i];
                    });
                }
                ge.stamp(15);
#pragma unroll
                for (int d = 0; d < D; ++d) gout[d] = splat(0.f);
                rl_linear_dt<ALG, NOG, LY::o_W1(1)>(gout, gy, lds, ge);
                ge.stamp(16);
                CSMPN_PHASE();
            }
            constexpr int NGL = GE::NGL0;
            f4 gx[NGL][D];
            {
                f4 gy[D];
                f4 xin[NT_IN][DQ];
                {
                    RlFwd<ALG> S;
                    f4 unused[D];
                    forward0(S, unused);
                    rl_block_backward<ALG, NOG, 0>(lds, sc, ge, S, gout, gy, tot_0, accRL_0, [&] { if constexpr (kEarly) load_input_pieces(xin); });
                }
                CSMPN_PHASE();
                if constexpr (!kEarly) load_input_pieces(xin);
                rl_w1_grad<ALG, NOG, 0>(sc, ge, gy, accW1_0, [&](auto d, float* sB) { write_input_slice(xin, d, sB); });
                ge.stamp(15);
#pragma unroll
                for (int t = 0; t < NGL; ++t)
#pragma unroll
                    for (int d = 0; d < D; ++d) gx[t][d] = splat(0.f);
                rl_linear_xt<ALG, NOG, LY::o_W1(0)>(gx, gy, lds, ge);
                ge.stamp(16);
            }
            // the next tile's first loads go out in front of this tile's stores / atomics
            first_loads(Tn, gout_n, in1_n);
            // gx[t] = d/d(input channels 4(og + NOG t) .. +3)
            if constexpr (MODE == MODE_EDGE) {
                if (io.gx[0]) {
                    if (io.row_store) {
                        if (valid) rl_store_t<ALG>(gx[0], io.gx[0] + (size_t)lrow * ROW + ge.og * PIECE, 1.0f);
                    } else {
                        RL_LDS_ORDER();
                        rl_store_t<ALG>(gx[0], sc + ge.g_st, 1.0f);
                        RL_LDS_ORDER();
                        rl_scatter<GE, ROW, true>(sc, valid ? i_dst : -1, valid ? i_src : -1, io.gx[0], ge.lane);
                    }
                }
                if constexpr (NA > 0) {
                    if (io.gx[1] && valid) {
                        float* base = io.gx[1] + (size_t)i_perm * (NA * D);
                        static_for<1, NGL>([&](auto t) { rl_store_group<ALG>(gx[t], base, 4 * (ge.og + NOG * t) - C, NA); });
                    }
                }
            } else if (valid) {
                // channel group cg = og + NOG t of the concatenated input -> its segment
                static_for<0, NGL>([&](auto t) {
                    const int cg = ge.og + NOG * t;
                    if constexpr (MODE == MODE_NODE) {
                        if (cg < NOG) {
                            if (io.gx[0]) {
                                f4 g0[D];
#pragma unroll
                                for (int d = 0; d < D; ++d) g0[d] = gx[t][d];
                                if (io.resid_bwd) {
                                    f4 res[D];
                                    rl_load_t<ALG>(res, io.gy + (size_t)row * ROW + cg * PIECE);
#pragma unroll
                                    for (int d = 0; d < D; ++d) g0[d] += res[d];
                                }
                                rl_store_t<ALG>(g0, io.gx[0] + (size_t)row * ROW + cg * PIECE, 1.0f);
                            }
                        } else if (cg < 2 * NOG) {
                            if (io.gx[1]) rl_store_t<ALG>(gx[t], io.gx[1] + (size_t)row * ROW + (cg - NOG) * PIECE, scale);
                        } else if (NA > 0 && io.gx[2]) {
                            rl_store_group<ALG>(gx[t], io.gx[2] + (size_t)row * (NA * D), 4 * (cg - 2 * NOG), NA);
                        }
                    } else {
                        if (io.gx[0]) rl_store_group<ALG>(gx[t], io.gx[0] + (size_t)row * (I0 * D), 4 * cg, I0);
                    }
                });
            }
            ge.stamp(18);
        }
    }

    if constexpr (BWD) {
        // block by block: every wave builds the image of its sums in its own scratch; the workgroup adds
        // the four images in wave order and writes its slice (coalesced 16-byte stores)
        float* part = io.rl_partials + (size_t)blockIdx.x * LY::part_total;
        const float* img = lds + LY::sc_bwd;
        static_for<0, NBLK>([&](auto kb) {
            constexpr int n = LY::part_blk(kb);
            static_assert(n <= LY::tot_off && n % 4 == 0, "slice image must fit below the totals");
            if constexpr (kb == 0) rl_partials_image<ALG, NOG, 0>(sc, ge, accW1_0, accRL_0, tot_0);
            else rl_partials_image<ALG, NOG, 1>(sc, ge, accW1_1, accRL_1, tot_1);
            __syncthreads();
            for (int e = 4 * threadIdx.x; e < n; e += 4 * 64 * kRlWaves) {
                f4 v = ld4(img + e);
#pragma unroll
                for (int w = 1; w < kRlWaves; ++w) v += ld4(img + w * LY::scratch + e);
                *reinterpret_cast<f4*>(part + LY::part_off(kb) + e) = v;
            }
            __syncthreads();
        });
    }
#ifdef CSMPN_STAMPS
    ge.stamp(19);
    if (io.stamps && ge.lane == 0) {
        for (int i = 0; i < GE::kStampSlots; ++i) atomicAdd(io.stamps + i, ge.acc[i]);
        atomicAdd(io.stamps + GE::kStampSlots, 1ull);
    }
#endif
}

}  // namespace csmpn
