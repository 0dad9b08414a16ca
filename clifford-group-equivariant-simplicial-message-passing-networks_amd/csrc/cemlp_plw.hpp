// Parity-lane row program for WIDE D = 32 layers (9 .. 32 channels; the convex-hulls width is 28 channels of
// Cl(5,0)): the distribution of cemlp_pl.hpp with one WAVE per group of 8 channels.
//
//   workgroup = NG waves on the same 4-row tile; wave g owns channels 8g .. 8g+7;
//   lane = (row q, channel c of the group, blade parity s) as in cemlp_pl.hpp: a tensor is float t[16].
//
// Wave-local (no communication): gates, normalisation, the geometric product and its backward, the per-channel
// parameter sums. Across the groups:
//   dense mixing    y_og = sum_ig M[og][ig] x_ig. The input groups travel through an LDS exchange buffer
//                   ([group][slot][lane], conflict-free); every (og, ig) block is the 8-rotation / 16-slot VALU
//                   product of cemlp_pl.hpp with its 24 weights per lane read as 6 x 16 bytes from rotation tables
//                   packed once per launch into the workspace (plw_pack_kernel; L2-resident - the weights of a
//                   28-channel CEMLP, 134 KB, do not fit LDS beside the gradient sums);
//   LayerNorm mean  one float per (row, group) through LDS;
//   weight gradients v_mfma_f32_16x16x4_f32 as in cemlp_pl.hpp with B = the input group read from the exchange buffer:
//                   3 x f4 accumulators per (og, ig) block in AGPRs. The backward is launched once per block
//                   (BLK = 1, then BLK = 0, d/d(block-1 input) handed over through a row scratch) so that one
//                   launch holds the accumulators of one block only.
// Block 0 reads its input "chunks" (8 channels of one input segment: h[dst] - h[src], attributes, aggregate, ...)
// straight from global memory in every wave, so input segments need no alignment to the groups.
#pragma once
#include "cemlp_pl.hpp"

namespace csmpn {

// compile-time description of one kernel family member
template <class ALG, int NG_, int C_, int MODE_, int NA_, int NBLK_ = 2>
struct PlwCfg {
    using P = PS<ALG>;
    static constexpr int NG = NG_, C = C_, MODE = MODE_, NA = NA_, CP = 8 * NG_, NBLK = NBLK_;
    static constexpr int D = ALG::D, DL = P::DL, GC = P::GC, G = ALG::G, NP = ALG::P, QP = P::QP, N = ALG::n;
    static constexpr int ROW = C * D;
    static_assert(C > 8 * (NG - 1) && C <= 8 * NG, "NG = ceil(C / 8)");
    static constexpr int WG_PER_CU_BWD = 4 / NG_ > 0 ? 4 / NG_ : 1;   // one wave per SIMD in the backward
    // HIP's second __launch_bounds__ argument counts WAVES PER SIMD (execution unit), not workgroups per CU: W workgroups of
    // NG waves per CU are W NG / 4 waves per SIMD. (Rounds 1-2 passed the workgroup count: the one- and two-wave variants were
    // compiled for 4 / 2 waves per SIMD in the backward - 128 / 256 registers, 0.5-1.3 KB of scratch per lane - instead of one.)
    static constexpr int waves_per_simd(int wgs_per_cu) { return wgs_per_cu * NG_ / 4 > 0 ? wgs_per_cu * NG_ / 4 : 1; }
    static_assert(NA >= 0 && NA <= 8, "attribute channels fit one chunk");
    static_assert(NBLK == 1 || NBLK == 2, "one or two blocks");
    static_assert(MODE != MODE_PLAIN || NA > 0, "MODE_PLAIN: NA = input channels (one chunk)");
    // input chunks of block 0: (first channel in the concatenated input, valid channels)
    static constexpr int NSEG = MODE == MODE_EDGE ? 1 : (MODE == MODE_NODE ? 2 : 0);   // full-width segments in front of the attribute chunk
                                                                                     // (MODE_PLAIN: the <= 8 input channels are that chunk)
    static constexpr int NCH0 = NSEG * NG + (NA > 0 ? 1 : 0);
    static constexpr int I0 = NSEG * C + NA;
    static constexpr int chunk_base(int j) { return j < NSEG * NG ? (j / NG) * C + 8 * (j % NG) : NSEG * C; }
    static constexpr int chunk_valid(int j) {
        return j < NSEG * NG ? (C - 8 * (j % NG) < 8 ? C - 8 * (j % NG) : 8) : NA;
    }
    static constexpr int Iof(int k) { return k == 0 ? I0 : C; }
    static constexpr int nch(int k) { return k == 0 ? NCH0 : NG; }
    // rotation tables in the workspace (floats): pair (a, b) of a table with NB columns at ((a * NB + b) * 16 + n) * 24
    static constexpr int PAIR = 16 * 24;
    static constexpr int t_W1(int k) { return k == 0 ? 0 : tab_blk(0); }            // [og][chunk]
    static constexpr int t_W1t(int k) { return t_W1(k) + NG * nch(k) * PAIR; }      // [chunk][og]
    static constexpr int t_WR(int k) { return t_W1t(k) + NG * nch(k) * PAIR; }      // [og][ig]
    static constexpr int t_WRt(int k) { return t_WR(k) + NG * NG * PAIR; }          // [ig][og]
    static constexpr int t_WL(int k) { return t_WRt(k) + NG * NG * PAIR; }
    static constexpr int t_WLt(int k) { return t_WL(k) + NG * NG * PAIR; }
    static constexpr int tab_blk(int k) { return (2 * NG * nch(k) + 4 * NG * NG) * PAIR; }
    static constexpr int tab_total = tab_blk(0) + (NBLK > 1 ? tab_blk(1) : 0);
    // LDS (floats)
    static constexpr int par_floats = 3 * CP + 3 * CP * G + CP * NP;
    static constexpr int p_b1(int k) { return k * par_floats; }
    static constexpr int p_bL(int k) { return p_b1(k) + CP; }
    static constexpr int p_la(int k) { return p_bL(k) + CP; }
    static constexpr int p_sa(int k) { return p_la(k) + CP; }
    static constexpr int p_sb(int k) { return p_sa(k) + CP * G; }
    static constexpr int p_sg(int k) { return p_sb(k) + CP * G; }
    static constexpr int p_w(int k) { return p_sg(k) + CP * G; }
    static constexpr int store_total = 2 * par_floats;
    // layout: parameters | LayerNorm exchange | staging tile | exchange buffers 0, 1 (| 2 | running sums: backward)
    static constexpr int RS = ROW + 4;
    static constexpr int ln_off = store_total;              // [2][NG][4 rows]
    static constexpr int st_off = ln_off + 2 * NG * kPlRows + 8;   // staging tile [4 rows][ROW + 4]
    static constexpr int XB = NG * DL * 64;                 // one exchange buffer [group][slot][lane]
    static constexpr int x_base = (st_off + kPlRows * RS + 3) & ~3;
    static constexpr int x_off(int b) { return x_base + b * XB; }
    static constexpr int NXB = 3;
    static constexpr int fwd_total = x_off(2);              // the forward uses buffers 0 and 1: two workgroups per CU
    static constexpr int n_sums = 3 + 3 * GC + 2 * QP;
    static constexpr int tot_off = x_off(NXB);              // backward: lane-private running sums [slot][thread]
    static constexpr int bwd_total = tot_off + n_sums * 64 * NG;
};

// ---------------------------------------------------------------------------------
// rotation tables -> workspace. One thread per table float.
template <class CF, class ALG>
__global__ void plw_pack_kernel(const DevCemlp Cd, float* tabs) {
    constexpr int NG = CF::NG, C = CF::C, G = CF::G, GC = CF::GC, PAIR = CF::PAIR;
    const int probe = pl_dpp_i<0x122>((int)(threadIdx.x & 15));
    const int dir = (((probe - (int)(threadIdx.x & 15)) & 15) == 2) ? 1 : -1;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= CF::tab_total) return;
    const int k = t >= CF::tab_blk(0) ? 1 : 0;
    int e = (int)(t - (k ? CF::tab_blk(0) : 0));
    const DevBlock& B = Cd.b[k];
    const int nch = k == 0 ? CF::NCH0 : NG, I = k == 0 ? CF::I0 : C;
    // which table
    const float* W;
    int Iw, NB, which;   // which: 0 W1, 1 W1t, 2 WR, 3 WRt, 4 WL, 5 WLt
    const int s1 = NG * nch * PAIR, s2 = NG * NG * PAIR;
    if (e < s1) { which = 0; }
    else if (e < 2 * s1) { which = 1; e -= s1; }
    else { e -= 2 * s1; which = 2 + e / s2; e %= s2; }
    const bool tr = which & 1;
    if (which < 2) { W = B.W1; Iw = I; } else { W = which < 4 ? B.WR : B.WL; Iw = C; }
    const bool chunked = which < 2;
    NB = which == 0 ? nch : (which == 1 ? NG : NG);
    const int f = e % 24, n = (e / 24) % 16, pair = e / PAIR;
    const int a = pair / NB, b = pair % NB;
    const int r = f / GC, cls = f % GC, c = n >> 1, s = n & 1;
    const int sc = (c + dir * r) & 7;
    const int g = s ? ALG::n - 2 * cls : 2 * cls;
    // forward table [og = a][in = b]: W[8a + c][base(b) + sc]; transposed [in = a][og = b]: W[8b + sc][base(a) + c]
    int o, cin;
    bool ok;
    if (!tr) {
        const int base = chunked ? (k == 0 ? CF::chunk_base(b) : 8 * b) : 8 * b;
        const int nv = chunked ? (k == 0 ? CF::chunk_valid(b) : (C - 8 * b < 8 ? C - 8 * b : 8)) : (C - 8 * b < 8 ? C - 8 * b : 8);
        o = 8 * a + c; cin = base + sc; ok = o < C && sc < nv;
    } else {
        const int base = chunked ? (k == 0 ? CF::chunk_base(a) : 8 * a) : 8 * a;
        const int nv = chunked ? (k == 0 ? CF::chunk_valid(a) : (C - 8 * a < 8 ? C - 8 * a : 8)) : (C - 8 * a < 8 ? C - 8 * a : 8);
        o = 8 * b + sc; cin = base + c; ok = o < C && c < nv;
    }
    tabs[t] = ok ? W[((size_t)o * Iw + cin) * G + g] : 0.f;
}

// the 24 weights of this lane for one (output group, input group) block: 6 x 16 bytes from the packed tables (L2)
CSMPN_DEV void plw_ldw(f4 (&wv)[6], const float* tp) {
#pragma unroll
    for (int e = 0; e < 6; ++e) wv[e] = pl_ld4(tp + 4 * e);
}
// acc[j] += sum_r T[r][class(j)] * rot_r(x[j])
template <class ALG>
CSMPN_DEV void plw_mix_w(float (&acc)[PS<ALG>::DL], const float (&x)[PS<ALG>::DL], const f4 (&wv)[6]) {
    using P = PS<ALG>;
    constexpr int GC = P::GC, DL = P::DL;
    pl_dpp_ready(x);
    static_for<0, 8>([&](auto r) {
        static_for<0, DL>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            constexpr int f = decltype(r)::value * GC + P::t.cls[j];
            pl_fmac_rot<decltype(r)::value>(acc[j], x[j], wv[f / 4][f % 4]);
        });
        pl_pin<0, DL>(acc);
    });
}
template <class ALG>
CSMPN_DEV void plw_mix2_w(float (&accA)[PS<ALG>::DL], float (&accB)[PS<ALG>::DL], const float (&x)[PS<ALG>::DL],
                          const f4 (&wa)[6], const f4 (&wb)[6]) {
    using P = PS<ALG>;
    constexpr int GC = P::GC, DL = P::DL;
    pl_dpp_ready(x);
    static_for<0, 8>([&](auto r) {
        static_for<0, DL>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            constexpr int f = decltype(r)::value * GC + P::t.cls[j];
            pl_fmac_rot<decltype(r)::value>(accA[j], x[j], wa[f / 4][f % 4]);
            pl_fmac_rot<decltype(r)::value>(accB[j], x[j], wb[f / 4][f % 4]);
        });
        pl_pin<0, DL>(accA);
        pl_pin<0, DL>(accB);
    });
}
// sum over `count` blocks whose tables are `stride` floats apart, the next block's weights in flight while the
// current one is computed; operand(i, x) delivers the i-th input group
template <class ALG, class F>
CSMPN_DEV void plw_mix_loop(float (&acc)[PS<ALG>::DL], const float* tp0, int stride, int count, F&& operand) {
    f4 wn[6];
    plw_ldw(wn, tp0);
    for (int i = 0; i < count; ++i) {
        f4 wv[6];
#pragma unroll
        for (int e = 0; e < 6; ++e) wv[e] = wn[e];
        if (i + 1 < count) plw_ldw(wn, tp0 + (size_t)(i + 1) * stride);
        float x[PS<ALG>::DL];
        operand(i, x);
        plw_mix_w<ALG>(acc, x, wv);
    }
}

// exchange buffer access: this wave's tensor -> [group][slot][lane]; any group's tensor at this lane position
template <class ALG>
CSMPN_DEV void plw_put(float* xb, int group, int lane, const float (&x)[PS<ALG>::DL]) {
    float* p = xb + group * (PS<ALG>::DL * 64) + lane;
#pragma unroll
    for (int j = 0; j < PS<ALG>::DL; ++j) p[j * 64] = x[j];
}
template <class ALG>
CSMPN_DEV void plw_get(float (&x)[PS<ALG>::DL], const float* xb, int group, int lane) {
    const float* p = xb + group * (PS<ALG>::DL * 64) + lane;
#pragma unroll
    for (int j = 0; j < PS<ALG>::DL; ++j) x[j] = p[j * 64];
}

// sum of one float per (row, group) over the groups; result in every wave. slot: which of the two LN buffers.
template <class CF>
CSMPN_DEV float plw_group_sum(float* lds, int slot, int wave, int q, float v) {
    float* p = lds + CF::ln_off + slot * CF::NG * kPlRows;
    p[wave * kPlRows + q] = v;          // all 16 lanes of a row hold the same value
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < CF::NG; ++g) t += p[g * kPlRows + q];
    return t;
}

// block forward behind the MVLinear (S.y = MVLinear output without bias); mixing through the exchange buffers.
// Barriers: every wave of the workgroup executes this function for the same tile.
// SAVED (backward under CSMPN_FLAG_SAVE_STATE): y, R and s come from the forward (pl_store_state / PlSaved, cemlp_pl.hpp): the
// recompute keeps the gates, the denominators and the layer norm's mean; z still goes to exchange buffer 0 (the weight
// gradients of linear_left / right read it there).
template <class ALG, class CF, int K, bool SAVED = false>
CSMPN_DEV void plw_block_tail(float* lds, const float* tabs, const PlGeo<ALG>& ge, int wave, bool cvalid,
                              PlState<ALG>& S, float (&out)[PS<ALG>::DL], const PlSaved<ALG>* sv = nullptr) {
    using P = PS<ALG>;
    constexpr int DL = P::DL, GC = P::GC, G = ALG::G, NG = CF::NG;
    const int c = 8 * wave + ge.c;       // channel in the (padded) layer
    if constexpr (SAVED) {
#pragma unroll
        for (int j = 0; j < DL; ++j) {
            S.y[j] = cvalid ? sv->y[j / 4][j % 4] : 0.f;
            S.R[j] = cvalid ? sv->R[j / 4][j % 4] : 0.f;
            S.s[j] = cvalid ? sv->s[j / 4][j % 4] : 0.f;
        }
    } else {
        if (ge.s == 0) S.y[0] += lds[CF::p_b1(K) + c];
    }
    float z[DL];
    static_for<0, GC>([&](auto k) {
        constexpr int j0 = P::t.cstart[k], j1 = P::t.cstart[k + 1];
        float u = 0.f;
        static_for<j0, j1>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            u += ge.template qs<j>() * S.y[j] * S.y[j];
        });
        if constexpr (k == 0) {
            if (ge.s == 0) u = S.y[0];
        }
        const int pg = c * G + ge.grade(k);
        S.gate[k] = sigmoidf(lds[CF::p_sa(K) + pg] * u + lds[CF::p_sb(K) + pg]);
        static_for<j0, j1>([&](auto jj) { z[decltype(jj)::value] = S.gate[k] * S.y[decltype(jj)::value]; });
    });
    CSMPN_PHASE();
    // linear_right / linear_left over all groups
    float L[DL];
    if constexpr (!SAVED) {
#pragma unroll
        for (int j = 0; j < DL; ++j) { S.R[j] = 0.f; L[j] = 0.f; }
    }
    float* xb = lds + CF::x_off(0);
    // no barrier in front of the put: every earlier reader of buffer 0 (the previous block's / tile's mixing and
    // weight-gradient reads) is followed by a workgroup barrier in program order (LayerNorm exchange, gradient
    // exchanges, row stores), so all waves have left those reads before any wave gets here
    plw_put<ALG>(xb, wave, ge.lane, z);
    __syncthreads();
    if constexpr (!SAVED) {
        const float* tr = tabs + CF::t_WR(K) + (wave * NG * 16 + ge.n) * 24;
        const float* tl = tabs + CF::t_WL(K) + (wave * NG * 16 + ge.n) * 24;
        f4 ran[6], lan[6];
        plw_ldw(ran, tr);
        plw_ldw(lan, tl);
        for (int ig = 0; ig < NG; ++ig) {
            f4 ra[6], la_[6];
#pragma unroll
            for (int e = 0; e < 6; ++e) { ra[e] = ran[e]; la_[e] = lan[e]; }
            if (ig + 1 < NG) {
                plw_ldw(ran, tr + (ig + 1) * CF::PAIR);
                plw_ldw(lan, tl + (ig + 1) * CF::PAIR);
            }
            float zi[DL];
            plw_get<ALG>(zi, xb, ig, ge.lane);
            plw_mix2_w<ALG>(S.R, L, zi, ra, la_);
        }
    }
    if constexpr (!SAVED) { if (ge.s == 0) L[0] += lds[CF::p_bL(K) + c]; }
    CSMPN_PHASE();
    float r[DL];
    static_for<0, GC>([&](auto k) {
        constexpr int j0 = P::t.cstart[k], j1 = P::t.cstart[k + 1];
        float qq = 0.f;
        static_for<j0, j1>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            qq += ge.template qs<j>() * S.R[j] * S.R[j];
        });
        const float sg = lds[CF::p_sg(K) + c * G + ge.grade(k)];
        const float m = sg * (smooth_abs_sqrt1(qq) - 1.0f) + 1.0f;
        S.invden[k] = fast_rcp(m + kEps);
        static_for<j0, j1>([&](auto jj) { r[decltype(jj)::value] = S.R[decltype(jj)::value] * S.invden[k]; });
    });
    CSMPN_PHASE();
    if constexpr (!SAVED) {
        pl_weighted_gp<ALG>(L, z, r, lds + CF::p_w(K) + c * ALG::P, ge);
#pragma unroll
        for (int j = 0; j < DL; ++j) S.s[j] = cvalid ? L[j] * kInvSqrt2 : 0.f;
    }
    CSMPN_PHASE();
    float qs = 0.f;
    static_for<0, DL>([&](auto jj) {
        constexpr int j = decltype(jj)::value;
        qs += ge.template qs<j>() * S.s[j] * S.s[j];
    });
    qs += pl_partner(qs);
    S.qs = qs;
    S.nl = smooth_abs_sqrt1(qs);
    const float part = pl_chan_sum(cvalid ? S.nl : 0.f);
    const float tot = plw_group_sum<CF>(lds, 0, wave, ge.q, part);
    S.invMn = fast_rcp(tot * (1.0f / float(CF::C)) + kEps);
    const float la = lds[CF::p_la(K) + c];
#pragma unroll
    for (int j = 0; j < DL; ++j) out[j] = la * S.s[j] * S.invMn;
}

// ---------------------------------------------------------------------------------
// forward kernel: two blocks of C channels. BWD kernels: see cemlp_plw_bwd below.
template <class ALG, class CF>
__global__ void __launch_bounds__(64 * CF::NG, CF::NG >= 3 ? (CF::fwd_total * 4 <= 80 * 1024 ? 2 * CF::WG_PER_CU_BWD : CF::WG_PER_CU_BWD)
                                                             : CF::waves_per_simd(CF::fwd_total * 4 <= 80 * 1024 ? 2 * CF::WG_PER_CU_BWD : CF::WG_PER_CU_BWD))
    cemlp_plw_fwd_kernel(const DevCemlp C_arg, const RowIO io_arg) {
    typedef const char __attribute__((address_space(4))) * KArgPtr;
    const KArgPtr ka = (KArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr size_t kIoOffset = (sizeof(DevCemlp) + alignof(RowIO) - 1) / alignof(RowIO) * alignof(RowIO);
    const DevCemlp& Cd = *(const DevCemlp*)(const char*)ka;
    const RowIO& io = *(const RowIO*)(const char*)(ka + kIoOffset);
    (void)C_arg; (void)io_arg;
    using P = PS<ALG>;
    constexpr int D = ALG::D, DL = P::DL, G = ALG::G, NG = CF::NG, C = CF::C, CP = CF::CP, ROW = CF::ROW, RS = CF::RS;
    constexpr int MODE = CF::MODE, NA = CF::NA, NT = 64 * NG, NCH0 = CF::NCH0;
    static_assert(CF::fwd_total * 4 <= 160 * 1024, "LDS footprint");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* lds = smem;
    const int wave = threadIdx.x >> 6;
    const PlGeo<ALG> ge(threadIdx.x & 63);
    const bool cvalid = 8 * wave + ge.c < C;
    const float* tabs = io.plw_tabs;
    float* stg = lds + CF::st_off;
    // per-channel parameters -> LDS (zero beyond C)
    static_for<0, CF::NBLK>([&](auto kk) {
        constexpr int K = decltype(kk)::value;
        const DevBlock& B = Cd.b[K];
        for (int e = threadIdx.x; e < CP; e += NT) {
            const bool ok = e < C;
            lds[CF::p_b1(K) + e] = (ok && B.has_b1) ? B.b1[e] : 0.f;
            lds[CF::p_bL(K) + e] = ok ? B.bL[e] : 0.f;
            lds[CF::p_la(K) + e] = ok ? B.la[e] : 0.f;
        }
        for (int e = threadIdx.x; e < CP * G; e += NT) {
            const bool ok = e < C * G;
            lds[CF::p_sa(K) + e] = ok ? B.sa[e] : 0.f;
            lds[CF::p_sb(K) + e] = ok ? B.sb[e] : 0.f;
            lds[CF::p_sg(K) + e] = ok ? sigmoidf(B.an[e]) : 0.5f;
        }
        for (int e = threadIdx.x; e < CP * ALG::P; e += NT) lds[CF::p_w(K) + e] = e < C * ALG::P ? B.w[e] : 0.f;
    });
    __syncthreads();

    const long ntiles = (io.rows + kPlRows - 1) / kPlRows;
    // Fused simplex embedding (MODE_PLAIN with io.emb_nperm > 0): a workgroup takes GT consecutive tiles = whole simplices
    // (GT * 4 rows = lcm(4, emb_nperm)), so that the sum over the emb_nperm vertex orders of a simplex stays with one
    // workgroup, in registers, in row order (no atomics).
    const int emb_np = MODE == MODE_PLAIN ? io.emb_nperm : 0;
    const int GT = emb_np == 6 ? 3 : 1;
    const long ngroups = (ntiles + GT - 1) / GT;
    for (long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    f4 eacc[kPlRows];
#pragma unroll
    for (int i = 0; i < kPlRows; ++i) eacc[i] = f4{0.f, 0.f, 0.f, 0.f};
    for (int gt_i = 0; gt_i < GT; ++gt_i) {
        const long tile = grp * GT + gt_i;
        if (tile >= ntiles) break;
        const long row = tile * kPlRows + ge.q;
        const bool valid = row < io.rows;
        const long lrow = valid ? row : 0;
        int i_dst = -1, i_src = -1, i_perm = 0;
        float scale = 1.0f;
        if (valid) {
            if constexpr (MODE == MODE_EDGE) {
                i_dst = io.seg[0].ia[row];
                i_src = io.seg[0].ib[row];
                if constexpr (NA > 0) i_perm = io.seg[1].ia[row];
            } else if constexpr (MODE == MODE_NODE) {
                if (io.seg[1].deg) {
                    const int dg = io.seg[1].deg[row];
                    scale = 1.0f / float(dg > 1 ? dg : 1);
                }
            }
        }
        // chunk j of the block-0 input (8 channels of one segment) at this lane's channel position
        auto load_chunk = [&](int j, float (&x)[DL]) {
            const int seg = j / NG, grp = j % NG;
            const bool attr = j >= CF::NSEG * NG;
            const int ch = attr ? ge.c : 8 * grp + ge.c;
            const bool on = valid && ch < (attr ? NA : C);
            const int co = (on ? ch : 0) * D;
            if constexpr (MODE == MODE_EDGE) {
                if (!attr) {
                    pl_load_diff<ALG>(x, io.seg[0].a + (size_t)(valid ? i_dst : 0) * ROW + co,
                                      io.seg[0].b + (size_t)(valid ? i_src : 0) * ROW + co, ge.s, on ? 1.0f : 0.0f);
                } else {
                    pl_load<ALG>(x, io.seg[1].a + (size_t)(valid ? i_perm : 0) * (NA * D) + co, ge.s, on ? 1.0f : 0.0f);
                }
            } else if constexpr (MODE == MODE_NODE) {
                if (attr) pl_load<ALG>(x, io.seg[2].a + (size_t)lrow * (NA * D) + co, ge.s, on ? 1.0f : 0.0f);
                else if (seg == 0) pl_load<ALG>(x, io.seg[0].a + (size_t)lrow * ROW + co, ge.s, on ? 1.0f : 0.0f);
                else pl_load<ALG>(x, io.seg[1].a + (size_t)lrow * ROW + co, ge.s, on ? scale : 0.0f);
            } else {
                (void)seg;
                if (emb_np) {   // channel ch = vertex ch / emb_k of this row's vertex order, its feature channel ch % emb_k
                    const int v = (on ? ch : 0) / io.emb_k, kk = (on ? ch : 0) % io.emb_k;
                    const long vr = min(max(io.emb_verts[lrow * io.emb_nv + v], 0), io.emb_nrows - 1);
                    pl_load<ALG>(x, io.seg[0].a + ((size_t)vr * io.emb_k + kk) * D, ge.s, on ? 1.0f : 0.0f);
                } else {
                    pl_load<ALG>(x, io.seg[0].a + (size_t)lrow * (NA * D) + co, ge.s, on ? 1.0f : 0.0f);
                }
            }
        };
        PlState<ALG> S;
#pragma unroll
        for (int j = 0; j < DL; ++j) S.y[j] = 0.f;
        // the first 2 NG input chunks are gathered once per workgroup (wave w: chunks w, w + NG) and handed round through
        // the two exchange buffers, free at the top of a tile; further chunks (the node program's attributes) per wave
        constexpr int NSH = NCH0 < 2 * NG ? NCH0 : 2 * NG;
        float* xbc = lds + CF::x_off(0);
        for (int j = wave; j < NSH; j += NG) {
            float x[DL];
            load_chunk(j, x);
            plw_put<ALG>(xbc + (j / NG) * CF::XB, j % NG, ge.lane, x);
        }
        __syncthreads();
        plw_mix_loop<ALG>(S.y, tabs + CF::t_W1(0) + (wave * NCH0 * 16 + ge.n) * 24, CF::PAIR, NCH0,
                          [&](int j, float (&x)[DL]) {
                              if (j < NSH) plw_get<ALG>(x, xbc + (j / NG) * CF::XB, j % NG, ge.lane);
                              else load_chunk(j, x);
                          });
        __syncthreads();                  // the chunk slots of buffer 0 are read no more (the tail puts z there)
        float out[DL];
        plw_block_tail<ALG, CF, 0>(lds, tabs, ge, wave, cvalid, S, out);
        // CSMPN_FLAG_SAVE_STATE (EGCL stages, fused two-block embedding): the blocks' s, y, R -> regions 2 + K, 4 + K, 6 + K of the saved buffer, lane
        // order, whole tiles (pl_store_state, cemlp_pl.hpp)
        const bool save_s = CF::NBLK > 1 && io.save_state != 0 && io.save != nullptr && valid && cvalid;
        constexpr int ROWP = CP * D;   // a tile slot = (tile, channel group): kPlRows x 8 channels x D floats
        const size_t s_off = ((size_t)tile * NG + wave) * (kPlRows * 8 * D) + 4 * ge.lane;
        if (save_s) pl_store_state<ALG, ROW, ROWP>(io.save, io.rows, 0, s_off, S);
        if constexpr (CF::NBLK > 1) {
            // block-1 input: to the exchange buffer (and to HBM for the backward)
            float* xb1 = lds + CF::x_off(1);
            plw_put<ALG>(xb1, wave, ge.lane, out);
            if (io.save) pl_stage<ALG>(stg + 8 * wave * D, out, ge, RS, cvalid);
            __syncthreads();
            if (io.save) {
                for (int r = 0; r < kPlRows; ++r) {
                    const long rr = tile * kPlRows + r;
                    if (rr < io.rows)
                        for (int e = 4 * threadIdx.x; e < ROW; e += 4 * NT)
                            *reinterpret_cast<f4*>(io.save + (size_t)rr * ROW + e) = pl_ld4(stg + r * RS + e);
                }
            }
#pragma unroll
            for (int j = 0; j < DL; ++j) S.y[j] = 0.f;
            plw_mix_loop<ALG>(S.y, tabs + CF::t_W1(1) + (wave * NG * 16 + ge.n) * 24, CF::PAIR, NG,
                              [&](int ig, float (&xi)[DL]) { plw_get<ALG>(xi, xb1, ig, ge.lane); });
            plw_block_tail<ALG, CF, 1>(lds, tabs, ge, wave, cvalid, S, out);
            if (save_s) pl_store_state<ALG, ROW, ROWP>(io.save, io.rows, 1, s_off, S);
        }
        if constexpr (MODE == MODE_NODE) {
            if (io.resid) {
                float res[DL];
                pl_load<ALG>(res, io.resid + (size_t)lrow * ROW + (cvalid ? 8 * wave + ge.c : 0) * D, ge.s, cvalid ? 1.0f : 0.0f);
#pragma unroll
                for (int j = 0; j < DL; ++j) out[j] += res[j];
            }
        }
        __syncthreads();                  // the staging tile's readers (save copy) are done
        pl_stage<ALG>(stg + 8 * wave * D, out, ge, RS, cvalid);
        __syncthreads();
        if constexpr (MODE == MODE_EDGE) {
            // scatter-add whole rows to agg[dst]; equal consecutive targets are summed first
            int tg[kPlRows];
#pragma unroll
            for (int r = 0; r < kPlRows; ++r) tg[r] = __builtin_amdgcn_readlane(i_dst, 16 * r);
            if (io.row_store) {   // deterministic mode: the message rows go to an [E, C, D] table in sorted edge order
                for (int r = 0; r < kPlRows; ++r) {
                    const long rr = tile * kPlRows + r;
                    if (rr < io.rows)
                        for (int e = 4 * threadIdx.x; e < ROW; e += 4 * NT)
                            *reinterpret_cast<f4*>(io.agg + (size_t)rr * ROW + e) = pl_ld4(stg + r * RS + e);
                }
            } else
            for (int col = threadIdx.x; col < ROW; col += NT) {
                float acc = 0.f;
                int cur = tg[0];
#pragma unroll
                for (int r = 0; r < kPlRows; ++r) {
                    if (tg[r] != cur) {
                        if (cur >= 0) atomicAdd(io.agg + (long)cur * ROW + col, acc);
                        cur = tg[r];
                        acc = 0.f;
                    }
                    acc += stg[r * RS + col];
                }
                if (cur >= 0) atomicAdd(io.agg + (long)cur * ROW + col, acc);
            }
        } else if (emb_np) {
            // rows of the group -> running sums of their simplices (slot = simplex index inside the group: <= 4 of them)
            static_assert(MODE != MODE_PLAIN || 4 * NT >= ROW, "one 16-byte column piece per thread");
            const int e = 4 * threadIdx.x;
            if (e < ROW) {
                const long s0 = grp * GT * kPlRows / emb_np;   // first simplex of the group
                for (int r = 0; r < kPlRows; ++r) {
                    const long rr = tile * kPlRows + r;
                    if (rr < io.rows) {
                        const int slot = (int)(rr / emb_np - s0);
                        const f4 v = pl_ld4(stg + r * RS + e);
#pragma unroll
                        for (int i = 0; i < kPlRows; ++i) if (i == slot) eacc[i] += v;
                    }
                }
            }
        } else {
            for (int r = 0; r < kPlRows; ++r) {
                const long rr = tile * kPlRows + r;
                if (rr < io.rows)
                    for (int e = 4 * threadIdx.x; e < ROW; e += 4 * NT)
                        *reinterpret_cast<f4*>(io.y + (size_t)rr * ROW + e) = pl_ld4(stg + r * RS + e);
            }
        }
        __syncthreads();                  // staging tile and exchange buffers free for the next tile
    }
    if (emb_np) {   // the group's simplices: one output row each
        const int e = 4 * threadIdx.x;
        const long s0 = grp * GT * kPlRows / emb_np, nsimp = io.rows / emb_np;
        const int per_group = GT * kPlRows / emb_np;
        if (e < ROW) {
#pragma unroll
            for (int i = 0; i < kPlRows; ++i)
                if (i < per_group && s0 + i < nsimp) *reinterpret_cast<f4*>(io.y + (size_t)(s0 + i) * ROW + e) = eacc[i];
        }
    }
    }
}


// ---------------------------------------------------------------------------------
// backward

template <int NT>
CSMPN_DEV void plw_sum_add(float* tot, int slot, float v) {
    float* p = tot + slot * NT;
    *p = *p + v;
}

// block backward from d/d(out) to d/d(MVLinear output). Needs: S (recomputed forward state), z of every group in
// exchange buffer 0 (left there by plw_block_tail). Accumulates the WR / WL gradient tiles per input group and the
// small sums.
template <class ALG, class CF, int K>
CSMPN_DEV void plw_block_backward(float* lds, const float* tabs, const PlGeo<ALG>& ge, int wave, bool cvalid,
                                  const PlState<ALG>& S, const float (&gout)[PS<ALG>::DL], float (&gy)[PS<ALG>::DL],
                                  float* tot, f4 (&accWR)[CF::NG][PS<ALG>::GC], f4 (&accWL)[CF::NG][PS<ALG>::GC]) {
    // the 3 + 3 GC per-channel running sums of this thread (LDS): read here, written at the end of the block - as
    // read-modify-writes where the values are produced each one stalled the wave for an LDS round trip
    float sums[3 + 3 * PS<ALG>::GC];
#pragma unroll
    for (int i = 0; i < 3 + 3 * PS<ALG>::GC; ++i) sums[i] = tot[i * (64 * CF::NG)];
    using P = PS<ALG>;
    using SI = PlSumIdx<ALG>;
    constexpr int DL = P::DL, GC = P::GC, G = ALG::G, NG = CF::NG, NT = 64 * NG;
    const int c = 8 * wave + ge.c;
    const float la = lds[CF::p_la(K) + c];
    float* xb0 = lds + CF::x_off(0);
    float* xb2 = lds + CF::x_off(2);
    // ---- MVLayerNorm backward
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < DL; ++j) dot += gout[j] * S.s[j];
    sums[SI::la] += (dot * S.invMn);
    dot += pl_partner(dot);
    const float part = pl_chan_sum(cvalid ? -(la * dot) * S.invMn * S.invMn : 0.f);
    const float gMn = plw_group_sum<CF>(lds, 1, wave, ge.q, part);
    const float inl = fast_rcp(S.nl);
    const float gqs = (gMn * (1.0f / float(CF::C))) * (0.5f * S.qs) * (inl * inl * inl);
    float ggp[DL];
    static_for<0, DL>([&](auto jj) {
        constexpr int j = decltype(jj)::value;
        const float gs = (la * gout[j]) * S.invMn + gqs * (2.0f * ge.template qs<j>()) * S.s[j];
        ggp[j] = cvalid ? gs * kInvSqrt2 : 0.f;
    });
    sums[SI::bL] += (ggp[0]);
    CSMPN_PHASE();
    // ---- linear_left: d/dz and gWL
    float gz[DL];
#pragma unroll
    for (int j = 0; j < DL; ++j) gz[j] = 0.f;
    // (no barrier in front: the previous readers of buffer 2 - the last tile's d/d(MVLinear output) exchange - are
    // followed by the row-store barriers and this tile's forward barriers)
    plw_put<ALG>(xb2, wave, ge.lane, ggp);
    __syncthreads();
    plw_mix_loop<ALG>(gz, tabs + CF::t_WLt(K) + (wave * NG * 16 + ge.n) * 24, CF::PAIR, NG,
                      [&](int og, float (&g)[DL]) { plw_get<ALG>(g, xb2, og, ge.lane); });
    static_for<0, NG>([&](auto ig) {
        float zi[DL];
        plw_get<ALG>(zi, xb0, decltype(ig)::value, ge.lane);
        pl_wgrad<ALG>(accWL[decltype(ig)::value], ggp, zi);
    });
    CSMPN_PHASE();
    // ---- geometric product backward
    pl_weighted_gp_bwd_z<ALG, NT>(ggp, S, lds + CF::p_w(K) + c * ALG::P, ge, gz, tot + SI::wA * NT, tot + SI::wB * NT);
    CSMPN_PHASE();
    float gr[DL];
    pl_weighted_gp_bwd_r<ALG>(ggp, S, lds + CF::p_w(K) + c * ALG::P, ge, gr);
    CSMPN_PHASE();
    // ---- NormalizationLayer backward -> gR
    float gR[DL];
    static_for<0, GC>([&](auto k) {
        constexpr int j0 = P::t.cstart[k], j1 = P::t.cstart[k + 1];
        float gden = 0.f, qR = 0.f;
        static_for<j0, j1>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            gden -= gr[j] * S.R[j];
            qR += ge.template qs<j>() * S.R[j] * S.R[j];
        });
        gden *= S.invden[k] * S.invden[k];
        const float sg = lds[CF::p_sg(K) + c * G + ge.grade(k)];
        const float nu = smooth_abs_sqrt1(qR);
        sums[SI::an + k] += (gden * (nu - 1.0f) * sg * (1.0f - sg));
        const float inu = fast_rcp(nu);
        const float gq = (gden * sg) * (0.5f * qR) * (inu * inu * inu);
        static_for<j0, j1>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            gR[j] = cvalid ? gr[j] * S.invden[k] + gq * (2.0f * ge.template qs<j>()) * S.R[j] : 0.f;
        });
    });
    __syncthreads();
    plw_put<ALG>(xb2, wave, ge.lane, gR);
    __syncthreads();
    plw_mix_loop<ALG>(gz, tabs + CF::t_WRt(K) + (wave * NG * 16 + ge.n) * 24, CF::PAIR, NG,
                      [&](int og, float (&g)[DL]) { plw_get<ALG>(g, xb2, og, ge.lane); });
    static_for<0, NG>([&](auto ig) {
        float zi[DL];
        plw_get<ALG>(zi, xb0, decltype(ig)::value, ge.lane);
        pl_wgrad<ALG>(accWR[decltype(ig)::value], gR, zi);
    });
    CSMPN_PHASE();
    // ---- MVSiLU backward -> gy
    static_for<0, GC>([&](auto k) {
        constexpr int j0 = P::t.cstart[k], j1 = P::t.cstart[k + 1];
        float ggate = 0.f, u = 0.f;
        static_for<j0, j1>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            ggate += gz[j] * S.y[j];
            u += ge.template qs<j>() * S.y[j] * S.y[j];
        });
        const bool scalar_inv = k == 0 && ge.s == 0;
        if (scalar_inv) u = S.y[0];
        const float gpre = ggate * S.gate[k] * (1.0f - S.gate[k]);
        sums[SI::sa + k] += (gpre * u);
        sums[SI::sb + k] += (gpre);
        const float gu = gpre * lds[CF::p_sa(K) + c * G + ge.grade(k)];
        static_for<j0, j1>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            float v = gz[j] * S.gate[k];
            const float quad = gu * (2.0f * ge.template qs<j>()) * S.y[j];
            if constexpr (k == 0 && j == 0) v += scalar_inv ? gu : quad;
            else v += quad;
            gy[j] = cvalid ? v : 0.f;
        });
    });
    sums[SI::b1] += (gy[0]);
#pragma unroll
    for (int i = 0; i < 3 + 3 * PS<ALG>::GC; ++i) tot[i * (64 * CF::NG)] = sums[i];
}

// end of a backward launch: the workgroup's MFMA tiles -> its slice of the partial buffer (plain coalesced stores);
// plw_reduce_kernel adds the slices in a fixed order (256 workgroups adding 19 k addresses each with atomics cost
// ~175 us per launch, more than the tile work of the small convex-hulls batches).
// Slice layout: [tile slot][thread], slot = ((table * MAXIN + input group) * GC + class) * 4 + v; tables: 0 W1, 1 WR, 2 WL.
template <class CF, int BLK>
struct PlwPart {
    static constexpr int NIN = BLK == 0 ? CF::NCH0 : CF::NG;
    static constexpr int GC = CF::GC, NG = CF::NG, NT = 64 * CF::NG;
    static constexpr int n_tiles = NIN + 2 * NG;                 // (W1: NIN input groups) + (WR, WL: NG each)
    static constexpr int slots = n_tiles * GC * 4;
    static constexpr int w_floats = slots * NT;                  // MFMA tiles
    // ... followed by the image of the per-channel sums: b1, bL, la [CP each], sa, sb, an [CP x G each], w [CP x P]
    static constexpr int CP = CF::CP, G = CF::G, NP = CF::NP;
    static constexpr int i_b1 = 0, i_bL = CP, i_la = 2 * CP, i_sa = 3 * CP, i_sb = i_sa + CP * G, i_an = i_sb + CP * G,
                         i_w = i_an + CP * G, i_tot = i_w + CP * NP;
    static constexpr int slice = (w_floats + i_tot + 3) & ~3;    // floats per workgroup
    static constexpr int tile_of(int table, int ig) { return table == 0 ? ig : (table == 1 ? NIN + ig : NIN + NG + ig); }
};
template <class ALG, class CF, int BLK>
CSMPN_DEV void plw_store_tile(float* slice, int tile_idx, const f4 (&acc)[PS<ALG>::GC], int tid) {
    constexpr int GC = PS<ALG>::GC, NT = 64 * CF::NG;
#pragma unroll
    for (int k = 0; k < GC; ++k)
#pragma unroll
        for (int v = 0; v < 4; ++v) slice[((tile_idx * GC + k) * 4 + v) * NT + tid] = acc[k][v];
}
// grads += sum over the workgroups' slices; one thread per (slot, thread position): a single writer per gradient
// element, fixed summation order
template <class ALG, class CF, int BLK>
__global__ void __launch_bounds__(64 * kPlReduceSubs) plw_reduce_kernel(const DevCemlp Cd, const float* part, int ngroups) {
    using PP = PlwPart<CF, BLK>;
    constexpr int GC = CF::GC, NG = CF::NG, NT = PP::NT, G = ALG::G, C = CF::C, NS = kPlReduceSubs;
    // 64 slice positions per workgroup x 16 interleaved group subsets (4 independent loads in flight each), combined in
    // LDS in a fixed order. (4 subsets: 13 us per launch on the 256 slices of a convex-hulls batch, 14 launches per step.)
    __shared__ float red[NS][64];
    const int sub = threadIdx.x >> 6;
    const long t = (long)blockIdx.x * 64 + (threadIdx.x & 63);
    const bool in = t < PP::w_floats + PP::i_tot;
    const bool small = t >= PP::w_floats;
    const int tid = (int)(t % NT), slot = small ? 0 : (int)(t / NT);
    const int v = slot & 3, k = (slot >> 2) % GC, tile_idx = (slot >> 2) / GC;
    int table, ig;
    if (tile_idx < PP::NIN) { table = 0; ig = tile_idx; }
    else if (tile_idx < PP::NIN + NG) { table = 1; ig = tile_idx - PP::NIN; }
    else { table = 2; ig = tile_idx - PP::NIN - NG; }
    const int wave = tid >> 6, lane = tid & 63, q = lane >> 4, n = lane & 15, c = n >> 1, s = n & 1;
    const int i = 4 * q + v, o = 8 * wave + (i >> 1), so = i & 1;
    int base, nvalid, I;
    if (table == 0 && BLK == 0) { base = CF::chunk_base(ig); nvalid = CF::chunk_valid(ig); I = CF::I0; }
    else { base = 8 * ig; nvalid = C - 8 * ig < 8 ? C - 8 * ig : 8; I = C; }
    const DevBlock& B = Cd.b[BLK];
    float* gW = table == 0 ? B.gW1 : (table == 1 ? B.gWR : B.gWL);
    bool live = in && !small && so == s && o < C && c < nvalid && gW != nullptr;
    float* dst = nullptr;
    if (live) {
        const int grade = s ? ALG::n - 2 * k : 2 * k;
        dst = gW + ((size_t)o * I + base + c) * G + grade;
    }
    if (in && small) {   // per-channel sums: image entry -> (tensor, element); channels beyond C are padding
        const int e = (int)(t - PP::w_floats);
        constexpr int CPc = CF::CP, NPc = ALG::P;
        if (e < PP::i_bL) { if (e < C && B.has_b1 && B.gb1) dst = B.gb1 + e; }
        else if (e < PP::i_la) { if (e - PP::i_bL < C && B.gbL) dst = B.gbL + (e - PP::i_bL); }
        else if (e < PP::i_sa) { if (e - PP::i_la < C && B.gla) dst = B.gla + (e - PP::i_la); }
        else if (e < PP::i_sb) { if (e - PP::i_sa < C * G && B.gsa) dst = B.gsa + (e - PP::i_sa); }
        else if (e < PP::i_an) { if (e - PP::i_sb < C * G && B.gsb) dst = B.gsb + (e - PP::i_sb); }
        else if (e < PP::i_w) { if (e - PP::i_an < C * G && B.gan) dst = B.gan + (e - PP::i_an); }
        else { if (e - PP::i_w < C * NPc && B.gw) dst = B.gw + (e - PP::i_w); }
        (void)CPc;
        live = dst != nullptr;
    }
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (live) {
        const float* p = part + t;
        int g = sub;
        for (; g + 3 * NS < ngroups; g += 4 * NS) {
            s0 += p[(size_t)g * PP::slice];
            s1 += p[(size_t)(g + NS) * PP::slice];
            s2 += p[(size_t)(g + 2 * NS) * PP::slice];
            s3 += p[(size_t)(g + 3 * NS) * PP::slice];
        }
        for (; g < ngroups; g += NS) s0 += p[(size_t)g * PP::slice];
    }
    red[sub][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sub == 0 && live) *dst += pl_reduce_combine(red, threadIdx.x);
}

// BLK = 1: gout -> block-1 backward -> d/d(block-1 input) rows to io.plw_g1.   BLK = 0: io.plw_g1 -> block-0
// backward -> input gradients (scatter / rows). Parameter gradients of block BLK.
// SAVES: the forward ran with CSMPN_FLAG_SAVE_STATE (regions 2 .. 7 of the saved buffer hold the blocks' s, y, R): a
// compile-time choice, as in cemlp_pl.hpp / cemlp_cl.hpp.
template <class ALG, class CF, int BLK, bool SAVES = false>
__global__ void __launch_bounds__(64 * CF::NG, CF::NG >= 3 ? CF::WG_PER_CU_BWD : CF::waves_per_simd(CF::WG_PER_CU_BWD)) cemlp_plw_bwd_kernel(const DevCemlp C_arg, const RowIO io_arg) {
    typedef const char __attribute__((address_space(4))) * KArgPtr;
    const KArgPtr ka = (KArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr size_t kIoOffset = (sizeof(DevCemlp) + alignof(RowIO) - 1) / alignof(RowIO) * alignof(RowIO);
    const DevCemlp& Cd = *(const DevCemlp*)(const char*)ka;
    const RowIO& io = *(const RowIO*)(const char*)(ka + kIoOffset);
    (void)C_arg; (void)io_arg;
    using P = PS<ALG>;
    using SI = PlSumIdx<ALG>;
    constexpr int D = ALG::D, DL = P::DL, G = ALG::G, GC = P::GC, NG = CF::NG, C = CF::C, CP = CF::CP, ROW = CF::ROW, RS = CF::RS;
    constexpr int MODE = CF::MODE, NA = CF::NA, NT = 64 * NG, NCH0 = CF::NCH0, NIN = BLK == 0 ? NCH0 : NG;
    static_assert(CF::bwd_total * 4 <= 160 * 1024, "LDS footprint");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* lds = smem;
    const int wave = threadIdx.x >> 6;
    const PlGeo<ALG> ge(threadIdx.x & 63);
    const bool cvalid = 8 * wave + ge.c < C;
    const int cch = (cvalid ? 8 * wave + ge.c : 0) * D;   // this lane's channel offset inside a C-wide row
    const float* tabs = io.plw_tabs;
    float* stg = lds + CF::st_off;
    float* tot = lds + CF::tot_off + threadIdx.x;
    {
        const DevBlock& B = Cd.b[BLK];
        for (int e = threadIdx.x; e < CP; e += NT) {
            const bool ok = e < C;
            lds[CF::p_b1(BLK) + e] = (ok && B.has_b1) ? B.b1[e] : 0.f;
            lds[CF::p_bL(BLK) + e] = ok ? B.bL[e] : 0.f;
            lds[CF::p_la(BLK) + e] = ok ? B.la[e] : 0.f;
        }
        for (int e = threadIdx.x; e < CP * G; e += NT) {
            const bool ok = e < C * G;
            lds[CF::p_sa(BLK) + e] = ok ? B.sa[e] : 0.f;
            lds[CF::p_sb(BLK) + e] = ok ? B.sb[e] : 0.f;
            lds[CF::p_sg(BLK) + e] = ok ? sigmoidf(B.an[e]) : 0.5f;
        }
        for (int e = threadIdx.x; e < CP * ALG::P; e += NT) lds[CF::p_w(BLK) + e] = e < C * ALG::P ? B.w[e] : 0.f;
        for (int e = threadIdx.x; e < CF::n_sums * NT; e += NT) lds[CF::tot_off + e] = 0.f;
    }
    __syncthreads();

    f4 aW1[NIN][GC], aWR[NG][GC], aWL[NG][GC];
#pragma unroll
    for (int k = 0; k < GC; ++k) {
#pragma unroll
        for (int i = 0; i < NIN; ++i) aW1[i][k] = splat(0.f);
#pragma unroll
        for (int i = 0; i < NG; ++i) { aWR[i][k] = splat(0.f); aWL[i][k] = splat(0.f); }
    }

    const long ntiles = (io.rows + kPlRows - 1) / kPlRows;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long row = tile * kPlRows + ge.q;
        const bool valid = row < io.rows;
        const long lrow = valid ? row : 0;
        int i_dst = -1, i_src = -1, i_perm = 0;
        float scale = 1.0f;
        if (valid) {
            if constexpr (MODE == MODE_EDGE) {
                i_dst = io.seg[0].ia[row];
                i_src = io.seg[0].ib[row];
                if constexpr (NA > 0) i_perm = io.seg[1].ia[row];
            } else if constexpr (MODE == MODE_NODE) {
                if (io.seg[1].deg) {
                    const int dg = io.seg[1].deg[row];
                    scale = 1.0f / float(dg > 1 ? dg : 1);
                }
            }
        }
        const float von = (valid && cvalid) ? 1.0f : 0.0f;
        float* xb1 = lds + CF::x_off(1);
        float* xb2 = lds + CF::x_off(2);
        if constexpr (BLK == 1) {
            float gout[DL], in1[DL], gy[DL];
            {
                long grow = MODE == MODE_EDGE ? (long)(valid ? i_dst : 0) : lrow;
                if (MODE == MODE_PLAIN && io.emb_nperm) grow = lrow / io.emb_nperm;   // fused embedding: the simplex's row
                pl_load<ALG>(gout, io.gy + (size_t)grow * ROW + cch, ge.s, von);
            }
            pl_load<ALG>(in1, io.saved + (size_t)lrow * ROW + cch, ge.s, von);
            {
                PlState<ALG> S;
                float unused[DL];
#pragma unroll
                for (int j = 0; j < DL; ++j) S.y[j] = 0.f;
                plw_put<ALG>(xb1, wave, ge.lane, in1);
                __syncthreads();
                if constexpr (SAVES) {
                    PlSaved<ALG> sv;
                    sv.template load<ROW, CP * D>(io.saved, io.rows, 1, ((size_t)(valid ? tile : 0) * NG + wave) * (kPlRows * 8 * D) + 4 * ge.lane);
                    plw_block_tail<ALG, CF, 1, true>(lds, tabs, ge, wave, cvalid, S, unused, &sv);
                } else {
                    plw_mix_loop<ALG>(S.y, tabs + CF::t_W1(1) + (wave * NG * 16 + ge.n) * 24, CF::PAIR, NG,
                                      [&](int ig, float (&xi)[DL]) { plw_get<ALG>(xi, xb1, ig, ge.lane); });
                    plw_block_tail<ALG, CF, 1>(lds, tabs, ge, wave, cvalid, S, unused);
                }
                plw_block_backward<ALG, CF, 1>(lds, tabs, ge, wave, cvalid, S, gout, gy, tot, aWR, aWL);
            }
            static_for<0, NG>([&](auto ig) {
                float xi[DL];
                plw_get<ALG>(xi, xb1, decltype(ig)::value, ge.lane);
                pl_wgrad<ALG>(aW1[decltype(ig)::value], gy, xi);
            });
            float g1[DL];
#pragma unroll
            for (int j = 0; j < DL; ++j) g1[j] = 0.f;
            __syncthreads();
            plw_put<ALG>(xb2, wave, ge.lane, gy);
            __syncthreads();
            plw_mix_loop<ALG>(g1, tabs + CF::t_W1t(1) + (wave * NG * 16 + ge.n) * 24, CF::PAIR, NG,
                              [&](int og, float (&g)[DL]) { plw_get<ALG>(g, xb2, og, ge.lane); });
            pl_stage<ALG>(stg + 8 * wave * D, g1, ge, RS, cvalid);
            __syncthreads();
            for (int r = 0; r < kPlRows; ++r) {
                const long rr = tile * kPlRows + r;
                if (rr < io.rows)
                    for (int e = 4 * threadIdx.x; e < ROW; e += 4 * NT)
                        *reinterpret_cast<f4*>(io.plw_g1 + (size_t)rr * ROW + e) = pl_ld4(stg + r * RS + e);
            }
            __syncthreads();
        } else {
            auto load_chunk = [&](int j, float (&x)[DL]) {
                const int seg = j / NG, grp = j % NG;
                const bool attr = j >= CF::NSEG * NG;
                const int ch = attr ? ge.c : 8 * grp + ge.c;
                const bool on = valid && ch < (attr ? NA : C);
                const int co = (on ? ch : 0) * D;
                if constexpr (MODE == MODE_EDGE) {
                    if (!attr) {
                        pl_load_diff<ALG>(x, io.seg[0].a + (size_t)(valid ? i_dst : 0) * ROW + co,
                                          io.seg[0].b + (size_t)(valid ? i_src : 0) * ROW + co, ge.s, on ? 1.0f : 0.0f);
                    } else {
                        pl_load<ALG>(x, io.seg[1].a + (size_t)(valid ? i_perm : 0) * (NA * D) + co, ge.s, on ? 1.0f : 0.0f);
                    }
                } else if constexpr (MODE == MODE_NODE) {
                    if (attr) pl_load<ALG>(x, io.seg[2].a + (size_t)lrow * (NA * D) + co, ge.s, on ? 1.0f : 0.0f);
                    else if (seg == 0) pl_load<ALG>(x, io.seg[0].a + (size_t)lrow * ROW + co, ge.s, on ? 1.0f : 0.0f);
                    else pl_load<ALG>(x, io.seg[1].a + (size_t)lrow * ROW + co, ge.s, on ? scale : 0.0f);
                } else {
                    (void)seg;
                    if (io.emb_nperm) {   // fused embedding: see the forward
                        const int v = (on ? ch : 0) / io.emb_k, kk = (on ? ch : 0) % io.emb_k;
                        const long vr = min(max(io.emb_verts[lrow * io.emb_nv + v], 0), io.emb_nrows - 1);
                        pl_load<ALG>(x, io.seg[0].a + ((size_t)vr * io.emb_k + kk) * D, ge.s, on ? 1.0f : 0.0f);
                    } else {
                        pl_load<ALG>(x, io.seg[0].a + (size_t)lrow * (NA * D) + co, ge.s, on ? 1.0f : 0.0f);
                    }
                }
            };
            float g1[DL], gy0[DL];
            if constexpr (CF::NBLK > 1) {
                pl_load<ALG>(g1, io.plw_g1 + (size_t)lrow * ROW + cch, ge.s, von);
            } else {   // single block: d/d(out) comes from the caller
                long grow = MODE == MODE_EDGE ? (long)(valid ? i_dst : 0) : lrow;
                if (MODE == MODE_PLAIN && io.emb_nperm) grow = lrow / io.emb_nperm;
                pl_load<ALG>(g1, io.gy + (size_t)grow * ROW + cch, ge.s, von);
            }
            {
                PlState<ALG> S;
                float unused[DL];
#pragma unroll
                for (int j = 0; j < DL; ++j) S.y[j] = 0.f;
                if constexpr (SAVES) {
                    PlSaved<ALG> sv;
                    sv.template load<ROW, CP * D>(io.saved, io.rows, 0, ((size_t)(valid ? tile : 0) * NG + wave) * (kPlRows * 8 * D) + 4 * ge.lane);
                    plw_block_tail<ALG, CF, 0, true>(lds, tabs, ge, wave, cvalid, S, unused, &sv);
                } else {
                    plw_mix_loop<ALG>(S.y, tabs + CF::t_W1(0) + (wave * NCH0 * 16 + ge.n) * 24, CF::PAIR, NCH0,
                                      [&](int j, float (&x)[DL]) { load_chunk(j, x); });
                    plw_block_tail<ALG, CF, 0>(lds, tabs, ge, wave, cvalid, S, unused);
                }
                plw_block_backward<ALG, CF, 0>(lds, tabs, ge, wave, cvalid, S, g1, gy0, tot, aWR, aWL);
            }
            // d/dW1 needs EVERY input chunk in EVERY wave (own output group x chunk j): the chunks are gathered ONCE per
            // workgroup - wave w takes the chunks w, w + NG, .. - and handed round through the exchange buffers (chunk j in
            // slot j % NG of buffer j / NG; all three are free here), instead of NG redundant gathers of NCH0 chunks, each
            // one exposed at one wave per SIMD
            static_assert(NCH0 <= CF::NXB * NG, "chunk slots");
            float* xb0c = lds + CF::x_off(0);
            __syncthreads();               // z (buffer 0) and the last gradient exchange (buffer 2) are read no more
            for (int j = wave; j < NCH0; j += NG) {
                float x[DL];
                load_chunk(j, x);
                plw_put<ALG>(xb0c + (j / NG) * CF::XB, j % NG, ge.lane, x);
            }
            __syncthreads();
            static_for<0, NCH0>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                float x[DL];
                plw_get<ALG>(x, xb0c + (j / NG) * CF::XB, j % NG, ge.lane);
                pl_wgrad<ALG>(aW1[j], gy0, x);
            });
            __syncthreads();
            plw_put<ALG>(xb2, wave, ge.lane, gy0);
            __syncthreads();
            // d/d(input chunk j): wave j % NG; segment by segment through the staging tile
            auto chunk_grad = [&](int j, float (&gx)[DL]) {
#pragma unroll
                for (int t = 0; t < DL; ++t) gx[t] = 0.f;
                plw_mix_loop<ALG>(gx, tabs + CF::t_W1t(0) + (j * NG * 16 + ge.n) * 24, CF::PAIR, NG,
                                  [&](int og, float (&g)[DL]) { plw_get<ALG>(g, xb2, og, ge.lane); });
            };
            auto copy_rows = [&](float* dst, int ncol, auto row_of) {
                for (int r = 0; r < kPlRows; ++r) {
                    const long rr = tile * kPlRows + r;
                    if (rr < io.rows) {
                        float* d = dst + (size_t)row_of(r, rr) * ncol;
                        for (int e = 4 * threadIdx.x; e < ncol; e += 4 * NT) *reinterpret_cast<f4*>(d + e) = pl_ld4(stg + r * RS + e);
                    }
                }
            };
            static_for<0, CF::NSEG>([&](auto sgc) {
                constexpr int sg_ = decltype(sgc)::value;
                float* dstp = io.gx[sg_];
                if (dstp) {
                    float gx[DL];
                    chunk_grad(sg_ * NG + wave, gx);
                    if constexpr (MODE == MODE_NODE) {
                        if constexpr (sg_ == 0) {
                            if (io.resid_bwd) {
                                float go[DL];
                                pl_load<ALG>(go, io.gy + (size_t)lrow * ROW + cch, ge.s, von);
#pragma unroll
                                for (int t = 0; t < DL; ++t) gx[t] += go[t];
                            }
                        } else {
#pragma unroll
                            for (int t = 0; t < DL; ++t) gx[t] *= scale;
                        }
                    }
                    pl_stage<ALG>(stg + 8 * wave * D, gx, ge, RS, cvalid);
                    __syncthreads();
                    if (MODE == MODE_EDGE && io.row_store) {
                        copy_rows(dstp, ROW, [&](int, long rr) { return rr; });   // deterministic mode: per-edge rows
                    } else if constexpr (MODE == MODE_EDGE) {
                        int td[kPlRows], ts[kPlRows];
#pragma unroll
                        for (int r = 0; r < kPlRows; ++r) {
                            td[r] = __builtin_amdgcn_readlane(i_dst, 16 * r);
                            ts[r] = __builtin_amdgcn_readlane(i_src, 16 * r);
                        }
                        for (int col = threadIdx.x; col < ROW; col += NT) {
                            float acc = 0.f;
                            int cur = td[0];
#pragma unroll
                            for (int r = 0; r < kPlRows; ++r) {
                                const float v = stg[r * RS + col];
                                if (td[r] != cur) {
                                    if (cur >= 0) atomicAdd(dstp + (long)cur * ROW + col, acc);
                                    cur = td[r];
                                    acc = 0.f;
                                }
                                acc += v;
                                if (ts[r] >= 0) atomicAdd(dstp + (long)ts[r] * ROW + col, -v);
                            }
                            if (cur >= 0) atomicAdd(dstp + (long)cur * ROW + col, acc);
                        }
                    } else {
                        copy_rows(dstp, ROW, [&](int, long rr) { return rr; });
                    }
                    __syncthreads();
                }
            });
            if constexpr (NA > 0) {
                float* dstp = io.gx[CF::NSEG];
                if (dstp) {
                    if (wave == 0) {
                        float gx[DL];
                        chunk_grad(CF::NSEG * NG, gx);
                        pl_stage<ALG>(stg, gx, ge, RS, ge.c < NA);
                    }
                    __syncthreads();
                    if constexpr (MODE == MODE_EDGE) {
                        int tp[kPlRows];
#pragma unroll
                        for (int r = 0; r < kPlRows; ++r) tp[r] = __builtin_amdgcn_readlane(i_perm, 16 * r);
                        copy_rows(dstp, NA * D, [&](int r, long) { return (long)tp[r]; });
                    } else {
                        copy_rows(dstp, NA * D, [&](int, long rr) { return rr; });
                    }
                    __syncthreads();
                }
            }
        }
    }

    // ---- parameter gradients of block BLK
    {
        const DevBlock& B = Cd.b[BLK];
        {
            using PP = PlwPart<CF, BLK>;
            float* slice = io.plw_part + (size_t)blockIdx.x * PP::slice;
            static_for<0, NIN>([&](auto jc) {
                plw_store_tile<ALG, CF, BLK>(slice, PP::tile_of(0, decltype(jc)::value), aW1[decltype(jc)::value], threadIdx.x);
            });
            static_for<0, NG>([&](auto jc) {
                plw_store_tile<ALG, CF, BLK>(slice, PP::tile_of(1, decltype(jc)::value), aWR[decltype(jc)::value], threadIdx.x);
                plw_store_tile<ALG, CF, BLK>(slice, PP::tile_of(2, decltype(jc)::value), aWL[decltype(jc)::value], threadIdx.x);
            });
        }
        // small sums: every (row quarter, lane column) writes its value to its own slot of a 4-fold LDS image (the
        // exchange buffers are free now; exactly one writer per slot, no atomics), the quarters are added in a fixed
        // order and the image goes to the tail of the workgroup's slice: plw_reduce_kernel adds the slices
        {
            using PP = PlwPart<CF, BLK>;
            constexpr int L_b1 = 0, L_bL = CP, L_la = 2 * CP, L_sa = 4 * CP, L_sb = L_sa + CP * G, L_an = L_sb + CP * G,
                          L_w = L_an + CP * G, L_tot = L_w + CP * ALG::P;      // la holds [c][parity] here
            static_assert(4 * L_tot <= CF::NXB * CF::XB, "4-fold image fits the exchange buffers");
            float* img = lds + CF::x_off(0);
            __syncthreads();
            for (int e = threadIdx.x; e < 4 * L_tot; e += NT) img[e] = 0.f;
            __syncthreads();
            float* mine = img + ge.q * L_tot;
            const int c = 8 * wave + ge.c;
            mine[L_la + 2 * c + ge.s] = tot[SI::la * NT];
            if (ge.s == 0) {
                mine[L_bL + c] = tot[SI::bL * NT];
                mine[L_b1 + c] = tot[SI::b1 * NT];
            }
#pragma unroll
            for (int k = 0; k < GC; ++k) {
                const int pg = c * G + ge.grade(k);
                mine[L_an + pg] = tot[(SI::an + k) * NT];
                mine[L_sa + pg] = tot[(SI::sa + k) * NT];
                mine[L_sb + pg] = tot[(SI::sb + k) * NT];
            }
            static_for<0, P::QP>([&](auto qq) {
                constexpr int q = decltype(qq)::value;
                mine[L_w + c * ALG::P + (ge.s ? P::t.pid[1][0][q] : P::t.pid[0][0][q])] = tot[(SI::wA + q) * NT];
                mine[L_w + c * ALG::P + (ge.s ? P::t.pid[1][1][q] : P::t.pid[0][1][q])] =
                    tot[(SI::wB + q) * NT] * (ge.s ? 1.0f : float(P::t.I2));
            });
            __syncthreads();
            float* tail = io.plw_part + (size_t)blockIdx.x * PP::slice + PP::w_floats;
            auto quarters = [&](int l) { return (img[l] + img[L_tot + l]) + (img[2 * L_tot + l] + img[3 * L_tot + l]); };
            for (int e = threadIdx.x; e < PP::i_tot; e += NT) {
                float v;
                if (e < PP::i_la) v = quarters(e);                                    // b1, bL: same offsets
                else if (e < PP::i_sa) v = quarters(L_la + 2 * (e - PP::i_la)) + quarters(L_la + 2 * (e - PP::i_la) + 1);
                else v = quarters(L_sa + (e - PP::i_sa));                           // sa, sb, an, w: same order
                tail[e] = v;
            }
        }
    }
}

}  // namespace csmpn
