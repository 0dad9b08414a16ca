// Parity-lane ("PL") row program for D = 32 algebras with an ODD number of generators (Cl(5,0), Cl(4,1))
// and 8-channel blocks: the S3 shape of BASELINE.json. Same arithmetic as cemlp_ps.hpp / cemlp_device.hpp
// (csmpn/models/cegnn_utils.py:34-155,287-338), different distribution over the wave:
//
//   lane = (row q of a 4-row tile, channel c, blade parity s):  lane = 16 q + 2 c + s.
//   Every activation tensor is float t[16]: the 16 even-grade blades of (row, channel) in the s = 0 lane,
//   their 16 Hodge partners in the s = 1 lane (slot order of PS<ALG>, cemlp_ps.hpp).
//
// Why: with the 16x16x4 accumulator layout of the other kernels a lane owns 4 rows, so a D = 32 tensor costs
// 64 (parity-split) to 128 VGPRs and the backward, which keeps about ten of them live, spills 4-6 KB per lane
// (profiles/r02_S3_pmc_summary.json: 8.9 GB of scratch traffic per launch for 0.59 GB of algorithmic bytes).
// Here a tensor is 16 VGPRs, nothing of a row lives in LDS between phases and there is no scratch.
//
//   channel mixing   8 DPP rotations (row_ror by 2 k lanes keeps the parity) x 16 slots of v_fmac per 8 x 8
//                    matrix; the weights sit in LDS as "rotation tables" [rotation][grade class][lane column],
//                    built once per workgroup from the reference layout [o][c][g]
//   products         per lane: two products of the even subalgebra on own / partner (lane ^ 1) operands
//                    (the X~ I form of cemlp_ps.hpp), compile-time signs
//   weight gradients v_mfma_f32_16x16x4_f32 with A = this lane's gradient slot, B = this lane's input slot:
//                    D[(o,s)][(c,s')] += sum over the 4 rows; accumulators persist over the tile loop
//   small gradients  per-lane sums over the tile loop; one LDS image + one round of global atomics per workgroup
#pragma once
#include "cemlp_ps.hpp"

namespace csmpn {

constexpr int kPlWaves = 4;
constexpr int kPlRows = 4;
#ifndef CSMPN_PL_BWD_WAVES
#define CSMPN_PL_BWD_WAVES 1
#endif
#ifndef CSMPN_PL_FWD_WAVES
#define CSMPN_PL_FWD_WAVES 2
#endif

template <int CTRL>
CSMPN_DEV int pl_dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }
// value of the lane that rotation R (channels) brings here; R = 0: the lane's own
template <int R>
CSMPN_DEV float pl_rot(float v) {
    if constexpr (R == 0) return v;
    else {
        const int i = __builtin_bit_cast(int, v);
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, 0x120 + 2 * R, 0xF, 0xF, true));
    }
}
// acc += w * (value of x in the lane that rotation R brings here): ONE v_fmac_f32_dpp. hipcc folds a DPP move into
// v_add / v_mul but not into v_fmac (tied accumulator), which would leave the mixing at two instructions per term.
// Inline asm is opaque to the hazard recognizer (a DPP read needs 2 wait states after a VALU write of its source,
// 5 after an EXEC write): pl_dpp_ready() in front of a run of these orders the producers of x before an s_nop 4,
// and the statements are volatile so that the run stays behind it.
template <int R>
CSMPN_DEV void pl_fmac_rot(float& acc, float x, float w) {
    if constexpr (R == 0) acc = __builtin_fmaf(w, x, acc);
    else asm volatile("v_fmac_f32_dpp %0, %1, %2 row_ror:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                      : "+v"(acc) : "v"(x), "v"(w), "n"(2 * R));
}
CSMPN_DEV void pl_dpp_ready(const float (&x)[16]) {
    asm volatile("s_nop 4" :: "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]),
                             "v"(x[8]), "v"(x[9]), "v"(x[10]), "v"(x[11]), "v"(x[12]), "v"(x[13]), "v"(x[14]), "v"(x[15]));
}
CSMPN_DEV float pl_partner(float v) { return dpp_mov<0xB1>(v); }   // quad_perm [1,0,3,2]: the other parity
CSMPN_DEV float pl_even(float v) { return dpp_mov<0xA0>(v); }      // quad_perm [0,0,2,2]: the even lane's value in both
CSMPN_DEV float pl_odd(float v) { return dpp_mov<0xF5>(v); }       // quad_perm [1,1,3,3]
// sum over the 8 channels of a row (lanes of equal parity); result in every lane
CSMPN_DEV float pl_chan_sum(float v) {
    v += dpp_mov<0x122>(v);   // row_ror 2
    v += dpp_mov<0x124>(v);   // row_ror 4
    v += dpp_mov<0x128>(v);   // row_ror 8
    return v;
}
CSMPN_DEV f4 pl_ld4(const float* p) { return *reinterpret_cast<const f4*>(p); }
// Ordering point for the instruction scheduler. A bare sched_barrier does not order pure arithmetic (the
// selection DAG sinks it below a run of barriers and the whole phase becomes one region with hundreds of
// values in flight); passing the values just produced through an empty volatile asm ties them to the barrier.
template <int LO, int HI, int N>
CSMPN_DEV void pl_pin(float (&a)[N]) {
#pragma unroll
    for (int i = LO; i < HI; ++i) asm volatile("" : "+v"(a[i]));
    CSMPN_PHASE();
}
CSMPN_DEV float smooth_abs_sqrt1(float q) { return sqrt_pos(sqrt_pos(q * q + kSmooth)); }

// ---------------------------------------------------------------------------------
// compile-time LDS layout (floats)
template <class ALG, int NBLK, int I0>
struct PlLay {
    using P = PS<ALG>;
    static constexpr int C = 8, D = ALG::D, DL = P::DL, GC = P::GC, G = ALG::G, NP = ALG::P, QP = P::QP, N = ALG::n;
    static constexpr int ROW = C * D, RS = ROW + 4;
    static constexpr int TAB = 8 * GC * 16;   // one rotation table
    static constexpr int Iof(int k) { return k == 0 ? I0 : C; }
    static constexpr int nch(int k) { return (Iof(k) + 7) / 8; }
    static constexpr int ntab(int k) { return 2 * nch(k) + 4; }
    // tables of block k: [W1 chunk 0.., W1^T chunk 0.., WR, WR^T, WL, WL^T], then the per-channel parameters
    static constexpr int par_floats = 3 * C + 3 * C * G + C * NP;
    static constexpr int blk_floats(int k) { return ntab(k) * TAB + par_floats; }
    static constexpr int blk_off(int k) { return k == 0 ? 0 : blk_off(k - 1) + blk_floats(k - 1); }
    static constexpr int t_W1(int k, int ch) { return blk_off(k) + ch * TAB; }
    static constexpr int t_W1t(int k, int ch) { return blk_off(k) + (nch(k) + ch) * TAB; }
    static constexpr int t_WR(int k) { return blk_off(k) + 2 * nch(k) * TAB; }
    static constexpr int t_WRt(int k) { return t_WR(k) + TAB; }
    static constexpr int t_WL(int k) { return t_WR(k) + 2 * TAB; }
    static constexpr int t_WLt(int k) { return t_WR(k) + 3 * TAB; }
    static constexpr int p_b1(int k) { return blk_off(k) + ntab(k) * TAB; }
    static constexpr int p_bL(int k) { return p_b1(k) + C; }
    static constexpr int p_la(int k) { return p_bL(k) + C; }
    static constexpr int p_sa(int k) { return p_la(k) + C; }
    static constexpr int p_sb(int k) { return p_sa(k) + C * G; }
    static constexpr int p_sg(int k) { return p_sb(k) + C * G; }
    static constexpr int p_w(int k) { return p_sg(k) + C * G; }
    static constexpr int store_total = (blk_off(NBLK - 1) + blk_floats(NBLK - 1) + 3) & ~3;
    // per wave: a staging tile [4 rows][ROW + 4]
    static constexpr int scratch = kPlRows * RS;
    static constexpr int sc_off = store_total;
    // backward: image of the workgroup's parameter-gradient sums, reference layouts back to back
    static constexpr int i_W1(int k) { return k == 0 ? 0 : i_W1(k - 1) + img_blk(k - 1); }
    static constexpr int i_WR(int k) { return i_W1(k) + C * Iof(k) * G; }
    static constexpr int i_WL(int k) { return i_WR(k) + C * C * G; }
    static constexpr int i_b1(int k) { return i_WL(k) + C * C * G; }
    static constexpr int i_bL(int k) { return i_b1(k) + C; }
    static constexpr int i_la(int k) { return i_bL(k) + C; }
    static constexpr int i_sa(int k) { return i_la(k) + C; }
    static constexpr int i_sb(int k) { return i_sa(k) + C * G; }
    static constexpr int i_an(int k) { return i_sb(k) + C * G; }
    static constexpr int i_w(int k) { return i_an(k) + C * G; }
    static constexpr int img_blk(int k) { return C * Iof(k) * G + 2 * C * C * G + 3 * C + 3 * C * G + C * NP; }
    static constexpr int img_total = (i_W1(NBLK - 1) + img_blk(NBLK - 1) + 3) & ~3;
    static constexpr int fwd_total = sc_off + kPlWaves * scratch;
    // backward: lane-private running sums [block][slot][thread]
    static constexpr int n_sums = 3 + 3 * GC + 2 * QP;
    static constexpr int tot_off = sc_off + kPlWaves * scratch;
    static constexpr int tot_blk = n_sums * 64 * kPlWaves;
    static constexpr int bwd_total = tot_off + NBLK * tot_blk;
};

template <class ALG>
struct PlGeo {
    using P = PS<ALG>;
    int lane, q, n, c, s;
    float tau;   // -1 in odd lanes: sign of the flipped slots when moving to / from the X~ basis
    CSMPN_DEV explicit PlGeo(int lane_) : lane(lane_), q(lane_ >> 4), n(lane_ & 15) {
        c = n >> 1;
        s = n & 1;
        tau = s ? -1.0f : 1.0f;
    }
    template <int J> CSMPN_DEV int blade() const { return s ? P::t.od[J] : P::t.ev[J]; }
    template <int J> CSMPN_DEV float qs() const {
        constexpr int qe = ALG::t.qsign[P::t.ev[J]], qo = ALG::t.qsign[P::t.od[J]];
        if constexpr (qe == qo) return float(qe);
        else return s ? float(qo) : float(qe);
    }
    CSMPN_DEV int grade(int k) const { return s ? P::N - 2 * k : 2 * k; }
};

// ---------------------------------------------------------------------------------
// parameters -> LDS (once per workgroup)
template <class LY, class ALG, int K>
CSMPN_DEV void pl_stage_block(float* lds, const DevBlock& B, int tid, int dir) {
    constexpr int GC = LY::GC, G = LY::G, I = LY::Iof(K), NCH = LY::nch(K), NT = LY::ntab(K), C = LY::C, NP = LY::NP;
    const int n = tid & 15, c = n >> 1, s = n & 1, grp = tid >> 4;
    float* base = lds + LY::blk_off(K);
    for (int e = grp; e < NT * 8 * GC; e += (64 * kPlWaves) / 16) {
        const int cls = e % GC, r = (e / GC) % 8, T = e / (8 * GC);
        const int sc = (c + dir * r) & 7;   // channel whose value rotation r brings to this lane
        const int g = s ? ALG::n - 2 * cls : 2 * cls;
        const float* W;
        int Iw, ch;
        bool tr;
        if (T < NCH) { W = B.W1; Iw = I; ch = T; tr = false; }
        else if (T < 2 * NCH) { W = B.W1; Iw = I; ch = T - NCH; tr = true; }
        else { W = (T - 2 * NCH) < 2 ? B.WR : B.WL; Iw = C; ch = 0; tr = ((T - 2 * NCH) & 1) != 0; }
        const int o = tr ? sc : c, cin = 8 * ch + (tr ? c : sc);
        base[e * 16 + n] = cin < Iw ? W[((size_t)o * Iw + cin) * G + g] : 0.f;
    }
    for (int e = tid; e < C; e += 64 * kPlWaves) {
        lds[LY::p_b1(K) + e] = B.has_b1 ? B.b1[e] : 0.f;
        lds[LY::p_bL(K) + e] = B.bL[e];
        lds[LY::p_la(K) + e] = B.la[e];
    }
    for (int e = tid; e < C * G; e += 64 * kPlWaves) {
        lds[LY::p_sa(K) + e] = B.sa[e];
        lds[LY::p_sb(K) + e] = B.sb[e];
        lds[LY::p_sg(K) + e] = sigmoidf(B.an[e]);
    }
    for (int e = tid; e < C * NP; e += 64 * kPlWaves) lds[LY::p_w(K) + e] = B.w[e];
}

// acc[j] += sum_r table[r][class(j)][n] * (slot j of the lane that rotation r brings here)
// The 24 weights are read first; every rotation is its own scheduling region, so that the rotated copies
// (16 values) die before the next rotation starts (left alone, the scheduler hoists all 128 DPP moves).
template <class ALG, int OFF>
CSMPN_DEV void pl_linear(float (&acc)[PS<ALG>::DL], const float (&x)[PS<ALG>::DL], const float* ldsn) {
    using P = PS<ALG>;
    constexpr int GC = P::GC, DL = P::DL;
    float w[8][GC];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int k = 0; k < GC; ++k) w[r][k] = ldsn[OFF + (r * GC + k) * 16];
    // (two instructions per term: the v_fmac_f32_dpp form of cemlp_plw.hpp pins its operands to arch VGPRs and
    // costs the 8-channel backward 400-570 B of scratch - measured slower: edge backward 883 -> 1299 us)
    static_for<0, 8>([&](auto r) {
        static_for<0, DL>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            acc[j] = __builtin_fmaf(w[decltype(r)::value][P::t.cls[j]], pl_rot<decltype(r)::value>(x[j]), acc[j]);
        });
        pl_pin<0, DL>(acc);
    });
}
// two matrices applied to the same input (linear_right and linear_left): the rotated copies are shared
template <class ALG, int OFFA, int OFFB>
CSMPN_DEV void pl_linear2(float (&accA)[PS<ALG>::DL], float (&accB)[PS<ALG>::DL], const float (&x)[PS<ALG>::DL],
                          const float* ldsn) {
    using P = PS<ALG>;
    constexpr int GC = P::GC, DL = P::DL;
    static_for<0, 8>([&](auto r) {
        float wa[GC], wb[GC];
#pragma unroll
        for (int k = 0; k < GC; ++k) {
            wa[k] = ldsn[OFFA + (decltype(r)::value * GC + k) * 16];
            wb[k] = ldsn[OFFB + (decltype(r)::value * GC + k) * 16];
        }
        static_for<0, DL>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            const float t = pl_rot<decltype(r)::value>(x[j]);
            accA[j] = __builtin_fmaf(wa[P::t.cls[j]], t, accA[j]);
            accB[j] = __builtin_fmaf(wb[P::t.cls[j]], t, accB[j]);
        });
        pl_pin<0, DL>(accA);
        pl_pin<0, DL>(accB);
    });
}

// dW accumulators of one 8 x 8 (chunk of a) matrix: acc[class][v], MFMA tile element
// (i = 4 (lane >> 4) + v = 2 o + s_o, j = lane & 15 = 2 c + s_c), valid where s_o == s_c
template <class ALG>
CSMPN_DEV void pl_wgrad(f4 (&acc)[PS<ALG>::GC], const float (&g)[PS<ALG>::DL], const float (&x)[PS<ALG>::DL]) {
    using P = PS<ALG>;
    static_for<0, P::DL>([&](auto jj) {
        constexpr int j = decltype(jj)::value;
        acc[P::t.cls[j]] = mfma16(g[j], x[j], acc[P::t.cls[j]]);
    });
}

// ---------------------------------------------------------------------------------
// row access. A channel row is D = 32 contiguous floats; the lane keeps its parity's 16.
template <class ALG>
CSMPN_DEV void pl_pick(float (&x)[PS<ALG>::DL], const f4 (&v)[ALG::D / 4], int s, float scale) {
    using P = PS<ALG>;
    static_for<0, P::DL>([&](auto jj) {
        constexpr int j = decltype(jj)::value;
        constexpr int ev = P::t.ev[j], od = P::t.od[j];
        // the two candidates pass through an (empty) asm so that they are register values: left alone, the
        // compiler turns the select between two elements of v into ONE dynamically indexed load and moves v
        // to scratch memory
        float a = v[ev / 4][ev % 4], b = v[od / 4][od % 4];
        asm volatile("" : "+v"(a), "+v"(b));
        x[j] = (s ? b : a) * scale;
    });
}
template <class ALG>
CSMPN_DEV void pl_load(float (&x)[PS<ALG>::DL], const float* p, int s, float scale) {
    f4 v[ALG::D / 4];
#pragma unroll
    for (int e = 0; e < ALG::D / 4; ++e) v[e] = pl_ld4(p + 4 * e);
    pl_pick<ALG>(x, v, s, scale);
}
template <class ALG>
CSMPN_DEV void pl_load_diff(float (&x)[PS<ALG>::DL], const float* pa, const float* pb, int s, float scale) {
    f4 v[ALG::D / 4];
#pragma unroll
    for (int e = 0; e < ALG::D / 4; ++e) v[e] = pl_ld4(pa + 4 * e) - pl_ld4(pb + 4 * e);
    pl_pick<ALG>(x, v, s, scale);
}
// this lane's half tensor -> the wave's staging tile [4 rows][nch * D (+4)], reference blade order
template <class ALG>
CSMPN_DEV void pl_stage(float* sc, const float (&x)[PS<ALG>::DL], const PlGeo<ALG>& ge, int rs, bool on) {
    using P = PS<ALG>;
    if (on) {
        float* p = sc + ge.q * rs + ge.c * ALG::D;
        static_for<0, P::DL>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            p[ge.template blade<j>()] = x[j];
        });
    }
}

// ---------------------------------------------------------------------------------
template <class ALG>
struct PlState {
    float y[PS<ALG>::DL];
    float gate[PS<ALG>::GC];
    float R[PS<ALG>::DL];
    float invden[PS<ALG>::GC];
    float s[PS<ALG>::DL];
    float qs, nl, invMn;
};

template <class ALG>
CSMPN_DEV void pl_tilde(float (&t)[PS<ALG>::DL], const PlGeo<ALG>& ge) {
    static_for<0, PS<ALG>::DL>([&](auto jj) {
        constexpr int j = decltype(jj)::value;
        if constexpr (PS<ALG>::t.eps[j] < 0) t[j] *= ge.tau;
    });
}

// out += weighted geometric product of z and r (own parity, reference basis); wrow: this channel's path weights (LDS)
template <class ALG>
CSMPN_DEV void pl_weighted_gp(float (&out)[PS<ALG>::DL], const float (&z)[PS<ALG>::DL], const float (&r)[PS<ALG>::DL],
                              const float* wrow, const PlGeo<ALG>& ge) {
    using P = PS<ALG>;
    constexpr int DL = P::DL, QP = P::QP;
    float zE[DL], zO[DL], rw[DL], ro[DL], gp[DL];
#pragma unroll
    for (int j = 0; j < DL; ++j) { zE[j] = z[j]; rw[j] = r[j]; gp[j] = 0.f; }
    pl_tilde<ALG>(zE, ge);
    pl_tilde<ALG>(rw, ge);
#pragma unroll
    for (int j = 0; j < DL; ++j) {
        zO[j] = pl_odd(zE[j]);
        zE[j] = pl_even(zE[j]);
        ro[j] = pl_partner(rw[j]);
    }
    // the path weights of class q + 1 are requested in front of the products of class q (every class is its own scheduling
    // region: left at their use, each pair of LDS reads stalled the wave for its latency - 2 x QP times per product)
    float wn[2];
    auto ldw = [&](auto qq, float (&w)[2]) {
        constexpr int q = decltype(qq)::value;
        w[0] = wrow[ge.s ? P::t.pid[1][0][q] : P::t.pid[0][0][q]];
        w[1] = wrow[ge.s ? P::t.pid[1][1][q] : P::t.pid[0][1][q]];
    };
    ldw(std::integral_constant<int, 0>{}, wn);
    static_for<0, QP>([&](auto qq) {
        constexpr int q = decltype(qq)::value;
        constexpr int ka = P::t.qg[q][0], kc = P::t.qg[q][1], kb = P::t.qg[q][2];
        constexpr int a0 = P::t.cstart[ka], a1 = P::t.cstart[ka + 1];
        constexpr int c0 = P::t.cstart[kc], nc = P::t.cstart[kc + 1] - c0;
        constexpr int b0 = P::t.cstart[kb], b1 = P::t.cstart[kb + 1];
        const float wA = wn[0];
        const float wB = wn[1] * (ge.s ? 1.0f : float(P::t.I2));
        if constexpr (q + 1 < QP) ldw(std::integral_constant<int, q + 1>{}, wn);
        float tA[nc], tB[nc];
#pragma unroll
        for (int t = 0; t < nc; ++t) { tA[t] = 0.f; tB[t] = 0.f; }
        static_for<a0, a1>([&](auto aa) {
            static_for<b0, b1>([&](auto bb) {
                constexpr int a = decltype(aa)::value, b = decltype(bb)::value;
                constexpr int c = P::t.pc[a][b];
                if constexpr (c >= c0 && c < c0 + nc) {
                    constexpr float sg = float(P::t.psg[a][b]);
                    tA[c - c0] += (sg * zE[a]) * rw[b];
                    tB[c - c0] += (sg * zO[a]) * ro[b];
                }
            });
        });
#pragma unroll
        for (int t = 0; t < nc; ++t) gp[c0 + t] += wA * tA[t] + wB * tB[t];
        pl_pin<c0, c0 + nc>(gp);   // one path class per scheduling region: bounds the accumulators in flight
    });
    pl_tilde<ALG>(gp, ge);
#pragma unroll
    for (int j = 0; j < DL; ++j) out[j] += gp[j];
}

// the four reference path weights of path class Q (LDS): [even lanes' product A, B, odd lanes' A, B]
template <class ALG, int Q>
CSMPN_DEV void pl_ld_paths(float (&w)[4], const float* wrow) {
    using P = PS<ALG>;
    w[0] = wrow[P::t.pid[0][0][Q]];
    w[1] = wrow[P::t.pid[0][1][Q]];
    w[2] = wrow[P::t.pid[1][0][Q]];
    w[3] = wrow[P::t.pid[1][1][Q]];
}

// backward of pl_weighted_gp in two passes (each keeps four operand copies live instead of eight).
// Pass Z: gz += d/dz, gwA / gwB += the gradients of the lane's two forward weight sets (the I^2 factor of wB is
// applied when the sums are written out). Pass R: gr = d/dr.
// totA / totB: this thread's running sums of the weight gradients of path class 0 (LDS, STRIDE floats between classes):
// a class's two sums are READ in front of its terms and written behind them - as a read-modify-write at the end each pair
// stalled the wave for an LDS round trip, 2 x QP times per product backward at one wave per SIMD.
template <class ALG, int STRIDE>
CSMPN_DEV void pl_weighted_gp_bwd_z(const float (&ggp)[PS<ALG>::DL], const PlState<ALG>& S, const float* wrow,
                                    const PlGeo<ALG>& ge, float (&gz)[PS<ALG>::DL], float* totA, float* totB) {
    using P = PS<ALG>;
    constexpr int DL = P::DL, QP = P::QP;
    const float i2 = float(P::t.I2);
    float zw[DL], zo[DL], rE[DL], rO[DL], Gw[DL], Go[DL], gzt[DL];
    static_for<0, DL>([&](auto jj) {
        constexpr int j = decltype(jj)::value;
        zw[j] = S.gate[P::t.cls[j]] * S.y[j];
        rE[j] = S.R[j] * S.invden[P::t.cls[j]];
        Gw[j] = ggp[j];
        gzt[j] = 0.f;
    });
    pl_tilde<ALG>(zw, ge);
    pl_tilde<ALG>(rE, ge);
    pl_tilde<ALG>(Gw, ge);
#pragma unroll
    for (int j = 0; j < DL; ++j) {
        zo[j] = pl_partner(zw[j]);
        rO[j] = pl_odd(rE[j]);
        rE[j] = pl_even(rE[j]);
        Go[j] = pl_partner(Gw[j]);
    }
    CSMPN_PHASE();
    float wn[4];
    pl_ld_paths<ALG, 0>(wn, wrow);
    static_for<0, QP>([&](auto qq) {
        constexpr int q = decltype(qq)::value;
        constexpr int ka = P::t.qg[q][0], kc = P::t.qg[q][1], kb = P::t.qg[q][2];
        constexpr int a0 = P::t.cstart[ka], na = P::t.cstart[ka + 1] - a0;
        constexpr int c0 = P::t.cstart[kc], nc = P::t.cstart[kc + 1] - c0;
        constexpr int b0 = P::t.cstart[kb], nb = P::t.cstart[kb + 1] - b0;
        // d/dz of the even lanes: w00 Ge (x) Re + w10 Go (x) Ro; of the odd lanes: w11 Go (x) Re + I^2 w01 Ge (x) Ro
        const float w00 = wn[0], w01 = wn[1] * i2, w10 = wn[2], w11 = wn[3];
        if constexpr (q + 1 < QP) pl_ld_paths<ALG, q + 1>(wn, wrow);   // next class: in front of this class's products
        const float oldA = totA[q * STRIDE], oldB = totB[q * STRIDE];
        const float u1 = ge.s ? w11 : w00, u2 = ge.s ? w01 : w10;
        float S1[na], S2[na], S3[na];
#pragma unroll
        for (int t = 0; t < na; ++t) { S1[t] = 0.f; S2[t] = 0.f; S3[t] = 0.f; }
        static_for<0, na>([&](auto aa) {
            static_for<0, nb>([&](auto bb) {
                constexpr int ai = decltype(aa)::value, bi = decltype(bb)::value;
                constexpr int a = a0 + ai, b = b0 + bi;
                constexpr int c = P::t.pc[a][b];
                if constexpr (c >= c0 && c < c0 + nc) {
                    constexpr float sg = float(P::t.psg[a][b]);
                    S1[ai] += (sg * Gw[c]) * rE[b];
                    S3[ai] += (sg * Gw[c]) * rO[b];
                    S2[ai] += (sg * Go[c]) * rO[b];
                }
            });
        });
        float k1 = 0.f, k3 = 0.f;
#pragma unroll
        for (int t = 0; t < na; ++t) {
            gzt[a0 + t] += u1 * S1[t] + u2 * S2[t];
            k1 += zw[a0 + t] * S1[t];
            k3 += zo[a0 + t] * S3[t];
        }
        // even lanes: K1 -> w00 (product A), K3 -> w01 (product B); odd lanes: K3 -> w10 (A), K1 -> w11 (B)
        totA[q * STRIDE] = oldA + (ge.s ? k3 : k1);
        totB[q * STRIDE] = oldB + (ge.s ? k1 : k3);
        pl_pin<a0, a0 + na>(gzt);
    });
    pl_tilde<ALG>(gzt, ge);
#pragma unroll
    for (int j = 0; j < DL; ++j) gz[j] += gzt[j];
}
template <class ALG>
CSMPN_DEV void pl_weighted_gp_bwd_r(const float (&ggp)[PS<ALG>::DL], const PlState<ALG>& S, const float* wrow,
                                    const PlGeo<ALG>& ge, float (&gr)[PS<ALG>::DL]) {
    using P = PS<ALG>;
    constexpr int DL = P::DL, QP = P::QP;
    const float i2 = float(P::t.I2);
    float zE[DL], zO[DL], Gw[DL], Go[DL], grt[DL];
    static_for<0, DL>([&](auto jj) {
        constexpr int j = decltype(jj)::value;
        zE[j] = S.gate[P::t.cls[j]] * S.y[j];
        Gw[j] = ggp[j];
        grt[j] = 0.f;
    });
    pl_tilde<ALG>(zE, ge);
    pl_tilde<ALG>(Gw, ge);
#pragma unroll
    for (int j = 0; j < DL; ++j) {
        zO[j] = pl_odd(zE[j]);
        zE[j] = pl_even(zE[j]);
        Go[j] = pl_partner(Gw[j]);
    }
    CSMPN_PHASE();
    float wn[4];
    pl_ld_paths<ALG, 0>(wn, wrow);
    static_for<0, QP>([&](auto qq) {
        constexpr int q = decltype(qq)::value;
        constexpr int ka = P::t.qg[q][0], kc = P::t.qg[q][1], kb = P::t.qg[q][2];
        constexpr int a0 = P::t.cstart[ka], na = P::t.cstart[ka + 1] - a0;
        constexpr int c0 = P::t.cstart[kc], nc = P::t.cstart[kc + 1] - c0;
        constexpr int b0 = P::t.cstart[kb], nb = P::t.cstart[kb + 1] - b0;
        // d/dr of the even lanes: w00 Ge (x) Ze + w11 Go (x) Zo; of the odd lanes: w10 Go (x) Ze + I^2 w01 Ge (x) Zo
        const float w00 = wn[0], w01 = wn[1] * i2, w10 = wn[2], w11 = wn[3];
        if constexpr (q + 1 < QP) pl_ld_paths<ALG, q + 1>(wn, wrow);
        const float v1 = ge.s ? w10 : w00, v2 = ge.s ? w01 : w11;
        float V1[nb], V2[nb];
#pragma unroll
        for (int t = 0; t < nb; ++t) { V1[t] = 0.f; V2[t] = 0.f; }
        static_for<0, na>([&](auto aa) {
            static_for<0, nb>([&](auto bb) {
                constexpr int ai = decltype(aa)::value, bi = decltype(bb)::value;
                constexpr int a = a0 + ai, b = b0 + bi;
                constexpr int c = P::t.pc[a][b];
                if constexpr (c >= c0 && c < c0 + nc) {
                    constexpr float sg = float(P::t.psg[a][b]);
                    V1[bi] += (sg * Gw[c]) * zE[a];
                    V2[bi] += (sg * Go[c]) * zO[a];
                }
            });
        });
#pragma unroll
        for (int t = 0; t < nb; ++t) grt[b0 + t] += v1 * V1[t] + v2 * V2[t];
        pl_pin<b0, b0 + nb>(grt);
    });
    pl_tilde<ALG>(grt, ge);
#pragma unroll
    for (int j = 0; j < DL; ++j) gr[j] = grt[j];
}

// running sums of the small-parameter gradients of one block. They live in lane-private LDS slots
// ([slot][thread], no conflicts, no atomics) between tiles: 40 values per block would otherwise be live in
// registers for the whole launch, on top of the MFMA accumulators.
template <class ALG>
struct PlSumIdx {
    static constexpr int GC = PS<ALG>::GC, QP = PS<ALG>::QP;
    static constexpr int la = 0, bL = 1, b1 = 2, an = 3, sa = 3 + GC, sb = 3 + 2 * GC, wA = 3 + 3 * GC, wB = wA + QP;
    static constexpr int count = wB + QP;
};
constexpr int kPlThreads = 64 * kPlWaves;
CSMPN_DEV void pl_sum_add(float* slot, float v) { *slot = *slot + v; }

// CSMPN_FLAG_SAVE_STATE at D = 32: the forward stores, per block, the three tensors the backward would otherwise recompute
// through two channel mixes and a geometric product - y (MVLinear output with its bias), R (linear_right output) and s (the
// block's output in front of its layer norm). These kernels run at 5-10 % of the HBM roofline: the extra rows travel under the
// arithmetic. State regions of the saved buffer (cemlp_device.hpp): a wave's 4-row tile of one tensor is 4 pieces x 64 lanes x
// 16 bytes in lane order - no staging tile, no pick of the parity's blades, every instruction 1 KB of contiguous memory.
// p: piece 0 of this lane (tile slot * 1024 + 4 lane floats into the region).
template <class ALG>
CSMPN_DEV void pl_store_lane(float* p, const float (&sv)[PS<ALG>::DL]) {
    static_assert(PS<ALG>::DL == 16, "four 16-byte pieces per lane");
#pragma unroll
    for (int e = 0; e < 4; ++e)   // streaming stores: these rows are read once, by the backward - they should not evict the gathered h rows from L2
        __builtin_nontemporal_store(f4{sv[4 * e], sv[4 * e + 1], sv[4 * e + 2], sv[4 * e + 3]}, reinterpret_cast<f4*>(p + 256 * e));
}
// base: the saved buffer; off: this lane's offset inside a state region; ROW / ROWP as in state_region()
template <class ALG, int ROW, int ROWP>
CSMPN_DEV void pl_store_state(float* base, long rows, int K, size_t off, const PlState<ALG>& S) {
    pl_store_lane<ALG>(base + state_region<ROW, ROWP>(rows, 0, K) + off, S.s);
    pl_store_lane<ALG>(base + state_region<ROW, ROWP>(rows, 1, K) + off, S.y);
    pl_store_lane<ALG>(base + state_region<ROW, ROWP>(rows, 2, K) + off, S.R);
}
template <class ALG>
struct PlSaved {
    f4 y[4], R[4], s[4];
    template <int ROW, int ROWP>
    CSMPN_DEV void load(const float* base, long rows, int K, size_t off) {
        const float *py = base + state_region<ROW, ROWP>(rows, 1, K) + off, *pR = base + state_region<ROW, ROWP>(rows, 2, K) + off,
                    *ps = base + state_region<ROW, ROWP>(rows, 0, K) + off;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(py + 256 * e));
#pragma unroll
        for (int e = 0; e < 4; ++e) R[e] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(pR + 256 * e));
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(ps + 256 * e));
    }
};

// block forward behind the MVLinear: S.y holds the MVLinear output (without bias).
// SAVED (backward under CSMPN_FLAG_SAVE_STATE): y, R and s come from the forward (sv); what is left of the recompute are the
// gates, the normalisation's denominators and the layer norm's mean - no channel mix, no geometric product.
template <class ALG, class LY, int K, bool SAVED = false>
CSMPN_DEV void pl_block_tail(const float* lds, const PlGeo<ALG>& ge, PlState<ALG>& S, float (&out)[PS<ALG>::DL],
                             const PlSaved<ALG>* sv = nullptr) {
    using P = PS<ALG>;
    constexpr int DL = P::DL, GC = P::GC, G = ALG::G;
    const float* ldsn = lds + ge.n;
    const int c = ge.c;
    if constexpr (SAVED) {
#pragma unroll
        for (int j = 0; j < DL; ++j) { S.y[j] = sv->y[j / 4][j % 4]; S.R[j] = sv->R[j / 4][j % 4]; S.s[j] = sv->s[j / 4][j % 4]; }
    } else {
        if (ge.s == 0) S.y[0] += lds[LY::p_b1(K) + c];
    }
    // MVSiLU (cegnn_utils.py:76-83)
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
        S.gate[k] = sigmoidf(lds[LY::p_sa(K) + pg] * u + lds[LY::p_sb(K) + pg]);
        static_for<j0, j1>([&](auto jj) { z[decltype(jj)::value] = S.gate[k] * S.y[decltype(jj)::value]; });
    });
    CSMPN_PHASE();
    // linear_right / linear_left (cegnn_utils.py:143-148)
    float L[DL];
    if constexpr (!SAVED) {
#pragma unroll
        for (int j = 0; j < DL; ++j) { S.R[j] = 0.f; L[j] = 0.f; }
        pl_linear2<ALG, LY::t_WR(K), LY::t_WL(K)>(S.R, L, z, ldsn);
        if (ge.s == 0) L[0] += lds[LY::p_bL(K) + c];
    }
    CSMPN_PHASE();
    // NormalizationLayer on the right operand (cegnn_utils.py:42-51)
    float r[DL];
    static_for<0, GC>([&](auto k) {
        constexpr int j0 = P::t.cstart[k], j1 = P::t.cstart[k + 1];
        float qq = 0.f;
        static_for<j0, j1>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            qq += ge.template qs<j>() * S.R[j] * S.R[j];
        });
        const float sg = lds[LY::p_sg(K) + c * G + ge.grade(k)];
        const float m = sg * (smooth_abs_sqrt1(qq) - 1.0f) + 1.0f;
        S.invden[k] = fast_rcp(m + kEps);
        static_for<j0, j1>([&](auto jj) { r[decltype(jj)::value] = S.R[decltype(jj)::value] * S.invden[k]; });
    });
    CSMPN_PHASE();
    // steerable geometric product + first-order term (cegnn_utils.py:126-152)
    if constexpr (!SAVED) {
        pl_weighted_gp<ALG>(L, z, r, lds + LY::p_w(K) + c * ALG::P, ge);
#pragma unroll
        for (int j = 0; j < DL; ++j) S.s[j] = L[j] * kInvSqrt2;
    }
    CSMPN_PHASE();
    // MVLayerNorm (cegnn_utils.py:93-96): q over all blades = own half + partner's half
    float qs = 0.f;
    static_for<0, DL>([&](auto jj) {
        constexpr int j = decltype(jj)::value;
        qs += ge.template qs<j>() * S.s[j] * S.s[j];
    });
    qs += pl_partner(qs);
    S.qs = qs;
    S.nl = smooth_abs_sqrt1(qs);
    const float tot = pl_chan_sum(S.nl);
    S.invMn = fast_rcp(tot * (1.0f / float(LY::C)) + kEps);
    const float la = lds[LY::p_la(K) + c];
#pragma unroll
    for (int j = 0; j < DL; ++j) out[j] = la * S.s[j] * S.invMn;
}

// block backward from d/d(out) to d/d(MVLinear output) gy; accumulates WR / WL gradients and the small sums
template <class ALG, class LY, int K>
CSMPN_DEV void pl_block_backward(const float* lds, const PlGeo<ALG>& ge, const PlState<ALG>& S,
                                 const float (&gout)[PS<ALG>::DL], float (&gy)[PS<ALG>::DL], float* tot,
                                 f4 (&accWR)[PS<ALG>::GC], f4 (&accWL)[PS<ALG>::GC]) {
    // the 3 + 3 GC per-channel running sums of this thread (LDS): read here, written at the end of the block - as
    // read-modify-writes where the values are produced each one stalled the wave for an LDS round trip
    float sums[3 + 3 * PS<ALG>::GC];
#pragma unroll
    for (int i = 0; i < 3 + 3 * PS<ALG>::GC; ++i) sums[i] = tot[i * kPlThreads];
    using P = PS<ALG>;
    constexpr int DL = P::DL, GC = P::GC, G = ALG::G;
    const float* ldsn = lds + ge.n;
    const int c = ge.c;
    const float la = lds[LY::p_la(K) + c];
    // ---- MVLayerNorm backward
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < DL; ++j) dot += gout[j] * S.s[j];
    using SI = PlSumIdx<ALG>;
    sums[SI::la] += (dot * S.invMn);   // own half; the partner lane adds its own
    dot += pl_partner(dot);
    const float gMn = pl_chan_sum(-(la * dot) * S.invMn * S.invMn);   // sum over the 8 channels (lanes of this parity)
    const float inl = fast_rcp(S.nl);
    const float gqs = (gMn * (1.0f / float(LY::C))) * (0.5f * S.qs) * (inl * inl * inl);
    float ggp[DL];
    static_for<0, DL>([&](auto jj) {
        constexpr int j = decltype(jj)::value;
        const float gs = (la * gout[j]) * S.invMn + gqs * (2.0f * ge.template qs<j>()) * S.s[j];
        ggp[j] = gs * kInvSqrt2;
    });
    sums[SI::bL] += (ggp[0]);   // (read back from the even lanes only)
    CSMPN_PHASE();
    // ---- d/dz from linear_left; gWL += GL (x) Z
    float z[DL];
    static_for<0, DL>([&](auto jj) { z[decltype(jj)::value] = S.gate[P::t.cls[decltype(jj)::value]] * S.y[decltype(jj)::value]; });
    float gz[DL];
#pragma unroll
    for (int j = 0; j < DL; ++j) gz[j] = 0.f;
    pl_linear<ALG, LY::t_WLt(K)>(gz, ggp, ldsn);
    pl_wgrad<ALG>(accWL, ggp, z);
    CSMPN_PHASE();
    // ---- geometric product backward
    pl_weighted_gp_bwd_z<ALG, kPlThreads>(ggp, S, lds + LY::p_w(K) + c * ALG::P, ge, gz, tot + SI::wA * kPlThreads, tot + SI::wB * kPlThreads);
    CSMPN_PHASE();
    float gr[DL];
    pl_weighted_gp_bwd_r<ALG>(ggp, S, lds + LY::p_w(K) + c * ALG::P, ge, gr);
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
        const float sg = lds[LY::p_sg(K) + c * G + ge.grade(k)];
        const float nu = smooth_abs_sqrt1(qR);
        sums[SI::an + k] += (gden * (nu - 1.0f) * sg * (1.0f - sg));
        const float inu = fast_rcp(nu);
        const float gq = (gden * sg) * (0.5f * qR) * (inu * inu * inu);
        static_for<j0, j1>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            gR[j] = gr[j] * S.invden[k] + gq * (2.0f * ge.template qs<j>()) * S.R[j];
        });
    });
    pl_linear<ALG, LY::t_WRt(K)>(gz, gR, ldsn);
    pl_wgrad<ALG>(accWR, gR, z);
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
        const float gu = gpre * lds[LY::p_sa(K) + c * G + ge.grade(k)];
        static_for<j0, j1>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            float v = gz[j] * S.gate[k];
            const float quad = gu * (2.0f * ge.template qs<j>()) * S.y[j];
            if constexpr (k == 0 && j == 0) v += scalar_inv ? gu : quad;
            else v += quad;
            gy[j] = v;
        });
    });
    sums[SI::b1] += (gy[0]);
#pragma unroll
    for (int i = 0; i < 3 + 3 * PS<ALG>::GC; ++i) tot[i * kPlThreads] = sums[i];
}

// ---------------------------------------------------------------------------------
// end of a backward launch: every WAVE writes its parameter-gradient sums to its own slice of a partial buffer in the
// workspace - the MFMA tiles as they are ([tile][class][v][lane], coalesced), the per-channel sums after a fixed-order
// sum over the 4 row quarters (one MFMA with A = 1 per value) - and pl_reduce_kernel adds the slices in a fixed order:
// no atomics anywhere, the gradients are bit-reproducible.
template <class LY>
struct PlPart {
    static constexpr int GC = LY::GC, G = LY::G, NP = LY::NP, C = LY::C, NCH0 = LY::nch(0);
    static constexpr int n_tiles = NCH0 + 5;     // W1_0 chunks, WR_0, WL_0, W1_1, WR_1, WL_1
    static constexpr int w_floats = n_tiles * GC * 4 * 64;
    static constexpr int i_b1 = 0, i_bL = C, i_la = 2 * C, i_sa = 3 * C, i_sb = i_sa + C * G, i_an = i_sb + C * G,
                         i_w = i_an + C * G, i_blk = i_w + C * NP;
    static constexpr int slice = (w_floats + 2 * i_blk + 3) & ~3;
};
CSMPN_DEV float pl_quarter_sum(float v) {
    const f4 r = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, v, splat(0.f), 0, 0, 0);   // D[i][j] = sum_k B[k][j]
    return r.x;
}
template <class ALG>
CSMPN_DEV void pl_store_tile(float* slice, int tile_idx, const f4 (&acc)[PS<ALG>::GC], int lane) {
    constexpr int GC = PS<ALG>::GC;
#pragma unroll
    for (int k = 0; k < GC; ++k)
#pragma unroll
        for (int v = 0; v < 4; ++v) slice[((tile_idx * GC + k) * 4 + v) * 64 + lane] = acc[k][v];
}
// per-channel sums of block K: quarter sums -> the wave's scratch image (single writer per entry) -> slice tail
template <class ALG, class LY, int K>
CSMPN_DEV void pl_store_small(float* slice_tail, float* sc, const float* tot, const PlGeo<ALG>& ge) {
    using P = PS<ALG>;
    using SI = PlSumIdx<ALG>;
    using PP = PlPart<LY>;
    constexpr int GC = P::GC, G = ALG::G, NP = ALG::P, C = LY::C;
    constexpr int L_la2 = PP::i_blk;     // la per (channel, parity) behind the block image
    const int c = ge.c;
    const bool w0 = ge.q == 0;
    tile_sync<VAR_WAVE>();
    for (int e = ge.lane; e < PP::i_blk + 2 * C; e += 64) sc[e] = 0.f;
    tile_sync<VAR_WAVE>();
    {
        const float la = pl_quarter_sum(tot[SI::la * kPlThreads]);
        const float bL = pl_quarter_sum(tot[SI::bL * kPlThreads]);
        const float b1 = pl_quarter_sum(tot[SI::b1 * kPlThreads]);
        if (w0) {
            sc[L_la2 + 2 * c + ge.s] = la;
            if (ge.s == 0) { sc[PP::i_bL + c] = bL; sc[PP::i_b1 + c] = b1; }
        }
    }
#pragma unroll
    for (int k = 0; k < GC; ++k) {
        const int pg = c * G + ge.grade(k);
        const float an = pl_quarter_sum(tot[(SI::an + k) * kPlThreads]);
        const float sa = pl_quarter_sum(tot[(SI::sa + k) * kPlThreads]);
        const float sb = pl_quarter_sum(tot[(SI::sb + k) * kPlThreads]);
        if (w0) { sc[PP::i_an + pg] = an; sc[PP::i_sa + pg] = sa; sc[PP::i_sb + pg] = sb; }
    }
    static_for<0, P::QP>([&](auto qq) {
        constexpr int q = decltype(qq)::value;
        const float a = pl_quarter_sum(tot[(SI::wA + q) * kPlThreads]);
        const float b = pl_quarter_sum(tot[(SI::wB + q) * kPlThreads]) * (ge.s ? 1.0f : float(P::t.I2));
        if (w0) {
            sc[PP::i_w + c * NP + (ge.s ? P::t.pid[1][0][q] : P::t.pid[0][0][q])] = a;
            sc[PP::i_w + c * NP + (ge.s ? P::t.pid[1][1][q] : P::t.pid[0][1][q])] = b;
        }
    });
    tile_sync<VAR_WAVE>();
    for (int e = ge.lane; e < PP::i_blk; e += 64) {
        float v = sc[e];
        if (e >= PP::i_la && e < PP::i_sa) v = sc[L_la2 + 2 * (e - PP::i_la)] + sc[L_la2 + 2 * (e - PP::i_la) + 1];
        slice_tail[e] = v;
    }
    tile_sync<VAR_WAVE>();
}
// slice subsets of the reduce kernels (cemlp_pl.hpp, cemlp_plw.hpp): 64 slice positions x kPlReduceSubs subsets per workgroup
constexpr int kPlReduceSubs = 16;
// fixed-order sum of the subsets' partial sums of position `pos`
CSMPN_DEV float pl_reduce_combine(const float (&red)[kPlReduceSubs][64], int pos) {
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < kPlReduceSubs; k += 4) a += (red[k][pos] + red[k + 1][pos]) + (red[k + 2][pos] + red[k + 3][pos]);
    return a;
}
// grads += sum over the waves' slices, fixed order; one thread per slice element (64 per workgroup x 16 slice subsets:
// with 4 subsets the 1 024 slices of an S3 launch took 23.7 us)
template <class ALG, int NBLK, int I0>
__global__ void __launch_bounds__(64 * kPlReduceSubs) pl_reduce_kernel(const DevCemlp Cd, const float* part, int nslices) {
    using LY = PlLay<ALG, NBLK, I0>;
    using PP = PlPart<LY>;
    constexpr int GC = LY::GC, G = ALG::G, C = LY::C, NCH0 = PP::NCH0, NP = ALG::P, NS = kPlReduceSubs;
    __shared__ float red[NS][64];
    const int sub = threadIdx.x >> 6;
    const long t = (long)blockIdx.x * 64 + (threadIdx.x & 63);
    float* dst = nullptr;
    if (t < PP::w_floats) {
        const int lane = (int)(t % 64), slot = (int)(t / 64);
        const int v = slot & 3, k = (slot >> 2) % GC, tile_idx = (slot >> 2) / GC;
        const int q = lane >> 4, n = lane & 15, c = n >> 1, s = n & 1;
        const int i = 4 * q + v, o = i >> 1, so = i & 1;
        int blk, table, ch;   // table: 0 W1, 1 WR, 2 WL
        if (tile_idx < NCH0) { blk = 0; table = 0; ch = tile_idx; }
        else if (tile_idx < NCH0 + 2) { blk = 0; table = 1 + (tile_idx - NCH0); ch = 0; }
        else { blk = 1; table = tile_idx - NCH0 - 2; ch = 0; }
        const int I = (blk == 0 && table == 0) ? I0 : C;
        const int cin = 8 * ch + c;
        const DevBlock& B = Cd.b[blk];
        float* gW = table == 0 ? B.gW1 : (table == 1 ? B.gWR : B.gWL);
        if (so == s && cin < I && gW) dst = gW + ((size_t)o * I + cin) * G + (s ? ALG::n - 2 * k : 2 * k);
    } else if (t < PP::w_floats + 2 * PP::i_blk) {
        const int e2 = (int)(t - PP::w_floats), blk = e2 / PP::i_blk, e = e2 % PP::i_blk;
        const DevBlock& B = Cd.b[blk];
        if (e < PP::i_bL) { if (B.has_b1 && B.gb1) dst = B.gb1 + e; }
        else if (e < PP::i_la) { if (B.gbL) dst = B.gbL + (e - PP::i_bL); }
        else if (e < PP::i_sa) { if (B.gla) dst = B.gla + (e - PP::i_la); }
        else if (e < PP::i_sb) { if (B.gsa) dst = B.gsa + (e - PP::i_sa); }
        else if (e < PP::i_an) { if (B.gsb) dst = B.gsb + (e - PP::i_sb); }
        else if (e < PP::i_w) { if (B.gan) dst = B.gan + (e - PP::i_an); }
        else { if (B.gw) dst = B.gw + (e - PP::i_w); }
        (void)NP;
    }
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (dst) {
        const float* p = part + t;
        int g = sub;
        for (; g + 3 * NS < nslices; g += 4 * NS) {
            s0 += p[(size_t)g * PP::slice];
            s1 += p[(size_t)(g + NS) * PP::slice];
            s2 += p[(size_t)(g + 2 * NS) * PP::slice];
            s3 += p[(size_t)(g + 3 * NS) * PP::slice];
        }
        for (; g < nslices; g += NS) s0 += p[(size_t)g * PP::slice];
    }
    red[sub][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sub == 0 && dst) *dst += pl_reduce_combine(red, threadIdx.x);
}

// staged rows -> global. The staging tile holds kPlRows rows of `ncol` floats (row stride rs).
// coalesced copy: out_row(r) = pointer to the destination of tile row r, or nullptr
template <class F>
CSMPN_DEV void pl_copy_rows(const float* sc, int rs, int ncol, int lane, F&& out_row) {
#pragma unroll
    for (int r = 0; r < kPlRows; ++r) {
        float* dst = out_row(r);
        if (dst)
            for (int e = 4 * lane; e < ncol; e += 256) *reinterpret_cast<f4*>(dst + e) = pl_ld4(sc + r * rs + e);
    }
}
// scatter-add of the staged rows to table[t_add[row]] (sorted targets: equal consecutive targets are summed first)
// and, when SUB, subtraction from table[t_sub[row]]. Negative targets are skipped. t_* live in lane 16 r of row r.
template <int ROWLEN, bool SUB>
CSMPN_DEV void pl_scatter(const float* sc, int rs, int t_add, int t_sub, float* table, int lane) {
    static_for<0, ROWLEN / 64>([&](auto cc) {
        const int col = 64 * decltype(cc)::value + lane;
        float val[kPlRows];
#pragma unroll
        for (int r = 0; r < kPlRows; ++r) val[r] = sc[r * rs + col];
        float acc = 0.f;
        int cur = __builtin_amdgcn_readlane(t_add, 0);
        static_for<0, kPlRows>([&](auto rr) {
            constexpr int r = decltype(rr)::value;
            const int t = __builtin_amdgcn_readlane(t_add, 16 * r);
            if (t != cur) {
                if (cur >= 0) atomicAdd(table + (long)cur * ROWLEN + col, acc);
                cur = t;
                acc = 0.f;
            }
            acc += val[r];
        });
        if (cur >= 0) atomicAdd(table + (long)cur * ROWLEN + col, acc);
        if constexpr (SUB) {
            static_for<0, kPlRows>([&](auto rr) {
                constexpr int r = decltype(rr)::value;
                const int t = __builtin_amdgcn_readlane(t_sub, 16 * r);
                if (t >= 0) atomicAdd(table + (long)t * ROWLEN + col, -val[r]);
            });
        }
    });
}

// ---------------------------------------------------------------------------------
// the kernel. NBLK = 2 blocks of 8 channels; block 0 has I0 input channels.
// MODE_EDGE: I0 = 8 + A; MODE_NODE: I0 = 16 + T.
// SAVES (backward): the forward ran with CSMPN_FLAG_SAVE_STATE - regions 2 .. 7 behind [block-1 inputs | hand-over] of the
// saved buffer hold s, y, R of block 0 / 1 (pl_store_state; a compile-time choice, as in cemlp_cl.hpp).
template <class ALG, int MODE, int NBLK, int I0, bool BWD, bool SAVES = false>
__global__ void __launch_bounds__(64 * kPlWaves, BWD ? CSMPN_PL_BWD_WAVES : CSMPN_PL_FWD_WAVES) cemlp_pl_kernel(const DevCemlp C_arg, const RowIO io_arg) {
    typedef const char __attribute__((address_space(4))) * KArgPtr;
    const KArgPtr ka = (KArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr size_t kIoOffset = (sizeof(DevCemlp) + alignof(RowIO) - 1) / alignof(RowIO) * alignof(RowIO);
    const DevCemlp& Cd = *(const DevCemlp*)(const char*)ka;
    const RowIO& io = *(const RowIO*)(const char*)(ka + kIoOffset);
    (void)C_arg; (void)io_arg;
    using LY = PlLay<ALG, NBLK, I0>;
    using P = PS<ALG>;
    constexpr int D = ALG::D, DL = P::DL, GC = P::GC, C = 8, ROW = C * D, RS = LY::RS, NCH0 = LY::nch(0);
    constexpr int NA = MODE == MODE_EDGE ? I0 - C : (MODE == MODE_NODE ? I0 - 2 * C : 0);
    static_assert(NBLK == 2, "two blocks");
    static_assert(MODE == MODE_EDGE || MODE == MODE_NODE, "edge / node programs");
    static_assert(NA >= 0 && NA <= 8, "attribute channels must fit one chunk");
    static_assert(LY::bwd_total * 4 <= 160 * 1024, "LDS footprint");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* lds0 = smem;
    const int wave = threadIdx.x >> 6;
    const PlGeo<ALG> ge(threadIdx.x & 63);
    float* sc = lds0 + LY::sc_off + wave * LY::scratch;
    float* lds = lds0;
    {
        const int probe = pl_dpp_i<0x122>(ge.n);
        const int dir = (((probe - ge.n) & 15) == 2) ? 1 : -1;
        pl_stage_block<LY, ALG, 0>(lds, Cd.b[0], threadIdx.x, dir);
        pl_stage_block<LY, ALG, 1>(lds, Cd.b[1], threadIdx.x, dir);
        if constexpr (BWD)
            for (int e = threadIdx.x; e < NBLK * LY::tot_blk; e += 64 * kPlWaves) lds0[LY::tot_off + e] = 0.f;   // running sums
    }
    __syncthreads();

    // parameter-gradient accumulators (backward): MFMA tiles over the whole launch, per-lane small sums
    f4 aW1_0[NCH0][GC], aWR_0[GC], aWL_0[GC], aW1_1[GC], aWR_1[GC], aWL_1[GC];
    // the sums sit beyond the 64 KB reach of a ds_* immediate offset: opaque per-block bases, so that every slot is
    // base + immediate (otherwise the compiler materialises, hoists and spills one address register per slot)
    unsigned tot0_a = (unsigned)(LY::tot_off + threadIdx.x) * 4u, tot1_a = tot0_a + LY::tot_blk * 4u;
    asm volatile("" : "+v"(tot0_a));
    asm volatile("" : "+v"(tot1_a));
    float* tot0 = reinterpret_cast<float*>(reinterpret_cast<char*>(lds0) + tot0_a);
    float* tot1 = reinterpret_cast<float*>(reinterpret_cast<char*>(lds0) + tot1_a);
    if constexpr (BWD) {
#pragma unroll
        for (int k = 0; k < GC; ++k) {
#pragma unroll
            for (int ch = 0; ch < NCH0; ++ch) aW1_0[ch][k] = splat(0.f);
            aWR_0[k] = splat(0.f); aWL_0[k] = splat(0.f); aW1_1[k] = splat(0.f); aWR_1[k] = splat(0.f); aWL_1[k] = splat(0.f);
        }
    }

    const long ntiles = (io.rows + kPlRows - 1) / kPlRows;
    const long tstep = (long)gridDim.x * kPlWaves;
    struct Idx { int dst, src, perm, deg; };
    auto load_idx = [&](long tile) -> Idx {
        Idx x{-1, -1, 0, 1};
        const long row = tile * kPlRows + ge.q;
        if (tile < ntiles && row < io.rows) {
            if constexpr (MODE == MODE_EDGE) {
                x.dst = io.seg[0].ia[row];
                x.src = io.seg[0].ib[row];
                if constexpr (NA > 0) x.perm = io.seg[1].ia[row];
            } else {
                if (io.seg[1].deg) x.deg = io.seg[1].deg[row];
            }
        }
        return x;
    };
    long tile = (long)blockIdx.x * kPlWaves + wave;
    Idx nxt = load_idx(tile);
    for (; tile < ntiles; tile += tstep) {
        // the parameter store is read through a base that is opaque per iteration: with compile-time offsets the
        // compiler proves the weight reads invariant, hoists hundreds of them out of the tile loop and spills them
        unsigned zoff = 0;
        asm volatile("" : "+v"(zoff));
        const float* lds = reinterpret_cast<const float*>(reinterpret_cast<const char*>(lds0) + zoff);
        const float* ldsn = lds + ge.n;
        const Idx ix = nxt;
        nxt = load_idx(tile + tstep);
        const long row = tile * kPlRows + ge.q;
        const bool valid = row < io.rows;
        const long lrow = valid ? row : 0;
        const float scale = (MODE == MODE_NODE && io.seg[1].deg) ? 1.0f / float(ix.deg > 1 ? ix.deg : 1) : 1.0f;
        const int cofs = ge.c * D;
        // chunk ch of the block-0 input of this lane's row
        auto load_chunk = [&](auto chc, float (&x)[DL]) {
            constexpr int ch = decltype(chc)::value;
            if constexpr (MODE == MODE_EDGE) {
                if constexpr (ch == 0) {
                    const int d = valid ? ix.dst : 0, s = valid ? ix.src : 0;
                    pl_load_diff<ALG>(x, io.seg[0].a + (size_t)d * ROW + cofs, io.seg[0].b + (size_t)s * ROW + cofs, ge.s,
                                      valid ? 1.0f : 0.0f);
                } else {
                    const bool on = valid && ge.c < NA;
                    pl_load<ALG>(x, io.seg[1].a + (size_t)(on ? ix.perm : 0) * (NA * D) + (on ? cofs : 0), ge.s, on ? 1.0f : 0.0f);
                }
            } else {
                if constexpr (ch == 0) pl_load<ALG>(x, io.seg[0].a + (size_t)lrow * ROW + cofs, ge.s, valid ? 1.0f : 0.0f);
                else if constexpr (ch == 1) pl_load<ALG>(x, io.seg[1].a + (size_t)lrow * ROW + cofs, ge.s, valid ? scale : 0.0f);
                else {
                    const bool on = valid && ge.c < NA;
                    pl_load<ALG>(x, io.seg[2].a + (size_t)(on ? lrow : 0) * (NA * D) + (on ? cofs : 0), ge.s, on ? 1.0f : 0.0f);
                }
            }
        };
        // MVLinear of block 0 (chunk by chunk: only one chunk of the input is live)
        auto mvlinear0 = [&](float (&y)[DL]) {
#pragma unroll
            for (int j = 0; j < DL; ++j) y[j] = 0.f;
            static_for<0, NCH0>([&](auto chc) {
                float x[DL];
                load_chunk(chc, x);
                pl_linear<ALG, LY::t_W1(0, decltype(chc)::value)>(y, x, ldsn);
            });
        };

        if constexpr (!BWD) {
            PlState<ALG> S;
            float out[DL];
            const bool save_s = io.save_state != 0 && io.save != nullptr;
            auto store_s = [&](int blk) {   // CSMPN_FLAG_SAVE_STATE: this block's s, y, R -> its state regions (tile slot = tile)
                if (valid) pl_store_state<ALG, ROW, ROW>(io.save, io.rows, blk, (size_t)tile * (kPlRows * ROW) + 4 * ge.lane, S);
            };
            mvlinear0(S.y);
            pl_block_tail<ALG, LY, 0>(lds, ge, S, out);
            if (save_s) store_s(0);
            if (io.save) {
                tile_sync<VAR_WAVE>();
                pl_stage<ALG>(sc, out, ge, RS, true);
                tile_sync<VAR_WAVE>();
                pl_copy_rows(sc, RS, ROW, ge.lane, [&](int r) -> float* {
                    const long rr = tile * kPlRows + r;
                    return rr < io.rows ? io.save + (size_t)rr * ROW : nullptr;
                });
            }
            {
                float in1[DL];
#pragma unroll
                for (int j = 0; j < DL; ++j) { in1[j] = out[j]; S.y[j] = 0.f; }
                pl_linear<ALG, LY::t_W1(1, 0)>(S.y, in1, ldsn);
            }
            pl_block_tail<ALG, LY, 1>(lds, ge, S, out);
            if (save_s) store_s(1);
            if constexpr (MODE == MODE_NODE) {
                if (io.resid) {
                    float res[DL];
                    pl_load<ALG>(res, io.resid + (size_t)lrow * ROW + cofs, ge.s, 1.0f);
#pragma unroll
                    for (int j = 0; j < DL; ++j) out[j] += res[j];
                }
            }
            tile_sync<VAR_WAVE>();
            pl_stage<ALG>(sc, out, ge, RS, true);
            tile_sync<VAR_WAVE>();
            if constexpr (MODE == MODE_EDGE) {
                if (io.row_store) {
                    pl_copy_rows(sc, RS, ROW, ge.lane, [&](int r) -> float* {
                        const long rr = tile * kPlRows + r;
                        return rr < io.rows ? io.agg + (size_t)rr * ROW : nullptr;
                    });
                } else {
                    pl_scatter<ROW, false>(sc, RS, valid ? ix.dst : -1, -1, io.agg, ge.lane);
                }
            } else {
                pl_copy_rows(sc, RS, ROW, ge.lane, [&](int r) -> float* {
                    const long rr = tile * kPlRows + r;
                    return rr < io.rows ? io.y + (size_t)rr * ROW : nullptr;
                });
            }
        } else {
            // ------------------------------------------------------------ backward
            float gout[DL];
            {
                const long grow = MODE == MODE_EDGE ? (long)(valid ? ix.dst : 0) : lrow;
                pl_load<ALG>(gout, io.gy + (size_t)grow * ROW + cofs, ge.s, valid ? 1.0f : 0.0f);
            }
            float g1[DL];   // d/d(block-1 input)
            {
                float in1[DL], gy[DL];
                pl_load<ALG>(in1, io.saved + (size_t)lrow * ROW + cofs, ge.s, valid ? 1.0f : 0.0f);
                {
                    PlState<ALG> S;
                    float unused[DL];
                    if constexpr (SAVES) {
                        PlSaved<ALG> sv;
                        sv.template load<ROW, ROW>(io.saved, io.rows, 1, (size_t)(valid ? tile : 0) * (kPlRows * ROW) + 4 * ge.lane);
                        pl_block_tail<ALG, LY, 1, true>(lds, ge, S, unused, &sv);
                    } else {
#pragma unroll
                        for (int j = 0; j < DL; ++j) S.y[j] = 0.f;
                        pl_linear<ALG, LY::t_W1(1, 0)>(S.y, in1, ldsn);
                        pl_block_tail<ALG, LY, 1>(lds, ge, S, unused);
                    }
                    pl_block_backward<ALG, LY, 1>(lds, ge, S, gout, gy, tot1, aWR_1, aWL_1);
                }
                pl_wgrad<ALG>(aW1_1, gy, in1);
#pragma unroll
                for (int j = 0; j < DL; ++j) g1[j] = 0.f;
                pl_linear<ALG, LY::t_W1t(1, 0)>(g1, gy, ldsn);
            }
            CSMPN_PHASE();
            float gy0[DL];
            {
                PlState<ALG> S;
                float unused[DL];
                if constexpr (SAVES) {
                    PlSaved<ALG> sv;
                    sv.template load<ROW, ROW>(io.saved, io.rows, 0, (size_t)(valid ? tile : 0) * (kPlRows * ROW) + 4 * ge.lane);
                    pl_block_tail<ALG, LY, 0, true>(lds, ge, S, unused, &sv);
                } else {
                    mvlinear0(S.y);
                    pl_block_tail<ALG, LY, 0>(lds, ge, S, unused);
                }
                pl_block_backward<ALG, LY, 0>(lds, ge, S, g1, gy0, tot0, aWR_0, aWL_0);
            }
            CSMPN_PHASE();
            // W1 gradient and d/d(input), chunk by chunk (the chunk is gathered again: it was not kept)
            static_for<0, NCH0>([&](auto chc) {
                constexpr int ch = decltype(chc)::value;
                {
                    float x[DL];
                    load_chunk(chc, x);
                    pl_wgrad<ALG>(aW1_0[ch], gy0, x);
                }
                float gx[DL];
#pragma unroll
                for (int j = 0; j < DL; ++j) gx[j] = 0.f;
                pl_linear<ALG, LY::t_W1t(0, ch)>(gx, gy0, ldsn);
                if constexpr (MODE == MODE_EDGE) {
                    if constexpr (ch == 0) {
                        if (io.gx[0]) {
                            tile_sync<VAR_WAVE>();
                            pl_stage<ALG>(sc, gx, ge, RS, true);
                            tile_sync<VAR_WAVE>();
                            if (io.row_store) {
                                pl_copy_rows(sc, RS, ROW, ge.lane, [&](int r) -> float* {
                                    const long rr = tile * kPlRows + r;
                                    return rr < io.rows ? io.gx[0] + (size_t)rr * ROW : nullptr;
                                });
                            } else {
                                pl_scatter<ROW, true>(sc, RS, valid ? ix.dst : -1, valid ? ix.src : -1, io.gx[0], ge.lane);
                            }
                        }
                    } else if (io.gx[1]) {
                        tile_sync<VAR_WAVE>();
                        pl_stage<ALG>(sc, gx, ge, RS, ge.c < NA);
                        tile_sync<VAR_WAVE>();
                        pl_copy_rows(sc, RS, NA * D, ge.lane, [&](int r) -> float* {
                            const long rr = tile * kPlRows + r;
                            const int pm = __builtin_amdgcn_readlane(ix.perm, 16 * r);
                            return rr < io.rows ? io.gx[1] + (size_t)pm * (NA * D) : nullptr;
                        });
                    }
                } else {
                    float* dstp = io.gx[ch];
                    if (dstp) {
                        if constexpr (ch == 0) {
                            if (io.resid_bwd) {
#pragma unroll
                                for (int j = 0; j < DL; ++j) gx[j] += gout[j];
                            }
                        }
                        if constexpr (ch == 1) {
#pragma unroll
                            for (int j = 0; j < DL; ++j) gx[j] *= scale;
                        }
                        constexpr int ncol = ch < 2 ? ROW : NA * D;
                        tile_sync<VAR_WAVE>();
                        pl_stage<ALG>(sc, gx, ge, RS, ch < 2 || ge.c < NA);
                        tile_sync<VAR_WAVE>();
                        pl_copy_rows(sc, RS, ncol, ge.lane, [&](int r) -> float* {
                            const long rr = tile * kPlRows + r;
                            return rr < io.rows ? dstp + (size_t)rr * ncol : nullptr;
                        });
                    }
                }
            });
        }
    }

    if constexpr (BWD) {
        using PP = PlPart<LY>;
        float* slice = io.plw_part + ((size_t)blockIdx.x * kPlWaves + wave) * PP::slice;
        static_for<0, NCH0>([&](auto chc) { pl_store_tile<ALG>(slice, decltype(chc)::value, aW1_0[decltype(chc)::value], ge.lane); });
        pl_store_tile<ALG>(slice, NCH0, aWR_0, ge.lane);
        pl_store_tile<ALG>(slice, NCH0 + 1, aWL_0, ge.lane);
        pl_store_tile<ALG>(slice, NCH0 + 2, aW1_1, ge.lane);
        pl_store_tile<ALG>(slice, NCH0 + 3, aWR_1, ge.lane);
        pl_store_tile<ALG>(slice, NCH0 + 4, aWL_1, ge.lane);
        pl_store_small<ALG, LY, 0>(slice + PP::w_floats, sc, tot0, ge);
        pl_store_small<ALG, LY, 1>(slice + PP::w_floats + PP::i_blk, sc, tot1, ge);
    }
}

}  // namespace csmpn
