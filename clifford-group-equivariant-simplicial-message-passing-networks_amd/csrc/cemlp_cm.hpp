// Channel-MFMA ("cm") CEMLP kernels for Cl(3,0) layers whose blocks are all C = 16 MB channels wide (16 and 32 channels:
// the S2 width and md17's).
//
// Same arithmetic as cemlp_device.hpp / cemlp_cl.hpp (csmpn/models/cegnn_utils.py:34-155,287-338).
// The mapping is chosen so that the dense channel mixing runs on v_mfma_f32_16x16x4_f32 WITHOUT any exchange between
// lanes in front of it (the row-per-lane-group kernels of cemlp_rl.hpp rotate every input value through the lanes of a
// row with one DPP move per MFMA operand; the (row, channel)-per-lane kernels of cemlp_cl.hpp pay one half-rate DPP
// FMA per multiply):
//
//  * a wave covers 16 rows; lane = (row = lane & 15, q = lane >> 4) holds the channels 16 m + 4 v + q (m < MB, v < 4)
//    of its row, all 8 blades: a tensor is  f4 t[MB][8]  (component v of t[m][d] = channel 16 m + 4 v + q, blade d).
//    (Any bijection (q, v) <-> channel serves the MFMA as long as the weight table follows it; this one puts the four
//    lanes of a row on four neighbouring 32-byte pieces of the row, so a gather / store instruction touches one 128-byte
//    line per row instead of four.)
//  * mixing  out[r, o, d] = sum_c W[o][c][grade d] x[r, c, d]:  D = A B with M = 16 output channels, N = the 16 rows,
//    K = 4 input channels per instruction. B[k][n] sits in lane n + 16 k: with k = q and the step (m, v) that IS the
//    layout above - the operand is the register t[m][d][v] as it stands. D[i][n] arrives in lane n + 16 (i / 4), register
//    i % 4: with output channel 16 m' + 4 (i % 4) + i / 4 in row i of A that is the same layout again, so the result of
//    one mixing feeds the next without a move. A[i][k] = W[16 m' + 4 (i % 4) + i / 4][16 m + 4 v + k][g]: one ds_read_b128 per (grade, m', m) from a
//    table [g][m'][chunk][lane] of f4 (v = 0..3) built once per workgroup from the reference layout [o][c][g]; it
//    serves the 1 or 3 blades of its grade. 32 MB^2 MFMAs per mixing of 16 rows, no VALU instruction at all.
//    Input segments narrower than 16 channels (the attribute channels) are one chunk whose slot (q, v) holds channel
//    q + 4 v: ceil(NA / 4) steps instead of 4.
//  * gates, normalisation, the geometric product and the layer norm are per (row, channel): the cemlp_cl.hpp code, once
//    per multivector the lane holds (4 MB of them), parameters read from LDS by channel (16 lanes share an address).
//    The one cross-lane sum (LayerNorm mean over the channels of a row) is the in-lane sum over (m, v) + ONE MFMA with
//    A = 1 over the four q.
//  * gathers: 128 contiguous bytes per (lane, m) (4 channels x 8 blades); scatters through a per-wave LDS tile so that one
//    atomic instruction covers whole rows, equal consecutive targets summed first (as cemlp_cl.hpp).
#pragma once
#include "cemlp_cl.hpp"

namespace csmpn {

constexpr int kCmWaves = 4;   // waves per workgroup
constexpr int kCmRows = 16;   // rows per wave tile
constexpr int kCmFwdStageRows = 8;   // rows of the forward's scatter staging tile (two passes per wave tile)

// Scheduling fence between the per-channel sections of the backward: left alone, the scheduler interleaves the four
// channels of a lane for instruction-level parallelism and quadruples the live temporaries (2 KB of scratch per lane).
#define CM_FENCE() __builtin_amdgcn_sched_barrier(0)
// Ordering point between LDS accesses of DIFFERENT lanes of one wave (a lane stores into the per-wave staging tile, another
// lane reads those bytes): the LDS executes a wave's operations in issue order, so no wait is needed - but the compiler
// sees each lane's own stores and loads as disjoint addresses and may move them across each other (it did move the second
// half's stores in front of the first half's reads: found by the 101-edge parity case). A compiler barrier + the wave
// barrier intrinsic (a scheduling barrier, no instruction) state the contract at every such hand-over.
#define CM_LDS_ORDER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); } while (0)
#ifndef CM_FWD_OCC
#define CM_FWD_OCC 3   // workgroups per CU the forward is compiled for (~160 registers, <= 53 KB of LDS)
#endif

// block K of a CEMLP in mode MODE with NA attribute channels: its 16-slot input chunks and its LDS tables (float offsets)
template <int C, int MODE, int NA, int K>
struct CmTab {
    static_assert(C % 16 == 0 && C >= 16 && C <= 32, "16 or 32 channels");
    static_assert(MODE == MODE_EDGE || MODE == MODE_NODE, "edge or node program");
    static_assert(NA >= 0 && NA <= 16, "attribute channels fit one chunk");
    static constexpr int MB = C / 16, NA_ = NA;
    static constexpr int NSEG = MODE == MODE_EDGE ? 1 : 2;                       // full-width segments in front of the attributes
    static constexpr int NCH = K > 0 ? MB : NSEG * MB + (NA > 0 ? 1 : 0);        // chunks
    static constexpr int I = K > 0 ? C : NSEG * C + NA;
    static constexpr bool attr(int ch) { return K == 0 && NA > 0 && ch == NCH - 1; }
    static constexpr int nstep(int ch) { return attr(ch) ? (NA + 3) / 4 : 4; }
    // channel of the concatenated input in slot (q, v) of chunk ch (-1: empty slot)
    static constexpr int chan(int ch, int q, int v) {
        if (attr(ch)) return q + 4 * v < NA ? NSEG * C + q + 4 * v : -1;
        return 16 * ch + 4 * v + q;
    }
    // output channel (inside its group of 16) that row i of the A operand / lane l16 of a table entry stands for
    static constexpr int orow(int i) { return 4 * (i & 3) + (i >> 2); }
    static constexpr int ENT = 64 * 4;   // floats of one (grade, m', chunk): one f4 per lane
    static constexpr int w1(int g, int mp, int ch) { return ((g * MB + mp) * NCH + ch) * ENT; }
    static constexpr int n1 = 4 * MB * NCH * ENT;
    static constexpr int nc = 4 * MB * MB * ENT;
    static constexpr int wc(int which, int g, int mp, int m) { return n1 + which * nc + ((g * MB + mp) * MB + m) * ENT; }   // 0: linear_right, 1: linear_left
    static constexpr int par = n1 + 2 * nc;
    static constexpr int total = par + C * kClParStride;
};

// 16-byte unit of lane (l16, q)'s vector inside a table entry of 64 units: lane order for the forward kernels (SWZ = false);
// the order that also serves the transposed reads of the backward kernels (SWZ = true, cemlp_cmb.hpp)
template <bool SWZ>
CSMPN_DEV constexpr int cm_unit(int i, int k) { return SWZ ? (i & 3) * 16 + (i >> 3) * 8 + k * 2 + ((i >> 2) & 1) : i + 16 * k; }

// parameters -> LDS tables of one block (once per workgroup of NT threads). A work item is one (output channel o, slot)
// pair: slot = (chunk, q, v) of the block's input for W1 / (m, q, v) for linear_right / left; the FOUR grades of its weight
// are 16 contiguous bytes in the reference layout [o][c][g] - one coalesced 16-byte load (consecutive items = consecutive
// input channels) and four 4-byte LDS stores, one per grade's table entry. (Round 3 staged one (entry, lane) vector per item from
// four scalar loads strided by the row length: 40 k cycles per block for 32 channels, a third of an md17-sized launch.)
// Then the per-channel parameter rows [b1, bL, la, 0 | sa[4] | sb[4] | sigmoid(an)[4] | w[P]] (as cemlp_cl.hpp).
template <class ALG, int C, class TB, int NT, bool SWZ>
__device__ void cm_stage_tables(const DevBlock& B, float* base, int tid) {
    constexpr int G = ALG::G, P = ALG::P, MB = TB::MB, NCH = TB::NCH;
    static_assert(G == 4, "Cl(3,0)-shaped algebra: the grades of one weight are one 16-byte vector");
    constexpr int N1 = C * 16 * NCH, NC = C * C, NITEMS = N1 + 2 * NC, NIT = (NITEMS + NT - 1) / NT;
    constexpr int GS1 = TB::w1(1, 0, 0) - TB::w1(0, 0, 0), GSC = TB::wc(0, 1, 0, 0) - TB::wc(0, 0, 0, 0);
    const float *pW1 = B.W1, *pWR = B.WR, *pWL = B.WL;
    const float* src[NIT];
    int dst[NIT], gs[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int e = tid + it * NT;
        src[it] = nullptr;
        dst[it] = -1;
        gs[it] = 0;
        if (e < N1) {
            const int o = e / (16 * NCH), cc = e % (16 * NCH), ch = cc >> 4, s16 = cc & 15;
            const bool at = TB::attr(ch);
            const int q = s16 & 3, v = s16 >> 2;          // full chunk: channel 16 ch + 4 v + q = 16 ch + s16; attributes: q + 4 v = s16
            const int c = at ? (s16 < TB::NA_ ? TB::NSEG * C + s16 : -1) : 16 * ch + s16;
            const int oo = o & 15, l16 = (oo >> 2) + 4 * (oo & 3);   // orow(l16) = oo
            if (c >= 0) src[it] = pW1 + (size_t)(o * TB::I + c) * G;
            dst[it] = TB::w1(0, o >> 4, ch) + 4 * cm_unit<SWZ>(l16, q) + v;
            gs[it] = GS1;
        } else if (e < NITEMS) {
            int f = e - N1;
            const int which = f / NC;
            f -= which * NC;
            const int o = f / C, c = f % C, m = c >> 4, s16 = c & 15, q = s16 & 3, v = s16 >> 2;
            const int oo = o & 15, l16 = (oo >> 2) + 4 * (oo & 3);
            src[it] = (which == 0 ? pWR : pWL) + (size_t)(o * C + c) * G;
            dst[it] = TB::wc(which, 0, o >> 4, m) + 4 * cm_unit<SWZ>(l16, q) + v;
            gs[it] = GSC;
        }
    }
    f4 val[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) val[it] = src[it] ? cl_ld4(src[it]) : f4{0.f, 0.f, 0.f, 0.f};
    constexpr int NPAR = C * kClParStride, NITP = (NPAR + NT - 1) / NT;
    static_assert(16 + P <= kClParStride, "parameter stride");
    const float *pb1 = B.b1, *pbL = B.bL, *pla = B.la, *psa = B.sa, *psb = B.sb, *pan = B.an, *pw = B.w;
    const bool has_b1 = B.has_b1 != 0;
    const float* ps[NITP];
    bool sig[NITP];
#pragma unroll
    for (int it = 0; it < NITP; ++it) {
        const int e = tid + it * NT;
        ps[it] = nullptr;
        sig[it] = false;
        if (e < NPAR) {
            const int ch = e / kClParStride, s = e % kClParStride;
            if (s == 0) { if (has_b1) ps[it] = pb1 + ch; }
            else if (s == 1) ps[it] = pbL + ch;
            else if (s == 2) ps[it] = pla + ch;
            else if (s >= 4 && s < 8) ps[it] = psa + ch * G + (s - 4);
            else if (s >= 8 && s < 12) ps[it] = psb + ch * G + (s - 8);
            else if (s >= 12 && s < 16) { ps[it] = pan + ch * G + (s - 12); sig[it] = true; }
            else if (s >= 16 && s < 16 + P) ps[it] = pw + ch * P + (s - 16);
        }
    }
    float pv[NITP];
#pragma unroll
    for (int it = 0; it < NITP; ++it) pv[it] = ps[it] ? *ps[it] : 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        if (dst[it] >= 0) {
#pragma unroll
            for (int g = 0; g < G; ++g) base[dst[it] + g * gs[it]] = val[it][g];
        }
    }
#pragma unroll
    for (int it = 0; it < NITP; ++it) {
        const int e = tid + it * NT;
        if (sig[it]) pv[it] = sigmoidf(pv[it]);
        if (e < NPAR) base[TB::par + e] = pv[it];
    }
}
// the forward's tables: lane order
template <class ALG, int C, class TB>
__device__ void cm_stage_block(const DevBlock& B, float* base, int tid) {
    cm_stage_tables<ALG, C, TB, 64 * kCmWaves, false>(B, base, tid);
}

// acc[m'][d] += (one 16-slot chunk, NSTEP steps) x (its table entries). ldsa = LDS + 4 lane + the float offset of entry
// (g = 0, m' = 0) of this chunk; GS / MS = float strides between grades / between m'.
template <class ALG, int MB, int NSTEP, int GS, int MS>
CSMPN_DEV void cm_mix_chunk(f4 (&acc)[MB][8], const f4 (&x)[8], const float* ldsa) {
    static_assert(ALG::D == 8 && ALG::G == 4, "Cl(3,0)-shaped algebra");
    static_for<0, MB>([&](auto mp) {
        f4 a[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) a[g] = cl_ld4(ldsa + (g * GS + mp * MS));
        static_for<0, 8>([&](auto d) {
            constexpr int g = ALG::grade(d);
            static_for<0, NSTEP>([&](auto v) { acc[mp][d] = mfma16(a[g][int(v)], x[d][int(v)], acc[mp][d]); });
        });
    });
}

// MVLinear output (bias not yet added) of one channel -> MVSiLU (cegnn_utils.py:76-83). pp = the channel's parameter row.
template <class ALG>
CSMPN_DEV void cm_silu(float (&y)[8], float (&z)[8], float (&gate)[4], const float* pp) {
    constexpr int G = ALG::G;
    const f4 p0 = cl_ld4(pp), sa = cl_ld4(pp + 4), sb = cl_ld4(pp + 8);
    y[0] += p0.x;
    static_for<0, G>([&](auto g) {
        constexpr int d0 = ALG::gstart(g), nd = ALG::gsize(g);
        float u;
        if constexpr (g == 0) {
            u = y[0];
        } else {
            u = 0.f;
            static_for<0, nd>([&](auto t) {
                constexpr int d = d0 + decltype(t)::value;
                u += qsf<ALG, d> * y[d] * y[d];
            });
        }
        gate[g] = sigmoidf(__builtin_fmaf(sa[int(g)], u, sb[int(g)]));
#pragma unroll
        for (int t = 0; t < nd; ++t) z[d0 + t] = gate[g] * y[d0 + t];
    });
}

// normalisation of the right operand, steerable geometric product + first-order term (cegnn_utils.py:42-51,126-152):
// L (linear_left output without bias) -> s = (L + bL + gp(z, r)) / sqrt 2; returns the smooth norm of s
template <class ALG>
CSMPN_DEV float cm_gp_tail(const float (&z)[8], const float (&R)[8], float (&L)[8], float (&invden)[4], const float* pp) {
    constexpr int D = ALG::D, G = ALG::G;
    L[0] += pp[1];
    const f4 sg = cl_ld4(pp + 12);
    float r[D];
    static_for<0, G>([&](auto g) {
        constexpr int d0 = ALG::gstart(g), nd = ALG::gsize(g);
        float qq = 0.f;
        static_for<0, nd>([&](auto t) {
            constexpr int d = d0 + decltype(t)::value;
            qq += qsf<ALG, d> * R[d] * R[d];
        });
        const float m = __builtin_fmaf(sg[int(g)], cl_smooth_abs_sqrt(qq) - 1.0f, 1.0f);
        invden[g] = fast_rcp(m + kEps);
#pragma unroll
        for (int t = 0; t < nd; ++t) r[d0 + t] = R[d0 + t] * invden[g];
    });
    cl_weighted_gp<ALG>(L, z, r, pp + 16);
    float qs = 0.f;
    static_for<0, D>([&](auto dd) {
        constexpr int d = decltype(dd)::value;
        L[d] *= kInvSqrt2;
        qs += qsf<ALG, d> * L[d] * L[d];
    });
    return cl_smooth_abs_sqrt(qs);
}

// sum over the four q (the lanes 16 apart): every lane receives the sum of its row
CSMPN_DEV float cm_q_sum(float v) { return mfma16(1.0f, v, f4{0.f, 0.f, 0.f, 0.f})[0]; }

// CSMPN_FLAG_SAVE_STATE at 32 channels (md17's width: kernels at 8 % of the HBM roofline, one wave per SIMD): the forward
// stores y (MVLinear output, no bias), R (linear_right output) and s (the block's output in front of its layer norm) of
// every block, the pair backward (cemlp_cmp.hpp) reads them instead of recomputing two channel mixes and the geometric
// product. State regions of the saved buffer (cemlp_device.hpp): a 16-row tile of one tensor and channel group is 8 blades
// x 64 lanes x 16 bytes in lane order (tile slot = 2 tile + group) - no transposition, 1 KB of contiguous memory per instruction.
// p: blade 0 of this lane.
CSMPN_DEV void cm_store_lane(float* p, const f4 (&t)[8]) {
#pragma unroll
    for (int d = 0; d < 8; ++d)   // streaming: read once, by the backward - no business in L2 next to the gathered h rows
        __builtin_nontemporal_store(t[d], reinterpret_cast<f4*>(p + 256 * d));
}
CSMPN_DEV void cm_load_lane(f4 (&t)[8], const float* p) {
#pragma unroll
    for (int d = 0; d < 8; ++d) t[d] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p + 256 * d));
}

// one block forward. x: the block's input chunks; out: its output. sv (32 channels, CSMPN_FLAG_SAVE_STATE; else null): blade 0
// of this lane, group 0, in the block's s region; step = floats from a block's s region to its y region (and from y to R). ldsa = table base + 4 lane, ldsp = table base + par +
// kClParStride * q (the parameter row of this lane's channel (0, 0); channel (m, v) is 16 m + 4 v rows further).
template <class ALG, int C, class TB>
CSMPN_DEV void cm_block_forward(const float* ldsa, const float* ldsp, const f4 (&x)[TB::NCH][8], f4 (&out)[TB::MB][8], ClStamp& stamp, int sid,
                                 float* sv = nullptr, size_t step = 0) {
    constexpr int D = ALG::D, MB = TB::MB;
    constexpr int GS1 = TB::w1(1, 0, 0) - TB::w1(0, 0, 0), MS1 = MB > 1 ? TB::w1(0, 1, 0) - TB::w1(0, 0, 0) : 0;
    constexpr int GSC = TB::wc(0, 1, 0, 0) - TB::wc(0, 0, 0, 0), MSC = MB > 1 ? TB::wc(0, 0, 1, 0) - TB::wc(0, 0, 0, 0) : 0;
    f4 y[MB][8];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int d = 0; d < D; ++d) y[m][d] = f4{0.f, 0.f, 0.f, 0.f};
    static_for<0, TB::NCH>([&](auto ch) { cm_mix_chunk<ALG, MB, TB::nstep(ch), GS1, MS1>(y, x[ch], ldsa + TB::w1(0, 0, ch)); });
    if constexpr (C == 32) {
        if (sv) {
#pragma unroll
            for (int m = 0; m < MB; ++m) cm_store_lane(sv + step + 2048 * m, y[m]);
        }
    }
    stamp(sid);
    f4 z[MB][8];
    static_for<0, MB>([&](auto m) {
        static_for<0, 4>([&](auto v) {
            float yy[D], zz[D], gate[4];
#pragma unroll
            for (int d = 0; d < D; ++d) yy[d] = y[m][d][int(v)];
            cm_silu<ALG>(yy, zz, gate, ldsp + (16 * m + 4 * v) * kClParStride);
#pragma unroll
            for (int d = 0; d < D; ++d) z[m][d][int(v)] = zz[d];
            CM_FENCE();
        });
    });
    stamp(sid + 1);
    f4 R[MB][8], L[MB][8];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int d = 0; d < D; ++d) R[m][d] = L[m][d] = f4{0.f, 0.f, 0.f, 0.f};
    static_for<0, MB>([&](auto m) {
        cm_mix_chunk<ALG, MB, 4, GSC, MSC>(R, z[m], ldsa + TB::wc(0, 0, 0, m));
        cm_mix_chunk<ALG, MB, 4, GSC, MSC>(L, z[m], ldsa + TB::wc(1, 0, 0, m));
    });
    if constexpr (C == 32) {
        if (sv) {
#pragma unroll
            for (int m = 0; m < MB; ++m) cm_store_lane(sv + 2 * step + 2048 * m, R[m]);
        }
    }
    stamp(sid + 2);
    float nlsum = 0.f;
    static_for<0, MB>([&](auto m) {
        static_for<0, 4>([&](auto v) {
            float zz[D], RR[D], LL[D], invden[4];
#pragma unroll
            for (int d = 0; d < D; ++d) { zz[d] = z[m][d][int(v)]; RR[d] = R[m][d][int(v)]; LL[d] = L[m][d][int(v)]; }
            nlsum += cm_gp_tail<ALG>(zz, RR, LL, invden, ldsp + (16 * m + 4 * v) * kClParStride);
#pragma unroll
            for (int d = 0; d < D; ++d) L[m][d][int(v)] = LL[d];
            CM_FENCE();
        });
    });
    if constexpr (C == 32) {
        if (sv) {
#pragma unroll
            for (int m = 0; m < MB; ++m) cm_store_lane(sv + 2048 * m, L[m]);
        }
    }
    stamp(sid + 3);
    // MVLayerNorm (cegnn_utils.py:93-96): mean over the C channels of the row
    const float invMn = fast_rcp(__builtin_fmaf(cm_q_sum(nlsum), 1.0f / float(C), kEps));
    static_for<0, MB>([&](auto m) {
        static_for<0, 4>([&](auto v) {
            const float k = ldsp[(16 * m + 4 * v) * kClParStride + 2] * invMn;
#pragma unroll
            for (int d = 0; d < D; ++d) out[m][d][int(v)] = k * L[m][d][int(v)];
        });
    });
    stamp(sid + 4);
}

// ---------------------------------------------------------------------------------
// tile bookkeeping: lane (r, q) works on row tile * 16 + r
template <int MODE>
struct CmTile {
    long row, lrow;
    bool valid;
    int i_dst, i_src, i_perm;
    float scale;
    template <int NA>
    CSMPN_DEV void load(const RowIO& io, long tile, int r) {
        row = tile * kCmRows + r;
        valid = row < io.rows;
        lrow = valid ? row : 0;   // lanes past the end compute on row 0 and contribute nothing
        i_dst = i_src = i_perm = 0;
        scale = 1.0f;
        if constexpr (MODE == MODE_EDGE) {
            i_dst = io.seg[0].ia[lrow];
            i_src = io.seg[0].ib[lrow];
            if constexpr (NA > 0) i_perm = io.seg[1].ia[lrow];
        }
        if constexpr (MODE == MODE_NODE) {   // mean aggregation
            if (io.seg[1].deg) { const int dg = io.seg[1].deg[lrow]; scale = 1.0f / float(dg > 1 ? dg : 1); }
        }
    }
};

// the lane's 4 channels (q, q + 4, q + 8, q + 12 of a group of 16) x 8 blades of one row: 8 x 16 bytes, p = the first
// float of channel q
struct CmPiece {
    f4 v[4][2];
    CSMPN_DEV void load(const float* p) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { v[c][0] = cl_ld4(p + 32 * c); v[c][1] = cl_ld4(p + 32 * c + 4); }
    }
};
// t[d][v] <-> memory [v][d]
CSMPN_DEV void cm_unpack(f4 (&t)[8], const CmPiece& a) {
#pragma unroll
    for (int d = 0; d < 8; ++d)
#pragma unroll
        for (int c = 0; c < 4; ++c) t[d][c] = a.v[c][d >> 2][d & 3];
}
CSMPN_DEV void cm_store_piece(float* p, const f4 (&t)[8]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        cl_st4(p + 32 * c, f4{t[0][c], t[1][c], t[2][c], t[3][c]});
        cl_st4(p + 32 * c + 4, f4{t[4][c], t[5][c], t[6][c], t[7][c]});
    }
}

// gathered input of block 0: requested in one go (issue), turned into chunks later (finish)
template <class ALG, int C, int MODE, int NA>
struct CmRaw {
    static constexpr int MB = C / 16, NSTEP = (NA + 3) / 4, D = ALG::D, ROW = C * D;
    CmPiece a[MB], b[MB];
    f4 t[NSTEP > 0 ? NSTEP : 1][2];
    CSMPN_DEV void issue(const RowIO& io, const CmTile<MODE>& T, int q) {
        const float *pa, *pb, *pt = nullptr;
        if constexpr (MODE == MODE_EDGE) {
            pa = io.seg[0].a + (size_t)T.i_dst * ROW + q * D;
            pb = io.seg[0].b + (size_t)T.i_src * ROW + q * D;
            if constexpr (NA > 0) pt = io.seg[1].a + (size_t)T.i_perm * (NA * D);
        } else {
            pa = io.seg[0].a + (size_t)T.lrow * ROW + q * D;
            pb = io.seg[1].a + (size_t)T.lrow * ROW + q * D;
            if constexpr (NA > 0) pt = io.seg[2].a + (size_t)T.lrow * (NA * D);
        }
#pragma unroll
        for (int m = 0; m < MB; ++m) { a[m].load(pa + 16 * m * D); b[m].load(pb + 16 * m * D); }
        if constexpr (NA > 0) {
#pragma unroll
            for (int v = 0; v < NSTEP; ++v) {
                const int ca = q + 4 * v;   // slot (q, v) = attribute channel q + 4 v; an empty slot reads a valid one (its weights are 0)
                const float* p = pt + (ca < NA ? ca : NA - 1) * D;
                t[v][0] = cl_ld4(p); t[v][1] = cl_ld4(p + 4);
            }
        }
    }
    template <class TB>
    CSMPN_DEV void finish(f4 (&x)[TB::NCH][8], const CmTile<MODE>& T) const {
        if constexpr (MODE == MODE_EDGE) {
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                CmPiece df;
#pragma unroll
                for (int c = 0; c < 4; ++c) { df.v[c][0] = a[m].v[c][0] - b[m].v[c][0]; df.v[c][1] = a[m].v[c][1] - b[m].v[c][1]; }
                cm_unpack(x[m], df);
            }
        } else {
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                cm_unpack(x[m], a[m]);
                CmPiece sc;
#pragma unroll
                for (int c = 0; c < 4; ++c) { sc.v[c][0] = b[m].v[c][0] * T.scale; sc.v[c][1] = b[m].v[c][1] * T.scale; }
                cm_unpack(x[MB + m], sc);
            }
        }
        if constexpr (NA > 0) {
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                x[TB::NCH - 1][d] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int v = 0; v < NSTEP; ++v) x[TB::NCH - 1][d][v] = t[v][d >> 2][d & 3];
            }
        }
    }
};

// Rows ROW0 .. ROW0 + NROWS - 1 of the wave's tile, staged as [NROWS][ROWLEN + 4] -> atomic adds into table rows of ROWLEN
// floats; lane = column. Adds the rows to table[t_add[row]] (rows sorted by that index: equal consecutive targets are
// summed first) and, when SUB, subtracts them from table[t_sub[row]] (unsorted). Negative targets are skipped. Lane r
// (< 16) holds the targets of row r.
template <int ROWLEN, bool SUB, int NROWS = kCmRows, int ROW0 = 0>
CSMPN_DEV void cm_scatter(const float* sc, int t_add, int t_sub, float* table, int lane) {
    constexpr int SS = ROWLEN + 4, NC = ROWLEN / 64;
    static_assert(ROWLEN % 64 == 0, "whole columns");
    static_for<0, NC>([&](auto cc) {
        const int colx = 64 * cc + lane;
        const float* col = sc + colx;
        auto flush = [&](int target, float a) {
#ifndef CM_X_NOATOM   // timing experiment only (results wrong): no atomics
            if (target >= 0) atomicAdd(table + (size_t)target * ROWLEN + colx, a);
#else
            if (target == -12345) atomicAdd(table + (size_t)target * ROWLEN + colx, a);
#endif
        };
        float val[NROWS];
#pragma unroll
        for (int i = 0; i < NROWS; ++i) val[i] = col[i * SS];
        float acc = 0.f;
        int cur = __builtin_amdgcn_readlane(t_add, ROW0);
        static_for<0, NROWS>([&](auto rr) {
            const int t = __builtin_amdgcn_readlane(t_add, ROW0 + rr);
            if (t != cur) {
                flush(cur, acc);
                cur = t;
                acc = 0.f;
            }
            acc += val[rr];
        });
        flush(cur, acc);
        if constexpr (SUB) {
            static_for<0, NROWS>([&](auto rr) { flush(__builtin_amdgcn_readlane(t_sub, ROW0 + rr), -val[rr]); });
        }
    });
}

// ---------------------------------------------------------------------------------
// forward kernel: NBLK blocks (1 or 2), all C channels wide. Tile t (16 rows) belongs to wave t mod (4 gridDim).
template <class ALG, int C, int MODE, int NBLK, int NA>
__global__ void __launch_bounds__(64 * kCmWaves, C == 16 ? CM_FWD_OCC : 1) cemlp_cm_fwd_kernel(const DevCemlp C_arg, const RowIO io_arg) {
    typedef const char __attribute__((address_space(4))) * KArgPtr;
    const KArgPtr ka = (KArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr size_t kIoOffset = (sizeof(DevCemlp) + alignof(RowIO) - 1) / alignof(RowIO) * alignof(RowIO);
    const DevCemlp& Cd = *(const DevCemlp*)(const char*)ka;
    const RowIO& io = *(const RowIO*)(const char*)(ka + kIoOffset);
    (void)C_arg; (void)io_arg;
    using T0 = CmTab<C, MODE, NA, 0>;
    using T1 = CmTab<C, MODE, NA, 1>;
    constexpr int D = ALG::D, ROW = C * D, MB = C / 16, SS = ROW + 4;
    static_assert(NBLK == 1 || NBLK == 2, "one or two blocks");
    constexpr int tab_floats = T0::total + (NBLK > 1 ? T1::total : 0);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* lds = smem;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 15, q = lane >> 4;
    float* sc = lds + tab_floats + wave * (kCmFwdStageRows * SS);   // scatter staging tile (edge program)
    const float* ldsa0 = lds + 4 * lane;
    const float* ldsp0 = lds + T0::par + kClParStride * q;
    const float* ldsa1 = ldsa0 + T0::total;
    const float* ldsp1 = lds + T0::total + T1::par + kClParStride * q;
    ClStamp stamp(0);

    const long ntiles = (io.rows + kCmRows - 1) / kCmRows;
    const long tstride = (long)gridDim.x * kCmWaves;
    const long tile0 = (long)blockIdx.x * kCmWaves + wave;
    // software pipeline over the wave's tiles (as cemlp_cl.hpp): the rows of tile t + 1 are requested BEFORE the stores and
    // atomics of tile t, the indices of tile t + 2 travel while tile t + 1 computes
    CmTile<MODE> T, Tn;
    T.template load<NA>(io, tile0, r);
    CmRaw<ALG, C, MODE, NA> raw;
    raw.issue(io, T, q);
    Tn.template load<NA>(io, tile0 + tstride, r);
    cm_stage_block<ALG, C, T0>(Cd.b[0], lds, threadIdx.x);
    if constexpr (NBLK > 1) cm_stage_block<ALG, C, T1>(Cd.b[1], lds + T0::total, threadIdx.x);
    __syncthreads();
    stamp(0);
    for (long tile = tile0; tile < ntiles; tile += tstride) {
        // the tables are loop invariant: without this the node program (no LDS write inside the loop) has every table
        // entry hoisted out of the loop and spilled (1.4 KB of scratch per lane)
        asm volatile("" ::: "memory");
        f4 out[MB][8], in1[MB][8];
        // CSMPN_FLAG_SAVE_STATE (32 channels, two blocks): this lane's blade 0 of group 0 in block 0's s region (cm_store_lane)
        const size_t s_step = (size_t)2 * state_rows(io.rows) * ROW;
        float* const sv0 = (C == 32 && NBLK > 1 && io.save_state != 0 && io.save != nullptr && T.valid)
                               ? io.save + state_region<ROW, ROW>(io.rows, 0, 0) + (size_t)tile * (kCmRows * ROW) + 4 * lane : nullptr;
        {
            f4 x[T0::NCH][8];
            raw.template finish<T0>(x, T);
#ifdef CSMPN_STAMPS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            stamp(1);
            cm_block_forward<ALG, C, T0>(ldsa0, ldsp0, x, out, stamp, 2, sv0, s_step);
        }
        if constexpr (NBLK > 1) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int d = 0; d < D; ++d) in1[m][d] = out[m][d];
            // the block-1 input rows leave right away (they are far in front of the next tile's loads by then)
#ifdef CM_X_NOSAVE   // timing experiment only (results wrong): block-1 inputs not saved
            if (io.save && T.row == -12345) {
#else
            if (io.save && T.valid) {
#endif
#pragma unroll
                for (int m = 0; m < MB; ++m) cm_store_piece(io.save + (size_t)T.row * ROW + (16 * m + q) * D, in1[m]);
            }
            cm_block_forward<ALG, C, T1>(ldsa1, ldsp1, in1, out, stamp, 8, sv0 ? sv0 + state_rows(io.rows) * ROW : nullptr, s_step);
        }
        // next tile's rows, then this tile's stores (edge program: loads queued behind atomics wait for them; the node
        // program has no atomics and asks for its rows after the stores - fewer live registers)
        const CmTile<MODE> Tc = T;
        T = Tn;
        if constexpr (MODE == MODE_EDGE) raw.issue(io, T, q);
        Tn.template load<NA>(io, tile + 2 * tstride, r);
        if constexpr (MODE == MODE_EDGE) {
            if (io.row_store) {
                if (Tc.valid) {
#pragma unroll
                    for (int m = 0; m < MB; ++m) cm_store_piece(io.agg + (size_t)Tc.lrow * ROW + (16 * m + q) * D, out[m]);
                }
            } else {
                // two halves of 8 rows through one 8-row staging tile (LDS for three workgroups per CU)
                const int tgt = Tc.valid ? Tc.i_dst : -1;
                static_for<0, 2>([&](auto hf) {
                    CM_LDS_ORDER();
                    if ((r >> 3) == hf) {
#pragma unroll
                        for (int m = 0; m < MB; ++m) cm_store_piece(sc + (r & 7) * SS + (16 * m + q) * D, out[m]);
                    }
                    CM_LDS_ORDER();
                    cm_scatter<ROW, false, 8, 8 * hf>(sc, tgt, -1, io.agg, lane);
                });
            }
        } else if (Tc.valid) {
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                if (io.resid) {
                    CmPiece res;
                    res.load(io.resid + (size_t)Tc.row * ROW + (16 * m + q) * D);
                    f4 rr[8];
                    cm_unpack(rr, res);
#pragma unroll
                    for (int d = 0; d < D; ++d) out[m][d] += rr[d];
                }
                cm_store_piece(io.y + (size_t)Tc.row * ROW + (16 * m + q) * D, out[m]);
            }
        }
        if constexpr (MODE != MODE_EDGE) raw.issue(io, T, q);
        stamp(14);
    }
    stamp.flush(io.stamps, lane);
}
template <class ALG, int C, int MODE, int NBLK, int NA>
constexpr size_t cm_fwd_lds_bytes() {
    return sizeof(float) * (CmTab<C, MODE, NA, 0>::total + (NBLK > 1 ? CmTab<C, MODE, NA, 1>::total : 0) +
                            (MODE == MODE_EDGE ? kCmWaves * kCmFwdStageRows * (C * ALG::D + 4) : 0));
}

// =================================================================================
// pieces of the backward shared by cemlp_cmb.hpp (16 channels, two waves per SIMD) and cemlp_cmp.hpp (32 channels, wave pairs)

// acc[d] += (table entry of ONE (m', chunk) pair) x (the chunk's NSTEP steps). ldsa = LDS + 4 lane + offset of the pair's
// grade-0 entry, GS = float stride between grades
template <class ALG, int NSTEP, int GS>
CSMPN_DEV void cm_mix_one(f4 (&acc)[8], const f4 (&x)[8], const float* ldsa) {
    f4 a[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) a[g] = cl_ld4(ldsa + g * GS);
    static_for<0, 8>([&](auto d) {
        constexpr int g = ALG::grade(d);
        static_for<0, NSTEP>([&](auto v) { acc[d] = mfma16(a[g][int(v)], x[d][int(v)], acc[d]); });
    });
}

// the gates of MVSiLU from the (biased) MVLinear output of one channel
template <class ALG>
CSMPN_DEV void cm_gate(const float (&y)[8], float (&gate)[4], const float* pp) {
    const f4 sa = cl_ld4(pp + 4), sb = cl_ld4(pp + 8);
    static_for<0, ALG::G>([&](auto g) {
        constexpr int d0 = ALG::gstart(g), nd = ALG::gsize(g);
        float u;
        if constexpr (g == 0) {
            u = y[0];
        } else {
            u = 0.f;
            static_for<0, nd>([&](auto t) {
                constexpr int d = d0 + decltype(t)::value;
                u += qsf<ALG, d> * y[d] * y[d];
            });
        }
        gate[g] = sigmoidf(__builtin_fmaf(sa[int(g)], u, sb[int(g)]));
    });
}

// one channel: MVSiLU backward. gz = d/dz; y = biased MVLinear output. gy = d/dy; gs = {sa0, sb0, sa1, sb1, ..., b1}
template <class ALG>
CSMPN_DEV void cm_silu_bwd(const float (&gz)[8], const float (&y)[8], float (&gy)[8], float (&gs)[9], const float* pp) {
    constexpr int G = ALG::G;
    float gate[4];
    cm_gate<ALG>(y, gate, pp);
    const f4 sa = cl_ld4(pp + 4);
    static_for<0, G>([&](auto g) {
        constexpr int d0 = ALG::gstart(g), nd = ALG::gsize(g);
        float ggate = 0.f;
#pragma unroll
        for (int t = 0; t < nd; ++t) ggate = __builtin_fmaf(gz[d0 + t], y[d0 + t], ggate);
        const float gpre = ggate * gate[g] * (1.0f - gate[g]);
        float u;
        if constexpr (g == 0) {
            u = y[0];
        } else {
            u = 0.f;
            static_for<0, nd>([&](auto t) {
                constexpr int d = d0 + decltype(t)::value;
                u += qsf<ALG, d> * y[d] * y[d];
            });
        }
        gs[2 * g] = gpre * u;
        gs[2 * g + 1] = gpre;
        const float gu = gpre * sa[int(g)];
        static_for<0, nd>([&](auto t) {
            constexpr int d = d0 + decltype(t)::value;
            float v = gz[d] * gate[g];
            if constexpr (g == 0) v += gu;
            else v = __builtin_fmaf(gu * (2.0f * qsf<ALG, d>), y[d], v);
            gy[d] = v;
        });
    });
    gs[8] = gy[0];
}

}  // namespace csmpn
