// Channel-MFMA backward for 32-channel Cl(3,0) CEMLPs (md17's width: csmpn/models/md17_cssmpnn.py:11-14,60), "pair" form
// (round 4): TWO waves per 16-row tile, wave p of a pair owns the channel group 16 p .. 16 p + 15 of every tensor.
//
// Same arithmetic (csmpn/models/cegnn_utils.py:34-155,287-338) and the same lane layout as
// cemlp_cm.hpp / cemlp_cmb.hpp: lane = (row = lane & 15, q = lane >> 4) holds the channels 16 p + 4 v + q (v < 4) of its
// row, all 8 blades: a tensor is  f4 t[8]  per wave. Why a pair: at 32 channels one wave would hold 64 registers per
// tensor and 112-144 registers of weight-gradient tiles; split by OUTPUT channel group each wave keeps 32 + 56-72 and
// the whole backward state (y, z, R, s / ggp, gz: 160 registers) stays in registers at one wave per SIMD - no parking,
// no second MVLinear as in cemlp_cmb.hpp. What crosses the pair goes through LDS:
//
//  * a dense mixing contracts over ALL 32 input channels: wave p takes its own group's operand from its registers and
//    the partner's from a slot the partner has just written ([blade][row][4 q + v]: 16-byte writes by the owner, 16-byte
//    reads of the same positions by the partner);
//  * the same slots are the transposition buffers of the weight-gradient MFMAs (contraction over rows: lane (i, k) reads
//    row 4 s + k, column i): d/dW[group p][group m] = (gradient of group p)^T (operand of group m), both from slots;
//  * the layer norm's mean over the 32 channels: one float per (wave, row).
//  Five slots per pair: X/Z halves (block input, later z, later the input again), one attribute / spare slot, G halves
//  (d/d(gp), then d/dR, then d/dy). A workgroup is two pairs (4 waves, one per SIMD); its barrier is the pair's
//  rendezvous (both pairs run the same program). ONE set of weight tables per block in cb_unit order (cemlp_cmb.hpp):
//  57-78 KB for 32 channels - the reason for one workgroup per CU.
//  Per-channel parameter gradients: the transposing butterfly of cemlp_cmb.hpp; weight-gradient tiles persistent in
//  registers; per-workgroup slices + cl_reduce_kernel; all blocks in one launch, last block first, hand-over rows through L2.
#pragma once
#include "cemlp_cmb.hpp"

namespace csmpn {

constexpr int kCpWaves = 4;        // waves per workgroup: two pairs, one wave per SIMD
constexpr int kCpSlots = 5;        // LDS tensor slots per pair (kCbSlot floats each)
#ifndef CP_OCC
#define CP_OCC 1
#endif

// __builtin_amdgcn_sched_barrier does not order pure arithmetic: without these pins the selection DAG interleaves the
// four per-channel sections of a phase (their inputs are all in registers from the start) and quadruples the live
// temporaries - 700 B of scratch per lane. An empty asm that "modifies" a section's inputs in front of it and its outputs
// behind it ties the section to its place (as pl_pin in cemlp_pl.hpp). No instruction.
CSMPN_DEV void cp_pin8(float (&x)[8]) {
    asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
}

CSMPN_DEV void cp_put(float* slot, int lane, const f4 (&t)[8]) {
#pragma unroll
    for (int d = 0; d < 8; ++d) cl_st4(slot + d * 256 + (lane & 15) * 16 + 4 * (lane >> 4), t[d]);
}
CSMPN_DEV void cp_get(const float* slot, int lane, f4 (&t)[8]) {
#pragma unroll
    for (int d = 0; d < 8; ++d) t[d] = cl_ld4(slot + d * 256 + (lane & 15) * 16 + 4 * (lane >> 4));
}
// acc[grade] += sum over the 16 rows and the blades of the grade of a^T b (operands in slots, see cb_wgrad)
template <class ALG>
CSMPN_DEV void cp_wgrad(f4 (&acc)[4], const float* slotA, const float* slotB, int lane) {
    static_for<0, 8>([&](auto d) {
        constexpr int g = ALG::grade(d);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[g] = mfma16(slotA[d * 256 + 64 * s + lane], slotB[d * 256 + 64 * s + lane], acc[g]);
    });
}

// two weight-gradient tiles that share the gradient operand: (a^T b0, a^T b1) - the A operand is read once and the two
// accumulation chains are independent (a dependent MFMA waits for its predecessor's result)
template <class ALG>
CSMPN_DEV void cp_wgrad2(f4 (&acc0)[4], f4 (&acc1)[4], const float* slotA, const float* slotB0, const float* slotB1, int lane) {
    static_for<0, 8>([&](auto d) {
        constexpr int g = ALG::grade(d);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float a = slotA[d * 256 + 64 * s + lane];
            acc0[g] = mfma16(a, slotB0[d * 256 + 64 * s + lane], acc0[g]);
            acc1[g] = mfma16(a, slotB1[d * 256 + 64 * s + lane], acc1[g]);
        }
    });
}

// cm_scatter for half rows: the staged tile holds HALF floats per row, the table rows are STRIDE floats long (the
// caller passes the table offset by its half)
template <int HALF, int STRIDE, bool SUB>
CSMPN_DEV void cp_scatter(const float* sc, int t_add, int t_sub, float* table, int lane) {
    constexpr int SS = HALF + 4, NC = HALF / 64;
    static_assert(HALF % 64 == 0, "whole columns");
    static_for<0, NC>([&](auto cc) {
        const int colx = 64 * cc + lane;
        const float* col = sc + colx;
        auto flush = [&](int target, float a) {
            if (target >= 0) atomicAdd(table + (size_t)target * STRIDE + colx, a);
        };
        float val[kCmRows];
#pragma unroll
        for (int i = 0; i < kCmRows; ++i) val[i] = col[i * SS];
        float acc = 0.f;
        int cur = __builtin_amdgcn_readlane(t_add, 0);
        static_for<0, kCmRows>([&](auto rr) {
            const int t = __builtin_amdgcn_readlane(t_add, rr);
            if (t != cur) {
                flush(cur, acc);
                cur = t;
                acc = 0.f;
            }
            acc += val[rr];
        });
        flush(cur, acc);
        if constexpr (SUB) {
            static_for<0, kCmRows>([&](auto rr) { flush(__builtin_amdgcn_readlane(t_sub, rr), -val[rr]); });
        }
    });
}

// this wave's half (channel group p) of the block's full-width input rows + (wave 0) the attribute channels
template <class ALG, int C, int MODE, int NA, int K>
struct CpIn {
    static constexpr int D = ALG::D, ROW = C * D, NSTEP = (NA + 3) / 4;
    CmPiece a, b;                         // edge: h[dst], h[src]; node: h, agg; later blocks: a = the saved block input
    f4 t[NSTEP > 0 ? NSTEP : 1][2];       // attribute slots (q, v) = channel q + 4 v
    CSMPN_DEV void issue(const RowIO& io, const CmTile<MODE>& T, int p, int q) {
        const int off = (16 * p + q) * D;
        if constexpr (K > 0) {
            a.load(io.saved + (size_t)T.lrow * ROW + off);
        } else {
            const float* pt = nullptr;
            if constexpr (MODE == MODE_EDGE) {
                a.load(io.seg[0].a + (size_t)T.i_dst * ROW + off);
                b.load(io.seg[0].b + (size_t)T.i_src * ROW + off);
                if constexpr (NA > 0) pt = io.seg[1].a + (size_t)T.i_perm * (NA * D);
            } else {
                a.load(io.seg[0].a + (size_t)T.lrow * ROW + off);
                b.load(io.seg[1].a + (size_t)T.lrow * ROW + off);
                if constexpr (NA > 0) pt = io.seg[2].a + (size_t)T.lrow * (NA * D);
            }
            if constexpr (NA > 0) {
                if (p == 0) {
#pragma unroll
                    for (int v = 0; v < NSTEP; ++v) {
                        const int ca = q + 4 * v;
                        const float* pp = pt + (ca < NA ? ca : NA - 1) * D;
                        t[v][0] = cl_ld4(pp); t[v][1] = cl_ld4(pp + 4);
                    }
                }
            }
        }
    }
    // x0: the first full-width segment's half (edge: h[dst] - h[src]); x1: node's aggregate half (scaled)
    CSMPN_DEV void finish(f4 (&x0)[8], f4 (&x1)[8], f4 (&xa)[8], const CmTile<MODE>& T) const {
        if constexpr (K > 0) {
            cm_unpack(x0, a);
        } else if constexpr (MODE == MODE_EDGE) {
            CmPiece df;
#pragma unroll
            for (int c = 0; c < 4; ++c) { df.v[c][0] = a.v[c][0] - b.v[c][0]; df.v[c][1] = a.v[c][1] - b.v[c][1]; }
            cm_unpack(x0, df);
        } else {
            cm_unpack(x0, a);
            CmPiece sc;
#pragma unroll
            for (int c = 0; c < 4; ++c) { sc.v[c][0] = b.v[c][0] * T.scale; sc.v[c][1] = b.v[c][1] * T.scale; }
            cm_unpack(x1, sc);
        }
        if constexpr (K == 0 && NA > 0) {
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                xa[d] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int v = 0; v < NSTEP; ++v) xa[d][v] = t[v][d >> 2][d & 3];
            }
        }
    }
};

// backward of block K over this pair's tiles. tab: the block's tables; slots: this pair's five slots; red: this pair's
// 4 x 2 x 16 floats of row sums; work: all slots of the workgroup (the end-of-block image lies over them).
// SAVES (CSMPN_FLAG_SAVE_STATE): y, R and s of the block come from the forward (cm_store_lane, cemlp_cm.hpp) - the tile
// starts at the gates; both W1 mixes, the linear_right / left mixes and the product's forward are gone, the input rows are
// only parked in the slots for d/dW1.
template <class ALG, int C, int MODE, int NBLK, int NA, int K, bool SAVES>
__device__ void cp_block(const RowIO& io, float* tab, float* slots, float* red, float* work, ClStamp& stamp) {
    static_assert(C == 32, "two channel groups");
    using TF = CmTab<C, MODE, NA, K>;
    using PT = ClPart<ALG, C, TF::I>;
    using SM = CbSmall<ALG>;
    constexpr int D = ALG::D, G = ALG::G, ROW = C * D, HALF = ROW / 2, SS = HALF + 4, NCH = TF::NCH, NSEG = TF::NSEG;
    constexpr bool kLast = K == NBLK - 1;
    constexpr bool kAttr = K == 0 && NA > 0;
    constexpr int NFULL = K > 0 ? 1 : NSEG;            // full-width input segments (each two chunks)
    constexpr int GS1 = TF::w1(1, 0, 0) - TF::w1(0, 0, 0), GSC = TF::wc(0, 1, 0, 0) - TF::wc(0, 0, 0, 0);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int p = wave & 1, pair = wave >> 1;
    const int r = lane & 15, q = lane >> 4;
    // Slot pointers are plain offsets from `slots` (NOT an array of pointers indexed by p: the compiler loses the LDS
    // address space through it and emits flat_load / flat_store - 320 + 96 per kernel, each waiting on vmcnt AND lgkmcnt).
    float* const XZ0 = slots;                          // halves of the block input / z
    float* const XZ1 = slots + kCbSlot;
    float* const XZp = slots + p * kCbSlot;            // this wave's half
    float* const XA = slots + 2 * kCbSlot;             // attribute chunk
    float* const GGp = slots + (3 + p) * kCbSlot;      // halves of the gradient tensor on its way through the mixes: own
    const float* ldsa = tab + 4 * cb_unit(r, q);
    const float* ldst = tab + cb_tofs(lane);
    const float* ldsp = tab + TF::par + kClParStride * (16 * p + q);
    auto PP = [&](int v) { return ldsp + 4 * v * kClParStride; };
#ifdef CSMPN_STAMPS   // diagnostic build: the time spent at the pair's rendezvous goes to slot 11, not to the phase
    auto pair_sync = [&]() {
        CM_FENCE();
        const unsigned long long ta = __builtin_amdgcn_s_memtime();
        __syncthreads();
        const unsigned long long tb = __builtin_amdgcn_s_memtime();
        stamp.acc[11] += tb - ta;
        stamp.t0 += tb - ta;
        CM_FENCE();
    };
#else
    auto pair_sync = [&]() { CM_FENCE(); __syncthreads(); CM_FENCE(); };
#endif

    // persistent sums: weight-gradient tiles of this wave's OUTPUT group against every input chunk / group
    f4 aW1[NCH][4], aWR[2][4], aWL[2][4];
    float sm[kCbGroups];
    {
        const f4 z4 = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) aW1[c][g] = z4;
#pragma unroll
            for (int m = 0; m < 2; ++m) aWR[m][g] = aWL[m][g] = z4;
        }
#pragma unroll
        for (int g = 0; g < kCbGroups; ++g) sm[g] = 0.f;
    }
    const long ntiles = (io.rows + kCmRows - 1) / kCmRows;
    const long tstride = (long)gridDim.x * (kCpWaves / 2);
    long tile = (long)blockIdx.x * (kCpWaves / 2) + pair;
    // Both pairs of a workgroup run the same number of iterations (the barrier is the workgroup's): a pair without a
    // tile computes on row 0 with d/d(out) = 0 and stores nothing.
    const long iters = (ntiles + tstride - 1) / tstride;
    CmTile<MODE> T;
    T.template load<NA>(io, tile, r);
    CpIn<ALG, C, MODE, NA, K> in;
    // SAVES: y one step behind s, R two steps
    const size_t s_step = (size_t)2 * state_rows(io.rows) * ROW;
    auto state_of = [&](const CmTile<MODE>& t) {   // blade 0 of this lane, group p, in block K's s region (cm_store_lane)
        return io.saved + state_region<ROW, ROW>(io.rows, 0, K) + ((size_t)(t.lrow >> 4) * 2 + p) * 2048 + 4 * (int)(t.lrow & 15) + 64 * q;
    };
    f4 ynext[8];   // SAVES: the next tile's y, requested in front of the current tile's stores
    if constexpr (SAVES) cm_load_lane(ynext, state_of(T) + s_step);
    else in.issue(io, T, p, q);
    for (long it = 0; it < iters; ++it, tile += tstride) {
        asm volatile("" ::: "memory");
        // ---- the block's input: own half -> XZp (+ attributes -> XA), y = W1 x over all chunks
        float* const XZo = slots + (1 - p) * kCbSlot;   // the partner's half
        float* const GGo = slots + (4 - p) * kCbSlot;
        // table entries: W1 (g, m', chunk) -> w1e(m', chunk); linear_right / left (which, g, m', m) -> wce(which, m', m)
        auto w1e = [&](int mp, int chunk) { return TF::w1(0, 0, 0) + (mp * NCH + chunk) * TF::ENT; };
        auto wce = [&](int which, int mp, int m) { return TF::wc(which, 0, 0, 0) + (mp * TF::MB + m) * TF::ENT; };
        f4 y[8];
        auto mix_w1 = [&](int seg, const f4 (&own)[8]) __attribute__((always_inline)) {   // the two chunks of full-width segment `seg`: own group first
            f4 oth[8];
            cp_get(XZo, lane, oth);
            cm_mix_one<ALG, 4, GS1>(y, own, ldsa + w1e(p, 2 * seg + p));
            cm_mix_one<ALG, 4, GS1>(y, oth, ldsa + w1e(p, 2 * seg + 1 - p));
        };
        // y = W1 x from the rows `in` holds; leaves the LAST full-width segment's halves in XZ and returns the first
        // segment's own half (node program: needed once more for d/dW1)
        auto mvlinear = [&](f4 (&keep)[8], bool put_attr) __attribute__((always_inline)) {
            f4 x1[8], xa[8];
            in.finish(keep, x1, xa, T);
            cp_put(XZp, lane, keep);
            if constexpr (kAttr) { if (put_attr && p == 0) cp_put(XA, lane, xa); }
            pair_sync();
#pragma unroll
            for (int d = 0; d < D; ++d) y[d] = f4{0.f, 0.f, 0.f, 0.f};
            mix_w1(0, keep);
            if constexpr (kAttr) {
                f4 ta[8];
                cp_get(XA, lane, ta);
                cm_mix_one<ALG, (NA + 3) / 4, GS1>(y, ta, ldsa + w1e(p, NCH - 1));
            }
            if constexpr (NFULL > 1) {   // node program: the aggregate's two chunks through the same slots
                pair_sync();
                cp_put(XZp, lane, x1);
                pair_sync();
                mix_w1(1, x1);
            }
        };
        const float* const sp = SAVES ? state_of(T) : nullptr;
        if constexpr (SAVES) {
#pragma unroll
            for (int d = 0; d < D; ++d) y[d] = ynext[d];
        } else {
            f4 x0[8];
            mvlinear(x0, true);
        }
        stamp(1); CB_MARK(1);
        CM_FENCE();
        // ---- forward again: z = gate(y) y; R = WR z, s = (WL z + bL + gp(z, n(R))) / sqrt 2
        f4 z[8], R[8], s[8];
        if constexpr (SAVES) {       // requested in front of the gates: they travel under them (requesting them a tile ahead,
            cm_load_lane(s, sp);     // with y, costs 23 AGPRs and 4 % at M32)
            cm_load_lane(R, sp + 2 * s_step);
            asm volatile("" ::: "memory");
        }
        static_for<0, 4>([&](auto v) {
            float yy[D], zz[D], gate[4];
#pragma unroll
            for (int d = 0; d < D; ++d) yy[d] = y[d][int(v)];
            cp_pin8(yy);
            cm_silu<ALG>(yy, zz, gate, PP(v));
            cp_pin8(zz);
#pragma unroll
            for (int d = 0; d < D; ++d) z[d][int(v)] = zz[d];
            CM_FENCE();
        });
        if constexpr (SAVES) {
            cp_put(XZp, lane, z);    // (the last readers of XZ - the previous tile's d/dW1 - are behind the tile's closing rendezvous)
            pair_sync();
        } else {
        pair_sync();                 // everybody is done with the input chunks
        cp_put(XZp, lane, z);
        pair_sync();
#pragma unroll
        for (int d = 0; d < D; ++d) R[d] = s[d] = f4{0.f, 0.f, 0.f, 0.f};
        {
            f4 zo[8];
            cp_get(XZo, lane, zo);
            cm_mix_one<ALG, 4, GSC>(R, z, ldsa + wce(0, p, p));
            cm_mix_one<ALG, 4, GSC>(R, zo, ldsa + wce(0, p, 1 - p));
            cm_mix_one<ALG, 4, GSC>(s, z, ldsa + wce(1, p, p));
            cm_mix_one<ALG, 4, GSC>(s, zo, ldsa + wce(1, p, 1 - p));
        }
        }
        // d/d(out), own half: it travels under the per-channel product phase
        CM_FENCE();
        CmPiece gp;
        gp.load((kLast ? io.gy + (size_t)(MODE == MODE_EDGE ? (long)T.i_dst : T.lrow) * ROW : io.plw_g1 + (size_t)T.lrow * ROW) +
                (16 * p + q) * D);
        asm volatile("" ::: "memory");
        float nlsum = 0.f;
        if constexpr (SAVES) {       // the smooth norms of the saved s rows
            static_for<0, 4>([&](auto v) {
                float qs = 0.f;
                static_for<0, D>([&](auto dd) {
                    constexpr int d = decltype(dd)::value;
                    qs += qsf<ALG, d> * s[d][int(v)] * s[d][int(v)];
                });
                nlsum += cl_smooth_abs_sqrt(qs);
            });
        } else {
        static_for<0, 4>([&](auto v) {
            float zz[D], RR[D], LL[D], invden[4];
#pragma unroll
            for (int d = 0; d < D; ++d) { zz[d] = z[d][int(v)]; RR[d] = R[d][int(v)]; LL[d] = s[d][int(v)]; }
            cp_pin8(zz); cp_pin8(RR); cp_pin8(LL);
            nlsum += cm_gp_tail<ALG>(zz, RR, LL, invden, PP(v));
            cp_pin8(LL);
#pragma unroll
            for (int d = 0; d < D; ++d) s[d][int(v)] = LL[d];
            CM_FENCE();
        });
        }
        // mean over the 32 channels: the partner's row sums through red[0]
        const float nl_own = cm_q_sum(nlsum);
        if (q == 0) red[(0 * 2 + p) * 16 + r] = nl_own;
        pair_sync();
        const float invMn = fast_rcp(__builtin_fmaf(nl_own + red[(0 * 2 + (1 - p)) * 16 + r], 1.0f / float(C), kEps));
        stamp(2); CB_MARK(2);
        CM_FENCE();
        // ---- MVLayerNorm backward -> ggp = d/d(gp + linear_left output)
        f4 ggp[8];
        cm_unpack(ggp, gp);
        if (!T.valid) {
#pragma unroll
            for (int d = 0; d < D; ++d) ggp[d] = f4{0.f, 0.f, 0.f, 0.f};
        }
        {
            float dot[4], S = 0.f;
            static_for<0, 4>([&](auto v) {
                float a = 0.f;
#pragma unroll
                for (int d = 0; d < D; ++d) a = __builtin_fmaf(ggp[d][int(v)], s[d][int(v)], a);
                dot[v] = a;
                S = __builtin_fmaf(PP(v)[2], a, S);
            });
            const float S_own = cm_q_sum(S);
            if (q == 0) red[(1 * 2 + p) * 16 + r] = S_own;
            pair_sync();
            const float gMn = -(S_own + red[(1 * 2 + (1 - p)) * 16 + r]) * invMn * invMn * (1.0f / float(C));
            float sums[8];
            static_for<0, 4>([&](auto v) {
                float qs = 0.f;
                static_for<0, D>([&](auto dd) {
                    constexpr int d = decltype(dd)::value;
                    qs += qsf<ALG, d> * s[d][int(v)] * s[d][int(v)];
                });
                const float inl = fast_rcp(cl_smooth_abs_sqrt(qs));
                const float gqs = gMn * (0.5f * qs) * (inl * inl * inl);
                const float k0 = PP(v)[2] * invMn;
                static_for<0, D>([&](auto dd) {
                    constexpr int d = decltype(dd)::value;
                    ggp[d][int(v)] = __builtin_fmaf(k0, ggp[d][int(v)], gqs * (2.0f * qsf<ALG, d>) * s[d][int(v)]) * kInvSqrt2;
                });
                sums[2 * v] = dot[v] * invMn;      // d/d(la)
                sums[2 * v + 1] = ggp[0][int(v)];  // d/d(bL)
                CM_FENCE();
            });
            sm[0] += cb_rows_sum<8>(sums, r);
        }
        cp_put(GGp, lane, ggp);
        pair_sync();
        stamp(3); CB_MARK(3);
        CM_FENCE();
        // ---- d/d(linear_left weight)[group p][group m] = ggp_p^T z_m; d/dz = WL^T ggp (both groups' ggp)
        cp_wgrad2<ALG>(aWL[0], aWL[1], GGp, XZ0, XZ1, lane);
        f4 gz[8];
#pragma unroll
        for (int d = 0; d < D; ++d) gz[d] = f4{0.f, 0.f, 0.f, 0.f};
        auto mix_t = [&](int which, const f4 (&own)[8]) __attribute__((always_inline)) {   // gz += W^T (gradient of both groups), W = linear_right / left
            f4 oth[8];
            cp_get(GGo, lane, oth);
            // entry (which, g, m' = the gradient's group, m = p)
            cb_mix_t<ALG, GSC>(gz, own, ldst + wce(which, p, p));
            cb_mix_t<ALG, GSC>(gz, oth, ldst + wce(which, 1 - p, p));
        };
        mix_t(1, ggp);
        stamp(4); CB_MARK(4);
        CM_FENCE();
        // ---- geometric product + normalisation backward, per channel: R becomes d/dR
        {
            CbCollect<1> col;
            static_for<0, 4>([&](auto v) {
                float gg[D], zf[D], RR[D], gzz[D], gRR[D];
#pragma unroll
                for (int d = 0; d < D; ++d) {   // d/d(gp) and z of this channel come back from the slots: 64 registers less in this phase
                    gg[d] = GGp[d * 256 + r * 16 + 4 * q + int(v)];
                    zf[d] = XZp[d * 256 + r * 16 + 4 * q + int(v)];
                    RR[d] = R[d][int(v)];
                }
                cp_pin8(RR);
                cb_gp_bwd<ALG, true>(gg, zf, RR, gzz, gRR, PP(v), [&](auto k, float val) {
                    col.template add<24 * decltype(v)::value + decltype(k)::value>(val, sm, r);
                });
                cp_pin8(gzz); cp_pin8(gRR);
#pragma unroll
                for (int d = 0; d < D; ++d) { gz[d][int(v)] += gzz[d]; R[d][int(v)] = gRR[d]; }
                asm volatile("" ::: "memory");
                CM_FENCE();
            });
        }
        stamp(5); CB_MARK(5);
        CM_FENCE();
        pair_sync();                 // everybody is done with ggp in GG
        cp_put(GGp, lane, R);
        // the input again (the gates' argument y = W1 x; d/dW1's operand): requested here, it travels under the MFMAs below
        asm volatile("" : "+v"(T.i_dst), "+v"(T.i_src), "+v"(T.i_perm), "+v"(T.lrow));
        in.issue(io, T, p, q);
        if constexpr (SAVES) cm_load_lane(y, sp + s_step);   // the gates' argument once more (not kept live over the product's backward)
        asm volatile("" ::: "memory");
        pair_sync();
        cp_wgrad2<ALG>(aWR[0], aWR[1], GGp, XZ0, XZ1, lane);
        mix_t(0, R);
        stamp(6); CB_MARK(6);
        CM_FENCE();
        pair_sync();                 // z and d/dR are no longer read
        f4 x0[8];
        if constexpr (SAVES) {       // the input rows go to the slots for d/dW1 only (XZ: the LAST full-width segment's halves)
            f4 x1[8], xa[8];
            in.finish(x0, x1, xa, T);
            if constexpr (NFULL > 1) cp_put(XZp, lane, x1);
            else cp_put(XZp, lane, x0);
            if constexpr (kAttr) { if (p == 0) cp_put(XA, lane, xa); }
        } else {
            mvlinear(x0, false);
        }
        stamp(7); CB_MARK(7);
        CM_FENCE();
        // ---- MVSiLU backward: gz becomes d/dy
        {
            CbCollect<7> col;
            float tail[4];
            static_for<0, 4>([&](auto v) {
                float yy[D], gzz[D], gyy[D], gs[9];
#pragma unroll
                for (int d = 0; d < D; ++d) { yy[d] = y[d][int(v)]; gzz[d] = gz[d][int(v)]; }
                cp_pin8(yy); cp_pin8(gzz);
                yy[0] += PP(v)[0];   // MVLinear bias
                cm_silu_bwd<ALG>(gzz, yy, gyy, gs, PP(v));
                cp_pin8(gyy);
#pragma unroll
                for (int d = 0; d < D; ++d) gz[d][int(v)] = gyy[d];
                static_for<0, 9>([&](auto k) {
                    constexpr int idx = 9 * decltype(v)::value + decltype(k)::value;
                    if constexpr (idx < 32) col.template add<idx>(gs[k], sm, r);
                    else tail[idx - 32] = gs[k];
                });
                CM_FENCE();
            });
            sm[9] += cb_rows_sum<4>(tail, r);
        }
        // ---- d/dy through the slots: d/dW1 against the input chunks in XZ / XA, d/d(input)
        cp_put(GGp, lane, gz);
        pair_sync();
        if constexpr (NFULL > 1) {   // node program: XZ holds the aggregate's halves; the first segment follows
            cp_wgrad2<ALG>(aW1[2], aW1[3], GGp, XZ0, XZ1, lane);
            pair_sync();
            cp_put(XZp, lane, x0);
            pair_sync();
        }
        cp_wgrad2<ALG>(aW1[0], aW1[1], GGp, XZ0, XZ1, lane);
        if constexpr (kAttr) cp_wgrad<ALG>(aW1[NCH - 1], GGp, XA, lane);
        // d/d(input): wave p emits the chunk p of every full-width segment: gx = sum over the gradient's groups W1^T gy
        f4 gyo[8];
        cp_get(GGo, lane, gyo);
        auto gx_of = [&](int chunk, f4 (&gx)[8]) __attribute__((always_inline)) {
#pragma unroll
            for (int d = 0; d < D; ++d) gx[d] = f4{0.f, 0.f, 0.f, 0.f};
            cb_mix_t<ALG, GS1>(gx, gz, ldst + w1e(p, chunk));
            cb_mix_t<ALG, GS1>(gx, gyo, ldst + w1e(1 - p, chunk));
        };
        f4 gx0[8];
        gx_of(p, gx0);
        stamp(8); CB_MARK(8);
        CM_FENCE();
        // the next tile's rows leave in front of this tile's stores / atomics
        const CmTile<MODE> Tc = T;
        T.template load<NA>(io, tile + tstride, r);
        if constexpr (SAVES) cm_load_lane(ynext, state_of(T) + s_step);
        else in.issue(io, T, p, q);
        asm volatile("" ::: "memory");
        const int coff = (16 * p + q) * D;
        if constexpr (K > 0) {
            if (Tc.valid) cm_store_piece(io.plw_g1 + (size_t)Tc.row * ROW + coff, gx0);
        } else if constexpr (MODE == MODE_EDGE) {
            if constexpr (NA > 0) {
                if (io.gx[1] && p == 1) {   // wave 1 takes the attribute chunk (wave 0 gathers it)
                    f4 gx[8];
                    gx_of(NCH - 1, gx);
                    static_for<0, (NA + 3) / 4>([&](auto v) {
                        if (Tc.valid && q + 4 * v < NA) {
                            float* pg = io.gx[1] + (size_t)Tc.i_perm * (NA * D) + (q + 4 * v) * D;
                            cl_st4(pg, f4{gx[0][int(v)], gx[1][int(v)], gx[2][int(v)], gx[3][int(v)]});
                            cl_st4(pg + 4, f4{gx[4][int(v)], gx[5][int(v)], gx[6][int(v)], gx[7][int(v)]});
                        }
                    });
                }
            }
            if (io.gx[0]) {
                if (io.row_store) {
                    if (Tc.valid) cm_store_piece(io.gx[0] + (size_t)Tc.lrow * ROW + coff, gx0);
                } else {
                    pair_sync();     // the slots are free: this wave stages its half rows in XZp (.. + SS rows fit one slot + 64 floats of XA's neighbour: use GG/XZ pair)
                    float* sc = slots + 3 * p * kCbSlot;   // XZ[0..1] | GG[0..1]: 16 x (HALF + 4) floats = 2112 <= two adjacent slots
                    cm_store_piece(sc + r * SS + q * D, gx0);
                    cb_sync();
                    cp_scatter<HALF, ROW, true>(sc, Tc.valid ? Tc.i_dst : -1, Tc.valid ? Tc.i_src : -1, io.gx[0] + 16 * p * D, lane);
                }
            }
        } else {
            if (io.gx[0]) {
                if (Tc.valid) {
                    if (io.resid_bwd) {
                        CmPiece res;
                        res.load(io.gy + (size_t)Tc.row * ROW + coff);
                        f4 rr[8];
                        cm_unpack(rr, res);
#pragma unroll
                        for (int d = 0; d < D; ++d) gx0[d] += rr[d];
                    }
                    cm_store_piece(io.gx[0] + (size_t)Tc.row * ROW + coff, gx0);
                }
            }
            if (io.gx[1]) {
                f4 gx[8];
                gx_of(2 + p, gx);
#pragma unroll
                for (int d = 0; d < D; ++d) gx[d] *= Tc.scale;
                if (Tc.valid) cm_store_piece(io.gx[1] + (size_t)Tc.row * ROW + coff, gx);
            }
            if constexpr (NA > 0) {
                if (io.gx[2] && p == 1) {
                    f4 gx[8];
                    gx_of(NCH - 1, gx);
                    static_for<0, (NA + 3) / 4>([&](auto v) {
                        if (Tc.valid && q + 4 * v < NA) {
                            float* pg = io.gx[2] + (size_t)Tc.row * (NA * D) + (q + 4 * v) * D;
                            cl_st4(pg, f4{gx[0][int(v)], gx[1][int(v)], gx[2][int(v)], gx[3][int(v)]});
                            cl_st4(pg + 4, f4{gx[4][int(v)], gx[5][int(v)], gx[6][int(v)], gx[7][int(v)]});
                        }
                    });
                }
            }
        }
        pair_sync();                 // the slots are free for the next tile
        stamp(10); CB_MARK(10);
        CM_FENCE();
    }

    // ---- end of the block: pair 0's waves write their sums into ONE image of the slice (wave p covers the output group
    // p: disjoint), pair 1's add theirs, the workgroup writes it out. The image lies over the slots.
    __syncthreads();
    float* img = work;
    static_assert(PT::total <= (kCpWaves / 2) * kCpSlots * kCbSlot, "the image fits the slots");
    for (int round = 0; round < 2; ++round) {
        if (pair == round) {
            const bool add = round != 0;
            const int j = lane & 15, qq = lane >> 4;
            // weight tiles: D[i = 4 qq + v][j] = d/dW[16 p + orow(i)][first channel + orow(j)] (4 grades = one 16-byte vector)
            auto put_tile = [&](const f4 (&acc)[4], int base, int I, int coff, int width) {
                const int c = TF::orow(j);
                if (c < width) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        float* p0 = img + base + ((16 * p + TF::orow(4 * qq + v)) * I + coff + c) * G;
                        const f4 val = f4{acc[0][v], acc[1][v], acc[2][v], acc[3][v]};
                        const f4 old = cl_ld4(p0);
                        cl_st4(p0, add ? old + val : val);
                    }
                }
            };
            static_for<0, NCH>([&](auto ch) {
                constexpr bool at = TF::attr(ch);
                put_tile(aW1[ch], 0, TF::I, at ? NSEG * C : 16 * ch, at ? NA : 16);
            });
            put_tile(aWR[0], PT::pWR, C, 0, 16);
            put_tile(aWR[1], PT::pWR, C, 16, 16);
            put_tile(aWL[0], PT::pWL, C, 0, 16);
            put_tile(aWL[1], PT::pWL, C, 16, 16);
#pragma unroll
            for (int g = 0; g < kCbGroups; ++g) {
                int v, idx;
                SM::decode(g, j, v, idx);
                if (v >= 0) {
                    float* p0 = img + PT::pS + PT::off(idx) + (16 * p + 4 * v + qq) * PT::stride(idx);
                    const float old = *p0;
                    *p0 = add ? old + sm[g] : sm[g];
                }
            }
        }
        __syncthreads();
    }
    float* part = io.rl_partials + (K == 0 ? 0 : (size_t)kClSliceCap * ClPart<ALG, C, CmTab<C, MODE, NA, 0>::I>::total) +
                  (size_t)blockIdx.x * PT::total;
    static_assert(PT::total % 4 == 0, "slice length");
    for (int e = 4 * threadIdx.x; e < PT::total; e += 4 * 64 * kCpWaves) cl_st4(part + e, cl_ld4(img + e));
    stamp(17); CB_MARK(17);
    CM_FENCE();
}

template <class ALG, int C, int MODE, int NBLK, int NA>
constexpr size_t cp_lds_bytes() {
    int tabs = CmTab<C, MODE, NA, 0>::total;
    if (NBLK > 1 && CmTab<C, MODE, NA, 1>::total > tabs) tabs = CmTab<C, MODE, NA, 1>::total;
    return sizeof(float) * (tabs + (kCpWaves / 2) * (kCpSlots * kCbSlot + 64));
}

template <class ALG, int C, int MODE, int NBLK, int NA, bool SAVES = false>
__global__ void __launch_bounds__(64 * kCpWaves, CP_OCC) cemlp_cmp_kernel(const DevCemlp C_arg, const RowIO io_arg) {
    typedef const char __attribute__((address_space(4))) * KArgPtr;
    const KArgPtr ka = (KArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr size_t kIoOffset = (sizeof(DevCemlp) + alignof(RowIO) - 1) / alignof(RowIO) * alignof(RowIO);
    const DevCemlp& Cd = *(const DevCemlp*)(const char*)ka;
    const RowIO& io = *(const RowIO*)(const char*)(ka + kIoOffset);
    (void)C_arg; (void)io_arg;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    ClStamp stamp(0);
    constexpr int tabs0 = CmTab<C, MODE, NA, 0>::total, tabs1 = NBLK > 1 ? CmTab<C, MODE, NA, 1>::total : 0;
    constexpr int tabs = tabs0 > tabs1 ? tabs0 : tabs1;
    constexpr int kSlotFloats = (kCpWaves / 2) * kCpSlots * kCbSlot;
    const int pair = threadIdx.x >> 7;
    float* work = smem + tabs;
    float* slots = work + pair * (kCpSlots * kCbSlot);
    float* red = work + kSlotFloats + pair * 64;
    if constexpr (NBLK > 1) {
        cb_stage_block<ALG, C, CmTab<C, MODE, NA, 1>, 64 * kCpWaves>(Cd.b[1], smem, threadIdx.x);
        __syncthreads();
        stamp(0);
        cp_block<ALG, C, MODE, NBLK, NA, 1, SAVES>(io, smem, slots, red, work, stamp);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's hand-over rows have left for L2
        __syncthreads();
    }
    cb_stage_block<ALG, C, CmTab<C, MODE, NA, 0>, 64 * kCpWaves>(Cd.b[0], smem, threadIdx.x);
    __syncthreads();
    stamp(0);
    cp_block<ALG, C, MODE, NBLK, NA, 0, SAVES>(io, smem, slots, red, work, stamp);
    stamp.flush(io.stamps, threadIdx.x & 63);
}

}  // namespace csmpn
