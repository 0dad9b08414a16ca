// Fused CEMLP block (MVLinear -> MVSiLU -> SteerableGeometricProduct(+Normalization)
// -> MVLayerNorm) forward and recompute-backward for gfx950, one 16-row tile at a time.
//
// Follows the arithmetic of csmpn/models/cegnn_utils.py:34-155,287-338 (SURVEY.md
// Appendix A), re-derived for the hardware:
//
//  * Lane layout of every activation tensor ("lane layout"): a wave owns a tile of
//    16 rows x 16 channels. lane l: channel c = 16*mt + (l & 15), rows 4*(l >> 4) + v,
//    v = 0..3, all D blades in registers:  f4 t[D]  (t[d][v]).  This is exactly the
//    C/D layout of v_mfma_f32_16x16x4_f32 with M = rows, N = channels.
//  * Dense channel mixing (MVLinear, linear_left/right and their transposes) runs on
//    the fp32 MFMA: A = activations read from an LDS tile [row][blade][channel]
//    (one ds_read_b128 = 4 k-steps), B = weights pre-packed in fragment order.
//  * Weight gradients are MFMAs whose A operand is the lane-layout gradient itself
//    (registers) and whose B operand is the input-side tile in LDS.
//  * Everything else (gates, norms, the sign-table geometric product with D^2
//    products per channel instead of the reference's dense D^3 einsum) is VALU work
//    in registers, fully unrolled with compile-time signs and indices.
//  * Backward stores no [rows, C, D] activations: it recomputes the block forward.
#pragma once
#include <hip/hip_runtime.h>

#include "algebra.hpp"

namespace csmpn {

typedef float f4 __attribute__((ext_vector_type(4)));

#define CSMPN_DEV __device__ __forceinline__
// stop the instruction scheduler from moving code across a phase boundary (keeps the live
// register set of one phase from overlapping the next one's)
#define CSMPN_PHASE() __builtin_amdgcn_sched_barrier(0)

constexpr float kInvSqrt2 = 0.70710678118654752440f;
constexpr float kEps = 1e-6f;        // cegnn_utils.py:5
constexpr float kSmooth = 1e-16f;    // cliffordalgebra.py:148

// ---------------------------------------------------------------------------------
// device-side descriptors (filled by the host, passed by value as kernel arguments)

struct DevBlock {
    int I, O;            // in / out channels
    int KKi, KKo;        // ceil(I/16), ceil(O/16)
    int CPi, CPo;        // channels padded to a multiple of 4 (LDS tile width)
    int has_b1;          // MVLinear bias present
    int lds_goff;        // float offset of this block's gradient mirror in LDS
    // small parameters, reference layouts
    const float *b1, *sa, *sb, *w, *an, *bL, *la;
    // packed weight fragments (f4 per lane): forward [g][nt][kk][64], transposed [g][it][kk][64]
    const f4 *pfW1, *pfWR, *pfWL, *pbW1, *pbWR, *pbWL;
    // gradient accumulators, reference layouts (global)
    float *gW1, *gb1, *gsa, *gsb, *gw, *gan, *gWR, *gWL, *gbL, *gla;
    int w1_sub;          // 1: W1 is [O,I,G]; 0: [O,I]
    int pad_;
};

struct DevCemlp {
    int nblk;
    int MT;              // waves cooperating on one row tile = ceil(max O / 16)
    int RT;              // row tiles per workgroup
    int grads_in_lds;    // gradient mirror lives in LDS (flushed once per workgroup)
    int off_in, off_p0, off_p1, off_z, off_g, off_red;  // float offsets inside one row tile's LDS region
    int tile_floats;     // LDS floats per row tile
    int mirror_floats;   // LDS floats of the gradient mirror (0 if not used)
    float* gtiles;       // non-null: row-tile buffers live in this global scratch (too big for LDS)
    DevBlock b[4];
};

// One concatenated input segment: x[:, off:off+ch, :] = scale * (a[ia[row]] - b[ib[row]])
struct Seg {
    const float* a;
    const float* b;        // nullable
    const int* ia;         // nullable: identity
    const int* ib;
    const int* deg;        // nullable: scale = 1/max(deg[row],1)
    int ch;
    int off;
};

enum { MODE_PLAIN = 0, MODE_EDGE = 1, MODE_NODE = 2 };

struct RowIO {
    long rows;
    int nseg;
    int pad_;
    Seg seg[3];
    // forward outputs
    float* y;               // PLAIN/NODE: [rows, O, D]
    const float* resid;     // NODE: h (or null)
    float* agg;             // EDGE: [N, O, D] atomically accumulated by dst
    const int* dst;         // EDGE
    const int* src;         // EDGE
    const int* perm;        // EDGE
    // backward
    const float* gy;        // PLAIN/NODE: [rows, O, D]; EDGE: g_agg [N, O, D] gathered by dst
    float* gx[3];           // per segment gradient target (nullable)
    int resid_bwd;          // NODE: add gy to gx[0]
    int pad2_;
};

// ---------------------------------------------------------------------------------
// small helpers

template <class ALG, int d> constexpr float qsf = float(ALG::t.qsign[d]);

CSMPN_DEV f4 mfma16(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

template <int CTRL>
CSMPN_DEV float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
// sum over the 16 lanes of a DPP row (= the 16 channels of a wave tile); result in every lane
CSMPN_DEV float row16_sum(float v) {
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x124>(v);  // row_ror 4
    v += dpp_mov<0x128>(v);  // row_ror 8
    return v;
}
CSMPN_DEV f4 row16_sum4(f4 v) {
    f4 o;
    o.x = row16_sum(v.x); o.y = row16_sum(v.y); o.z = row16_sum(v.z); o.w = row16_sum(v.w);
    return o;
}
CSMPN_DEV float hsum(f4 v) { return (v.x + v.y) + (v.z + v.w); }
// sum over the 4 row-quarters (lanes l, l^16, l^32, l^48)
CSMPN_DEV float quarters_sum(float v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
// Accuracy: the parity bar is 1e-5 relative against the reference's fp32 CPU path. The
// hardware approximations (v_rcp_f32, v_exp_f32, v_rsq_f32: ~1 ulp) are each refined by one
// Newton / compensation step (2-4 FMAs), which brings them to <=1 ulp of the exact value at
// a fraction of the IEEE division / expf / sqrtf sequences. -DCSMPN_FAST_MATH drops the
// refinement (measured: up to 1.1e-5 on small gradients, i.e. over the bar).
#ifdef CSMPN_FAST_MATH
CSMPN_DEV float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
CSMPN_DEV float exp_neg(float x) { return __builtin_amdgcn_exp2f(-1.44269504088896340736f * x); }
CSMPN_DEV float sqrt_pos(float x) { return __builtin_amdgcn_sqrtf(x); }
#else
CSMPN_DEV float fast_rcp(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(r, __builtin_fmaf(-x, r, 1.0f), r);
}
// exp(-x) = 2^(t + tl) with t = fl(-x*L), tl = product error + x*(log2e - L)
CSMPN_DEV float exp_neg(float x) {
    const float L = 1.44269502162933349609375f, Llo = 1.925963033500011e-08f;
    const float t = -x * L;
    const float tl = __builtin_fmaf(-x, L, -t) - x * Llo;
    const float e = __builtin_amdgcn_exp2f(t);
    return __builtin_fmaf(e, tl * 0.693147180559945309417f, e);
}
// sqrt for x >= ~1e-16 (never denormal here): rsq + one Newton step
CSMPN_DEV float sqrt_pos(float x) {
    const float r = __builtin_amdgcn_rsqf(x);
    const float s = x * r;
    return __builtin_fmaf(__builtin_fmaf(-s, s, x), 0.5f * r, s);
}
#endif
CSMPN_DEV float sigmoidf(float x) { return fast_rcp(1.0f + exp_neg(x)); }
CSMPN_DEV f4 rcp4(f4 x) { return f4{fast_rcp(x.x), fast_rcp(x.y), fast_rcp(x.z), fast_rcp(x.w)}; }
CSMPN_DEV f4 sigmoid4(f4 x) { return f4{sigmoidf(x.x), sigmoidf(x.y), sigmoidf(x.z), sigmoidf(x.w)}; }
CSMPN_DEV f4 sqrt4(f4 x) { return f4{sqrt_pos(x.x), sqrt_pos(x.y), sqrt_pos(x.z), sqrt_pos(x.w)}; }
// (q^2 + 1e-16)^(1/4)  (cliffordalgebra.py:148-149)
CSMPN_DEV f4 smooth_abs_sqrt4(f4 q) { return sqrt4(sqrt4(q * q + kSmooth)); }
CSMPN_DEV f4 splat(float v) { return f4{v, v, v, v}; }

// Storage variants of the row-tile buffers (compile time, so that the LDS variants use
// pure LDS addressing: ds_* instructions instead of flat_*):
//   VAR_WAVE      one wave owns a row tile; buffers + gradient mirror in LDS; no barriers
//   VAR_GROUP     MT waves share a row tile; buffers + mirror in LDS; workgroup barriers
//   VAR_GROUP_NM  as VAR_GROUP, but the gradient mirror does not fit beside the tiles:
//                 parameter gradients go to the global accumulators directly
//   VAR_GLOBAL    buffers in a global scratch (tiles beyond 160 KB of LDS), gradients by
//                 global atomics, workgroup barriers
enum { VAR_WAVE = 0, VAR_GROUP = 1, VAR_GROUP_NM = 2, VAR_GLOBAL = 3 };
template <int VAR> constexpr bool kVarBarrier = VAR != VAR_WAVE;
template <int VAR> constexpr bool kVarMirror = VAR == VAR_WAVE || VAR == VAR_GROUP;

template <int VAR>
CSMPN_DEV void tile_sync() {
    if constexpr (kVarBarrier<VAR>) {
        __syncthreads();
    } else {
        // one wave owns the tile: LDS operations of a wave execute in order, only
        // the compiler must not move them across this point
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// ---------------------------------------------------------------------------------
// MFMA pieces

// acc[d][v] += sum_in  T[row][d][in] * W[c][in][grade(d)]
// tile: LDS [16][D][CP] with row stride RS; frags: [g][kk][64] f4 for this wave's channel tile
template <class ALG>
CSMPN_DEV void linear_from_tile(f4 (&acc)[ALG::D], const float* tile, int RS, int CP, int KK,
                                const f4* frags, int lane) {
    constexpr int D = ALG::D, G = ALG::G;
    const int row = lane & 15, kq = lane >> 4;
    for (int kk = 0; kk < KK; ++kk) {
        const int c0 = 16 * kk + 4 * kq;
        const bool valid = c0 < CP;
        const float* ap = tile + row * RS + c0;
        static_for<0, G>([&](auto g) {
            const f4 b = frags[(g * KK + kk) * 64 + lane];
            constexpr int d0 = ALG::gstart(g), nd = ALG::gsize(g);
            f4 a[nd];
#pragma unroll
            for (int t = 0; t < nd; ++t)
                a[t] = valid ? *reinterpret_cast<const f4*>(ap + (d0 + t) * CP) : splat(0.f);
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int t = 0; t < nd; ++t) acc[d0 + t] = mfma16(a[t][v], b[v], acc[d0 + t]);
        });
    }
}

// Weight gradient tiles for one linear: for every input-channel tile it and grade g
//   gW[o = 16*mt + 4*(l>>4) + v][c = 16*it + (l&15)][g] += sum_{rows, d in g} Gr[row][d][o] * T[row][d][c]
// A operand = lane-layout gradient (registers), B operand = input-side LDS tile.
// Accumulates into `dstp` (LDS mirror laid out [g][O][I], or global reference layout).
template <class ALG, bool dst_is_mirror>
CSMPN_DEV void weight_grad(const f4 (&gr)[ALG::D], const float* tile, int RS, int CP, int I, int O, int KKin,
                           int mt, int lane, float* dstp, bool has_grades) {
    constexpr int D = ALG::D, G = ALG::G;
    const int n = lane & 15, q = lane >> 4;
    for (int it = 0; it < KKin; ++it) {
        const int c = 16 * it + n;
        const bool cvalid = c < CP;
        f4 accg[G];
        static_for<0, G>([&](auto g) {
            f4 acc = splat(0.f);
            constexpr int d0 = ALG::gstart(g), nd = ALG::gsize(g);
#pragma unroll
            for (int t = 0; t < nd; ++t) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float b = cvalid ? tile[(4 * q + v) * RS + (d0 + t) * CP + c] : 0.f;
                    acc = mfma16(gr[d0 + t][v], b, acc);
                }
            }
            accg[g] = acc;
        });
        if (c < I) {
            if (has_grades) {
                static_for<0, G>([&](auto g) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int o = 16 * mt + 4 * q + v;
                        if (o < O) {
                            float* p = dst_is_mirror ? dstp + (g * O + o) * I + c : dstp + (o * I + c) * G + g;
                            atomicAdd(p, accg[g][v]);
                        }
                    }
                });
            } else {
                f4 tot = accg[0];
                static_for<1, G>([&](auto g) { tot += accg[g]; });
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int o = 16 * mt + 4 * q + v;
                    if (o < O) atomicAdd(dstp + o * I + c, tot[v]);
                }
            }
        }
    }
}

// write a lane-layout tensor into an LDS tile [row][d][CP]
template <class ALG>
CSMPN_DEV void store_tile(const f4 (&t)[ALG::D], float* tile, int RS, int CP, int mt, int lane) {
    constexpr int D = ALG::D;
    const int c = 16 * mt + (lane & 15), q = lane >> 4;
    if (c < CP) {
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int v = 0; v < 4; ++v) tile[(4 * q + v) * RS + d * CP + c] = t[d][v];
    }
}

// ---------------------------------------------------------------------------------
// per-lane small parameters of one block (channel c of this lane)
template <class ALG>
struct LaneParams {
    float b1, bL, la;
    float sa[ALG::G], sb[ALG::G], sg[ALG::G];  // sg = sigmoid(norm.a)
    bool cvalid;
};

template <class ALG>
CSMPN_DEV LaneParams<ALG> load_lane_params(const DevBlock& B, int c) {
    constexpr int G = ALG::G;
    LaneParams<ALG> p;
    p.cvalid = c < B.O;
    const int cc = p.cvalid ? c : 0;
    p.b1 = (p.cvalid && B.has_b1) ? B.b1[cc] : 0.f;
    p.bL = p.cvalid ? B.bL[cc] : 0.f;
    p.la = p.cvalid ? B.la[cc] : 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        p.sa[g] = p.cvalid ? B.sa[cc * G + g] : 0.f;
        p.sb[g] = p.cvalid ? B.sb[cc * G + g] : 0.f;
        p.sg[g] = p.cvalid ? sigmoidf(B.an[cc * G + g]) : 0.f;
    }
    return p;
}

// forward intermediates kept for the backward (dead-code-eliminated in the pure forward)
template <class ALG>
struct FwdState {
    f4 y[ALG::D];        // MVLinear output
    f4 gate[ALG::G];     // sigmoid gates of MVSiLU
    f4 R[ALG::D];        // linear_right output
    f4 invden[ALG::G];   // 1 / (interpolated norm + eps) of NormalizationLayer
    f4 s[ALG::D];        // (left + gp)/sqrt2, input of MVLayerNorm
    f4 qs, nl, invMn;
};

// sign-table geometric product with per-path weights:
//   out[j] += sum_{(i,k)->j} sign(i,k) * w[path(g_i, g_j, g_k)] * z[i] * r[k]
template <class ALG>
CSMPN_DEV void weighted_gp(f4 (&out)[ALG::D], const f4 (&z)[ALG::D], const f4 (&r)[ALG::D], const float* wrow) {
    constexpr int P = ALG::P;
    static_for<0, P>([&](auto p) {
        constexpr int gi = ALG::t.path_g[p][0], gj = ALG::t.path_g[p][1], gk = ALG::t.path_g[p][2];
        constexpr int i0 = ALG::gstart(gi), ni = ALG::gsize(gi);
        constexpr int j0 = ALG::gstart(gj), nj = ALG::gsize(gj);
        constexpr int k0 = ALG::gstart(gk), nk = ALG::gsize(gk);
        const float w = wrow[p];
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

// backward of weighted_gp: gz, gr accumulate; gw[p] is reduced over the tile's rows and
// added to gw_dst. The left/right operands z = gate*y and r = R*invden are rebuilt per path
// from the kept forward state instead of being held in registers (64 VGPRs less).
template <class ALG>
CSMPN_DEV void weighted_gp_bwd(const f4 (&ggp)[ALG::D], const f4 (&y)[ALG::D], const f4 (&gate)[ALG::G],
                               const f4 (&R)[ALG::D], const f4 (&invden)[ALG::G], const float* wrow,
                               f4 (&gz)[ALG::D], f4 (&gr)[ALG::D], float* gw_dst, bool lane0q, bool cvalid) {
    constexpr int P = ALG::P;
    static_for<0, P>([&](auto p) {
        constexpr int gi = ALG::t.path_g[p][0], gj = ALG::t.path_g[p][1], gk = ALG::t.path_g[p][2];
        constexpr int i0 = ALG::gstart(gi), ni = ALG::gsize(gi);
        constexpr int j0 = ALG::gstart(gj), nj = ALG::gsize(gj);
        constexpr int k0 = ALG::gstart(gk), nk = ALG::gsize(gk);
        const float w = wrow[p];
        f4 U[ni];    // U[i] = sum_{k,j} sign * ggp[j] * r[k]   (unweighted d/dz)
        f4 zi[ni];   // z[i]
        f4 rk[nk];   // r[k]
#pragma unroll
        for (int t = 0; t < ni; ++t) { U[t] = splat(0.f); zi[t] = gate[gi] * y[i0 + t]; }
#pragma unroll
        for (int t = 0; t < nk; ++t) rk[t] = R[k0 + t] * invden[gk];
        static_for<0, ni>([&](auto ii) {
            static_for<0, nk>([&](auto kk) {
                constexpr int i = i0 + ii, k = k0 + kk;
                constexpr int j = ALG::t.out[i][k];
                if constexpr (j >= j0 && j < j0 + nj) {
                    constexpr float sg = float(ALG::t.sign[i][k]);
                    U[ii] += (sg * ggp[j]) * rk[kk];
                    gr[k] += (sg * w) * (ggp[j] * zi[ii]);
                }
            });
        });
        f4 gwv = splat(0.f);
#pragma unroll
        for (int t = 0; t < ni; ++t) { gz[i0 + t] += w * U[t]; gwv += zi[t] * U[t]; }
        const float tot = quarters_sum(hsum(gwv));
        if (lane0q && cvalid) atomicAdd(gw_dst + p, tot);
    });
}

// LDS mirror offsets (floats) of one block's gradients
struct MirrorOff { int W1, WR, WL, b1, sa, sb, w, an, bL, la, total; };
CSMPN_DEV MirrorOff mirror_offsets(int I, int O, int G, int P, bool w1_sub) {
    MirrorOff m;
    int o = 0;
    m.W1 = o; o += (w1_sub ? G : 1) * O * I;
    m.WR = o; o += G * O * O;
    m.WL = o; o += G * O * O;
    m.b1 = o; o += O;
    m.sa = o; o += O * G;
    m.sb = o; o += O * G;
    m.w = o; o += O * P;
    m.an = o; o += O * G;
    m.bL = o; o += O;
    m.la = o; o += O;
    m.total = o;
    return m;
}

// ---------------------------------------------------------------------------------
// block forward. Input tile in LDS (xin), output in lane layout (out) for this wave's
// channel tile. zbuf: LDS tile for the gated activations (feeds linear_left/right).
// red: LDS scratch [MT][16] floats for cross-wave LayerNorm sums (MULTI only).
template <class ALG, int VAR>
CSMPN_DEV void block_forward(const DevBlock& B, const LaneParams<ALG>& lp, const float* xin, float* zbuf,
                             float* red, int MT, int mt, int lane, FwdState<ALG>& S, f4 (&out)[ALG::D]) {
    constexpr int D = ALG::D, G = ALG::G;
    const int RSi = D * B.CPi + 4, RSo = D * B.CPo + 4;
    const int c = 16 * mt + (lane & 15), q = lane >> 4;
    const bool tile_active = 16 * mt < B.O;

    // 1. MVLinear (cegnn_utils.py:326-338)
#pragma unroll
    for (int d = 0; d < D; ++d) S.y[d] = splat(0.f);
    if (tile_active) linear_from_tile<ALG>(S.y, xin, RSi, B.CPi, B.KKi, B.pfW1 + (size_t)mt * G * B.KKi * 64, lane);
    S.y[0] += lp.b1;

    CSMPN_PHASE();
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
        S.gate[g] = sigmoid4(lp.sa[g] * u + lp.sb[g]);
#pragma unroll
        for (int t = 0; t < nd; ++t) z[d0 + t] = S.gate[g] * S.y[d0 + t];
    });
    store_tile<ALG>(z, zbuf, RSo, B.CPo, mt, lane);
    tile_sync<VAR>();

    CSMPN_PHASE();
    // 3. linear_right / linear_left (cegnn_utils.py:143-148)
    f4 L[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { S.R[d] = splat(0.f); L[d] = splat(0.f); }
    if (tile_active) {
        linear_from_tile<ALG>(S.R, zbuf, RSo, B.CPo, B.KKo, B.pfWR + (size_t)mt * G * B.KKo * 64, lane);
        linear_from_tile<ALG>(L, zbuf, RSo, B.CPo, B.KKo, B.pfWL + (size_t)mt * G * B.KKo * 64, lane);
    }
    L[0] += lp.bL;

    CSMPN_PHASE();
    // 4. NormalizationLayer on the right operand (cegnn_utils.py:42-51)
    f4 r[D];
    static_for<0, G>([&](auto g) {
        constexpr int d0 = ALG::gstart(g), nd = ALG::gsize(g);
        f4 qq = splat(0.f);
        static_for<0, nd>([&](auto t) {
            constexpr int d = d0 + decltype(t)::value;
            qq += qsf<ALG, d> * S.R[d] * S.R[d];
        });
        const f4 m = lp.sg[g] * (smooth_abs_sqrt4(qq) - 1.0f) + 1.0f;
        S.invden[g] = rcp4(m + kEps);
#pragma unroll
        for (int t = 0; t < nd; ++t) r[d0 + t] = S.R[d0 + t] * S.invden[g];
    });

    CSMPN_PHASE();
    // 5. steerable geometric product + first-order term (cegnn_utils.py:126-152)
    if (lp.cvalid) weighted_gp<ALG>(L, z, r, B.w + (size_t)c * ALG::P);
#pragma unroll
    for (int d = 0; d < D; ++d) S.s[d] = L[d] * kInvSqrt2;

    CSMPN_PHASE();
    // 6. MVLayerNorm (cegnn_utils.py:93-96): mean over the channels of the row
    f4 qs = splat(0.f);
    static_for<0, D>([&](auto dd) {
        constexpr int d = decltype(dd)::value;
        qs += qsf<ALG, d> * S.s[d] * S.s[d];
    });
    S.qs = qs;
    S.nl = smooth_abs_sqrt4(qs);
    f4 tot = row16_sum4(lp.cvalid ? S.nl : splat(0.f));
    if constexpr (kVarBarrier<VAR>) {
        if ((lane & 15) == 0) *reinterpret_cast<f4*>(red + mt * 16 + 4 * q) = tot;
        __syncthreads();
        tot = splat(0.f);
        for (int m = 0; m < MT; ++m) tot += *reinterpret_cast<const f4*>(red + m * 16 + 4 * q);
        __syncthreads();
    }
    S.invMn = rcp4(tot * (1.0f / float(B.O)) + kEps);
#pragma unroll
    for (int d = 0; d < D; ++d) out[d] = lp.la * S.s[d] * S.invMn;
}

// ---------------------------------------------------------------------------------
// block backward: given the forward state S of this tile and gout (lane layout),
// accumulate all parameter gradients and leave d/d(MVLinear output) in gy (lane
// layout) AND in the LDS tile gbuf (so the caller can run the transposed MVLinear).
template <class ALG, int VAR>
CSMPN_DEV void block_backward(const DevBlock& B, const LaneParams<ALG>& lp, const FwdState<ALG>& S,
                              const f4 (&gout)[ALG::D], const float* xin, const float* zbuf, float* gbuf,
                              float* red, float* mirror, int MT, int mt, int lane, f4 (&gy)[ALG::D]) {
    constexpr int D = ALG::D, G = ALG::G, P = ALG::P;
    const int RSi = D * B.CPi + 4, RSo = D * B.CPo + 4;
    const int c = 16 * mt + (lane & 15), q = lane >> 4;
    const bool tile_active = 16 * mt < B.O;
    const bool lane0q = q == 0;
    const bool cv = lp.cvalid;
    const MirrorOff mo = mirror_offsets(B.I, B.O, G, P, B.w1_sub != 0);
    float* mir = mirror + B.lds_goff;
    // gradient destinations: LDS mirror (flushed once per workgroup) or, in the global-tile
    // variant, the global reference-layout accumulators directly
    constexpr bool in_lds = kVarMirror<VAR>;
    float *d_b1, *d_sa, *d_sb, *d_w, *d_an, *d_bL, *d_la, *d_W1, *d_WR, *d_WL;
    if constexpr (in_lds) {
        d_b1 = mir + mo.b1; d_sa = mir + mo.sa; d_sb = mir + mo.sb; d_w = mir + mo.w; d_an = mir + mo.an;
        d_bL = mir + mo.bL; d_la = mir + mo.la; d_W1 = mir + mo.W1; d_WR = mir + mo.WR; d_WL = mir + mo.WL;
    } else {
        d_b1 = B.gb1; d_sa = B.gsa; d_sb = B.gsb; d_w = B.gw; d_an = B.gan;
        d_bL = B.gbL; d_la = B.gla; d_W1 = B.gW1; d_WR = B.gWR; d_WL = B.gWL;
    }

    // ---- MVLayerNorm backward
    f4 dot = splat(0.f), gla = splat(0.f);
#pragma unroll
    for (int d = 0; d < D; ++d) { dot += gout[d] * S.s[d]; }
    gla = dot * S.invMn;
    {
        const float t = quarters_sum(hsum(gla));
        if (lane0q && cv) atomicAdd(d_la + c, t);
    }
    f4 gMn = row16_sum4(-(lp.la * dot) * S.invMn * S.invMn);
    if constexpr (kVarBarrier<VAR>) {
        if ((lane & 15) == 0) *reinterpret_cast<f4*>(red + mt * 16 + 4 * q) = gMn;
        __syncthreads();
        gMn = splat(0.f);
        for (int m = 0; m < MT; ++m) gMn += *reinterpret_cast<const f4*>(red + m * 16 + 4 * q);
        __syncthreads();
    }
    // d nl/d qs = 0.5 * qs / nl^3
    const f4 inl = rcp4(S.nl);
    const f4 gqs = (gMn * (1.0f / float(B.O))) * (0.5f * S.qs) * (inl * inl * inl);
    f4 ggp[D];   // = d/d(left) = d/d(gp)
    static_for<0, D>([&](auto dd) {
        constexpr int d = decltype(dd)::value;
        const f4 gs = (lp.la * gout[d]) * S.invMn + gqs * (2.0f * qsf<ALG, d>) * S.s[d];
        ggp[d] = cv ? gs * kInvSqrt2 : splat(0.f);
    });
    {
        const float t = quarters_sum(hsum(ggp[0]));
        if (lane0q && cv) atomicAdd(d_bL + c, t);
    }

    CSMPN_PHASE();
    // ---- d/dz from linear_left: gz = GL . WL^T ; gWL += GL (x) Z
    store_tile<ALG>(ggp, gbuf, RSo, B.CPo, mt, lane);
    tile_sync<VAR>();
    f4 gz[D];
#pragma unroll
    for (int d = 0; d < D; ++d) gz[d] = splat(0.f);
    if (tile_active) {
        linear_from_tile<ALG>(gz, gbuf, RSo, B.CPo, B.KKo, B.pbWL + (size_t)mt * G * B.KKo * 64, lane);
        weight_grad<ALG, in_lds>(ggp, zbuf, RSo, B.CPo, B.O, B.O, B.KKo, mt, lane, d_WL, true);
    }

    CSMPN_PHASE();
    // ---- geometric product backward
    f4 gr[D];
#pragma unroll
    for (int d = 0; d < D; ++d) gr[d] = splat(0.f);
    if (tile_active)
        weighted_gp_bwd<ALG>(ggp, S.y, S.gate, S.R, S.invden, B.w + (size_t)(cv ? c : 0) * P, gz, gr,
                             d_w + (size_t)(cv ? c : 0) * P, lane0q, cv);

    CSMPN_PHASE();
    // ---- NormalizationLayer backward -> gR (q_g and nu_g are rebuilt from R)
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
        {
            const float t = quarters_sum(hsum(gden * (nu - 1.0f))) * lp.sg[g] * (1.0f - lp.sg[g]);
            if (lane0q && cv) atomicAdd(d_an + c * G + g, t);
        }
        const f4 inu = rcp4(nu);
        const f4 gq = (gden * lp.sg[g]) * (0.5f * qR) * (inu * inu * inu);
        static_for<0, nd>([&](auto t) {
            constexpr int d = d0 + decltype(t)::value;
            gR[d] = cv ? gr[d] * S.invden[g] + gq * (2.0f * qsf<ALG, d>) * S.R[d] : splat(0.f);
        });
    });
    tile_sync<VAR>();   // all reads of gbuf (GL) done
    store_tile<ALG>(gR, gbuf, RSo, B.CPo, mt, lane);
    tile_sync<VAR>();
    if (tile_active) {
        linear_from_tile<ALG>(gz, gbuf, RSo, B.CPo, B.KKo, B.pbWR + (size_t)mt * G * B.KKo * 64, lane);
        weight_grad<ALG, in_lds>(gR, zbuf, RSo, B.CPo, B.O, B.O, B.KKo, mt, lane, d_WR, true);
    }

    CSMPN_PHASE();
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
        {
            const float ta = quarters_sum(hsum(gpre * u));
            const float tb = quarters_sum(hsum(gpre));
            if (lane0q && cv) { atomicAdd(d_sa + c * G + g, ta); atomicAdd(d_sb + c * G + g, tb); }
        }
        const f4 gu = gpre * lp.sa[g];
        static_for<0, nd>([&](auto t) {
            constexpr int d = d0 + decltype(t)::value;
            f4 v = gz[d] * S.gate[g];
            if constexpr (g == 0) v += gu;
            else v += gu * (2.0f * qsf<ALG, d>) * S.y[d];
            gy[d] = cv ? v : splat(0.f);
        });
    });
    if (B.has_b1) {
        const float t = quarters_sum(hsum(gy[0]));
        if (lane0q && cv) atomicAdd(d_b1 + c, t);
    }
    CSMPN_PHASE();
    // ---- MVLinear weight gradient; gy tile to LDS for the transposed MVLinear
    tile_sync<VAR>();   // all reads of gbuf (GR) done
    store_tile<ALG>(gy, gbuf, RSo, B.CPo, mt, lane);
    if (tile_active) weight_grad<ALG, in_lds>(gy, xin, RSi, B.CPi, B.I, B.O, B.KKi, mt, lane, d_W1, B.w1_sub != 0);
    tile_sync<VAR>();
}

// flush one block's LDS gradient mirror into the global reference-layout accumulators
template <class ALG>
__device__ void flush_mirror(const DevBlock& B, const float* mirror, int tid, int nthreads) {
    constexpr int G = ALG::G, P = ALG::P;
    const MirrorOff mo = mirror_offsets(B.I, B.O, G, P, B.w1_sub != 0);
    const float* mir = mirror + B.lds_goff;
    const int I = B.I, O = B.O;
    const int nW1 = (B.w1_sub ? G : 1) * O * I;
    for (int e = tid; e < nW1; e += nthreads) {
        // mirror [g][o][i] -> reference [o][i][g]
        const int g = e / (O * I), rem = e % (O * I);
        const float v = mir[mo.W1 + e];
        if (v != 0.f) atomicAdd(B.gW1 + (B.w1_sub ? rem * G + g : rem), v);
    }
    for (int e = tid; e < G * O * O; e += nthreads) {
        const int g = e / (O * O), rem = e % (O * O);
        const float vr = mir[mo.WR + e], vl = mir[mo.WL + e];
        if (vr != 0.f) atomicAdd(B.gWR + rem * G + g, vr);
        if (vl != 0.f) atomicAdd(B.gWL + rem * G + g, vl);
    }
    for (int e = tid; e < O; e += nthreads) {
        if (B.has_b1) atomicAdd(B.gb1 + e, mir[mo.b1 + e]);
        atomicAdd(B.gbL + e, mir[mo.bL + e]);
        atomicAdd(B.gla + e, mir[mo.la + e]);
    }
    for (int e = tid; e < O * G; e += nthreads) {
        atomicAdd(B.gsa + e, mir[mo.sa + e]);
        atomicAdd(B.gsb + e, mir[mo.sb + e]);
        atomicAdd(B.gan + e, mir[mo.an + e]);
    }
    for (int e = tid; e < O * P; e += nthreads) atomicAdd(B.gw + e, mir[mo.w + e]);
}

}  // namespace csmpn
