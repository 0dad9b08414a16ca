// Fused CEMLP block (MVLinear -> MVSiLU -> SteerableGeometricProduct(+Normalization)
// -> MVLayerNorm) forward and recompute-backward for gfx950, one row tile at a time.
//
// Follows the arithmetic of csmpn/models/cegnn_utils.py:34-155,287-338 (SURVEY.md
// Appendix A), re-derived for the hardware:
//
//  * Lane layout of every activation tensor: a wave owns a tile of R = 16*H rows and 16
//    lane columns. lane l: column n = l & 15, quarter q = l >> 4.
//      H = 1: channel c = 16*mt + n,          rows 4q + v           (16 rows x 16 channels)
//      H = 2: half h = n >> 3, c = n & 7,     rows 16h + 4q + v     (32 rows x 8 channels)
//    v = 0..3 are the four accumulator registers of v_mfma_f32_16x16x4_f32, all D blades
//    of a (row, channel) pair sit in one lane:  f4 t[D]  (t[d][v]).
//    H = 2 exists for channel counts <= 8: with H = 1 half of the lane columns would carry
//    padding channels through all the VALU work.
//  * Dense channel mixing (MVLinear, linear_left/right and their transposes) runs on the
//    fp32 MFMA: A = activations from an LDS tile, B = weights pre-packed in fragment order
//    (for H = 2 one fragment per row half, zero in the other half's columns).
//  * LDS tile layout [channel][blade][row]: every stride is a compile-time constant of the
//    algebra (independent of the channel counts), so ds_* instructions use immediate
//    offsets; a lane stores its 4 rows with one ds_write_b128 and weight-gradient B operands
//    are read with one ds_read_b128 per 4 k-steps.
//  * Weight gradients are MFMAs whose A operand is the lane-layout gradient itself
//    (registers) and whose B operand is the input-side tile in LDS.
//  * Everything else (gates, norms, the sign-table geometric product with D^2 products per
//    channel instead of the reference's dense D^3 einsum) is VALU work in registers, fully
//    unrolled with compile-time signs and indices.
//  * Backward stores no [rows, C, D] activations: it recomputes the block forward.
#pragma once
#include <hip/hip_runtime.h>

#include "algebra.hpp"

namespace csmpn {

typedef float f4 __attribute__((ext_vector_type(4)));

#define CSMPN_DEV __device__ __forceinline__
// stop the instruction scheduler from moving code across a phase boundary
#define CSMPN_PHASE() __builtin_amdgcn_sched_barrier(0)

constexpr float kInvSqrt2 = 0.70710678118654752440f;
constexpr float kEps = 1e-6f;        // cegnn_utils.py:5
constexpr float kSmooth = 1e-16f;    // cliffordalgebra.py:148

// ---------------------------------------------------------------------------------
// device-side descriptors (filled by the host, passed by value as kernel arguments)

struct DevBlock {
    int I, O;            // in / out channels
    int KKi, KKo;        // ceil(I/16), ceil(O/16): k-blocks of 16 channels
    int NTi, NTo;        // ceil(I/NW), ceil(O/NW): N tiles of NW = 16/H channels
    int CPi, CPo;        // channels padded to a multiple of 4
    int has_b1;          // MVLinear bias present
    int lds_goff;        // float offset of this block's gradient mirror in LDS
    int lds_woff;        // float offset of this block's weight store in LDS (VAR_WAVE)
    int w1_sub;          // 1: W1 is [O,I,G]; 0: [O,I]
    // dense weights, reference layouts (staged into LDS by the VAR_WAVE kernels)
    const float *W1, *WR, *WL;
    // small parameters, reference layouts
    const float *b1, *sa, *sb, *w, *an, *bL, *la;
    // packed weight fragments (f4 per lane): forward [nt][kk][hp][g][64], transposed [it][kk][hp][g][64]
    const f4 *pfW1, *pfWR, *pfWL, *pbW1, *pbWR, *pbWL;
    // gradient accumulators, reference layouts (global)
    float *gW1, *gb1, *gsa, *gsb, *gw, *gan, *gWR, *gWL, *gbL, *gla;
};

struct DevCemlp {
    int nblk;
    int MT;              // waves cooperating on one row tile = ceil(max O / NW)
    int RT;              // row tiles per workgroup
    int H;               // row halves per tile (1 or 2)
    int off_in, off_p0, off_p1, off_z, off_g, off_red, off_idx;  // float offsets inside one row tile's buffers
    int tile_floats;     // floats per row tile
    int mirror_floats;   // LDS floats of the gradient mirror (0 if not used)
    int wstore_floats;   // LDS floats of the weight store (VAR_WAVE, else 0)
    int share_inz;       // backward: the z buffer aliases the input buffer; the input tile is staged
                         // again in front of the MVLinear weight gradient (one more row tile per CU
                         // where LDS, not registers, limits the resident waves)
    float* gtiles;       // non-null: row-tile buffers live in this global scratch (too big for LDS)
    int det_slice_floats;  // > 0: deterministic mode of the general kernels - the g* pointers are slice 0 of a per-workgroup region
    int phased;            // 1: backward block by block (outer loop over blocks, last first): mirror of ONE block in LDS, d/d(block input) rows through io.plw_g1
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
    // inputs of blocks 1..nblk-1 ([rows, O_{k-1}, D] each, back to back): written by the
    // forward when non-null, read by the backward instead of recomputing the earlier blocks
    float* save;
    const float* saved;
    unsigned long long* stamps;   // diagnostic (-DCSMPN_STAMPS) cycle accumulators, else null
    float* rl_partials;     // row-per-lane backward (cemlp_rl.hpp): one slice of parameter-gradient sums per workgroup
    // deterministic mode (CSMPN_FLAG_DETERMINISTIC, row-per-lane kernels only): EDGE rows are not
    // scattered; `agg` (forward) / `gx[0]` (backward) is an [rows, C, D] table in sorted edge order
    // that a segmented reduction sums afterwards in a fixed order
    int row_store;
    int save_state;         // CSMPN_FLAG_SAVE_STATE (cemlp_cl.hpp): the blocks' outputs in front of the layer norm are saved / available
    const float* plw_tabs;  // wide parity-lane kernels (cemlp_plw.hpp): rotation tables packed into the workspace
    float* plw_g1;          // ... backward: d/d(block-1 input) rows handed from the block-1 launch to the block-0 launch
    float* plw_part;        // ... backward: one slice of weight-gradient tiles per workgroup (added by plw_reduce_kernel)
    // Fused simplex embedding (round 3, MODE_PLAIN of the wide parity-lane kernels; hulls_cssmpnn.py:96-125): row r is vertex
    // order r % emb_nperm of simplex r / emb_nperm; its input channel v * emb_k + k is channel k of the embedded vertex
    // row seg[0].a[emb_verts[r * emb_nv + v]] ([S, emb_k, D]); the emb_nperm consecutive output rows of a simplex are summed
    // and stored as row r / emb_nperm of y; the backward reads d/d(out) there. emb_nperm = 0: off.
    const int* emb_verts;
    int emb_nperm, emb_nv, emb_k, emb_nrows;   // emb_nrows = S: vertex ids are clamped into [0, S) (memory safety; the entry point range-checks)
};

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

// geometry of a row tile
template <class ALG, int H>
struct Geo {
    static constexpr int D = ALG::D, G = ALG::G;
    static constexpr int R = 16 * H;        // rows per tile
    static constexpr int NW = 16 / H;       // channels per N tile (lane columns per half)
    static constexpr int CS = R * D + 4;    // LDS channel stride (floats); +4 breaks the bank pattern
    int lane, n, q, h, cn, r0;
    CSMPN_DEV explicit Geo(int lane_) : lane(lane_), n(lane_ & 15), q(lane_ >> 4) {
        h = H == 1 ? 0 : (n >> 3);
        cn = H == 1 ? n : (n & 7);
        r0 = 16 * h + 4 * q;                // first of the lane's 4 rows
#ifdef CSMPN_STAMPS
        for (int i = 0; i < kStampSlots; ++i) acc[i] = 0;
        t0 = __builtin_amdgcn_s_memtime();
#endif
    }
#ifdef CSMPN_STAMPS
    // diagnostic build only: shader-clock cycles per phase, summed per wave
    static constexpr int kStampSlots = 24;
    mutable unsigned long long t0, acc[kStampSlots];
    CSMPN_DEV void stamp(int id) const {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        acc[id] += t1 - t0;
        t0 = t1;
        __builtin_amdgcn_sched_barrier(0);
    }
#else
    CSMPN_DEV void stamp(int) const {}
#endif
};

// ---------------------------------------------------------------------------------
// small helpers

template <class ALG, int d> constexpr float qsf = float(ALG::t.qsign[d]);

// ---- CSMPN_FLAG_SAVE_STATE: the state regions of the saved buffer ------------------------------------------------------
// [rows x ROW block-1 inputs][rows x ROW hand-over][state regions ...]: a state region holds ONE tensor (s, y or R) of ONE
// block for all rows, as whole row TILES in the owning kernels' lane order - piece e of lane l of a wave's tile lies at
// ((tile_slot * PIECES + e) * 64 + l) * 4 floats, so one store / load instruction of a wave covers 1 KB of contiguous
// memory. (The first version kept a lane's pieces contiguous - 16 bytes per lane at a 64 / 128-byte stride, four to eight
// times the memory transactions: S3 edge forward 212 -> 325 us with the stores, M32 206 -> 299 us.)
// Region k = 2 t + K (t: 0 s, 1 y, 2 R; K: block) starts at  saved + 2 rows ROW + k * state_rows(rows) * ROWP  (ROWP: the row
// length with the channels padded to the kernels' lane groups); csmpn_cemlp_saved_floats sizes the buffer accordingly.
CSMPN_DEV size_t state_rows(long rows) { return (size_t)((rows + 15) & ~15L); }
template <int ROW, int ROWP>
CSMPN_DEV size_t state_region(long rows, int t, int K) { return (size_t)2 * rows * ROW + (size_t)(2 * t + K) * state_rows(rows) * ROWP; }

CSMPN_DEV f4 mfma16(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

template <int CTRL>
CSMPN_DEV float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
// sum over the NW lane columns that hold the channels of one row; result in every lane
template <int H>
CSMPN_DEV float chan_sum(float v) {
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    if constexpr (H == 1) {
        v += dpp_mov<0x124>(v);  // row_ror 4
        v += dpp_mov<0x128>(v);  // row_ror 8
    } else {
        v += dpp_mov<0x141>(v);  // row_half_mirror: the other quad of the same 8 lanes
    }
    return v;
}
template <int H>
CSMPN_DEV f4 chan_sum4(f4 v) {
    return f4{chan_sum<H>(v.x), chan_sum<H>(v.y), chan_sum<H>(v.z), chan_sum<H>(v.w)};
}
CSMPN_DEV float hsum(f4 v) { return (v.x + v.y) + (v.z + v.w); }
// sum over all lanes that hold the same channel: the 4 row-quarters (lanes l^32 by
// ds_bpermute, l^16 by ds_swizzle: the LDS crossbar, no memory access) and, for
// H = 2, the two row halves (lane columns n and n^8: one DPP rotate). Result in every lane.
template <int H>
CSMPN_DEV float channel_rows_sum(float v) {
    // One MFMA with A = 1: D[i][j] = sum_k B[k][j], and B[k = q][j = n] is this value in lane
    // (n, q) - every lane receives the sum over the 4 row quarters of its own column (exact
    // fp32 FMA chain). The LDS-crossbar form (ds_bpermute + ds_swizzle per value, 35 values per
    // block) cost 7 % of the edge backward in exposed LDS latency.
    const f4 r = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, v, f4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    float t = r.x;
    if constexpr (H == 2) t += dpp_mov<0x128>(t);   // row_ror 8: the other half's column
    return t;
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
// the argument is clamped at -87: exp(87) is the last finite power before v_exp_f32 returns inf, and inf poisons
// both refinement steps (inf * 0 in exp_neg, -inf * 0 in fast_rcp) - a gate whose pre-activation is below -88
// (large indefinite quadratic forms: Cl(4,1), hub nodes) came out NaN instead of 0. sigmoid(-87) = 1.6e-38.
CSMPN_DEV float sigmoidf(float x) { return fast_rcp(1.0f + exp_neg(__builtin_fmaxf(x, -87.0f))); }
CSMPN_DEV f4 rcp4(f4 x) { return f4{fast_rcp(x.x), fast_rcp(x.y), fast_rcp(x.z), fast_rcp(x.w)}; }
CSMPN_DEV f4 sigmoid4(f4 x) { return f4{sigmoidf(x.x), sigmoidf(x.y), sigmoidf(x.z), sigmoidf(x.w)}; }
CSMPN_DEV f4 sqrt4(f4 x) { return f4{sqrt_pos(x.x), sqrt_pos(x.y), sqrt_pos(x.z), sqrt_pos(x.w)}; }
// (q^2 + 1e-16)^(1/4)  (cliffordalgebra.py:148-149)
CSMPN_DEV f4 smooth_abs_sqrt4(f4 q) { return sqrt4(sqrt4(q * q + kSmooth)); }
CSMPN_DEV f4 splat(float v) { return f4{v, v, v, v}; }

template <int VAR>
CSMPN_DEV void tile_sync() {
    if constexpr (kVarBarrier<VAR>) {
        __syncthreads();
    } else {
        // one wave owns the tile: the LDS executes a wave's operations in issue order, so no
        // hardware wait is needed - only the compiler must not move LDS accesses across this
        // point (a wavefront-scope fence would also drain outstanding global loads/atomics).
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
    }
}

// ---------------------------------------------------------------------------------
// MFMA pieces

// LDS weight store of one block (VAR_WAVE): three dense arrays [g][O][IP], IP = input
// channels padded to 4 (padding zero-filled), so that a forward B fragment is ONE
// ds_read_b128 and a transposed one four ds_read_b32 - no pre-packed, per-lane duplicated
// fragments and no global-memory latency in front of the MFMAs.
// ... followed by the per-channel parameters (b1, silu a/b, sigmoid(norm a), left bias, layer
// norm a, path weights), so that a tile reads ALL its parameters from LDS.
struct WOff { int W1, WR, WL, b1, sa, sb, sg, bL, la, w, total; };
CSMPN_DEV WOff wstore_offsets(int O, int CPi, int CPo, int G, int P, bool w1_sub) {
    WOff w;
    int o = 0;
    w.W1 = o; o += (w1_sub ? G : 1) * O * CPi;
    w.WR = o; o += G * O * CPo;
    w.WL = o; o += G * O * CPo;
    w.b1 = o; o += O;
    w.sa = o; o += O * G;
    w.sb = o; o += O * G;
    w.sg = o; o += O * G;
    w.bL = o; o += O;
    w.la = o; o += O;
    w.w = o; o += O * P;
    w.total = o;
    return w;
}

// where the B fragments of one linear come from
struct WSrc {
    const f4* frags;   // packed global fragments [nt][kk][hp][g][64] (barrier variants)
    const float* w;    // LDS array [g][O][IP] (VAR_WAVE)
    int O, IP;         // rows and row stride of w
    int grades;        // 0: one matrix for all grades (MVLinear subspaces=False)
};

// acc[d][v] += sum_in  T[in][d][row] * W[out][in][grade(d)]        (TRANS = false)
// acc[d][v] += sum_out T[out][d][row] * W[out][in][grade(d)]       (TRANS = true, acc over in)
// tile: [channel][D][R] (channel stride CS); CP = valid channels (multiple of 4); nt = this
// wave's N tile; KK = k-blocks of 16 contracted channels.
template <class ALG, int H, bool WLDS, bool TRANS, bool SPEC = true>
CSMPN_DEV void linear_from_tile(f4 (&acc)[ALG::D], const float* tile, int CP, int KK, const WSrc& ws, int nt,
                                const Geo<ALG, H>& ge) {
    using GE = Geo<ALG, H>;
    constexpr int G = ALG::G, R = GE::R, CS = GE::CS, NW = GE::NW;
    const f4* fbase = ws.frags + (size_t)nt * KK * (H * G * 64) + ge.lane;
    const int ncol = NW * nt + ge.cn;   // this lane's output channel of the product
    // k-slot (q, v) of k-block kk is contracted channel 16*kk + 4*v + q: MFMA number v covers
    // the four CONSECUTIVE channels 4v..4v+3, so a block of 8 valid channels (the 8-channel
    // layers) issues 2 MFMAs per blade, not 4 with half of every k dimension on padding.
    // All LDS reads below are UNCONDITIONAL from clamped (always valid, finite) addresses:
    // a k-slot beyond the tile's channels contributes nothing because its B fragment (the
    // weight) is zero there. Predicated reads would compile to an exec-mask branch plus a
    // full s_waitcnt in front of every single MFMA.
    for (int kk = 0; kk < KK; ++kk) {
        const int c0 = 16 * kk + ge.q;
        const int left = CP - 16 * kk;               // valid channels of this k-block (multiple of 4)
        const int nv = left >= 16 ? 4 : left / 4;    // MFMAs per blade
        const float* ap = tile + (c0 < CP ? c0 : 0) * CS + ge.n;
        // one straight-line body per MFMA count (branches inside it would stop the scheduler
        // from batching the LDS reads in front of the MFMAs)
        auto body = [&](auto NVc) {
            constexpr int NV = decltype(NVc)::value;
#pragma unroll
            for (int hp = 0; hp < H; ++hp) {
                static_for<0, G>([&](auto g) {
                    f4 b;
                    if constexpr (!WLDS) {
                        b = fbase[(size_t)kk * (H * G * 64) + (hp * G + g) * 64];
                    } else {
                        const int gi = ws.grades ? int(g) : 0;
                        const bool half_ok = H == 1 || ge.h == hp;
                        if constexpr (!TRANS) {
                            const bool okc = half_ok && ncol < ws.O;
                            const float* wp = ws.w + (gi * ws.O + (ncol < ws.O ? ncol : 0)) * ws.IP;
#pragma unroll
                            for (int v = 0; v < NV; ++v) {
                                const int i = c0 + 4 * v;
                                b[v] = wp[i < ws.IP ? i : 0] * ((okc && i < ws.IP) ? 1.0f : 0.0f);
                            }
                        } else {
                            const bool okc = half_ok && ncol < ws.IP;
                            const float* wp = ws.w + gi * ws.O * ws.IP + (ncol < ws.IP ? ncol : 0);
#pragma unroll
                            for (int v = 0; v < NV; ++v) {
                                const int o = c0 + 4 * v;
                                b[v] = wp[(o < ws.O ? o : 0) * ws.IP] * ((okc && o < ws.O) ? 1.0f : 0.0f);
                            }
                        }
                    }
                    constexpr int d0 = ALG::gstart(g), nd = ALG::gsize(g);
                    float a[NV][nd];
#pragma unroll
                    for (int v = 0; v < NV; ++v)
#pragma unroll
                        for (int t = 0; t < nd; ++t) a[v][t] = ap[(4 * v < left ? 4 * v : 0) * CS + (d0 + t) * R + 16 * hp];
#pragma unroll
                    for (int v = 0; v < NV; ++v)
#pragma unroll
                        for (int t = 0; t < nd; ++t) acc[d0 + t] = mfma16(a[v][t], b[v], acc[d0 + t]);
                });
            }
        };
        // SPEC (backward kernels, one wave per SIMD): two straight-line bodies; the 4-MFMA one
        // also serves 1 and 3 (clamped reads times zero weights). Otherwise (forward kernels,
        // tight on registers and code size) one body with wave-uniform guards per MFMA group.
        if constexpr (SPEC) {
            if (nv == 2) body(IC<2>{});
            else body(IC<4>{});
        } else {
#pragma unroll
            for (int hp = 0; hp < H; ++hp) {
                static_for<0, G>([&](auto g) {
                    f4 b;
                    if constexpr (!WLDS) {
                        b = fbase[(size_t)kk * (H * G * 64) + (hp * G + g) * 64];
                    } else {
                        const int gi = ws.grades ? int(g) : 0;
                        const bool half_ok = H == 1 || ge.h == hp;
                        if constexpr (!TRANS) {
                            const bool okc = half_ok && ncol < ws.O;
                            const float* wp = ws.w + (gi * ws.O + (ncol < ws.O ? ncol : 0)) * ws.IP;
#pragma unroll
                            for (int v = 0; v < 4; ++v) {
                                const int i = c0 + 4 * v;
                                b[v] = wp[i < ws.IP ? i : 0] * ((okc && i < ws.IP) ? 1.0f : 0.0f);
                            }
                        } else {
                            const bool okc = half_ok && ncol < ws.IP;
                            const float* wp = ws.w + gi * ws.O * ws.IP + (ncol < ws.IP ? ncol : 0);
#pragma unroll
                            for (int v = 0; v < 4; ++v) {
                                const int o = c0 + 4 * v;
                                b[v] = wp[(o < ws.O ? o : 0) * ws.IP] * ((okc && o < ws.O) ? 1.0f : 0.0f);
                            }
                        }
                    }
                    constexpr int d0 = ALG::gstart(g), nd = ALG::gsize(g);
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        if (v < nv) {   // wave-uniform
                            float a[nd];
#pragma unroll
                            for (int t = 0; t < nd; ++t) a[t] = ap[4 * v * CS + (d0 + t) * R + 16 * hp];
#pragma unroll
                            for (int t = 0; t < nd; ++t) acc[d0 + t] = mfma16(a[t], b[v], acc[d0 + t]);
                        }
                    }
                });
            }
        }
    }
}

// Weight gradient tiles of one linear: for every input-channel tile it and grade g
//   gW[o][cin][g] += sum_{rows, d in g} Gr[row][d][o] * T[cin][d][row]
// A operand = lane-layout gradient (registers), B operand = input-side LDS tile (one
// ds_read_b128 = the lane's 4 rows = 4 k-steps). MIRROR: dst is the LDS mirror laid out
// [g][O][I]; otherwise the global reference layout [O][I][G] (or [O][I]).
template <class ALG, int H, bool MIRROR>
CSMPN_DEV void weight_grad(const f4 (&gr)[ALG::D], const float* tile, int CP, int I, int O, int NTin, int mt,
                           const Geo<ALG, H>& ge, float* dstp, bool has_grades) {
    using GE = Geo<ALG, H>;
    constexpr int G = ALG::G, R = GE::R, CS = GE::CS, NW = GE::NW;
    // output element (i = 4q + v', j = n) of the MFMA: i -> (row half, out channel)
    const int hi = H == 1 ? 0 : (ge.q >> 1);
    const int ob = NW * mt + (H == 1 ? 4 * ge.q : 4 * (ge.q & 1));
    for (int it = 0; it < NTin; ++it) {
        const int cin = NW * it + ge.cn;
        // clamped, unconditional read: columns cin >= CP produce garbage outputs that the
        // bounds check below discards
        const float* bp = tile + (cin < CP ? cin : 0) * CS + ge.r0;
        f4 accg[G];
        static_for<0, G>([&](auto g) {
            f4 acc = splat(0.f);
            constexpr int d0 = ALG::gstart(g), nd = ALG::gsize(g);
#pragma unroll
            for (int t = 0; t < nd; ++t) {
                const f4 b = *reinterpret_cast<const f4*>(bp + (d0 + t) * R);
#pragma unroll
                for (int v = 0; v < 4; ++v) acc = mfma16(gr[d0 + t][v], b[v], acc);
            }
            accg[g] = acc;
        });
        // one predicate for the whole store: lanes outside the tile skip it, inside it every
        // (o, cin) pair is owned by exactly this lane (plus its twin in the other row half)
        if (cin < I && hi == ge.h && ob < O) {
            const bool full = ob + 3 < O;
            if (has_grades) {
                static_for<0, G>([&](auto g) {
                    float* p = MIRROR ? dstp + (g * O + ob) * I + cin : dstp + (ob * I + cin) * G + g;
                    const int so = MIRROR ? I : I * G;   // stride between consecutive out channels
                    atomicAdd(p, accg[g][0]);
                    if (full) {
                        atomicAdd(p + so, accg[g][1]); atomicAdd(p + 2 * so, accg[g][2]); atomicAdd(p + 3 * so, accg[g][3]);
                    } else {
                        if (ob + 1 < O) atomicAdd(p + so, accg[g][1]);
                        if (ob + 2 < O) atomicAdd(p + 2 * so, accg[g][2]);
                    }
                });
            } else {
                f4 tot = accg[0];
                static_for<1, G>([&](auto g) { tot += accg[g]; });
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    if (ob + v < O) atomicAdd(dstp + (ob + v) * I + cin, tot[v]);
            }
        }
    }
}

// write a lane-layout tensor into an LDS tile [channel][D][R]: one b128 per blade
template <class ALG, int H>
CSMPN_DEV void store_tile(const f4 (&t)[ALG::D], float* tile, int CP, int mt, const Geo<ALG, H>& ge) {
    using GE = Geo<ALG, H>;
    constexpr int D = ALG::D, R = GE::R, CS = GE::CS, NW = GE::NW;
    const int c = NW * mt + ge.cn;
    if (c < CP) {
        float* p = tile + c * CS + ge.r0;
#pragma unroll
        for (int d = 0; d < D; ++d) *reinterpret_cast<f4*>(p + d * R) = t[d];
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

// same, from the LDS parameter store (VAR_WAVE): no global-memory latency per tile
template <class ALG>
CSMPN_DEV LaneParams<ALG> load_lane_params_lds(const DevBlock& B, const float* ws, const WOff& wo, int c) {
    constexpr int G = ALG::G;
    LaneParams<ALG> p;
    p.cvalid = c < B.O;
    const int cc = p.cvalid ? c : 0;
    const float m = p.cvalid ? 1.0f : 0.0f;
    p.b1 = ws[wo.b1 + cc] * m;
    p.bL = ws[wo.bL + cc] * m;
    p.la = ws[wo.la + cc] * m;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        p.sa[g] = ws[wo.sa + cc * G + g] * m;
        p.sb[g] = ws[wo.sb + cc * G + g] * m;
        p.sg[g] = ws[wo.sg + cc * G + g] * m;
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

// backward of weighted_gp: gz, gr accumulate; gw[p] (summed over the lane's 4 rows) is
// returned per lane. The operands z = gate*y and r = R*invden are rebuilt per path from
// the kept forward state instead of being held in registers.
template <class ALG>
CSMPN_DEV void weighted_gp_bwd(const f4 (&ggp)[ALG::D], const f4 (&y)[ALG::D], const f4 (&gate)[ALG::G],
                               const f4 (&R)[ALG::D], const f4 (&invden)[ALG::G], const float* wrow,
                               f4 (&gz)[ALG::D], f4 (&gr)[ALG::D], float (&gw)[ALG::P]) {
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
        gw[p] = hsum(gwv);
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
// red: scratch [MT][16] floats for cross-wave LayerNorm sums (barrier variants only).
template <class ALG, int H, int VAR, bool SPEC = true>
CSMPN_DEV void block_forward(const DevBlock& B, const LaneParams<ALG>& lp, const float* xin, float* zbuf,
                             float* red, const float* wstore, int MT, int mt, const Geo<ALG, H>& ge,
                             FwdState<ALG>& S, f4 (&out)[ALG::D], bool alias_in_z = false) {
    using GE = Geo<ALG, H>;
    static_assert(H == 1 || !kVarBarrier<VAR>, "multi-wave row tiles use H = 1");
    constexpr int D = ALG::D, G = ALG::G, NW = GE::NW;
    const int c = NW * mt + ge.cn;
    const bool tile_active = NW * mt < B.O;
    constexpr bool WLDS = VAR == VAR_WAVE;
    const WOff wo = wstore_offsets(B.O, B.CPi, B.CPo, G, ALG::P, B.w1_sub != 0);
    const WSrc sW1{B.pfW1, wstore + B.lds_woff + wo.W1, B.O, B.CPi, B.w1_sub};
    const WSrc sWR{B.pfWR, wstore + B.lds_woff + wo.WR, B.O, B.CPo, 1};
    const WSrc sWL{B.pfWL, wstore + B.lds_woff + wo.WL, B.O, B.CPo, 1};

    // 1. MVLinear (cegnn_utils.py:326-338)
#pragma unroll
    for (int d = 0; d < D; ++d) S.y[d] = splat(0.f);
    if (tile_active) linear_from_tile<ALG, H, WLDS, false, SPEC>(S.y, xin, B.CPi, B.KKi, sW1, mt, ge);
    S.y[0] += lp.b1;
    ge.stamp(3);

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
    // single-wave forward tiles keep z in the buffer the MVLinear just read (host layout):
    // nothing may move the stores above those reads
    if (VAR == VAR_WAVE || alias_in_z) tile_sync<VAR>();
    store_tile<ALG, H>(z, zbuf, B.CPo, mt, ge);
    tile_sync<VAR>();
    ge.stamp(4);

    CSMPN_PHASE();
    // 3. linear_right / linear_left (cegnn_utils.py:143-148)
    f4 L[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { S.R[d] = splat(0.f); L[d] = splat(0.f); }
    if (tile_active) {
        linear_from_tile<ALG, H, WLDS, false, SPEC>(S.R, zbuf, B.CPo, B.KKo, sWR, mt, ge);
        linear_from_tile<ALG, H, WLDS, false, SPEC>(L, zbuf, B.CPo, B.KKo, sWL, mt, ge);
    }
    L[0] += lp.bL;
    ge.stamp(5);

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

    ge.stamp(6);
    CSMPN_PHASE();
    // 5. steerable geometric product + first-order term (cegnn_utils.py:126-152)
    if (lp.cvalid) weighted_gp<ALG>(L, z, r, (WLDS ? wstore + B.lds_woff + wo.w : B.w) + (size_t)c * ALG::P);
#pragma unroll
    for (int d = 0; d < D; ++d) S.s[d] = L[d] * kInvSqrt2;

    ge.stamp(7);
    CSMPN_PHASE();
    // 6. MVLayerNorm (cegnn_utils.py:93-96): mean over the channels of the row
    f4 qs = splat(0.f);
    static_for<0, D>([&](auto dd) {
        constexpr int d = decltype(dd)::value;
        qs += qsf<ALG, d> * S.s[d] * S.s[d];
    });
    S.qs = qs;
    S.nl = smooth_abs_sqrt4(qs);
    f4 tot = chan_sum4<H>(lp.cvalid ? S.nl : splat(0.f));
    if constexpr (kVarBarrier<VAR>) {
        if (ge.n == 0) *reinterpret_cast<f4*>(red + mt * 16 + 4 * ge.q) = tot;
        __syncthreads();
        tot = splat(0.f);
        for (int m = 0; m < MT; ++m) tot += *reinterpret_cast<const f4*>(red + m * 16 + 4 * ge.q);
        __syncthreads();
    }
    S.invMn = rcp4(tot * (1.0f / float(B.O)) + kEps);
#pragma unroll
    for (int d = 0; d < D; ++d) out[d] = lp.la * S.s[d] * S.invMn;
    ge.stamp(8);
}

// ---------------------------------------------------------------------------------
// block backward: given the forward state S of this tile and gout (lane layout),
// accumulate all parameter gradients and leave d/d(MVLinear output) in gy (lane
// layout) AND in the LDS tile gbuf (so the caller can run the transposed MVLinear).
template <class ALG, int H, int VAR>
CSMPN_DEV void block_backward(const DevBlock& B, const LaneParams<ALG>& lp, const FwdState<ALG>& S,
                              const f4 (&gout)[ALG::D], const float* xin, const float* zbuf, float* gbuf,
                              float* red, float* mirror, const float* wstore, int MT, int mt,
                              const Geo<ALG, H>& ge, f4 (&gy)[ALG::D], bool defer_w1 = false, size_t goff = 0) {
    // goff: deterministic mode of the general kernels - float offset of this WORKGROUP's private copy of the gradient
    // accumulators (the host points g* at slice 0 of a zeroed per-workgroup region); 0 otherwise
    using GE = Geo<ALG, H>;
    constexpr int D = ALG::D, G = ALG::G, P = ALG::P, NW = GE::NW;
    const int c = NW * mt + ge.cn;
    const bool tile_active = NW * mt < B.O;
    const bool cv = lp.cvalid;
    const int cc = cv ? c : 0;
    constexpr bool WLDS = VAR == VAR_WAVE;
    const WOff wo = wstore_offsets(B.O, B.CPi, B.CPo, G, ALG::P, B.w1_sub != 0);
    const WSrc sWRt{B.pbWR, wstore + B.lds_woff + wo.WR, B.O, B.CPo, 1};
    const WSrc sWLt{B.pbWL, wstore + B.lds_woff + wo.WL, B.O, B.CPo, 1};
    const MirrorOff mo = mirror_offsets(B.I, B.O, G, P, B.w1_sub != 0);
    float* mir = mirror + B.lds_goff;
    // gradient destinations: LDS mirror (flushed once per workgroup) or, in the variants
    // without mirror, the global reference-layout accumulators directly
    constexpr bool in_lds = kVarMirror<VAR>;
    float *d_b1, *d_sa, *d_sb, *d_w, *d_an, *d_bL, *d_la, *d_W1, *d_WR, *d_WL;
    if constexpr (in_lds) {
        d_b1 = mir + mo.b1; d_sa = mir + mo.sa; d_sb = mir + mo.sb; d_w = mir + mo.w; d_an = mir + mo.an;
        d_bL = mir + mo.bL; d_la = mir + mo.la; d_W1 = mir + mo.W1; d_WR = mir + mo.WR; d_WL = mir + mo.WL;
    } else {
        d_b1 = B.gb1 + goff; d_sa = B.gsa + goff; d_sb = B.gsb + goff; d_w = B.gw + goff; d_an = B.gan + goff;
        d_bL = B.gbL + goff; d_la = B.gla + goff; d_W1 = B.gW1 + goff; d_WR = B.gWR + goff; d_WL = B.gWL + goff;
    }
    // per-lane partial sums (over the lane's 4 rows) of the small-parameter gradients; all
    // lanes of a channel add them to the accumulators in one predicated region at the end
    float p_la, p_bL, p_b1, p_an[G], p_sa[G], p_sb[G], p_w[P];

    // ---- MVLayerNorm backward
    f4 dot = splat(0.f);
#pragma unroll
    for (int d = 0; d < D; ++d) dot += gout[d] * S.s[d];
    p_la = hsum(dot * S.invMn);
    f4 gMn = chan_sum4<H>(-(lp.la * dot) * S.invMn * S.invMn);
    if constexpr (kVarBarrier<VAR>) {
        if (ge.n == 0) *reinterpret_cast<f4*>(red + mt * 16 + 4 * ge.q) = gMn;
        __syncthreads();
        gMn = splat(0.f);
        for (int m = 0; m < MT; ++m) gMn += *reinterpret_cast<const f4*>(red + m * 16 + 4 * ge.q);
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
    p_bL = hsum(ggp[0]);
    ge.stamp(9);

    CSMPN_PHASE();
    // ---- d/dz from linear_left: gz = GL . WL^T ; gWL += GL (x) Z
    store_tile<ALG, H>(ggp, gbuf, B.CPo, mt, ge);
    tile_sync<VAR>();
    f4 gz[D];
#pragma unroll
    for (int d = 0; d < D; ++d) gz[d] = splat(0.f);
    if (tile_active) {
        linear_from_tile<ALG, H, WLDS, true>(gz, gbuf, B.CPo, B.KKo, sWLt, mt, ge);
        weight_grad<ALG, H, in_lds>(ggp, zbuf, B.CPo, B.O, B.O, B.NTo, mt, ge, d_WL, true);
    }

    ge.stamp(10);
    CSMPN_PHASE();
    // ---- geometric product backward
    f4 gr[D];
#pragma unroll
    for (int d = 0; d < D; ++d) gr[d] = splat(0.f);
#pragma unroll
    for (int p = 0; p < P; ++p) p_w[p] = 0.f;
    if (tile_active)
        weighted_gp_bwd<ALG>(ggp, S.y, S.gate, S.R, S.invden, (WLDS ? wstore + B.lds_woff + wo.w : B.w) + (size_t)cc * P,
                             gz, gr, p_w);

    ge.stamp(11);
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
        p_an[g] = hsum(gden * (nu - 1.0f)) * lp.sg[g] * (1.0f - lp.sg[g]);
        const f4 inu = rcp4(nu);
        const f4 gq = (gden * lp.sg[g]) * (0.5f * qR) * (inu * inu * inu);
        static_for<0, nd>([&](auto t) {
            constexpr int d = d0 + decltype(t)::value;
            gR[d] = cv ? gr[d] * S.invden[g] + gq * (2.0f * qsf<ALG, d>) * S.R[d] : splat(0.f);
        });
    });
    ge.stamp(12);
    tile_sync<VAR>();   // all reads of gbuf (GL) done
    store_tile<ALG, H>(gR, gbuf, B.CPo, mt, ge);
    tile_sync<VAR>();
    if (tile_active) {
        linear_from_tile<ALG, H, WLDS, true>(gz, gbuf, B.CPo, B.KKo, sWRt, mt, ge);
        weight_grad<ALG, H, in_lds>(gR, zbuf, B.CPo, B.O, B.O, B.NTo, mt, ge, d_WR, true);
    }

    ge.stamp(13);
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
        p_sa[g] = hsum(gpre * u);
        p_sb[g] = hsum(gpre);
        const f4 gu = gpre * lp.sa[g];
        static_for<0, nd>([&](auto t) {
            constexpr int d = d0 + decltype(t)::value;
            f4 v = gz[d] * S.gate[g];
            if constexpr (g == 0) v += gu;
            else v += gu * (2.0f * qsf<ALG, d>) * S.y[d];
            gy[d] = cv ? v : splat(0.f);
        });
    });
    p_b1 = hsum(gy[0]);
    ge.stamp(14);

    CSMPN_PHASE();
    // ---- small-parameter gradients: reduce the per-lane partials over the lanes of the
    // channel with permlane swaps (no LDS, no bank conflicts), then ONE lane per channel adds
    // them in a single predicated region (conflict-free atomics)
    p_la = channel_rows_sum<H>(p_la);
    p_bL = channel_rows_sum<H>(p_bL);
    p_b1 = channel_rows_sum<H>(p_b1);
#pragma unroll
    for (int g = 0; g < G; ++g) {
        p_an[g] = channel_rows_sum<H>(p_an[g]);
        p_sa[g] = channel_rows_sum<H>(p_sa[g]);
        p_sb[g] = channel_rows_sum<H>(p_sb[g]);
    }
#pragma unroll
    for (int p = 0; p < P; ++p) p_w[p] = channel_rows_sum<H>(p_w[p]);
    if (cv && ge.q == 0 && ge.h == 0) {
        atomicAdd(d_la + c, p_la);
        atomicAdd(d_bL + c, p_bL);
        if (B.has_b1) atomicAdd(d_b1 + c, p_b1);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            atomicAdd(d_an + c * G + g, p_an[g]);
            atomicAdd(d_sa + c * G + g, p_sa[g]);
            atomicAdd(d_sb + c * G + g, p_sb[g]);
        }
        if (tile_active) {
#pragma unroll
            for (int p = 0; p < P; ++p) atomicAdd(d_w + (size_t)c * P + p, p_w[p]);
        }
    }

    ge.stamp(15);
    // ---- MVLinear weight gradient; gy tile to LDS for the transposed MVLinear
    tile_sync<VAR>();   // all reads of gbuf (GR) done
    store_tile<ALG, H>(gy, gbuf, B.CPo, mt, ge);
    // defer_w1: the input tile is gone (its buffer holds z): the caller stages it again and runs
    // block_w1_grad itself
    if (tile_active && !defer_w1) weight_grad<ALG, H, in_lds>(gy, xin, B.CPi, B.I, B.O, B.NTi, mt, ge, d_W1, B.w1_sub != 0);
    tile_sync<VAR>();
    ge.stamp(16);
}

// the MVLinear weight gradient of block_backward, for callers that deferred it (defer_w1)
template <class ALG, int H, int VAR>
CSMPN_DEV void block_w1_grad(const DevBlock& B, const f4 (&gy)[ALG::D], const float* xin, float* mirror, int mt,
                             const Geo<ALG, H>& ge, size_t goff = 0) {
    constexpr int G = ALG::G, NW = Geo<ALG, H>::NW;
    constexpr bool in_lds = kVarMirror<VAR>;
    const MirrorOff mo = mirror_offsets(B.I, B.O, G, ALG::P, B.w1_sub != 0);
    float* d_W1 = in_lds ? mirror + B.lds_goff + mo.W1 : B.gW1 + goff;
    if (NW * mt < B.O) weight_grad<ALG, H, in_lds>(gy, xin, B.CPi, B.I, B.O, B.NTi, mt, ge, d_W1, B.w1_sub != 0);
}

// flush one block's LDS gradient mirror into the global reference-layout accumulators
template <class ALG>
__device__ void flush_mirror(const DevBlock& B, const float* mirror, int tid, int nthreads, size_t goff = 0) {
    constexpr int G = ALG::G, P = ALG::P;
    const MirrorOff mo = mirror_offsets(B.I, B.O, G, P, B.w1_sub != 0);
    const float* mir = mirror + B.lds_goff;
    const int I = B.I, O = B.O;
    const int nW1 = (B.w1_sub ? G : 1) * O * I;
    for (int e = tid; e < nW1; e += nthreads) {
        // mirror [g][o][i] -> reference [o][i][g]
        const int g = e / (O * I), rem = e % (O * I);
        const float v = mir[mo.W1 + e];
        if (v != 0.f) atomicAdd(B.gW1 + goff + (B.w1_sub ? rem * G + g : rem), v);
    }
    for (int e = tid; e < G * O * O; e += nthreads) {
        const int g = e / (O * O), rem = e % (O * O);
        const float vr = mir[mo.WR + e], vl = mir[mo.WL + e];
        if (vr != 0.f) atomicAdd(B.gWR + goff + rem * G + g, vr);
        if (vl != 0.f) atomicAdd(B.gWL + goff + rem * G + g, vl);
    }
    for (int e = tid; e < O; e += nthreads) {
        if (B.has_b1) atomicAdd(B.gb1 + goff + e, mir[mo.b1 + e]);
        atomicAdd(B.gbL + goff + e, mir[mo.bL + e]);
        atomicAdd(B.gla + goff + e, mir[mo.la + e]);
    }
    for (int e = tid; e < O * G; e += nthreads) {
        atomicAdd(B.gsa + goff + e, mir[mo.sa + e]);
        atomicAdd(B.gsb + goff + e, mir[mo.sb + e]);
        atomicAdd(B.gan + goff + e, mir[mo.an + e]);
    }
    for (int e = tid; e < O * P; e += nthreads) atomicAdd(B.gw + goff + e, mir[mo.w + e]);
}

}  // namespace csmpn
