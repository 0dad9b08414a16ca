// (row, channel)-per-lane ("cl") CEMLP kernels for Cl(3,0) layers whose blocks are all C = 8 channels wide.
//
// Same arithmetic as cemlp_device.hpp / cemlp_rl.hpp (csmpn/models/cegnn_utils.py:34-155,287-338; SURVEY.md
// Appendix A); the mapping onto the hardware is chosen for OCCUPANCY: the row-per-lane-group kernels (cemlp_rl.hpp)
// keep 4 channels x 8 blades of every live tensor in a lane (32 registers per tensor, ~500 in the backward, one
// wave per SIMD, 45 % issue-active). Here
//
//  * lane = (row r, channel c): a tensor is `float t[8]` (the 8 blades). A wave covers 64 / C rows (8 rows at
//    8 channels); two rows are interleaved inside a 16-lane DPP row (lane = r0 | c << 1 | r_hi << 4) so that
//    `row_ror:2k` rotates the channels of a row. Forward ~100 registers (4 waves per SIMD), one block of the
//    backward ~200 including its persistent gradient sums (2 waves per SIMD).
//  * dense channel mixing  out[r,o,d] = sum_c W[o][c][grade d] x[r,c,d]  is C rotations x 8 blades of
//    `v_fmac_f32_dpp` (the rotated operand costs no instruction of its own): lane (r,o) multiplies the value of
//    lane (r, o + k) with W[o][o + k][g], read as ONE ds_read_b128 (4 grades) per rotation from a rotation-ordered
//    LDS table [k][o] built once per workgroup from the reference layout [o][c][g]. Measured (tools/dpp_probe.hip,
//    profiles/r03_dpp_probe.log): a DPP VALU instruction issues at HALF the plain VALU rate on gfx950 (4.2 against
//    2.4 cycles), so the mixing runs at ~15 MAC/cycle/SIMD where the 4x4x1 MFMA of cemlp_rl.hpp reaches ~26 - the
//    price of one channel per lane; it is paid back by occupancy, by the absence of any LDS round trip between
//    phases and by the weight-gradient form below. Input segments narrower than C (the attribute channels) are
//    replicated with their power-of-two period, so a 6-channel segment takes 8 and a 3-channel segment 4 rotations.
//  * weight gradients take the lane layout AS IT IS: v_mfma_f32_16x16x4_f32 with A = this lane's gradient blade and
//    B = this lane's input blade contracts over the lane bits 4-5 = r_hi; D[(o,r0)][(c,r0')] is the wanted sum on
//    r0 = r0' (half of the tile, the other half is discarded). One MFMA per (matrix, blade), no LDS transposition,
//    accumulators (4 grades x f4 per matrix) persistent over the tile loop.
//  * per-channel parameter gradients are lane-private running sums in LDS (the lane's channel is fixed; ClSums);
//    the sum over the rows of a wave happens ONCE per block, at its end.
//  * the backward is ONE launch for all blocks, run block by block (last block first): 48-80 accumulator registers +
//    35 running sums are alive per block instead of both blocks' at once; d/d(block input) rows travel through a
//    [rows, C, D] region behind the saved block inputs (through L2: a wave reads in block k - 1 what it wrote itself
//    in block k) or, with one tile per wave, stay in registers. Every workgroup writes one slice of partial sums per
//    block; cl_reduce_kernel adds the slices of all blocks in a fixed order (no atomics on parameters: bit-reproducible).
//  * gathers: a lane loads its own 32 bytes of a row (2 x 16 bytes); scatters go through a per-wave LDS tile so that
//    one atomic instruction covers whole 256-byte rows, equal consecutive targets summed first.
#pragma once
#include "cemlp_device.hpp"

namespace csmpn {

constexpr int kClWaves = 4;          // waves per workgroup
constexpr int kClSliceCap = 512;     // workgroups of a backward launch = slices of partial sums per block
constexpr int kClParStride = 36;     // floats per channel in the per-channel parameter table

// Ordering point between LDS accesses of DIFFERENT lanes of one wave (a lane stores into the per-wave staging tile, another
// lane reads those bytes; and the next tile's stores behind those reads): the LDS executes a wave's operations in issue
// order, so no wait is needed - but the compiler sees each lane's own stores and loads as disjoint addresses and may move
// them across each other (cemlp_cm.hpp met exactly that). A compiler barrier + the wave-barrier intrinsic (a scheduling
// barrier, no instruction) state the contract at every such hand-over.
#define CL_LDS_ORDER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); } while (0)
CSMPN_DEV f4 cl_ld4(const float* p) { return *reinterpret_cast<const f4*>(p); }
CSMPN_DEV void cl_st4(float* p, f4 v) { *reinterpret_cast<f4*>(p) = v; }

// diagnostic build only (-DCSMPN_STAMPS, never shipped, never timed): shader-clock cycles per phase, summed per wave.
// The scheduling barriers at a stamp forbid overlaps across phases (memory operations stay in flight: a phase is
// charged with the waits it executes): the output is a breakdown, not a duration.
struct ClStamp {
#ifdef CSMPN_STAMPS
    static constexpr int kSlots = 24;
    unsigned long long t0, acc[kSlots];
    CSMPN_DEV explicit ClStamp(int) {
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
    CSMPN_DEV explicit ClStamp(int) {}
    CSMPN_DEV void operator()(int) {}
    CSMPN_DEV void flush(unsigned long long*, int) {}
#endif
};

// ---------------------------------------------------------------------------------
// lane <-> (row, channel)
template <int C>
struct ClMap {
    static_assert(C == 8 || C == 16, "8 or 16 channels");
    static constexpr int RPW = 64 / C;             // rows per wave
    static constexpr int ROTL = C == 8 ? 2 : 1;    // lanes per channel step inside a DPP row
    static CSMPN_DEV int chan(int lane) { return C == 8 ? (lane >> 1) & 7 : lane & 15; }
    static CSMPN_DEV int row(int lane) { return C == 8 ? (lane & 1) | ((lane >> 4) << 1) : lane >> 4; }
    static constexpr int lane_of_row(int r) { return C == 8 ? ((r >> 1) << 4) | (r & 1) : r << 4; }   // its channel-0 lane
};

template <int CTRL>
CSMPN_DEV int cl_dpp_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, true); }

// sum over the C lanes that hold the channels of one row; result in every lane
template <int C>
CSMPN_DEV float cl_chan_sum(float v) {
    if constexpr (C == 16) v += dpp_mov<0x121>(v);
    v += dpp_mov<0x122>(v);
    v += dpp_mov<0x124>(v);
    v += dpp_mov<0x128>(v);
    return v;
}

// acc[d] += w[grade d] * (x[d] of the lane ROT lanes away in the DPP row), the 8 blades of Cl(3,0): grades 0 1 1 1 2 2 2 3.
// Inline asm: hipcc folds a DPP move into v_add / v_mul but not into v_fmac. The statement is opaque to the hazard
// recognizer (a VALU write of a VGPR needs two wait states before a DPP read of it), hence the s_nop in front: whatever
// the compiler places before the statement - the producers of x or a register copy of its own - is covered.
template <int ROT>
CSMPN_DEV void cl_fmac8(float (&acc)[8], const float (&x)[8], f4 w) {
    static_assert(ROT >= 0 && ROT < 16, "rotation inside a DPP row");
    if constexpr (ROT == 0) {
        acc[0] = __builtin_fmaf(x[0], w.x, acc[0]);
        acc[1] = __builtin_fmaf(x[1], w.y, acc[1]); acc[2] = __builtin_fmaf(x[2], w.y, acc[2]); acc[3] = __builtin_fmaf(x[3], w.y, acc[3]);
        acc[4] = __builtin_fmaf(x[4], w.z, acc[4]); acc[5] = __builtin_fmaf(x[5], w.z, acc[5]); acc[6] = __builtin_fmaf(x[6], w.z, acc[6]);
        acc[7] = __builtin_fmaf(x[7], w.w, acc[7]);
    } else {
        asm("s_nop 1\n\t"
            "v_fmac_f32_dpp %0, %8, %16 row_ror:%20 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %1, %9, %17 row_ror:%20 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %2, %10, %17 row_ror:%20 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %3, %11, %17 row_ror:%20 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %4, %12, %18 row_ror:%20 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %5, %13, %18 row_ror:%20 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %6, %14, %18 row_ror:%20 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %7, %15, %19 row_ror:%20 row_mask:0xf bank_mask:0xf"
            : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
            : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]),
              "v"(w.x), "v"(w.y), "v"(w.z), "v"(w.w), "n"(ROT));
    }
}

// acc += (table at float offset TOFF: NROT rotations x C lanes of f4) applied to x. ldsw = LDS + 4 * (this lane's channel).
// The weight vectors are read BATCH at a time, all of a batch in flight before the first is used (the empty volatile
// asm statement takes their results: left alone, the scheduler reads each vector right in front of its rotation and
// waits out an LDS round trip per rotation - 49 exposed round trips per backward tile at two waves per SIMD).
template <int NK>
CSMPN_DEV void cl_pin(f4 (&w)[NK]) {
    if constexpr (NK == 8) asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]));
    else if constexpr (NK == 4) asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
    else if constexpr (NK == 2) asm volatile("" : "+v"(w[0]), "+v"(w[1]));
    else static_assert(NK == 1 || NK == 2 || NK == 4 || NK == 8, "batch size");
}
template <int C, int NROT, int TOFF, int BATCH = 4>
CSMPN_DEV void cl_mix(float (&acc)[8], const float (&x)[8], const float* ldsw) {
    static_assert(NROT % BATCH == 0, "whole batches");
#ifdef CL_X_NOMIX   // timing experiment only (results wrong): rotation 0 alone
    cl_fmac8<0>(acc, x, cl_ld4(ldsw + TOFF));
    return;
#endif
    constexpr int NB = NROT / BATCH;
    f4 w[BATCH];
#pragma unroll
    for (int i = 0; i < BATCH; ++i) w[i] = cl_ld4(ldsw + (TOFF + 4 * C * i));
    cl_pin<BATCH>(w);
    static_for<0, NB>([&](auto b) {
        // the next batch travels while this one is used
        f4 wn[BATCH];
        if constexpr (b + 1 < NB) {
#pragma unroll
            for (int i = 0; i < BATCH; ++i) wn[i] = cl_ld4(ldsw + (TOFF + 4 * C * (BATCH * (b + 1) + i)));
        }
        static_for<0, BATCH>([&](auto i) { cl_fmac8<ClMap<C>::ROTL * (BATCH * b + i)>(acc, x, w[i]); });
        if constexpr (b + 1 < NB) {
            cl_pin<BATCH>(wn);
#pragma unroll
            for (int i = 0; i < BATCH; ++i) w[i] = wn[i];
        }
    });
}

// ---------------------------------------------------------------------------------
// compile-time description of one block's input passes and of its LDS tables (float offsets from the block's base)
constexpr int cl_pow2_ge(int w) { int p = 1; while (p < w) p <<= 1; return p; }

// block K of a CEMLP in mode MODE with NA attribute channels: K = 0 concatenates the mode's input segments (each one a
// "pass" of at most C channels), later blocks have one pass of C channels
template <int C, int MODE, int NA, int K, bool BWD>
struct ClTab {
    static constexpr int NSEG = MODE == MODE_EDGE ? 1 : (MODE == MODE_NODE ? 2 : 0);   // full-width segments in front of the attributes
    static_assert(MODE == MODE_EDGE || MODE == MODE_NODE, "edge or node program");
    static constexpr int NP = K > 0 ? 1 : NSEG + (NA > 0 ? 1 : 0);
    static constexpr int width(int p) { return (K > 0 || p < NSEG) ? C : NA; }
    static constexpr int period(int p) { return (K > 0 || p < NSEG) ? C : cl_pow2_ge(NA); }
    static constexpr int coff(int p) { return C * p; }
    static constexpr int I = K > 0 ? C : NSEG * C + NA;
    static_assert(NA <= C, "attribute segment wider than the layer");
    static constexpr int W1(int p) { int o = 0; for (int q = 0; q < p; ++q) o += 4 * C * period(q); return o; }
    static constexpr int WR = W1(NP), WL = WR + 4 * C * C;
    static constexpr int fwd_end = WL + 4 * C * C;
    static constexpr int W1T(int p) { return fwd_end + 4 * C * C * p; }
    static constexpr int WRT = W1T(NP), WLT = WRT + 4 * C * C;
    static constexpr int par = BWD ? WLT + 4 * C * C : fwd_end;
    static constexpr int total = par + C * kClParStride;
    // number of f4 table entries (everything in front of `par`)
    static constexpr int n_entries = par / 4;
};

// forward state of one block kept for its backward
// offset of (row, channel)'s first piece inside a state region (CSMPN_FLAG_SAVE_STATE, cemlp_device.hpp): a tile is 64 / C rows
// = 64 (row, channel) positions; piece e (4 blades) of position l lies at (tile * 2 + e) * 256 + 4 l floats
template <int C>
CSMPN_DEV size_t cl_state_off(long row, int c) {
    constexpr int RPT = 64 / C;
    return (size_t)(row / RPT) * 512 + 4 * ((int)(row % RPT) * C + c);
}
struct ClFwd {
    float y[8], gate[4], R[8], invden[4], s[8];
    float qs, nl, invMn;
};

CSMPN_DEV float cl_smooth_abs_sqrt(float q) { return sqrt_pos(sqrt_pos(__builtin_fmaf(q, q, kSmooth))); }

// ---------------------------------------------------------------------------------
// parameters -> LDS tables of one block (once per workgroup). dir = +1 / -1: the lane `row_ror:ROTL` reads from holds
// channel c + dir (probed on the device).
template <class ALG, int C, class TB, bool BWD>
__device__ void cl_stage_block(const DevBlock& B, float* base, int tid, int dir) {
    constexpr int G = ALG::G, P = ALG::P, NE = TB::n_entries, NIT = (NE + 64 * kClWaves - 1) / (64 * kClWaves);
    static_assert(G == 4, "the weight of one (o, c) pair is one 16-byte vector of 4 grades");
    const float *pW1 = B.W1, *pWR = B.WR, *pWL = B.WL;
    const float* src[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int e = tid + it * 64 * kClWaves;   // f4 entry
        src[it] = nullptr;
        // forward tables: entry (k, o) = W[o][first channel + ((o + dir k) mod period)]
        static_for<0, TB::NP>([&](auto p) {
            constexpr int e0 = TB::W1(p) / 4, PER = TB::period(p), n = PER * C;
            if (e >= e0 && e < e0 + n) {
                const int k = (e - e0) / C, o = (e - e0) % C, cs = (o + dir * k) & (PER - 1);
                if (cs < TB::width(p)) src[it] = pW1 + ((o * TB::I + TB::coff(p) + cs) * G);
            }
        });
        if (e >= TB::WR / 4 && e < TB::fwd_end / 4) {
            const int f = e - TB::WR / 4, which = f / (C * C), k = (f % (C * C)) / C, o = f % C, cs = (o + dir * k) & (C - 1);
            src[it] = (which == 0 ? pWR : pWL) + ((o * C + cs) * G);
        }
        if constexpr (BWD) {
            // transposed tables: entry (k, c) = W[(c + dir k) mod C][first channel + (c mod period)]
            static_for<0, TB::NP>([&](auto p) {
                constexpr int e0 = TB::W1T(p) / 4, PER = TB::period(p);
                if (e >= e0 && e < e0 + C * C) {
                    const int k = (e - e0) / C, c = (e - e0) % C, o = (c + dir * k) & (C - 1), cs = c & (PER - 1);
                    if (cs < TB::width(p)) src[it] = pW1 + ((o * TB::I + TB::coff(p) + cs) * G);
                }
            });
            if (e >= TB::WRT / 4 && e < TB::par / 4) {
                const int f = e - TB::WRT / 4, which = f / (C * C), k = (f % (C * C)) / C, c = f % C, o = (c + dir * k) & (C - 1);
                src[it] = (which == 0 ? pWR : pWL) + ((o * C + c) * G);
            }
        }
    }
    f4 v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) v[it] = src[it] ? cl_ld4(src[it]) : f4{0.f, 0.f, 0.f, 0.f};
    // per-channel parameters: [b1, bL, la, 0 | sa[4] | sb[4] | sigmoid(an)[4] | w[P]] per channel
    constexpr int NPAR = C * kClParStride, NITP = (NPAR + 64 * kClWaves - 1) / (64 * kClWaves);
    static_assert(16 + P <= kClParStride, "parameter stride");
    const float *pb1 = B.b1, *pbL = B.bL, *pla = B.la, *psa = B.sa, *psb = B.sb, *pan = B.an, *pw = B.w;
    const bool has_b1 = B.has_b1 != 0;
    const float* ps[NITP];
    bool sig[NITP];
#pragma unroll
    for (int it = 0; it < NITP; ++it) {
        const int e = tid + it * 64 * kClWaves;
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
        const int e = tid + it * 64 * kClWaves;
        if (e < NE) cl_st4(base + 4 * e, v[it]);
    }
#pragma unroll
    for (int it = 0; it < NITP; ++it) {
        const int e = tid + it * 64 * kClWaves;
        if (sig[it]) pv[it] = sigmoidf(pv[it]);
        if (e < NPAR) base[TB::par + e] = pv[it];
    }
}

// ---------------------------------------------------------------------------------
// sign-table geometric product with per-path weights, one channel:
//   out[j] += sum_{(i,k)->j} sign(i,k) * w[path(g_i, g_j, g_k)] * z[i] * r[k]
template <class ALG>
CSMPN_DEV void cl_weighted_gp(float (&out)[8], const float (&z)[8], const float (&r)[8], const float* wp) {
    constexpr int P = ALG::P;
    f4 wv[(P + 3) / 4];
#pragma unroll
    for (int q = 0; q < (P + 3) / 4; ++q) wv[q] = cl_ld4(wp + 4 * q);
    static_for<0, P>([&](auto p) {
        constexpr int gi = ALG::t.path_g[p][0], gj = ALG::t.path_g[p][1], gk = ALG::t.path_g[p][2];
        constexpr int i0 = ALG::gstart(gi), ni = ALG::gsize(gi);
        constexpr int j0 = ALG::gstart(gj), nj = ALG::gsize(gj);
        constexpr int k0 = ALG::gstart(gk), nk = ALG::gsize(gk);
        const float w = wv[p / 4][p % 4];
        float tmp[nj];
#pragma unroll
        for (int t = 0; t < nj; ++t) tmp[t] = 0.f;
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
        for (int t = 0; t < nj; ++t) out[j0 + t] = __builtin_fmaf(w, tmp[t], out[j0 + t]);
    });
}

// the part of a block forward behind its MVLinear: S.y holds the MVLinear output (without bias)
template <class ALG, int C, class TB, int BATCH>
CSMPN_DEV void cl_block_tail(const float* ldsw, const float* ldsp, ClFwd& S, float (&out)[8], ClStamp& stamp, int sid, bool have_s = false) {
    constexpr int D = ALG::D, G = ALG::G;
    static_assert(D == 8 && G == 4, "Cl(3,0)-shaped algebra");
    const f4 p0 = cl_ld4(ldsp + TB::par);   // b1, bL, la
    const f4 sa = cl_ld4(ldsp + (TB::par + 4)), sb = cl_ld4(ldsp + (TB::par + 8));
    S.y[0] += p0.x;
    // MVSiLU, invariant "mag2" (cegnn_utils.py:76-83)
    float z[D];
    static_for<0, G>([&](auto g) {
        constexpr int d0 = ALG::gstart(g), nd = ALG::gsize(g);
        float u;
        if constexpr (g == 0) {
            u = S.y[0];
        } else {
            u = 0.f;
            static_for<0, nd>([&](auto t) {
                constexpr int d = d0 + decltype(t)::value;
                u += qsf<ALG, d> * S.y[d] * S.y[d];
            });
        }
        S.gate[g] = sigmoidf(__builtin_fmaf(sa[int(g)], u, sb[int(g)]));
#pragma unroll
        for (int t = 0; t < nd; ++t) z[d0 + t] = S.gate[g] * S.y[d0 + t];
    });
    stamp(sid);
    // linear_right / linear_left (cegnn_utils.py:143-148)
    float L[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { S.R[d] = 0.f; L[d] = 0.f; }
    cl_mix<C, C, TB::WR, BATCH>(S.R, z, ldsw);
    // CSMPN_FLAG_SAVE_STATE: with s given (have_s: the forward saved it) the backward's recompute stops behind linear_right
    // and the normalisation denominators - no linear_left mix, no geometric product
    if (have_s) {
        const f4 sg = cl_ld4(ldsp + (TB::par + 12));
        static_for<0, G>([&](auto g) {
            constexpr int d0 = ALG::gstart(g), nd = ALG::gsize(g);
            float qq = 0.f;
            static_for<0, nd>([&](auto t) {
                constexpr int d = d0 + decltype(t)::value;
                qq += qsf<ALG, d> * S.R[d] * S.R[d];
            });
            const float m = __builtin_fmaf(sg[int(g)], cl_smooth_abs_sqrt(qq) - 1.0f, 1.0f);
            S.invden[g] = fast_rcp(m + kEps);
        });
        float qs = 0.f;
        static_for<0, D>([&](auto dd) {
            constexpr int d = decltype(dd)::value;
            qs += qsf<ALG, d> * S.s[d] * S.s[d];
        });
        S.qs = qs;
        S.nl = cl_smooth_abs_sqrt(qs);
        S.invMn = fast_rcp(__builtin_fmaf(cl_chan_sum<C>(S.nl), 1.0f / float(C), kEps));
        return;
    }
    cl_mix<C, C, TB::WL, BATCH>(L, z, ldsw);
    stamp(sid + 1);
    L[0] += p0.y;
    // NormalizationLayer on the right operand (cegnn_utils.py:42-51)
    const f4 sg = cl_ld4(ldsp + (TB::par + 12));
    float r[D];
    static_for<0, G>([&](auto g) {
        constexpr int d0 = ALG::gstart(g), nd = ALG::gsize(g);
        float qq = 0.f;
        static_for<0, nd>([&](auto t) {
            constexpr int d = d0 + decltype(t)::value;
            qq += qsf<ALG, d> * S.R[d] * S.R[d];
        });
        const float m = __builtin_fmaf(sg[int(g)], cl_smooth_abs_sqrt(qq) - 1.0f, 1.0f);
        S.invden[g] = fast_rcp(m + kEps);
#pragma unroll
        for (int t = 0; t < nd; ++t) r[d0 + t] = S.R[d0 + t] * S.invden[g];
    });
    stamp(sid + 2);
    // steerable geometric product + first-order term (cegnn_utils.py:126-152)
    cl_weighted_gp<ALG>(L, z, r, ldsp + (TB::par + 16));
    stamp(sid + 3);
#pragma unroll
    for (int d = 0; d < D; ++d) S.s[d] = L[d] * kInvSqrt2;
    // MVLayerNorm (cegnn_utils.py:93-96): mean over the C channels of the row
    float qs = 0.f;
    static_for<0, D>([&](auto dd) {
        constexpr int d = decltype(dd)::value;
        qs += qsf<ALG, d> * S.s[d] * S.s[d];
    });
    S.qs = qs;
    S.nl = cl_smooth_abs_sqrt(qs);
    S.invMn = fast_rcp(__builtin_fmaf(cl_chan_sum<C>(S.nl), 1.0f / float(C), kEps));
    const float k = p0.z * S.invMn;
#pragma unroll
    for (int d = 0; d < D; ++d) out[d] = k * S.s[d];
    stamp(sid + 4);
}


// ---------------------------------------------------------------------------------
// order of the per-channel parameter gradients in the lane-private running sums
template <class ALG>
struct ClRed {
    static constexpr int G = ALG::G, P = ALG::P;
    static constexpr int i_la = 0, i_bL = 1, i_w = 2, i_an = 2 + P, i_sa = 2 + P + G, i_b1 = 2 + P + 3 * G, n = 3 + 3 * G + P;
};
// slice of partial sums of one block launch: reference layouts [W1 [C][I][G] | WR [C][C][G] | WL | small], small =
// b1 [C], sa [C][G], sb [C][G], w [C][P], an [C][G], bL [C], la [C]
template <class ALG, int C, int I>
struct ClPart {
    static constexpr int G = ALG::G, P = ALG::P;
    static constexpr int pWR = C * I * G, pWL = pWR + C * C * G, pS = pWL + C * C * G;
    static constexpr int qb1 = 0, qsa = qb1 + C, qsb = qsa + C * G, qw = qsb + C * G, qan = qw + C * P, qbL = qan + C * G, qla = qbL + C;
    static constexpr int m_small = C * (3 + 3 * G + P);
    static constexpr int total = pS + m_small;
    using RM = ClRed<ALG>;
    // running-sum index -> (offset of the parameter's first element inside `small`, stride between channels)
    static constexpr int off(int idx) {
        if (idx == RM::i_la) return qla;
        if (idx == RM::i_bL) return qbL;
        if (idx < RM::i_an) return qw + (idx - RM::i_w);
        if (idx < RM::i_sa) return qan + (idx - RM::i_an);
        if (idx < RM::i_b1) return ((idx - RM::i_sa) & 1 ? qsb : qsa) + (idx - RM::i_sa) / 2;
        return qb1;
    }
    static constexpr int stride(int idx) {
        if (idx == RM::i_la || idx == RM::i_bL) return 1;
        if (idx < RM::i_an) return P;
        if (idx < RM::i_b1) return G;
        return 1;
    }
};

// Lane-private running sums of the per-channel parameter gradients, kept in LDS ([group of 4][thread] f4: registers are
// the scarce resource of the backward - 35 of them spilled MFMA accumulators to scratch). Values arrive in index
// order; every fourth one triggers a 16-byte read-modify-write (measured, tools/lds_probe.hip: 9 of them cost 63 ns per
// tile at two waves per SIMD; ds_add_f32 takes ~640 ns EACH).
template <int N>
struct ClSums {
    static constexpr int kGroups = (N + 3) / 4;
    static constexpr int floats_per_wg = kGroups * 4 * 64 * kClWaves;
    float* base;   // LDS, this thread's f4 of group 0; group g is 4 * 64 * kClWaves floats further
    f4 pend;
    CSMPN_DEV explicit ClSums(float* b) : base(b), pend{0.f, 0.f, 0.f, 0.f} {}
    CSMPN_DEV void zero() {
#pragma unroll
        for (int g = 0; g < kGroups; ++g) cl_st4(base + g * (4 * 64 * kClWaves), f4{0.f, 0.f, 0.f, 0.f});
    }
    template <int IDX>
    CSMPN_DEV void add(float v) {
        pend[IDX % 4] = v;
        if constexpr (IDX % 4 == 3 || IDX == N - 1) {
            if constexpr (IDX % 4 != 3) {
#pragma unroll
                for (int i = IDX % 4 + 1; i < 4; ++i) pend[i] = 0.f;
            }
            float* p = base + (IDX / 4) * (4 * 64 * kClWaves);
            cl_st4(p, cl_ld4(p) + pend);
        }
    }
    template <int IDX>
    CSMPN_DEV float get() const { return base[(IDX / 4) * (4 * 64 * kClWaves) + IDX % 4]; }
};

// block backward. S: the block's recomputed forward state; gout: d/d(out) of this lane's (row, channel).
// Adds this tile's contributions to the persistent sums (sm: per-channel parameters, accR / accL: linear_right /
// linear_left weight tiles per grade) and leaves d/d(MVLinear output) in gy. The MVLinear weight gradient and the
// transposed MVLinear are the caller's (they need the block's input).
template <class ALG, int C, class TB>
CSMPN_DEV void cl_block_backward(const float* ldsw, const float* ldsp, const ClFwd& S, const float (&gout)[8], float (&gy)[8],
                                 ClSums<ClRed<ALG>::n>& sm, f4 (&accR)[ALG::G], f4 (&accL)[ALG::G], ClStamp& stamp, int sid) {
    constexpr int D = ALG::D, G = ALG::G, P = ALG::P;
    using RM = ClRed<ALG>;
    const f4 p0 = cl_ld4(ldsp + TB::par);
    // ---- MVLayerNorm backward
    const float la = p0.z;
    float dot = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) dot = __builtin_fmaf(gout[d], S.s[d], dot);
    sm.template add<RM::i_la>(dot * S.invMn);
    const float gMn = -cl_chan_sum<C>(la * dot) * S.invMn * S.invMn * (1.0f / float(C));   // d/d(mean norm) / C
    float ggp[D];   // = d/d(left) = d/d(gp)
    {
        const float inl = fast_rcp(S.nl);
        const float gqs = gMn * (0.5f * S.qs) * (inl * inl * inl);   // d nl / d qs = 0.5 qs / nl^3
        const float k0 = la * S.invMn;
        static_for<0, D>([&](auto dd) {
            constexpr int d = decltype(dd)::value;
            const float gs = __builtin_fmaf(k0, gout[d], gqs * (2.0f * qsf<ALG, d>) * S.s[d]);
            ggp[d] = gs * kInvSqrt2;
        });
    }
    sm.template add<RM::i_bL>(ggp[0]);
    // ---- d/dz from linear_left
    float gz[D];
#pragma unroll
    for (int d = 0; d < D; ++d) gz[d] = 0.f;
    stamp(sid);
    cl_mix<C, C, TB::WLT>(gz, ggp, ldsw);
    stamp(sid + 1);
    // ---- geometric product backward (gz, gr accumulate; d/dw per path)
    float gr[D], zf[D], rf[D];
#pragma unroll
    for (int d = 0; d < D; ++d) gr[d] = 0.f;
    static_for<0, D>([&](auto d) {
        zf[d] = S.gate[ALG::grade(d)] * S.y[d];
        rf[d] = S.R[d] * S.invden[ALG::grade(d)];
    });
    {
        f4 wv[(P + 3) / 4];
#pragma unroll
        for (int q = 0; q < (P + 3) / 4; ++q) wv[q] = cl_ld4(ldsp + (TB::par + 16 + 4 * q));
        static_for<0, P>([&](auto p) {
            constexpr int gi = ALG::t.path_g[p][0], gj = ALG::t.path_g[p][1], gk = ALG::t.path_g[p][2];
            constexpr int i0 = ALG::gstart(gi), ni = ALG::gsize(gi);
            constexpr int j0 = ALG::gstart(gj), nj = ALG::gsize(gj);
            constexpr int k0 = ALG::gstart(gk), nk = ALG::gsize(gk);
            const float w = wv[p / 4][p % 4];
            // U[i] = sum sign ggp[j] r[k] (unweighted d/dz), V[k] = sum sign ggp[j] z[i] (unweighted d/dr)
            float U[ni], V[nk];
#pragma unroll
            for (int t = 0; t < ni; ++t) U[t] = 0.f;
#pragma unroll
            for (int t = 0; t < nk; ++t) V[t] = 0.f;
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
            float gwv = 0.f;
#pragma unroll
            for (int t = 0; t < ni; ++t) { gz[i0 + t] = __builtin_fmaf(w, U[t], gz[i0 + t]); gwv = __builtin_fmaf(zf[i0 + t], U[t], gwv); }
#pragma unroll
            for (int t = 0; t < nk; ++t) gr[k0 + t] = __builtin_fmaf(w, V[t], gr[k0 + t]);
            sm.template add<RM::i_w + p>(gwv);
        });
    }
    stamp(sid + 2);
    // ---- NormalizationLayer backward -> gR
    const f4 sgv = cl_ld4(ldsp + (TB::par + 12));
    float gR[D];
    static_for<0, G>([&](auto g) {
        constexpr int d0 = ALG::gstart(g), nd = ALG::gsize(g);
        float gden = 0.f, qR = 0.f;
        static_for<0, nd>([&](auto t) {
            constexpr int d = d0 + decltype(t)::value;
            gden -= gr[d] * S.R[d];
            qR += qsf<ALG, d> * S.R[d] * S.R[d];
        });
        gden *= S.invden[g] * S.invden[g];      // d/d(den): -sum gr * R / den^2
        const float nu = cl_smooth_abs_sqrt(qR);
        const float sg = sgv[int(g)];
        sm.template add<RM::i_an + g>(gden * (nu - 1.0f) * sg * (1.0f - sg));
        const float inu = fast_rcp(nu);
        const float gq = (gden * sg) * (0.5f * qR) * (inu * inu * inu);
        static_for<0, nd>([&](auto t) {
            constexpr int d = d0 + decltype(t)::value;
            gR[d] = __builtin_fmaf(gr[d], S.invden[g], gq * (2.0f * qsf<ALG, d>) * S.R[d]);
        });
    });
    stamp(sid + 3);
    cl_mix<C, C, TB::WRT>(gz, gR, ldsw);
    stamp(sid + 4);
    // ---- weight gradients of linear_right and linear_left on the MFMA, B = z = gate * y
#ifndef CL_X_NOMFMA
    static_for<0, D>([&](auto d) {
        constexpr int g = ALG::grade(d);
        accR[g] = mfma16(gR[d], zf[d], accR[g]);
        accL[g] = mfma16(ggp[d], zf[d], accL[g]);
    });
#else
    static_for<0, D>([&](auto d) { constexpr int g = ALG::grade(d); accR[g][0] += gR[d] * zf[d]; accL[g][0] += ggp[d] * zf[d]; });
#endif
    stamp(sid + 5);
    // ---- MVSiLU backward -> gy
    const f4 sa = cl_ld4(ldsp + (TB::par + 4));
    static_for<0, G>([&](auto g) {
        constexpr int d0 = ALG::gstart(g), nd = ALG::gsize(g);
        float ggate = 0.f;
#pragma unroll
        for (int t = 0; t < nd; ++t) ggate = __builtin_fmaf(gz[d0 + t], S.y[d0 + t], ggate);
        const float gpre = ggate * S.gate[g] * (1.0f - S.gate[g]);
        float u;
        if constexpr (g == 0) {
            u = S.y[0];
        } else {
            u = 0.f;
            static_for<0, nd>([&](auto t) {
                constexpr int d = d0 + decltype(t)::value;
                u += qsf<ALG, d> * S.y[d] * S.y[d];
            });
        }
        sm.template add<RM::i_sa + 2 * g>(gpre * u);
        sm.template add<RM::i_sa + 2 * g + 1>(gpre);
        const float gu = gpre * sa[int(g)];
        static_for<0, nd>([&](auto t) {
            constexpr int d = d0 + decltype(t)::value;
            float v = gz[d] * S.gate[g];
            if constexpr (g == 0) v += gu;
            else v = __builtin_fmaf(gu * (2.0f * qsf<ALG, d>), S.y[d], v);
            gy[d] = v;
        });
    });
    sm.template add<RM::i_b1>(gy[0]);
    stamp(sid + 6);
}

// ---------------------------------------------------------------------------------
// per-lane row access: the lane's 8 blades = 32 contiguous bytes
CSMPN_DEV void cl_ld8(float (&x)[8], const float* p) {
    const f4 a = cl_ld4(p), b = cl_ld4(p + 4);
    x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
}
CSMPN_DEV void cl_st8(float* p, const float (&x)[8]) {
    cl_st4(p, f4{x[0], x[1], x[2], x[3]});
    cl_st4(p + 4, f4{x[4], x[5], x[6], x[7]});
}
#ifdef CL_X_NOOUT   // timing experiment only (results wrong): row stores go to the wave's LDS scratch
#define CL_GST8(gp, lp, x) cl_st8(lp, x)
#else
#define CL_GST8(gp, lp, x) cl_st8(gp, x)
#endif

// Rows of a staged tile [RPW][ROWLEN + 4] -> atomic adds into table rows of ROWLEN floats; lane = column. The row
// targets travel through SGPRs (v_readlane of the channel-0 lane of the row). Adds the rows to table[t_add[row]] (rows
// sorted by that index: equal consecutive targets are summed first) and, when SUB, subtracts them from
// table[t_sub[row]] (unsorted). Negative targets are skipped.
template <int C, int ROWLEN, bool SUB>
CSMPN_DEV void cl_scatter(const float* sc, int t_add, int t_sub, float* table, int lane) {
    using MP = ClMap<C>;
    constexpr int RPW = MP::RPW, SS = ROWLEN + 4, NC = ROWLEN / 64;
    static_assert(ROWLEN % 64 == 0, "whole columns");
    static_for<0, NC>([&](auto cc) {
        const int colx = 64 * cc + lane;
        const float* col = sc + colx;
        auto flush = [&](int target, float a) {
#ifndef CL_X_NOOUT   // timing experiment only (results wrong): no atomics
            if (target >= 0) atomicAdd(table + (size_t)target * ROWLEN + colx, a);
#else
            if (target == -12345) atomicAdd(table + (size_t)target * ROWLEN + colx, a);
#endif
        };
        float val[RPW];
#pragma unroll
        for (int i = 0; i < RPW; ++i) val[i] = col[i * SS];
        float acc = 0.f;
        int cur = __builtin_amdgcn_readlane(t_add, MP::lane_of_row(0));
        static_for<0, RPW>([&](auto rr) {
            const int t = __builtin_amdgcn_readlane(t_add, MP::lane_of_row(rr));
            if (t != cur) {
                flush(cur, acc);
                cur = t;
                acc = 0.f;
            }
            acc += val[rr];
        });
        flush(cur, acc);
        if constexpr (SUB) {
            static_for<0, RPW>([&](auto rr) { flush(__builtin_amdgcn_readlane(t_sub, MP::lane_of_row(rr)), -val[rr]); });
        }
    });
}

// ---------------------------------------------------------------------------------
// tile bookkeeping shared by the kernels
template <int C, int MODE>
struct ClTile {
    long row, lrow;
    bool valid;
    int i_dst, i_src, i_perm;
    float scale;
    template <int NA>
    CSMPN_DEV void load(const RowIO& io, long tile, int r) {
        row = tile * ClMap<C>::RPW + r;
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

// The gathered input passes of block 0, in two steps so that every load of a tile is in flight at once (left to the
// scheduler, the attribute loads were issued behind the wait for the feature rows: three dependent round trips per
// tile at two waves per SIMD). issue: raw 16-byte pieces; finish: x[p] = this lane's channel of pass p (attribute
// passes: channel c mod period, clamped to a valid one - its table entries are zero where it is not).
template <class ALG, int C, int MODE, int NA>
struct ClRaw {
    static constexpr int NSEG = MODE == MODE_EDGE ? 2 : 2;   // edge: h[dst], h[src]; node: h, agg
    f4 v[2 * (NSEG + (NA > 0 ? 1 : 0))];
    template <class TB>
    CSMPN_DEV void issue(const RowIO& io, const ClTile<C, MODE>& T, int c) {
        constexpr int D = ALG::D, ROW = C * D;
        const float *p0, *p1, *p2 = nullptr;
        if constexpr (MODE == MODE_EDGE) {
            p0 = io.seg[0].a + (size_t)T.i_dst * ROW + c * D;
            p1 = io.seg[0].b + (size_t)T.i_src * ROW + c * D;
            if constexpr (NA > 0) {
                const int ca = c & (TB::period(1) - 1);
                p2 = io.seg[1].a + (size_t)T.i_perm * (NA * D) + (ca < NA ? ca : NA - 1) * D;
            }
        } else {
            p0 = io.seg[0].a + (size_t)T.lrow * ROW + c * D;
            p1 = io.seg[1].a + (size_t)T.lrow * ROW + c * D;
            if constexpr (NA > 0) {
                const int ca = c & (TB::period(2) - 1);
                p2 = io.seg[2].a + (size_t)T.lrow * (NA * D) + (ca < NA ? ca : NA - 1) * D;
            }
        }
        v[0] = cl_ld4(p0); v[1] = cl_ld4(p0 + 4);
        v[2] = cl_ld4(p1); v[3] = cl_ld4(p1 + 4);
        if constexpr (NA > 0) { v[4] = cl_ld4(p2); v[5] = cl_ld4(p2 + 4); }
    }
    // all of them requested before the first is used
    CSMPN_DEV void pin() {
        if constexpr (NA > 0) asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]));
        else asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
    }
    template <class TB>
    CSMPN_DEV void finish(float (&x)[TB::NP][8], const ClTile<C, MODE>& T) const {
        auto unpack = [&](float (&t)[8], const f4& a, const f4& b) {
            t[0] = a.x; t[1] = a.y; t[2] = a.z; t[3] = a.w; t[4] = b.x; t[5] = b.y; t[6] = b.z; t[7] = b.w;
        };
        if constexpr (MODE == MODE_EDGE) {
            const f4 a = v[0] - v[2], b = v[1] - v[3];
            unpack(x[0], a, b);
            if constexpr (NA > 0) unpack(x[1], v[4], v[5]);
        } else {
            unpack(x[0], v[0], v[1]);
            unpack(x[1], v[2] * T.scale, v[3] * T.scale);
            if constexpr (NA > 0) unpack(x[2], v[4], v[5]);
        }
    }
};

template <int C>
CSMPN_DEV int cl_probe_dir(int c) {
    const int probe = cl_dpp_i<0x120 + ClMap<C>::ROTL>(c);
    return (((probe - c) & (C - 1)) == 1) ? 1 : -1;
}

// ---------------------------------------------------------------------------------
// forward kernel: NBLK blocks (1 or 2), all C channels wide. Tile t (64 / C rows) belongs to wave t mod (4 gridDim).
template <class ALG, int C, int MODE, int NBLK, int NA>
__global__ void __launch_bounds__(64 * kClWaves, 4) cemlp_cl_fwd_kernel(const DevCemlp C_arg, const RowIO io_arg) {
    typedef const char __attribute__((address_space(4))) * KArgPtr;
    const KArgPtr ka = (KArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr size_t kIoOffset = (sizeof(DevCemlp) + alignof(RowIO) - 1) / alignof(RowIO) * alignof(RowIO);
    const DevCemlp& Cd = *(const DevCemlp*)(const char*)ka;
    const RowIO& io = *(const RowIO*)(const char*)(ka + kIoOffset);
    (void)C_arg; (void)io_arg;
    using MP = ClMap<C>;
    using T0 = ClTab<C, MODE, NA, 0, false>;
    using T1 = ClTab<C, MODE, NA, 1, false>;
    constexpr int D = ALG::D, ROW = C * D, RPW = MP::RPW, SS = ROW + 4;
    static_assert(NBLK == 1 || NBLK == 2, "one or two blocks");
    constexpr int tab_floats = T0::total + (NBLK > 1 ? T1::total : 0);
    constexpr int kBatch = 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* lds = smem;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = MP::chan(lane), r = MP::row(lane);
    float* sc = lds + tab_floats + wave * (RPW * SS);   // scatter staging tile (edge program)
    ClStamp stamp(0);
    const float* ldsw0 = lds + 4 * c;
    const float* ldsp0 = lds + kClParStride * c;
    const float* ldsw1 = ldsw0 + T0::total;
    const float* ldsp1 = ldsp0 + T0::total;

    const long ntiles = (io.rows + RPW - 1) / RPW;
    const long tstride = (long)gridDim.x * kClWaves;
    const bool save_s = NBLK > 1 && io.save_state != 0 && io.save != nullptr;
    // Software pipeline over the wave's tiles: the row pieces of tile t + 1 are requested (all together) BEFORE the stores
    // and atomics of tile t - vmcnt counts in issue order, a load behind an atomic waits for its acknowledgement
    // (~3000 cycles under load) - and the indices of tile t + 2 travel while tile t + 1 computes.
    const long tile0 = (long)blockIdx.x * kClWaves + wave;
    ClTile<C, MODE> T, Tn;
    T.template load<NA>(io, tile0, r);
    ClRaw<ALG, C, MODE, NA> raw;
    raw.template issue<T0>(io, T, c);
    Tn.template load<NA>(io, tile0 + tstride, r);
    // the first tile's rows travel while the tables are staged
    {
        const int dir = cl_probe_dir<C>(c);
        cl_stage_block<ALG, C, T0, false>(Cd.b[0], lds, threadIdx.x, dir);
        if constexpr (NBLK > 1) cl_stage_block<ALG, C, T1, false>(Cd.b[1], lds + T0::total, threadIdx.x, dir);
    }
    __syncthreads();
    stamp(0);
    for (long tile = tile0; tile < ntiles; tile += tstride) {
        raw.pin();
        stamp(1);
        float out[D], in1[D];
        {
            float x[T0::NP][D];
            raw.template finish<T0>(x, T);
            ClFwd S;
#pragma unroll
            for (int d = 0; d < D; ++d) S.y[d] = 0.f;
            static_for<0, T0::NP>([&](auto p) { cl_mix<C, T0::period(p), T0::W1(p), kBatch>(S.y, x[p], ldsw0); });
            stamp(2);
            cl_block_tail<ALG, C, T0, kBatch>(ldsw0, ldsp0, S, out, stamp, 3);
            if (save_s && T.valid) {   // CSMPN_FLAG_SAVE_STATE: s of block 0 -> its state region (cemlp_device.hpp): whole tiles, two pieces per lane
                float* ps_ = io.save + state_region<ROW, ROW>(io.rows, 0, 0) + cl_state_off<C>(T.row, c);
                __builtin_nontemporal_store(f4{S.s[0], S.s[1], S.s[2], S.s[3]}, reinterpret_cast<f4*>(ps_));       // streaming: read once, by the backward
                __builtin_nontemporal_store(f4{S.s[4], S.s[5], S.s[6], S.s[7]}, reinterpret_cast<f4*>(ps_ + 256));
            }
        }
        if constexpr (NBLK > 1) {
#pragma unroll
            for (int d = 0; d < D; ++d) in1[d] = out[d];
            ClFwd S;
#pragma unroll
            for (int d = 0; d < D; ++d) S.y[d] = 0.f;
            cl_mix<C, C, T1::W1(0), kBatch>(S.y, in1, ldsw1);
            stamp(8);
            cl_block_tail<ALG, C, T1, kBatch>(ldsw1, ldsp1, S, out, stamp, 9);
            if (save_s && T.valid) {   // ... s of block 1
                float* ps_ = io.save + state_region<ROW, ROW>(io.rows, 0, 1) + cl_state_off<C>(T.row, c);
                __builtin_nontemporal_store(f4{S.s[0], S.s[1], S.s[2], S.s[3]}, reinterpret_cast<f4*>(ps_));       // streaming: read once, by the backward
                __builtin_nontemporal_store(f4{S.s[4], S.s[5], S.s[6], S.s[7]}, reinterpret_cast<f4*>(ps_ + 256));
            }
        }
        // next tile's rows, then this tile's stores
        const ClTile<C, MODE> Tc = T;
        T = Tn;
        raw.template issue<T0>(io, T, c);
        Tn.template load<NA>(io, tile + 2 * tstride, r);
        if constexpr (NBLK > 1) {
            if (io.save && Tc.valid) CL_GST8(io.save + (size_t)Tc.row * ROW + c * D, sc + r * SS + c * D, in1);
        }
        if constexpr (MODE == MODE_EDGE) {
            if (io.row_store) {
                if (Tc.valid) CL_GST8(io.agg + (size_t)Tc.lrow * ROW + c * D, sc + r * SS + c * D, out);
            } else {
                CL_LDS_ORDER();
                cl_st8(sc + r * SS + c * D, out);
                CL_LDS_ORDER();
                cl_scatter<C, ROW, false>(sc, Tc.valid ? Tc.i_dst : -1, -1, io.agg, lane);
            }
        } else if (Tc.valid) {
            if (io.resid) {
                float res[D];
                cl_ld8(res, io.resid + (size_t)Tc.row * ROW + c * D);
#pragma unroll
                for (int d = 0; d < D; ++d) out[d] += res[d];
            }
            CL_GST8(io.y + (size_t)Tc.row * ROW + c * D, sc + r * SS + c * D, out);
        }
        stamp(14);
    }
    stamp.flush(io.stamps, lane);
}
template <class ALG, int C, int MODE, int NBLK, int NA>
constexpr size_t cl_fwd_lds_bytes() {
    return sizeof(float) * (ClTab<C, MODE, NA, 0, false>::total + (NBLK > 1 ? ClTab<C, MODE, NA, 1, false>::total : 0) +
                            (MODE == MODE_EDGE ? kClWaves * ClMap<C>::RPW * (C * ALG::D + 4) : 0));
}

// ---------------------------------------------------------------------------------
// backward of block K (launched last block first). d/d(out of block K) comes from io.gy (last block; the edge program
// gathers it by target) or from the hand-over rows io.plw_g1; d/d(input of block K) goes to the hand-over rows (K > 0)
// or to the program's gradient targets (K = 0). Block K > 0 reads its input from the saved rows.
struct ClCarry { f4 a, b; };
template <class ALG, int C, int MODE, int NBLK, int NA, int K, bool SAVES>
CSMPN_DEV ClCarry cl_bwd_block(const RowIO& io, float* tab, float* work, const ClCarry carry_in, bool single, ClStamp& stamp) {
    f4 carry_a = carry_in.a, carry_b = carry_in.b;
    using MP = ClMap<C>;
    using TB = ClTab<C, MODE, NA, K, true>;
    using PT = ClPart<ALG, C, TB::I>;
    using RM = ClRed<ALG>;
    constexpr int D = ALG::D, G = ALG::G, ROW = C * D, RPW = MP::RPW, SS = ROW + 4, NP = TB::NP;
    static_assert(K >= 0 && K < NBLK && NBLK <= 2, "block index");
    constexpr bool kLast = K == NBLK - 1;
    // per-wave scratch: staging tile / image of the slice (the larger of the blocks' images: one layout for the launch)
    constexpr int img_max = NBLK > 1 && ClPart<ALG, C, C>::total > ClPart<ALG, C, ClTab<C, MODE, NA, 0, true>::I>::total
                                ? ClPart<ALG, C, C>::total : ClPart<ALG, C, ClTab<C, MODE, NA, 0, true>::I>::total;
    constexpr int scratch = (RPW * SS > img_max ? RPW * SS : img_max);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = MP::chan(lane), r = MP::row(lane);
    float* sc = work + wave * scratch;
    const float* ldsw = tab + 4 * c;
    const float* ldsp = tab + kClParStride * c;

    // persistent sums of this wave
    f4 accW1[NP][G], accR[G], accL[G];
    ClSums<RM::n> sm(work + kClWaves * scratch + 4 * threadIdx.x);
    sm.zero();
#pragma unroll
    for (int g = 0; g < G; ++g) {
        accR[g] = accL[g] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < NP; ++p) accW1[p][g] = f4{0.f, 0.f, 0.f, 0.f};
    }

    const long ntiles = (io.rows + RPW - 1) / RPW;
    const long tstride = (long)gridDim.x * kClWaves;
    // Software pipeline as in the forward: the rows of tile t + 1 are requested in front of the stores / atomics of tile t.
    f4 g0, g1, s0, s1;
    ClRaw<ALG, C, MODE, NA> raw;
    auto issue = [&](const ClTile<C, MODE>& Tl) {
        const float* gsrc = (kLast ? io.gy + (size_t)(MODE == MODE_EDGE ? (long)Tl.i_dst : Tl.lrow) * ROW
                                   : io.plw_g1 + (size_t)Tl.lrow * ROW) + c * D;
        if (kLast || !single) {
            g0 = cl_ld4(gsrc); g1 = cl_ld4(gsrc + 4);
        } else {   // one tile per wave: d/d(out) of this block stayed in registers
            g0 = carry_a; g1 = carry_b;
        }
        if constexpr (K == 0) {
            raw.template issue<TB>(io, Tl, c);
        } else {
            const float* sp = io.saved + (size_t)Tl.lrow * ROW + c * D;
            s0 = cl_ld4(sp); s1 = cl_ld4(sp + 4);
        }
    };
    const long tile0 = (long)blockIdx.x * kClWaves + wave;
    ClTile<C, MODE> T, Tn;
    T.template load<NA>(io, tile0, r);
    issue(T);
    Tn.template load<NA>(io, tile0 + tstride, r);
    for (long tile = tile0; tile < ntiles; tile += tstride) {
        if constexpr (K == 0) {
            asm volatile("" : "+v"(g0), "+v"(g1));
            raw.pin();
        } else {
            asm volatile("" : "+v"(g0), "+v"(g1), "+v"(s0), "+v"(s1));
        }
        stamp(1);
        const ClTile<C, MODE> Tc = T;
        float gout[D] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        float x[NP][D];
        if constexpr (K == 0) {
            raw.template finish<TB>(x, Tc);
        } else {
            x[0][0] = s0.x; x[0][1] = s0.y; x[0][2] = s0.z; x[0][3] = s0.w; x[0][4] = s1.x; x[0][5] = s1.y; x[0][6] = s1.z; x[0][7] = s1.w;
        }
        if (!Tc.valid) {
#pragma unroll
            for (int d = 0; d < D; ++d) gout[d] = 0.f;
        }
        float gy[D];
        {
            ClFwd S;
#pragma unroll
            for (int d = 0; d < D; ++d) S.y[d] = 0.f;
            constexpr bool have_s = SAVES;   // compile time: a run-time switch between the two recomputes spills the node program
            if constexpr (have_s) {   // CSMPN_FLAG_SAVE_STATE: the block's output in front of the layer norm, saved by the forward
                const float* ps_ = io.saved + state_region<ROW, ROW>(io.rows, 0, K) + cl_state_off<C>(Tc.lrow, c);
                const f4 a_ = __builtin_nontemporal_load(reinterpret_cast<const f4*>(ps_)), b_ = __builtin_nontemporal_load(reinterpret_cast<const f4*>(ps_ + 256));
#pragma unroll
                for (int i = 0; i < 4; ++i) { S.s[i] = a_[i]; S.s[4 + i] = b_[i]; }
            }
            static_for<0, NP>([&](auto p) { cl_mix<C, TB::period(p), TB::W1(p)>(S.y, x[p], ldsw); });
            stamp(2);
            float unused[D];
            cl_block_tail<ALG, C, TB, 4>(ldsw, ldsp, S, unused, stamp, 3, have_s);
            cl_block_backward<ALG, C, TB>(ldsw, ldsp, S, gout, gy, sm, accR, accL, stamp, 8);
        }
        // MVLinear weight gradient: A = gy, B = the pass's input blade
#ifndef CL_X_NOMFMA
        static_for<0, NP>([&](auto p) {
            static_for<0, D>([&](auto d) { accW1[p][ALG::grade(d)] = mfma16(gy[d], x[p][d], accW1[p][ALG::grade(d)]); });
        });
#else
        static_for<0, NP>([&](auto p) { static_for<0, D>([&](auto d) { accW1[p][ALG::grade(d)][0] += gy[d] * x[p][d]; }); });
#endif
        stamp(15);
        // next tile's rows, then this tile's stores / atomics
        T = Tn;
        issue(T);
        Tn.template load<NA>(io, tile + 2 * tstride, r);
        // d/d(input)
        if constexpr (K > 0) {
            float gx[D];
#pragma unroll
            for (int d = 0; d < D; ++d) gx[d] = 0.f;
            cl_mix<C, C, TB::W1T(0)>(gx, gy, ldsw);
            if (single) {
                carry_a = f4{gx[0], gx[1], gx[2], gx[3]}; carry_b = f4{gx[4], gx[5], gx[6], gx[7]};
            } else if (Tc.valid) {
                CL_GST8(io.plw_g1 + (size_t)Tc.row * ROW + c * D, sc + r * SS + c * D, gx);
            }
        } else if constexpr (MODE == MODE_EDGE) {
            if (io.gx[0]) {
                float gx[D];
#pragma unroll
                for (int d = 0; d < D; ++d) gx[d] = 0.f;
                cl_mix<C, C, TB::W1T(0)>(gx, gy, ldsw);
                if (io.row_store) {
                    if (Tc.valid) CL_GST8(io.gx[0] + (size_t)Tc.lrow * ROW + c * D, sc + r * SS + c * D, gx);
                } else {
                    CL_LDS_ORDER();
                    cl_st8(sc + r * SS + c * D, gx);
                    CL_LDS_ORDER();
                    cl_scatter<C, ROW, true>(sc, Tc.valid ? Tc.i_dst : -1, Tc.valid ? Tc.i_src : -1, io.gx[0], lane);
                }
            }
            if constexpr (NA > 0) {
                if (io.gx[1]) {
                    float gx[D];
#pragma unroll
                    for (int d = 0; d < D; ++d) gx[d] = 0.f;
                    cl_mix<C, C, TB::W1T(1)>(gx, gy, ldsw);
                    if (Tc.valid && c < NA) CL_GST8(io.gx[1] + (size_t)Tc.i_perm * (NA * D) + c * D, sc + r * SS + c * D, gx);
                }
            }
        } else {
            if (io.gx[0]) {
                float gx[D];
#pragma unroll
                for (int d = 0; d < D; ++d) gx[d] = 0.f;
                cl_mix<C, C, TB::W1T(0)>(gx, gy, ldsw);
                if (Tc.valid) {
                    if (io.resid_bwd) {
                        float res[D];
                        cl_ld8(res, io.gy + (size_t)Tc.row * ROW + c * D);
#pragma unroll
                        for (int d = 0; d < D; ++d) gx[d] += res[d];
                    }
                    CL_GST8(io.gx[0] + (size_t)Tc.row * ROW + c * D, sc + r * SS + c * D, gx);
                }
            }
            if (io.gx[1]) {
                float gx[D];
#pragma unroll
                for (int d = 0; d < D; ++d) gx[d] = 0.f;
                cl_mix<C, C, TB::W1T(1)>(gx, gy, ldsw);
#pragma unroll
                for (int d = 0; d < D; ++d) gx[d] *= Tc.scale;
                if (Tc.valid) CL_GST8(io.gx[1] + (size_t)Tc.row * ROW + c * D, sc + r * SS + c * D, gx);
            }
            if constexpr (NA > 0) {
                if (io.gx[2]) {
                    float gx[D];
#pragma unroll
                    for (int d = 0; d < D; ++d) gx[d] = 0.f;
                    cl_mix<C, C, TB::W1T(2)>(gx, gy, ldsw);
                    if (Tc.valid && c < NA) CL_GST8(io.gx[2] + (size_t)Tc.row * (NA * D) + c * D, sc + r * SS + c * D, gx);
                }
            }
        }
        stamp(16);
    }

#ifdef CL_X_NOEND   // timing experiment only (results wrong): the sums are kept alive, nothing else
    {
        f4 t = accR[0] + accL[0];
#pragma unroll
        for (int g = 0; g < G; ++g) { t += accR[g] + accL[g];
#pragma unroll
            for (int p = 0; p < NP; ++p) t += accW1[p][g]; }
        cl_st4(sc + 4 * lane, t);
    }
    if (false)
#endif
    // ---- end of the launch: this wave's sums -> image of the slice in its scratch; the workgroup adds its four
    // images in wave order and writes its slice (coalesced 16-byte stores)
    {
        float* img = sc;
        const int j = lane & 15, q = lane >> 4;
        // MFMA tiles. C = 8: tile element (i = 4q + v, j) = (o = i >> 1, r0 = i & 1) x (c = j >> 1, r0' = j & 1); the
        // wanted sum is on r0 = r0': even lanes take v = 0, 2 of their own and v = 1, 3 of their odd neighbour.
        auto put_tile = [&](const f4 (&acc)[G], int base, int I, int coff, int width) {
            if constexpr (C == 8) {
                float lo[G], hi[G];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    lo[g] = acc[g][0] + dpp_mov<0xB1>(acc[g][1]);   // quad_perm [1,0,3,2]: the odd neighbour's value
                    hi[g] = acc[g][2] + dpp_mov<0xB1>(acc[g][3]);
                }
                const int cc = j >> 1;
                if ((j & 1) == 0 && cc < width) {
                    float* p0 = img + base + ((2 * q) * I + coff + cc) * G;
                    cl_st4(p0, f4{lo[0], lo[1], lo[2], lo[3]});
                    cl_st4(p0 + I * G, f4{hi[0], hi[1], hi[2], hi[3]});
                }
            } else {
                if (j < width) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        float* p0 = img + base + ((4 * q + v) * I + coff + j) * G;
                        cl_st4(p0, f4{acc[0][v], acc[1][v], acc[2][v], acc[3][v]});
                    }
                }
            }
        };
        static_for<0, NP>([&](auto p) { put_tile(accW1[p], 0, TB::I, TB::coff(p), TB::width(p)); });
        put_tile(accR, PT::pWR, C, 0, C);
        put_tile(accL, PT::pWL, C, 0, C);
        // per-channel sums over the rows of the wave: one MFMA with A = 1 adds the lane bits 4-5 (every lane receives
        // its column's sum), one DPP add the interleaved row pair (C = 8)
        f4 sv[ClSums<RM::n>::kGroups];
#pragma unroll
        for (int gq = 0; gq < ClSums<RM::n>::kGroups; ++gq) sv[gq] = cl_ld4(sm.base + gq * (4 * 64 * kClWaves));
        static_for<0, RM::n>([&](auto idx) {
            const f4 t = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, sv[idx / 4][idx % 4], f4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            float s = t[0];
            if constexpr (C == 8) s += dpp_mov<0xB1>(s);
            if (q == 0 && (C == 16 || (lane & 1) == 0)) img[PT::pS + PT::off(idx) + c * PT::stride(idx)] = s;
        });
        __syncthreads();
        // block k's slices start behind those of the blocks 0 .. k - 1 (kClSliceCap slices each)
        float* part = io.rl_partials + (K == 0 ? 0 : (size_t)kClSliceCap * ClPart<ALG, C, ClTab<C, MODE, NA, 0, true>::I>::total) +
                      (size_t)blockIdx.x * PT::total;
        const float* img0 = work;
        static_assert(PT::total % 4 == 0, "slice length");
        for (int e = 4 * threadIdx.x; e < PT::total; e += 4 * 64 * kClWaves) {
            f4 v = cl_ld4(img0 + e);
#pragma unroll
            for (int w = 1; w < kClWaves; ++w) v += cl_ld4(img0 + w * scratch + e);
            cl_st4(part + e, v);
        }
    }
    stamp(17);
    return ClCarry{carry_a, carry_b};
}
template <class ALG, int C, int MODE, int NBLK, int NA>
constexpr size_t cl_bwd_lds_bytes() {
    using TB0 = ClTab<C, MODE, NA, 0, true>;
    using TB1 = ClTab<C, MODE, NA, 1, true>;
    constexpr int tile = ClMap<C>::RPW * (C * ALG::D + 4);
    int img = ClPart<ALG, C, TB0::I>::total;
    if (NBLK > 1 && ClPart<ALG, C, C>::total > img) img = ClPart<ALG, C, C>::total;
    return sizeof(float) * (TB0::total + (NBLK > 1 ? TB1::total : 0) + kClWaves * (tile > img ? tile : img) +
                            ClSums<ClRed<ALG>::n>::floats_per_wg);
}
// The backward kernel: the blocks one after the other (last block first) in ONE launch. A wave keeps its tiles from
// block to block, so the hand-over rows d/d(block input) it reads in block k - 1 are the ones it wrote itself in block k
// (through L2: the region is not read earlier in the launch, its lines cannot sit stale in this CU's L1; the stores are
// drained before the workgroup's barrier) - no grid-wide synchronisation, one prologue less, and the node program
// (one tile per wave) is a single launch.
template <class ALG, int C, int MODE, int NBLK, int NA, bool SAVES = false>
__global__ void __launch_bounds__(64 * kClWaves, 2) cemlp_cl_bwd_kernel(const DevCemlp C_arg, const RowIO io_arg) {
    typedef const char __attribute__((address_space(4))) * KArgPtr;
    const KArgPtr ka = (KArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr size_t kIoOffset = (sizeof(DevCemlp) + alignof(RowIO) - 1) / alignof(RowIO) * alignof(RowIO);
    const DevCemlp& Cd = *(const DevCemlp*)(const char*)ka;
    const RowIO& io = *(const RowIO*)(const char*)(ka + kIoOffset);
    (void)C_arg; (void)io_arg;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    ClStamp stamp(0);
    using TB0 = ClTab<C, MODE, NA, 0, true>;
    using TB1 = ClTab<C, MODE, NA, 1, true>;
    constexpr int tabs = TB0::total + (NBLK > 1 ? TB1::total : 0);
    // the tables of every block are staged up front (one prologue per launch)
    {
        const int dir = cl_probe_dir<C>(ClMap<C>::chan(threadIdx.x & 63));
        cl_stage_block<ALG, C, TB0, true>(Cd.b[0], smem, threadIdx.x, dir);
        if constexpr (NBLK > 1) cl_stage_block<ALG, C, TB1, true>(Cd.b[1], smem + TB0::total, threadIdx.x, dir);
    }
    __syncthreads();
    stamp(0);
    const long ntiles = (io.rows + ClMap<C>::RPW - 1) / ClMap<C>::RPW;
    const bool single = ntiles <= (long)gridDim.x * kClWaves;   // one tile per wave: the hand-over stays in registers
    ClCarry carry{f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
    if constexpr (NBLK > 1) {
        carry = cl_bwd_block<ALG, C, MODE, NBLK, NA, 1, SAVES>(io, smem + TB0::total, smem + tabs, carry, single, stamp);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's hand-over rows have left for L2
        __syncthreads();                                   // ... and every wave is done with the images
    }
    cl_bwd_block<ALG, C, MODE, NBLK, NA, 0, SAVES>(io, smem, smem + tabs, carry, single, stamp);
    stamp.flush(io.stamps, threadIdx.x & 63);
}

// last kernel of a backward: grads += sum over the workgroups' slices of EVERY block, fixed order (deterministic).
// The slices of block 1 start slice_cap * (slice length of block 0) floats behind those of block 0. A workgroup takes 16
// consecutive elements (64-byte pieces of every slice); thread (j = tid & 15, w = tid >> 4) adds the slices w, w + 16,
// ... sixteen loads in flight at a time; the 16 partial sums of an element meet in LDS and are added in order.
template <class ALG, int C, int I0, int NBLK>
__global__ void __launch_bounds__(256) cl_reduce_kernel(const DevCemlp Cd, const float* part, int nslices, int slice_cap) {
    constexpr int G = ALG::G;
    using P0 = ClPart<ALG, C, I0>;
    using P1 = ClPart<ALG, C, C>;
    constexpr int total = P0::total + (NBLK > 1 ? P1::total : 0);
    __shared__ float red[16][17];
    const int j = threadIdx.x & 15, w = threadIdx.x >> 4;
    int e = blockIdx.x * 16 + j;
    const bool live = e < total;
    if (!live) e = 0;
    const int k = (NBLK > 1 && e >= P0::total) ? 1 : 0;
    const int len = k == 0 ? P0::total : P1::total;
    const float* p = part + (k == 0 ? 0 : (size_t)slice_cap * P0::total);
    if (k == 1) e -= P0::total;
    float s = 0.f;
    for (int s0 = w; s0 < nslices; s0 += 256) {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int sl = s0 + 16 * i;
            v[i] = sl < nslices ? p[(size_t)sl * len + e] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) s += v[i];
    }
    red[w][j] = s;
    __syncthreads();
    if (w != 0 || !live) return;
#pragma unroll
    for (int i = 1; i < 16; ++i) s += red[i][j];
    const DevBlock& B = Cd.b[k];
    const int I = k == 0 ? I0 : C;
    int f = e;
    float* dst = nullptr;
    const int nW1 = C * I * G, nWC = C * C * G;
    if (f < nW1) dst = B.gW1 + f;
    else if ((f -= nW1) < nWC) dst = B.gWR + f;
    else if ((f -= nWC) < nWC) dst = B.gWL + f;
    else {
        f -= nWC;
        if (f < P0::qsa) dst = B.has_b1 ? B.gb1 + f : nullptr;
        else if (f < P0::qsb) dst = B.gsa + (f - P0::qsa);
        else if (f < P0::qw) dst = B.gsb + (f - P0::qsb);
        else if (f < P0::qan) dst = B.gw + (f - P0::qw);
        else if (f < P0::qbL) dst = B.gan + (f - P0::qan);
        else if (f < P0::qla) dst = B.gbL + (f - P0::qbL);
        else dst = B.gla + (f - P0::qla);
    }
    if (dst) *dst += s;
}

}  // namespace csmpn
