// Channel-MFMA backward, "parked" form (round 4): the backward of Cl(3,0) CEMLPs whose blocks are 16 channels wide (S2's
// layer, the motion task model) at TWO waves per SIMD - 512-thread workgroups, <= 256 registers, no scratch.
//
// Same arithmetic as cemlp_cm.hpp's forward (csmpn/models/cegnn_utils.py:34-155,287-338) and the same
// lane layout: a wave covers 16 rows, lane = (row = lane & 15, q = lane >> 4) holds the channels 4 v + q (v < 4) of its row,
// all 8 blades - a tensor is  f4 t[8]  - so every dense mixing, forward or transposed, is v_mfma_f32_16x16x4_f32 on the
// registers as they stand. The round-3 backward in this layout kept y, R, d/d(gp) and d/dz (128 registers), the
// weight-gradient tiles (64-80) and one channel's product backward (~110) alive at once: ~330 registers, 0.6 KB of
// scratch per lane at one wave per SIMD. Here a tensor that is not needed by the running phase lives in LDS:
//
//  * two slots of 8 KB per wave (P, W). A slot holds one 16-row tensor in ONE layout that serves all three readers
//    without bank conflicts (ds_*_b32, banks = address mod 32 over 32-lane halves):
//        element (blade d, row r, q, v)  at  512 v + 64 d + 16 q + (r ^ (v << 2) ^ ((q >> 1) << 1))          [floats]
//      - the owner lane (r, q) writes / reads back its own 32 values (parking),
//      - the per-channel phases read the 8 blades of ONE channel v of their row,
//      - the weight-gradient MFMAs (contraction over ROWS) read it transposed: lane (i, k) takes row 4 s + k,
//        column i = (q = i >> 2, v = i & 3) for step s - the k-slice of an MFMA operand - straight from the slot.
//  * flow of one block, per 16-row tile (tensors in registers | in P | in W):
//        x -> y = W1 x                                   | -        | d/d(out) (loaded with x, parked)
//        z = gate(y) y          (y dies)                 | z        |
//        R = WR z, L = WL z -> s = (L + gp(z, n(R)))/sqrt2 ; z leaves the registers
//        LayerNorm backward: ggp per channel             | z        | ggp
//        dWL += ggp^T z  (both operands from the slots)
//        product backward per channel: R -> gR, gz       | z        | ggp     (ggp_c, z_c read per channel)
//        gz += WL^T ggp (ggp read back)                  | z        | -> gR
//        gz += WR^T gR ; dWR += gR^T z                    | -        | gR
//        x again -> y = W1 x (the gates' argument) ; MVSiLU backward: gz -> gy   | x chunk | gy
//        dW1 += gy^T x  per input chunk ; d/dx = W1^T gy -> scatter / store
//    At most three tensors are in registers beside the persistent sums; the recompute of y costs one MVLinear's MFMAs
//    (+8 %) and saves a slot.
//  * ONE set of weight tables per block: the transposed mixes read the forward tables. Entry (grade, m', chunk) holds
//    f4 F[(i, k)] = W[orow(i)][channel of slot (k, v)], v = 0..3; the transposed operand of lane (i', k') is
//    T[v'] = W[4 v' + k'][orow(i')] = F[(4 k' + v', i' >> 2)][i' & 3]. With lane (i, k)'s vector stored at 16-byte unit
//    (i & 3) 16 + (i >> 3) 8 + 2 k + ((i >> 2) & 1) the forward read is one conflict-free ds_read_b128 and the transposed
//    one four ds_read_b32 at 64-float strides whose addresses are CONSECUTIVE across the wave.
//  * per-channel parameter gradients (35 per channel): summed over the 16 rows of a q-group - a DPP row - by a
//    transposing butterfly: 16 values in, lane j of the row keeps the total of value j (15 DPP adds + 30 selects per 16
//    values instead of 64 DPP adds or an LDS pass), accumulated in 10 registers over the tile loop.
//  * weight gradients: MFMA tiles per grade, persistent over the tile loop (AGPRs), as before.
//  * one launch for all blocks, last block first, hand-over rows through L2, per-workgroup slices + cl_reduce_kernel:
//    the launch structure of cemlp_cl.hpp / the round-3 backward.
#pragma once
#include "cemlp_cm.hpp"

namespace csmpn {

// phase marker in the generated code (a comment: no instruction); tools/asm_phases.py counts instructions and scratch
// accesses between markers
#define CB_MARK(n) asm volatile("; cb-phase " #n ::: "memory")

constexpr int kCbWaves = 8;        // waves per workgroup: two per SIMD
constexpr int kCbSlot = 2048;      // floats of one tensor slot (16 rows x 16 channels x 8 blades)

// Ordering point between LDS accesses of different lanes of ONE wave (store by the owner, read by another lane): the LDS
// executes a wave's operations in issue order, so no wait is needed - but the compiler, which sees each lane's own
// stores and loads as disjoint addresses, must not move them across this point.
CSMPN_DEV void cb_sync() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// 16-byte unit of lane (i, k)'s vector inside a table entry of 64 units
CSMPN_DEV constexpr int cb_unit(int i, int k) { return cm_unit<true>(i, k); }

// parameters -> the block's LDS tables (once per workgroup): CmTab's entries, every entry in cb_unit order, then the
// per-channel parameter rows (cm_stage_tables, cemlp_cm.hpp)
template <class ALG, int C, class TB, int NT = 64 * kCbWaves>
__device__ void cb_stage_block(const DevBlock& B, float* base, int tid) {
    cm_stage_tables<ALG, C, TB, NT, true>(B, base, tid);
}

// float offsets of this lane inside a tensor slot (see the header): wr[v] - its own element (blade 0) of channel slot v;
// mr[s] - the element (blade 0) it feeds to step s of a rows-contracting MFMA
struct CbAddr {
    int w0, m0;   // wr(v) = (w0 ^ 4 v) + 512 v,  mr(s) = m0 ^ 4 s: two registers instead of eight
    CSMPN_DEV explicit CbAddr(int lane) {
        const int r = lane & 15, q = lane >> 4;
        w0 = 16 * q + (r ^ ((q >> 1) << 1));
        const int i = lane & 15, k = lane >> 4;
        m0 = 512 * (i & 3) + 16 * (i >> 2) + (k ^ ((i & 3) << 2) ^ ((i >> 3) << 1));
    }
    CSMPN_DEV int wr(int v) const { return (w0 ^ (v << 2)) + 512 * v; }
    CSMPN_DEV int mr(int s) const { return m0 ^ (s << 2); }
};
CSMPN_DEV void cb_put(float* slot, const CbAddr& A, const f4 (&t)[8]) {
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int d = 0; d < 8; ++d) slot[A.wr(v) + 64 * d] = t[d][v];
}
CSMPN_DEV void cb_get(const float* slot, const CbAddr& A, f4 (&t)[8]) {
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int d = 0; d < 8; ++d) t[d][v] = slot[A.wr(v) + 64 * d];
}
template <int V>
CSMPN_DEV void cb_get_chan(const float* slot, const CbAddr& A, float (&x)[8]) {
#pragma unroll
    for (int d = 0; d < 8; ++d) x[d] = slot[A.wr(V) + 64 * d];
}
template <int V>
CSMPN_DEV void cb_put_chan(float* slot, const CbAddr& A, const float (&x)[8]) {
#pragma unroll
    for (int d = 0; d < 8; ++d) slot[A.wr(V) + 64 * d] = x[d];
}
// acc[grade] += sum over the 16 rows and the blades of the grade of a^T b: acc[g][v] of lane (j, qq) = D[4 qq + v][j] =
// sum_rows a[row][column 4 qq + v] b[row][column j]   (column c = channel slot (q = c >> 2, v = c & 3) = channel orow(c))
template <class ALG>
CSMPN_DEV void cb_wgrad(f4 (&acc)[4], const float* slotA, const float* slotB, const CbAddr& A) {
    static_for<0, 8>([&](auto d) {
        constexpr int g = ALG::grade(d);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[g] = mfma16(slotA[A.mr(s) + 64 * d], slotB[A.mr(s) + 64 * d], acc[g]);
    });
}

// acc[d] += W^T-mix of x through the FORWARD table entry of the pair: ldst = table base + the lane's transposed offset
// (cb_tofs) + the float offset of the pair's grade-0 entry, GS = float stride between grades
CSMPN_DEV int cb_tofs(int lane) {
    const int i = lane & 15, k = lane >> 4;
    return 32 * (k >> 1) + 8 * (i >> 2) + 4 * (k & 1) + (i & 3);
}
template <class ALG, int GS>
CSMPN_DEV void cb_mix_t(f4 (&acc)[8], const f4 (&x)[8], const float* ldst) {
    f4 a[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float* p = ldst + g * GS;
        a[g] = f4{p[0], p[64], p[128], p[192]};
    }
    static_for<0, 8>([&](auto d) {
        constexpr int g = ALG::grade(d);
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[d] = mfma16(a[g][v], x[d][v], acc[d]);
    });
}

// Transposing sum over the 16 lanes of a DPP row (the 16 rows of a q-group): NV values in, lane j of the row returns the
// sum over the row's lanes of x[j % NV] (NV = 16: one value per lane; NV = 8 / 4: every value in 2 / 4 lanes). Step b
// pairs the values that differ in bit b of their index: a lane keeps the one selected by bit b of its own index and adds
// the kept one's other half from the lane 2^b away (lower bits equal, bit b opposite - true for the xor of the quad
// permutations and for a rotation in either direction, so the rotation direction does not matter).
template <int NV>
CSMPN_DEV float cb_rows_sum(float (&x)[NV], int l16) {
    static_assert(NV == 16 || NV == 8 || NV == 4, "values per call");
    const bool b0 = l16 & 1, b1 = l16 & 2, b2 = l16 & 4, b3 = l16 & 8;
    float y[NV / 2];
#pragma unroll
    for (int j = 0; j < NV / 2; ++j) {
        const float keep = b0 ? x[2 * j + 1] : x[2 * j], send = b0 ? x[2 * j] : x[2 * j + 1];
        y[j] = keep + dpp_mov<0xB1>(send);   // quad_perm [1,0,3,2]
    }
    float z[NV / 4];
#pragma unroll
    for (int j = 0; j < NV / 4; ++j) {
        const float keep = b1 ? y[2 * j + 1] : y[2 * j], send = b1 ? y[2 * j] : y[2 * j + 1];
        z[j] = keep + dpp_mov<0x4E>(send);   // quad_perm [2,3,0,1]
    }
    if constexpr (NV == 4) {
        float t = z[0];
        t += dpp_mov<0x124>(t);              // row_ror 4
        t += dpp_mov<0x128>(t);              // row_ror 8
        return t;
    } else {
        float u[NV / 8];
#pragma unroll
        for (int j = 0; j < NV / 8; ++j) {
            const float keep = b2 ? z[2 * j + 1] : z[2 * j], send = b2 ? z[2 * j] : z[2 * j + 1];
            u[j] = keep + dpp_mov<0x124>(send);
        }
        if constexpr (NV == 8) {
            float t = u[0];
            t += dpp_mov<0x128>(t);
            return t;
        } else {
            const float keep = b3 ? u[1] : u[0], send = b3 ? u[0] : u[1];
            return keep + dpp_mov<0x128>(send);
        }
    }
}

// the per-channel parameter gradients of a tile in the order they are produced, 16 per butterfly:
//   group 0        : (la, bL) of the channel slots v = 0..3 (8 values)
//   groups 1 .. 6  : [w 0..19 | an 0..3] of v = 0..3 (4 x 24 = 6 x 16 values)
//   groups 7 .. 9  : [sa0 sb0 sa1 sb1 sa2 sb2 sa3 sb3 | b1] of v = 0..3 (4 x 9 = 36 values: 16 + 16 + 4)
// lane j of a q-group's row accumulates value j of each group over the tiles of the launch.
constexpr int kCbGroups = 10;
template <class ALG>
struct CbSmall {
    using RM = ClRed<ALG>;
    static constexpr int P = ALG::P;
    static_assert(P == 20 && ALG::G == 4, "Cl(3,0)-shaped algebra");
    // (channel slot v, ClRed index) of lane j's value in group g; v = -1: none
    static CSMPN_DEV void decode(int g, int j, int& v, int& idx) {
        v = -1;
        idx = 0;
        if (g == 0) {
            if (j < 8) { v = j >> 1; idx = (j & 1) ? RM::i_bL : RM::i_la; }
        } else if (g <= 6) {
            const int t = 16 * (g - 1) + j, k = t % 24;
            v = t / 24;
            idx = k < P ? RM::i_w + k : RM::i_an + (k - P);
        } else {
            const int t = 16 * (g - 7) + j;
            if (t < 36 && (g < 9 || j < 4)) {
                const int k = t % 9;
                v = t / 9;
                idx = k < 8 ? RM::i_sa + k : RM::i_b1;
            }
        }
    }
};

// collects values in production order and runs a butterfly whenever 16 are there
template <int BASE_GROUP>
struct CbCollect {
    float buf[16];
    template <int IDX>   // IDX: position in this collector's sequence
    CSMPN_DEV void add(float v, float (&acc)[kCbGroups], int l16) {
        buf[IDX % 16] = v;
        if constexpr (IDX % 16 == 15) acc[BASE_GROUP + IDX / 16] += cb_rows_sum<16>(buf, l16);
    }
};

// one channel: geometric product + normalisation backward (as cm_gp_bwd), the parameter gradients handed to `emit`
// in the order [w 0..19 | an 0..3] as soon as they are final
template <class ALG, bool PIN = false, class EMIT>
CSMPN_DEV void cb_gp_bwd(const float (&ggp_in)[8], const float (&zf_in)[8], const float (&R)[8], float (&gz)[8], float (&gR)[8],
                         const float* pp, EMIT&& emit) {
    // PIN (cemlp_cmp.hpp, 512-register budget): the paths in program order - every path's operands pass through an empty asm
    float ggp[8], zf[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) { ggp[d] = ggp_in[d]; zf[d] = zf_in[d]; }
    auto pin8 = [](float (&x)[8]) {
        asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
    };
    constexpr int D = ALG::D, G = ALG::G, P = ALG::P;
    const f4 sgv = cl_ld4(pp + 12);
    float rf[D], invden[G], nu[G], qR[G];
    static_for<0, G>([&](auto g) {
        constexpr int d0 = ALG::gstart(g), nd = ALG::gsize(g);
        float qq = 0.f;
        static_for<0, nd>([&](auto t) {
            constexpr int d = d0 + decltype(t)::value;
            qq += qsf<ALG, d> * R[d] * R[d];
        });
        qR[g] = qq;
        nu[g] = cl_smooth_abs_sqrt(qq);
        invden[g] = fast_rcp(__builtin_fmaf(sgv[int(g)], nu[g] - 1.0f, 1.0f) + kEps);
#pragma unroll
        for (int t = 0; t < nd; ++t) rf[d0 + t] = R[d0 + t] * invden[g];
    });
    float gr[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { gr[d] = 0.f; gz[d] = 0.f; }
    static_for<0, P>([&](auto p) {
        constexpr int gi = ALG::t.path_g[p][0], gj = ALG::t.path_g[p][1], gk = ALG::t.path_g[p][2];
        constexpr int i0 = ALG::gstart(gi), ni = ALG::gsize(gi);
        constexpr int j0 = ALG::gstart(gj), nj = ALG::gsize(gj);
        constexpr int k0 = ALG::gstart(gk), nk = ALG::gsize(gk);
        const float w = pp[16 + p];
        if constexpr (PIN) { pin8(ggp); pin8(rf); pin8(zf); pin8(gz); pin8(gr); }
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
        emit(p, gwv);
    });
    static_for<0, G>([&](auto g) {
        constexpr int d0 = ALG::gstart(g), nd = ALG::gsize(g);
        float gden = 0.f;
        static_for<0, nd>([&](auto t) {
            constexpr int d = d0 + decltype(t)::value;
            gden -= gr[d] * R[d];
        });
        gden *= invden[g] * invden[g];
        const float sg = sgv[int(g)];
        emit(IC<P + g>{}, gden * (nu[g] - 1.0f) * sg * (1.0f - sg));
        const float inu = fast_rcp(nu[g]);
        const float gq = (gden * sg) * (0.5f * qR[g]) * (inu * inu * inu);
        static_for<0, nd>([&](auto t) {
            constexpr int d = d0 + decltype(t)::value;
            gR[d] = __builtin_fmaf(gr[d], invden[g], gq * (2.0f * qsf<ALG, d>) * R[d]);
        });
    });
}

// the block's input rows of one tile: requested in one go (issue), turned into chunks later (finish)
template <class ALG, int C, int MODE, int NA, int K>
struct CbIn {
    using TF = CmTab<C, MODE, NA, K>;
    static constexpr int ROW = C * ALG::D;
    CmRaw<ALG, C, MODE, NA> raw;   // block 0: gathered / concatenated rows
    CmPiece sp;                    // later blocks: the saved block input
    CSMPN_DEV void issue(const RowIO& io, const CmTile<MODE>& T, int q) {
        if constexpr (K == 0) raw.issue(io, T, q);
        else sp.load(io.saved + (size_t)T.lrow * ROW + q * ALG::D);
    }
    CSMPN_DEV void finish(f4 (&x)[TF::NCH][8], const CmTile<MODE>& T) const {
        if constexpr (K == 0) raw.template finish<TF>(x, T);
        else cm_unpack(x[0], sp);
    }
};

// persistent sums of one wave over its tiles of one block
template <int NCH>
struct CbAcc {
    f4 w1[NCH][4], wr[4], wl[4];   // weight-gradient tiles per grade
    float sm[kCbGroups];           // per-channel parameter sums (CbSmall)
    CSMPN_DEV void zero() {
        const f4 z = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            wr[g] = wl[g] = z;
#pragma unroll
            for (int c = 0; c < NCH; ++c) w1[c][g] = z;
        }
#pragma unroll
        for (int g = 0; g < kCbGroups; ++g) sm[g] = 0.f;
    }
};

// backward of block K over this wave's tiles. tab: the block's tables, P / W: this wave's slots, work: all waves' slots
// (the end-of-block image lies over them).
template <class ALG, int C, int MODE, int NBLK, int NA, int K>
__device__ void cb_block(const RowIO& io, float* tab, float* work, int* ctr, ClStamp& stamp) {
    static_assert(C == 16, "one channel group");
    using TF = CmTab<C, MODE, NA, K>;
    using PT = ClPart<ALG, C, TF::I>;
    using SM = CbSmall<ALG>;
    constexpr int D = ALG::D, G = ALG::G, P = ALG::P, ROW = C * D, SS = ROW + 4, NCH = TF::NCH;
    constexpr bool kLast = K == NBLK - 1;
    constexpr int GS1 = TF::w1(1, 0, 0) - TF::w1(0, 0, 0), GSC = TF::wc(0, 1, 0, 0) - TF::wc(0, 0, 0, 0);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 15, q = lane >> 4;
    float* Pq = work + wave * (2 * kCbSlot);
    float* Wq = Pq + kCbSlot;
    const CbAddr AD(lane);
    const float* ldsa = tab + 4 * cb_unit(r, q);
    const float* ldst = tab + cb_tofs(lane);
    const float* ldsp = tab + TF::par + kClParStride * q;
    auto PP = [&](int v) { return ldsp + 4 * v * kClParStride; };

    CbAcc<NCH> A;
    A.zero();
    const long ntiles = (io.rows + kCmRows - 1) / kCmRows;
    const long tstride = (long)gridDim.x * kCbWaves;
#ifdef CSMPN_STAMPS
    const unsigned long long kstart = __builtin_amdgcn_s_memtime();
#endif
    // Tiles of a workgroup: 8 consecutive ones per round of the grid. WHICH wave takes which is decided at run time (a
    // counter in LDS) unless the deterministic flag is set: the two waves of a SIMD run the same program and the vector
    // issue goes to the older one first - with a static map waves 0-3 finished their tiles ~20 % early and idled at the
    // end-of-block barrier while waves 4-7 ran alone (measured with per-wave stamps; alternating s_setprio did not level
    // it). Dynamic claims keep every wave busy to the end; the price is that a wave's partial sums - and so the last
    // bits of the parameter gradients - depend on the timing (as the float atomics of the scatter already do).
    // Software pipeline (as the forward): the rows of tile t + 1 are requested BEFORE the stores and atomics of tile t (a
    // load queued behind an atomic waits for its acknowledgement); its indices were requested at the start of tile t.
    const bool dynamic = !io.row_store;
    long seq = wave;   // static map: this wave's next position in the workgroup's tile sequence
    auto claim = [&]() -> long {
        long n;
        if (dynamic) {
            int c = 0;
            if (lane == 0) c = atomicAdd(ctr, 1);
            n = __builtin_amdgcn_readfirstlane(c);
        } else {
            n = seq;
            seq += kCbWaves;
        }
        return (n >> 3) * tstride + (long)blockIdx.x * kCbWaves + (n & 7);
    };
    long tile = claim();
    CmTile<MODE> T, Tn;
    T.template load<NA>(io, tile, r);
    CbIn<ALG, C, MODE, NA, K> in;
    in.issue(io, T, q);
    while (tile < ntiles) {
        asm volatile("" ::: "memory");   // the tables are loop invariant: keep their reads inside the loop
        const long tile_next = claim();
        Tn.template load<NA>(io, tile_next, r);
        // ---- the block's input -> y = W1 x; then d/d(out) is requested (it travels under the gates) and parked in W
        f4 y[8];
#pragma unroll
        for (int d = 0; d < D; ++d) y[d] = f4{0.f, 0.f, 0.f, 0.f};
        {
            f4 x[NCH][8];
            in.finish(x, T);
            static_for<0, NCH>([&](auto ch) { cm_mix_one<ALG, TF::nstep(ch), GS1>(y, x[ch], ldsa + TF::w1(0, 0, ch)); });
        }
        CM_FENCE();
        CmPiece gp;
        gp.load((kLast ? io.gy + (size_t)(MODE == MODE_EDGE ? (long)T.i_dst : T.lrow) * ROW : io.plw_g1 + (size_t)T.lrow * ROW) + q * D);
        asm volatile("" ::: "memory");
        stamp(1);
        CB_MARK(1);
        // ---- forward again: z = gate(y) y -> P; R = WR z, s = (WL z + bL + gp(z, n(R))) / sqrt 2
        f4 R[8], s[8];
        float invMn;
        {
            f4 z[8];
            static_for<0, 4>([&](auto v) {
                float yy[D], zz[D], gate[4];
#pragma unroll
                for (int d = 0; d < D; ++d) yy[d] = y[d][int(v)];
                cm_silu<ALG>(yy, zz, gate, PP(v));
#pragma unroll
                for (int d = 0; d < D; ++d) z[d][int(v)] = zz[d];
                CM_FENCE();
            });
            cb_put(Pq, AD, z);
            {
                f4 g0[8];
                cm_unpack(g0, gp);
                if (!T.valid) {
#pragma unroll
                    for (int d = 0; d < D; ++d) g0[d] = f4{0.f, 0.f, 0.f, 0.f};
                }
                cb_put(Wq, AD, g0);
            }
            CM_FENCE();
#pragma unroll
            for (int d = 0; d < D; ++d) R[d] = s[d] = f4{0.f, 0.f, 0.f, 0.f};
            cm_mix_one<ALG, 4, GSC>(R, z, ldsa + TF::wc(0, 0, 0, 0));
            cm_mix_one<ALG, 4, GSC>(s, z, ldsa + TF::wc(1, 0, 0, 0));
            float nlsum = 0.f;
            static_for<0, 4>([&](auto v) {
                float zz[D], RR[D], LL[D], invden[4];
#pragma unroll
                for (int d = 0; d < D; ++d) { zz[d] = z[d][int(v)]; RR[d] = R[d][int(v)]; LL[d] = s[d][int(v)]; }
                nlsum += cm_gp_tail<ALG>(zz, RR, LL, invden, PP(v));
#pragma unroll
                for (int d = 0; d < D; ++d) s[d][int(v)] = LL[d];
                CM_FENCE();
            });
            invMn = fast_rcp(__builtin_fmaf(cm_q_sum(nlsum), 1.0f / float(C), kEps));
        }
        stamp(2);
        CB_MARK(2);
        // ---- MVLayerNorm backward: d/d(out) (W) and s -> ggp = d/d(gp + linear_left output), written back to W per channel
        {
            float S = 0.f, dot[4];
            static_for<0, 4>([&](auto v) {
                float gg[D];
                cb_get_chan<decltype(v)::value>(Wq, AD, gg);
                float a = 0.f;
#pragma unroll
                for (int d = 0; d < D; ++d) a = __builtin_fmaf(gg[d], s[d][int(v)], a);
                dot[v] = a;
                S = __builtin_fmaf(PP(v)[2], a, S);
            });
            const float gMn = -cm_q_sum(S) * invMn * invMn * (1.0f / float(C));
            float sums[8];
            static_for<0, 4>([&](auto v) {
                float gg[D];
                cb_get_chan<decltype(v)::value>(Wq, AD, gg);
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
                    gg[d] = __builtin_fmaf(k0, gg[d], gqs * (2.0f * qsf<ALG, d>) * s[d][int(v)]) * kInvSqrt2;
                });
                cb_put_chan<decltype(v)::value>(Wq, AD, gg);
                sums[2 * v] = dot[v] * invMn;   // d/d(la)
                sums[2 * v + 1] = gg[0];        // d/d(bL)
                CM_FENCE();
            });
            A.sm[0] += cb_rows_sum<8>(sums, r);
        }
        cb_sync();
        stamp(3);
        CB_MARK(3);
        // ---- d/d(linear_left weight) = ggp^T z, both operands from the slots
        cb_wgrad<ALG>(A.wl, Wq, Pq, AD);
        stamp(4);
        CB_MARK(4);
        // ---- geometric product + normalisation backward, per channel: R becomes d/dR, s becomes the product's d/dz
        {
            CbCollect<1> col;
            static_for<0, 4>([&](auto v) {
                float gg[D], zf[D], RR[D], gzz[D], gRR[D];
                cb_get_chan<decltype(v)::value>(Wq, AD, gg);
                cb_get_chan<decltype(v)::value>(Pq, AD, zf);
#pragma unroll
                for (int d = 0; d < D; ++d) RR[d] = R[d][int(v)];
                cb_gp_bwd<ALG>(gg, zf, RR, gzz, gRR, PP(v), [&](auto k, float val) {
                    col.template add<24 * decltype(v)::value + decltype(k)::value>(val, A.sm, r);
                });
#pragma unroll
                for (int d = 0; d < D; ++d) { s[d][int(v)] = gzz[d]; R[d][int(v)] = gRR[d]; }
                CM_FENCE();
            });
        }
        stamp(5);
        CB_MARK(5);
        // ---- d/dz += WL^T ggp (read back) + WR^T gR; gR -> W; d/d(linear_right weight) = gR^T z
        {
            f4 ggp[8];
            cb_get(Wq, AD, ggp);
            cb_mix_t<ALG, GSC>(s, ggp, ldst + TF::wc(1, 0, 0, 0));
        }
        cb_sync();
        cb_put(Wq, AD, R);
        cb_mix_t<ALG, GSC>(s, R, ldst + TF::wc(0, 0, 0, 0));
        cb_sync();
        // the input again (the gates' argument y = W1 x; d/dW1's operand): requested here, it travels under the MFMAs below
        asm volatile("" : "+v"(T.i_dst), "+v"(T.i_src), "+v"(T.i_perm), "+v"(T.lrow));
        in.issue(io, T, q);
        asm volatile("" ::: "memory");
        cb_wgrad<ALG>(A.wr, Wq, Pq, AD);
        cb_sync();
        cb_put(Wq, AD, s);   // d/dz waits in W meanwhile
        CM_FENCE();
        stamp(6);
        CB_MARK(6);
        f4 x[NCH][8];
        in.finish(x, T);
#pragma unroll
        for (int d = 0; d < D; ++d) y[d] = f4{0.f, 0.f, 0.f, 0.f};
        static_for<0, NCH>([&](auto ch) { cm_mix_one<ALG, TF::nstep(ch), GS1>(y, x[ch], ldsa + TF::w1(0, 0, ch)); });
        cb_put(Pq, AD, x[0]);   // z is no longer needed
        CM_FENCE();
        stamp(7);
        CB_MARK(7);
        // ---- MVSiLU backward, per channel, in place in W: d/dz becomes d/dy
        {
            CbCollect<7> col;
            float tail[4];
            static_for<0, 4>([&](auto v) {
                float yy[D], gzz[D], gyy[D], gs[9];
                cb_get_chan<decltype(v)::value>(Wq, AD, gzz);
#pragma unroll
                for (int d = 0; d < D; ++d) yy[d] = y[d][int(v)];
                yy[0] += PP(v)[0];   // MVLinear bias
                cm_silu_bwd<ALG>(gzz, yy, gyy, gs, PP(v));
                cb_put_chan<decltype(v)::value>(Wq, AD, gyy);
                static_for<0, 9>([&](auto k) {
                    constexpr int idx = 9 * decltype(v)::value + decltype(k)::value;
                    if constexpr (idx < 32) col.template add<idx>(gs[k], A.sm, r);
                    else tail[idx - 32] = gs[k];
                });
                CM_FENCE();
            });
            A.sm[9] += cb_rows_sum<4>(tail, r);
        }
        cb_sync();
        stamp(8);
        CB_MARK(8);
        // ---- d/d(MVLinear weight) = gy^T x, chunk by chunk through P
        static_for<0, NCH>([&](auto ch) {
            if constexpr (ch > 0) {
                cb_sync();
                cb_put(Pq, AD, x[ch]);
                cb_sync();
            }
            cb_wgrad<ALG>(A.w1[ch], Wq, Pq, AD);
        });
        stamp(9);
        CB_MARK(9);
        // ---- d/d(input) = W1^T gy (gy read back from W)
        cb_get(Wq, AD, s);
        auto gx_of = [&](auto ch, f4 (&gx)[8]) {
#pragma unroll
            for (int d = 0; d < D; ++d) gx[d] = f4{0.f, 0.f, 0.f, 0.f};
            cb_mix_t<ALG, GS1>(gx, s, ldst + TF::w1(0, 0, ch));
        };
        const CmTile<MODE> Tc = T;
        auto next_tile = [&]() {   // the next tile's rows leave in front of this tile's stores / atomics
            T = Tn;
            in.issue(io, T, q);
            asm volatile("" ::: "memory");
        };
        if constexpr (K > 0) {
            f4 gx[8];
            gx_of(IC<0>{}, gx);
            next_tile();
            if (Tc.valid) cm_store_piece(io.plw_g1 + (size_t)Tc.row * ROW + q * D, gx);
        } else if constexpr (MODE == MODE_EDGE) {
            if constexpr (NA > 0) {
                if (io.gx[1]) {
                    f4 gx[8];
                    gx_of(IC<1>{}, gx);
                    static_for<0, (NA + 3) / 4>([&](auto v) {
                        if (Tc.valid && q + 4 * v < NA) {
                            float* p = io.gx[1] + (size_t)Tc.i_perm * (NA * D) + (q + 4 * v) * D;
                            cl_st4(p, f4{gx[0][int(v)], gx[1][int(v)], gx[2][int(v)], gx[3][int(v)]});
                            cl_st4(p + 4, f4{gx[4][int(v)], gx[5][int(v)], gx[6][int(v)], gx[7][int(v)]});
                        }
                    });
                }
            }
            if (io.gx[0]) {
                f4 gx[8];
                gx_of(IC<0>{}, gx);
                if (io.row_store) {
                    next_tile();
                    if (Tc.valid) cm_store_piece(io.gx[0] + (size_t)Tc.lrow * ROW + q * D, gx);
                } else {
                    float* sc = Pq;   // both slots: a 16 x (ROW + 4) staging tile
                    static_assert(kCmRows * SS <= 2 * kCbSlot, "the staging tile fits the two slots");
                    cb_sync();
                    cm_store_piece(sc + r * SS + q * D, gx);
                    cb_sync();
                    next_tile();
                    cm_scatter<ROW, true>(sc, Tc.valid ? Tc.i_dst : -1, Tc.valid ? Tc.i_src : -1, io.gx[0], lane);
                    cb_sync();
                }
            } else {
                next_tile();
            }
        } else {
            if (io.gx[0]) {
                f4 gx[8];
                gx_of(IC<0>{}, gx);
                if (Tc.valid) {
                    if (io.resid_bwd) {
                        CmPiece res;
                        res.load(io.gy + (size_t)Tc.row * ROW + q * D);
                        f4 rr[8];
                        cm_unpack(rr, res);
#pragma unroll
                        for (int d = 0; d < D; ++d) gx[d] += rr[d];
                    }
                    cm_store_piece(io.gx[0] + (size_t)Tc.row * ROW + q * D, gx);
                }
            }
            if (io.gx[1]) {
                f4 gx[8];
                gx_of(IC<1>{}, gx);
#pragma unroll
                for (int d = 0; d < D; ++d) gx[d] *= Tc.scale;
                if (Tc.valid) cm_store_piece(io.gx[1] + (size_t)Tc.row * ROW + q * D, gx);
            }
            if constexpr (NA > 0) {
                if (io.gx[2]) {
                    f4 gx[8];
                    gx_of(IC<2>{}, gx);
                    static_for<0, (NA + 3) / 4>([&](auto v) {
                        if (Tc.valid && q + 4 * v < NA) {
                            float* p = io.gx[2] + (size_t)Tc.row * (NA * D) + (q + 4 * v) * D;
                            cl_st4(p, f4{gx[0][int(v)], gx[1][int(v)], gx[2][int(v)], gx[3][int(v)]});
                            cl_st4(p + 4, f4{gx[4][int(v)], gx[5][int(v)], gx[6][int(v)], gx[7][int(v)]});
                        }
                    });
                }
            }
            next_tile();
        }
        cb_sync();
        tile = tile_next;
        stamp(10);
        CB_MARK(10);
    }

#ifdef CSMPN_STAMPS
    stamp.acc[18 + (wave >> 2)] += __builtin_amdgcn_s_memtime() - kstart;   // duration of the tile loop: waves 0-3 | waves 4-7
#endif
    // ---- end of the block: waves 0-3 write their sums as four images of the slice (the slots of two waves hold one
    // image; every element of the slice has exactly one writer per wave), waves 4-7 add theirs on top, all threads add the
    // four images in a fixed order and write the workgroup's slice. Two rounds and three barriers (the first version had
    // the eight waves add into ONE image one after the other: 18 % of a 3-tiles-per-wave launch).
    __syncthreads();
    constexpr int IMG = 4 * kCbSlot;
    static_assert(PT::total <= IMG, "one image fits the slots of two waves");
    float* img = work + (wave & 3) * IMG;
    for (int round = 0; round < 2; ++round) {
        if ((wave >> 2) == round) {
            const bool add = round != 0;
            const int j = lane & 15, qq = lane >> 4;
            // weight tiles: D[i = 4 qq + v][j] = d/dW[orow(i)][first channel + orow(j)] (4 grades = one 16-byte vector)
            auto put_tile = [&](const f4 (&acc)[4], int base, int I, int coff, int width) {
                const int c = TF::orow(j);
                if (c < width) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        float* p0 = img + base + (TF::orow(4 * qq + v) * I + coff + c) * G;
                        const f4 val = f4{acc[0][v], acc[1][v], acc[2][v], acc[3][v]};
                        const f4 old = cl_ld4(p0);
                        cl_st4(p0, add ? old + val : val);
                    }
                }
            };
            static_for<0, NCH>([&](auto ch) {
                constexpr bool at = TF::attr(ch);
                put_tile(A.w1[ch], 0, TF::I, at ? TF::NSEG * C : 16 * ch, at ? NA : 16);
            });
            put_tile(A.wr, PT::pWR, C, 0, C);
            put_tile(A.wl, PT::pWL, C, 0, C);
#pragma unroll
            for (int g = 0; g < kCbGroups; ++g) {
                int v, idx;
                SM::decode(g, j, v, idx);
                if (v >= 0) {
                    float* p0 = img + PT::pS + PT::off(idx) + (4 * v + qq) * PT::stride(idx);
                    const float old = *p0;
                    *p0 = add ? old + A.sm[g] : A.sm[g];
                }
            }
        }
        __syncthreads();
    }
    float* part = io.rl_partials + (K == 0 ? 0 : (size_t)kClSliceCap * ClPart<ALG, C, CmTab<C, MODE, NA, 0>::I>::total) +
                  (size_t)blockIdx.x * PT::total;
    static_assert(PT::total % 4 == 0, "slice length");
    for (int e = 4 * threadIdx.x; e < PT::total; e += 4 * 64 * kCbWaves)
        cl_st4(part + e, (cl_ld4(work + e) + cl_ld4(work + IMG + e)) + (cl_ld4(work + 2 * IMG + e) + cl_ld4(work + 3 * IMG + e)));
    stamp(17);
        CB_MARK(17);
}

template <class ALG, int C, int MODE, int NBLK, int NA>
constexpr size_t cb_lds_bytes() {
    int tabs = CmTab<C, MODE, NA, 0>::total;
    if (NBLK > 1 && CmTab<C, MODE, NA, 1>::total > tabs) tabs = CmTab<C, MODE, NA, 1>::total;
    return sizeof(float) * (tabs + kCbWaves * 2 * kCbSlot + 4);   // + the tile counter
}

// The backward kernel: the blocks one after the other (last block first) in ONE launch, each with its own tables staged
// in front of it. A wave keeps its tiles from block to block: the hand-over rows d/d(block input) it reads in block
// k - 1 are the ones it wrote itself in block k (through L2; drained before the barrier) - no grid-wide synchronisation.
template <class ALG, int C, int MODE, int NBLK, int NA>
__global__ void __launch_bounds__(64 * kCbWaves) cemlp_cmb_kernel(const DevCemlp C_arg, const RowIO io_arg) {
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
    int* ctr = reinterpret_cast<int*>(smem + tabs + kCbWaves * 2 * kCbSlot);
    if constexpr (NBLK > 1) {
        cb_stage_block<ALG, C, CmTab<C, MODE, NA, 1>>(Cd.b[1], smem, threadIdx.x);
        if (threadIdx.x == 0) *ctr = 0;
        __syncthreads();
        stamp(0);
        cb_block<ALG, C, MODE, NBLK, NA, 1>(io, smem, smem + tabs, ctr, stamp);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's hand-over rows have left for L2
        __syncthreads();                                   // ... every wave's have, and every wave is done with block 1's tables
    }
    cb_stage_block<ALG, C, CmTab<C, MODE, NA, 0>>(Cd.b[0], smem, threadIdx.x);
    if (threadIdx.x == 0) *ctr = 0;
    __syncthreads();
    stamp(0);
    cb_block<ALG, C, MODE, NBLK, NA, 0>(io, smem, smem + tabs, ctr, stamp);
    stamp.flush(io.stamps, threadIdx.x & 63);
}

}  // namespace csmpn
