// Row-program kernel around the fused CEMLP block: gathers/concatenates the input
// rows of a row tile into LDS, runs the CEMLP blocks, and writes / scatters the
// result. One template serves the three callers of the path:
//   MODE_PLAIN  CEMLP.forward on contiguous rows            (cegnn_utils.py:210-213)
//   MODE_EDGE   EGCL.message + PyG gather/scatter           (cegnn_utils.py:254-262,279)
//   MODE_NODE   EGCL.update (+ residual, + mean scale)      (cegnn_utils.py:264-275)
// BWD = recompute-forward + backward for the same three.
#pragma once
#include "cemlp_device.hpp"

namespace csmpn {

// The gathered row indices of one tile (3 x R ints in LDS): [0,R) first index array of
// segment 0 (edge: dst), [R,2R) its second one (edge: src), [2R,3R) segment 1's (edge: perm);
// -1 marks rows beyond the end. Loaded one tile AHEAD into registers (TileIdx) so that the
// index -> row dependent load chain is off the critical path.
struct TileIdx { int a, b, c; };

template <int R>
CSMPN_DEV TileIdx load_tile_indices(const RowIO& io, long row0, int tid) {
    TileIdx t{-1, -1, -1};
    const long grow = row0 + tid;
    if (tid < R && grow < io.rows) {
        const Seg& s0 = io.seg[0];
        t.a = t.b = t.c = (int)grow;
        if (s0.ia) t.a = s0.ia[grow];
        if (s0.b && s0.ib) t.b = s0.ib[grow];
        if (io.nseg > 1) { const int* ia1 = io.seg[1].ia; if (ia1) t.c = ia1[grow]; }
    }
    return t;
}
template <int R>
CSMPN_DEV void store_tile_indices(const TileIdx& t, int* tidx, int tid) {
    if (tid < R) { tidx[tid] = t.a; tidx[R + tid] = t.b; tidx[2 * R + tid] = t.c; }
}

// cooperative gather of the concatenated input rows of one tile into LDS [channel][D][R].
// Thread mapping: 16 consecutive threads take the (up to 16) 16-byte pieces of ONE row, the
// thread's row group rg = tid / 16 takes rows rg, rg + RG, rg + 2 RG, ...: a load instruction
// covers whole rows (256-byte rows of 8 channels x 8 blades: 4 rows = 8 cache lines per wave
// instruction; a row-per-lane mapping touches 32-64 different lines per instruction and is
// bound by the address unit), and everything that depends on the piece (channel, blade
// quarter, LDS column) is computed ONCE per thread and segment - per slot there is one index
// read, one 64-bit multiply-add and the load (the former element-wise decomposition spent
// ~100 VALU instructions per slot: 37 % of all instructions of the edge forward).
// The loads of ALL segments are in flight before the first one is consumed. The transposing
// LDS writes are at most 2-way bank conflicts, free for ds_write_b32.
template <class ALG, int H, int NS>
CSMPN_DEV void stage_input(const RowIO& io, float* tile, const int* tidx, int CP, long row0, int tid, int nthreads) {
    using GE = Geo<ALG, H>;
    constexpr int D = ALG::D, R = GE::R, CS = GE::CS;
    constexpr int DQ = D / 4;            // float4 pieces per channel (power of two)
    constexpr int U = R / 4;             // row slots per thread (single-wave tile); NS = segments of the mode
    const int pl = tid & 15, rg = tid >> 4, RG = nthreads >> 4;
    int covered = 0, most = 0;
    for (int s = 0; s < io.nseg; ++s) {
        const int ppr = io.seg[s].ch * DQ;
        most = ppr > most ? ppr : most;
        covered = io.seg[s].off + io.seg[s].ch;
    }
    for (int pb = 0; pb < most; pb += 16) {
        const int p = pb + pl;
        f4 v[NS][U], w[U];
        int dg[U];   // in-degrees of the (single) segment with a mean scale
#pragma unroll
        for (int u = 0; u < U; ++u) { w[u] = splat(0.f); dg[u] = 1; }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s < io.nseg) {
                const Seg& sg = io.seg[s];
                const int ppr = sg.ch * DQ;            // pieces per row
                const bool act = p < ppr;
                const int* ta = s == 0 ? tidx : tidx + 2 * R;
                const float* pa = sg.a + (act ? p : 0) * 4;
                const float* pbp = sg.b ? sg.b + (act ? p : 0) * 4 : nullptr;
                const unsigned stride = (unsigned)ppr * 4u;   // floats per row
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int row = rg + RG * u;
                    const long grow = row0 + row;
                    const bool ok = act && row < R && grow < io.rows;
                    // the LDS index copies exist for the (at most three) index arrays of
                    // segments 0/1; segments without an index array are read by row number.
                    // Invalid slots read row 0 of the table (always there) and are zeroed.
                    const unsigned ra = ok ? (sg.ia ? (unsigned)ta[row < R ? row : 0] : (unsigned)grow) : 0u;
                    v[s][u] = *reinterpret_cast<const f4*>(pa + (size_t)ra * stride);
                    if (s == 0 && pbp) {
                        const unsigned rb = ok ? (sg.ib ? (unsigned)tidx[R + (row < R ? row : 0)] : (unsigned)grow) : 0u;
                        w[u] = *reinterpret_cast<const f4*>(pbp + (size_t)rb * stride);
                    }
                    if (sg.deg) dg[u] = sg.deg[ra];
                    if (!ok) v[s][u] = splat(0.f);
                }
            }
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s < io.nseg) {
                const Seg& sg = io.seg[s];
                const int ppr = sg.ch * DQ;
                if (p < ppr) {
                    const int dq = p % DQ, ch = p / DQ;
                    float* q = tile + (sg.off + ch) * CS + (dq * 4) * R + rg;
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int row = rg + RG * u;
                        if (row < R) {
                            const long grow = row0 + row;
                            f4 val = v[s][u];
                            if (s == 0 && sg.b) val = grow < io.rows ? val - w[u] : splat(0.f);
                            if (sg.deg) val *= 1.0f / float(dg[u] > 1 ? dg[u] : 1);
                            float* qq = q + RG * u;
                            qq[0] = val.x; qq[R] = val.y; qq[2 * R] = val.z; qq[3 * R] = val.w;
                        }
                    }
                }
            }
        }
    }
    // zero the channel padding
    for (int e = tid; e < (CP - covered) * D * R; e += nthreads) {
        const int row2 = e % R, rem = e / R;
        const int d = rem % D, pc = rem / D;
        tile[(covered + pc) * CS + d * R + row2] = 0.f;
    }
}

// contiguous rows [rows][ch][D] (a saved block input) -> LDS tile [channel][D][R]; a tile's
// rows are one contiguous span, read fully coalesced; same thread mapping as stage_input
template <class ALG, int H>
CSMPN_DEV void stage_plain(const float* src, int ch, long rows, float* tile, int CP, long row0, int tid, int nthreads) {
    using GE = Geo<ALG, H>;
    constexpr int D = ALG::D, R = GE::R, CS = GE::CS;
    constexpr int DQ = D / 4;
    constexpr int U = R / 4;
    const int pl = tid & 15, rg = tid >> 4, RG = nthreads >> 4;
    const int ppr = ch * DQ;
    for (int pb = 0; pb < ppr; pb += 16) {
        const int p = pb + pl;
        const bool act = p < ppr;
        const float* pa = src + (act ? p : 0) * 4;
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int row = rg + RG * u;
            const long grow = row0 + row;
            const bool ok = act && row < R && grow < rows;
            v[u] = *reinterpret_cast<const f4*>(pa + (size_t)(ok ? grow : 0) * (size_t)(ppr * 4));
            if (!ok) v[u] = splat(0.f);
        }
        if (act) {
            const int dq = p % DQ, c = p / DQ;
            float* q = tile + c * CS + (dq * 4) * R + rg;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (rg + RG * u < R) {
                    float* qq = q + RG * u;
                    qq[0] = v[u].x; qq[R] = v[u].y; qq[2 * R] = v[u].z; qq[3 * R] = v[u].w;
                }
            }
        }
    }
    for (int e = tid; e < (CP - ch) * D * R; e += nthreads) {
        const int row2 = e % R, rem = e / R;
        const int d = rem % D, pc = rem / D;
        tile[(ch + pc) * CS + d * R + row2] = 0.f;
    }
}

// lane-layout tensor -> dense staging [R][nch*D] (row-major, channel, blade)
template <class ALG, int H>
CSMPN_DEV void store_dense(const f4 (&t)[ALG::D], float* stage, int nch, int ch, const Geo<ALG, H>& ge) {
    constexpr int D = ALG::D;
    if (ch < nch) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            float* p = stage + ((ge.r0 + v) * nch + ch) * D;
#pragma unroll
            for (int d4 = 0; d4 < D; d4 += 4)
                *reinterpret_cast<f4*>(p + d4) = f4{t[d4][v], t[d4 + 1][v], t[d4 + 2][v], t[d4 + 3][v]};
        }
    }
}

// rows of a dense staged tile -> rows row0 .. of a [rows, rowlen] table (deterministic mode: the edge rows are not
// scattered; a segmented reduction sums them afterwards in a fixed order)
template <class ALG, int H>
__device__ void store_rows_dense(const float* stage, int rowlen, long row0, long rows, float* table, int tid, int nthreads) {
    constexpr int R = 16 * H;
    const long nr = rows - row0 < R ? rows - row0 : R;
    for (long e = tid; e < nr * rowlen; e += nthreads) table[row0 * rowlen + e] = stage[e];
}

// rows of a dense staged tile -> atomic adds into table rows selected by lidx (LDS copy of
// the tile's row indices, -1 = masked row), sign * value.
// SEGMENTED (single-wave tiles, rows sorted by index): equal consecutive targets are summed first.
template <class ALG, int H, bool SEGMENTED>
__device__ void scatter_rows(const float* stage, int rowlen, const int* lidx, float* table, float sign, int tid,
                             int nthreads) {
    constexpr int D = ALG::D, R = 16 * H;
    if constexpr (SEGMENTED) {
        constexpr int NPER = D / 4;   // rowlen <= 16*D  ->  <= D/4 elements per lane
        float acc[NPER];
#pragma unroll
        for (int j = 0; j < NPER; ++j) acc[j] = 0.f;
        int cur = -1;
        for (int row = 0; row < R; ++row) {
            const int target = __builtin_amdgcn_readfirstlane(lidx[row]);
            if (target < 0) break;
            if (target != cur) {
                if (cur >= 0) {
#pragma unroll
                    for (int j = 0; j < NPER; ++j) {
                        const int e = tid + 64 * j;
                        if (e < rowlen) atomicAdd(table + (long)cur * rowlen + e, sign * acc[j]);
                        acc[j] = 0.f;
                    }
                }
                cur = target;
            }
#pragma unroll
            for (int j = 0; j < NPER; ++j) {
                const int e = tid + 64 * j;
                if (e < rowlen) acc[j] += stage[row * rowlen + e];
            }
        }
        if (cur >= 0) {
#pragma unroll
            for (int j = 0; j < NPER; ++j) {
                const int e = tid + 64 * j;
                if (e < rowlen) atomicAdd(table + (long)cur * rowlen + e, sign * acc[j]);
            }
        }
    } else {
        for (int row = 0; row < R; ++row) {
            const long target = lidx[row];
            if (target < 0) break;
            for (int f = tid; f < rowlen; f += nthreads) atomicAdd(table + target * rowlen + f, sign * stage[row * rowlen + f]);
        }
    }
}

// Single-wave tiles: the same scatter from registers. The staged tile is read once (every
// ds_read in flight together), the row targets travel through SGPRs (v_readlane with constant
// lane), so the loop over the rows has no LDS round trip and only scalar branches. Adds the
// rows to table[add_idx[row]] (rows sorted by that index: equal consecutive targets are summed
// first) and, when sub_idx is given, subtracts them from table[sub_idx[row]] (unsorted).
template <class ALG, int H>
CSMPN_DEV void scatter_tile(const float* stage, int rowlen, const int* add_idx, const int* sub_idx, float* table,
                            int lane) {
    constexpr int R = 16 * H;
    const int ta = add_idx[lane & (R - 1)];
    const int tb = sub_idx ? sub_idx[lane & (R - 1)] : -1;
    // 64 columns of the staged rows at a time, 16 rows per batch of LDS reads (one element per
    // lane and row in registers)
    for (int e0 = 0; e0 < rowlen; e0 += 64) {
        const int e = e0 + lane;
        const bool in = e < rowlen;
        const float* col = stage + (in ? e : 0);
        auto flush = [&](int target, float a) {
            if (target >= 0 && in) atomicAdd(table + (long)target * rowlen + e, a);
        };
        float acc = 0.f;
        int cur = __builtin_amdgcn_readlane(ta, 0);
        static_for<0, R / 16>([&](auto gg) {
            constexpr int r0 = 16 * decltype(gg)::value;
            float val[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) val[i] = col[(r0 + i) * rowlen];
            static_for<0, 16>([&](auto rr) {
                constexpr int row = r0 + decltype(rr)::value;
                const int t = __builtin_amdgcn_readlane(ta, row);
                if (t != cur) {
                    flush(cur, acc);
                    cur = t;
                    acc = 0.f;
                }
                acc += val[decltype(rr)::value];
            });
            if (sub_idx) {
                static_for<0, 16>([&](auto rr) {
                    constexpr int row = r0 + decltype(rr)::value;
                    flush(__builtin_amdgcn_readlane(tb, row), -val[decltype(rr)::value]);
                });
            }
        });
        flush(cur, acc);
    }
}

// copy one block's dense weights (reference layout [O][I][G] or [O][I]) into the LDS store
// [g][O][IP]; padding columns were zero-filled by the caller
__device__ inline void stage_weight(const float* w, float* dst, int O, int I, int IP, int G, bool grades, int tid,
                                    int nthreads) {
    const int per = I * (grades ? G : 1);
    for (int e = tid; e < O * per; e += nthreads) {
        const int o = e / per, rem = e - o * per;
        const int i = grades ? rem / G : rem, g = grades ? rem - i * G : 0;
        dst[(g * O + o) * IP + i] = w[e];
    }
}

// park / fetch a lane-layout tensor in a wave-private LDS area (one b128 per blade per lane,
// conflict-free): used to keep the incoming gradient out of the registers while a block's
// forward is recomputed
template <class ALG>
CSMPN_DEV void park(const f4 (&t)[ALG::D], float* area, int lane) {
#pragma unroll
    for (int d = 0; d < ALG::D; ++d) *reinterpret_cast<f4*>(area + (d * 64 + lane) * 4) = t[d];
}
template <class ALG>
CSMPN_DEV void unpark(f4 (&t)[ALG::D], const float* area, int lane) {
#pragma unroll
    for (int d = 0; d < ALG::D; ++d) t[d] = *reinterpret_cast<const f4*>(area + (d * 64 + lane) * 4);
}

// concatenated input segments per mode: CEMLP rows / [h_i - h_j | edge_attr] / [h | agg | node_attr]
template <int MODE> constexpr int kModeSegs = MODE == MODE_PLAIN ? 1 : (MODE == MODE_EDGE ? 2 : 3);

// Forward: 512 threads (2 waves/SIMD at <=256 VGPRs). Backward keeps the whole forward
// state of a block live: 256 threads (1 wave/SIMD, up to 512 VGPRs).
template <class ALG, int MODE, int VAR, int H, bool BWD>
__global__ void __launch_bounds__(BWD ? 256 : 512) cemlp_kernel(const DevCemlp C_arg, const RowIO io_arg) {
    // Read the descriptors in place from the kernarg segment (constant address space, scalar
    // loads). Indexing the by-value arguments dynamically (C.b[k]) makes the compiler copy
    // the whole struct to scratch and turns every field access into a scratch load.
    typedef const char __attribute__((address_space(4))) * KArgPtr;
    const KArgPtr ka = (KArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr size_t kIoOffset = (sizeof(DevCemlp) + alignof(RowIO) - 1) / alignof(RowIO) * alignof(RowIO);
    const DevCemlp& C = *(const DevCemlp*)(const char*)ka;
    const RowIO& io = *(const RowIO*)(const char*)(ka + kIoOffset);
    (void)C_arg; (void)io_arg;
    using GE = Geo<ALG, H>;
    constexpr int D = ALG::D, G = ALG::G, R = GE::R, NW = GE::NW;
    constexpr bool MULTI = kVarBarrier<VAR>;
    constexpr bool GT = VAR == VAR_GLOBAL;
    constexpr bool in_lds = kVarMirror<VAR>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int MT = C.MT, RT = C.RT;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rt = wave / MT, mt = wave - rt * MT;
    const int tid_rt = mt * 64 + lane, nthr_rt = MT * 64;
    const size_t det_goff = (size_t)blockIdx.x * (size_t)C.det_slice_floats;   // deterministic mode: this workgroup's accumulator copy
    const GE ge(lane);
    constexpr bool WLDS = VAR == VAR_WAVE;
    float* mirror = smem;
    float* wstore = smem + C.mirror_floats;
    float* base;
    if constexpr (GT) base = C.gtiles + ((size_t)blockIdx.x * RT + rt) * C.tile_floats;
    else base = smem + C.mirror_floats + C.wstore_floats + (size_t)rt * C.tile_floats;
    float* buf_in = base + C.off_in;
    auto buf_p = [&](int i) -> float* { return base + ((i & 1) ? C.off_p1 : C.off_p0); };
    float* buf_z = base + C.off_z;
    float* buf_g = base + C.off_g;
    float* red = base + C.off_red;
    int* tidx = reinterpret_cast<int*>(base + C.off_idx);

    if constexpr ((BWD && in_lds) || WLDS) {
        // zero the gradient mirror and the weight store (its padding columns stay zero)
        for (int e = threadIdx.x; e < C.mirror_floats + C.wstore_floats; e += blockDim.x) smem[e] = 0.f;
        __syncthreads();
    }
    if constexpr (WLDS) {
        for (int k = 0; k < C.nblk; ++k) {
            const DevBlock& B = C.b[k];
            const WOff wo = wstore_offsets(B.O, B.CPi, B.CPo, G, ALG::P, B.w1_sub != 0);
            float* ws = wstore + B.lds_woff;
            stage_weight(B.W1, ws + wo.W1, B.O, B.I, B.CPi, G, B.w1_sub != 0, threadIdx.x, blockDim.x);
            stage_weight(B.WR, ws + wo.WR, B.O, B.O, B.CPo, G, true, threadIdx.x, blockDim.x);
            stage_weight(B.WL, ws + wo.WL, B.O, B.O, B.CPo, G, true, threadIdx.x, blockDim.x);
            for (int e = threadIdx.x; e < B.O; e += blockDim.x) {
                ws[wo.b1 + e] = B.has_b1 ? B.b1[e] : 0.f;
                ws[wo.bL + e] = B.bL[e];
                ws[wo.la + e] = B.la[e];
            }
            for (int e = threadIdx.x; e < B.O * G; e += blockDim.x) {
                ws[wo.sa + e] = B.sa[e];
                ws[wo.sb + e] = B.sb[e];
                ws[wo.sg + e] = sigmoidf(B.an[e]);
            }
            for (int e = threadIdx.x; e < B.O * ALG::P; e += blockDim.x) ws[wo.w + e] = B.w[e];
        }
        __syncthreads();
    }

    auto lane_params = [&](const DevBlock& B, int c) -> LaneParams<ALG> {
        if constexpr (WLDS) {
            const WOff wo = wstore_offsets(B.O, B.CPi, B.CPo, G, ALG::P, B.w1_sub != 0);
            return load_lane_params_lds<ALG>(B, wstore + B.lds_woff, wo, c);
        } else {
            return load_lane_params<ALG>(B, c);
        }
    };
    const DevBlock& B0 = C.b[0];
    const DevBlock& BL = C.b[C.nblk - 1];
    const long ntiles = (io.rows + R - 1) / R;
    const long tiles_per_iter = (long)gridDim.x * RT;
    const long niter = (ntiles + tiles_per_iter - 1) / tiles_per_iter;

    // element offset of block k's saved input (k >= 1) inside io.save / io.saved
    auto save_off = [&](int kb) -> size_t {
        size_t o = 0;
        for (int j = 0; j + 1 < kb; ++j) o += (size_t)C.b[j].O;
        return o * (size_t)io.rows * D;
    };
    const bool use_saved = BWD && io.saved != nullptr && C.nblk > 1;
    // Phased backward (round 3; plan.C.phased, needs the saved block inputs): the blocks one after the other, last first,
    // each over ALL row tiles of the workgroup - the LDS mirror holds ONE block's gradient tensors (flushed between the
    // phases), which leaves room for a second row tile per workgroup where both blocks' mirrors did not (md17's 32
    // channels: 2 instead of 4 waves per CU). A wave keeps its tiles from phase to phase: the rows d/d(block input) it
    // reads in phase k - 1 are the ones it wrote itself in phase k (io.plw_g1, laid out like the saved inputs).
    const bool phased = BWD && C.phased != 0 && use_saved;
  for (int ph = phased ? C.nblk - 1 : 0; ph >= 0; --ph) {
    TileIdx nidx = load_tile_indices<R>(io, ((long)blockIdx.x * RT + rt) * R, tid_rt);

    for (long iter = 0; iter < niter; ++iter) {
        const long tile = iter * tiles_per_iter + (long)blockIdx.x * RT + rt;
        const long row0 = tile * R;   // may be >= rows: fully masked tile
        store_tile_indices<R>(nidx, tidx, tid_rt);
        tile_sync<VAR>();
        // the next tile's indices travel while this tile computes
        nidx = load_tile_indices<R>(io, row0 + tiles_per_iter * R, tid_rt);
        if (use_saved && (!phased || ph > 0)) {
            const int kin = phased ? ph : C.nblk - 1;
            const DevBlock& Bl = C.b[kin];
            stage_plain<ALG, H>(io.saved + save_off(kin), Bl.I, io.rows, buf_in, Bl.CPi, row0, tid_rt, nthr_rt);
        } else {
            stage_input<ALG, H, kModeSegs<MODE>>(io, buf_in, tidx, B0.CPi, row0, tid_rt, nthr_rt);
        }
        tile_sync<VAR>();
        ge.stamp(0);

        if constexpr (!BWD) {
            // ------------------------------------------------------------ forward
            const float* in = buf_in;
            f4 out[D];
            for (int k = 0; k < C.nblk; ++k) {
                const DevBlock& B = C.b[k];
                const LaneParams<ALG> lp = lane_params(B, NW * mt + ge.cn);
                FwdState<ALG> S;
                block_forward<ALG, H, VAR, BWD>(B, lp, in, buf_z, red, wstore, MT, mt, ge, S, out);
                if (k + 1 < C.nblk) {
                    tile_sync<VAR>();
                    store_tile<ALG, H>(out, buf_p(k), B.CPo, mt, ge);
                    if (io.save) {   // keep the next block's input for the backward
                        const int cs = NW * mt + ge.cn;
                        if (cs < B.O) {
                            float* sp = io.save + save_off(k + 1);
#pragma unroll
                            for (int v = 0; v < 4; ++v) {
                                const long grow = row0 + ge.r0 + v;
                                if (grow < io.rows) {
#pragma unroll
                                    for (int d4 = 0; d4 < D; d4 += 4)
                                        *reinterpret_cast<f4*>(sp + (grow * B.O + cs) * D + d4) =
                                            f4{out[d4][v], out[d4 + 1][v], out[d4 + 2][v], out[d4 + 3][v]};
                                }
                            }
                        }
                    }
                    tile_sync<VAR>();
                    in = buf_p(k);
                }
            }
            const int O = BL.O;
            const int c = NW * mt + ge.cn;
            if constexpr (MODE == MODE_EDGE) {
                tile_sync<VAR>();
                store_dense<ALG, H>(out, buf_g, O, c, ge);
                tile_sync<VAR>();
                if (io.row_store) store_rows_dense<ALG, H>(buf_g, O * D, row0, io.rows, io.agg, tid_rt, nthr_rt);
                else if constexpr (!MULTI) scatter_tile<ALG, H>(buf_g, O * D, tidx, nullptr, io.agg, lane);
                else scatter_rows<ALG, H, false>(buf_g, O * D, tidx, io.agg, 1.0f, tid_rt, nthr_rt);
                tile_sync<VAR>();
                ge.stamp(18);
            } else {
                if (c < O) {
                    // all residual loads first, then all stores: a load behind a store would
                    // wait for the store's acknowledgement (one vmcnt for both on gfx9)
                    f4 res[4][D / 4];
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const long grow = row0 + ge.r0 + v;
                        const bool ok = MODE == MODE_NODE && io.resid && grow < io.rows;
#pragma unroll
                        for (int d4 = 0; d4 < D; d4 += 4)
                            res[v][d4 / 4] = ok ? *reinterpret_cast<const f4*>(io.resid + (grow * O + c) * D + d4) : splat(0.f);
                    }
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const long grow = row0 + ge.r0 + v;
                        if (grow < io.rows) {
                            float* p = io.y + (grow * O + c) * D;
#pragma unroll
                            for (int d4 = 0; d4 < D; d4 += 4)
                                *reinterpret_cast<f4*>(p + d4) =
                                    f4{out[d4][v], out[d4 + 1][v], out[d4 + 2][v], out[d4 + 3][v]} + res[v][d4 / 4];
                        }
                    }
                }
            }
        } else {
            // ------------------------------------------------------------ backward
            const int OL = BL.O;
            const bool handed = phased && ph + 1 < C.nblk;   // d/d(out of block ph) = the rows phase ph + 1 wrote
            const int OG = handed ? C.b[ph].O : OL;
            const float* gsrc = handed ? io.plw_g1 + save_off(ph + 1) : io.gy;
            f4 gout[D];
            {
                const int c = NW * mt + ge.cn;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const long grow = row0 + ge.r0 + v;
                    const bool ok = grow < io.rows && c < OG;
                    long srow = grow;
                    if (MODE == MODE_EDGE && ok && !handed) srow = tidx[ge.r0 + v];
                    const float* p = gsrc + (srow * OG + c) * D;
#pragma unroll
                    for (int d4 = 0; d4 < D; d4 += 4) {
                        const f4 val = ok ? *reinterpret_cast<const f4*>(p + d4) : splat(0.f);
                        gout[d4][v] = val.x; gout[d4 + 1][v] = val.y; gout[d4 + 2][v] = val.z; gout[d4 + 3][v] = val.w;
                    }
                }
            }
            // single-wave tiles: the incoming gradient waits in the (still unused) g buffer
            // while the block forward is recomputed, instead of occupying 4*D VGPRs
            constexpr bool PARK = VAR == VAR_WAVE;
            if constexpr (PARK) park<ALG>(gout, buf_g, lane);
            ge.stamp(1);
            for (int k = phased ? ph : C.nblk - 1; k >= (phased ? ph : 0); --k) {
                const DevBlock& B = C.b[k];
                const float* in = buf_in;
                if (use_saved && k + 1 < C.nblk && !phased) {   // (phased: the tile's staging above brought this block's input)
                    // this block's input replaces the previous one in the single input buffer
                    if (k == 0) stage_input<ALG, H, kModeSegs<MODE>>(io, buf_in, tidx, B0.CPi, row0, tid_rt, nthr_rt);
                    else stage_plain<ALG, H>(io.saved + save_off(k), B.I, io.rows, buf_in, B.CPi, row0, tid_rt, nthr_rt);
                    tile_sync<VAR>();
                    ge.stamp(2);
                }
                // without saved inputs: recompute the input tile of block k from the tile's input
                for (int j = 0; !use_saved && j < k; ++j) {
                    const DevBlock& Bj = C.b[j];
                    const LaneParams<ALG> lpj = lane_params(Bj, NW * mt + ge.cn);
                    FwdState<ALG> Sj;
                    f4 oj[D];
                    block_forward<ALG, H, VAR, BWD>(Bj, lpj, in, buf_z, red, wstore, MT, mt, ge, Sj, oj);
                    tile_sync<VAR>();
                    store_tile<ALG, H>(oj, buf_p(j), Bj.CPo, mt, ge);
                    tile_sync<VAR>();
                    ge.stamp(2);
                    in = buf_p(j);
                }
                const LaneParams<ALG> lp = lane_params(B, NW * mt + ge.cn);
                f4 gy[D];
                {
                    FwdState<ALG> S;
                    f4 unused[D];
                    // 16-row tiles only: the 32-row kernels are register-bound (4 waves per CU fit their LDS)
                    const bool share = H == 1 && C.share_inz != 0;
                    block_forward<ALG, H, VAR, BWD>(B, lp, in, buf_z, red, wstore, MT, mt, ge, S, unused, share);
                    if constexpr (PARK) { tile_sync<VAR>(); unpark<ALG>(gout, buf_g, lane); tile_sync<VAR>(); }
                    block_backward<ALG, H, VAR>(B, lp, S, gout, in, buf_z, buf_g, red, mirror, wstore, MT, mt, ge, gy, share, det_goff);
                    if (share) {
                        // z lived in the input buffer: bring the block's input tile back for the MVLinear weight gradient
                        if (k == 0) stage_input<ALG, H, kModeSegs<MODE>>(io, buf_in, tidx, B0.CPi, row0, tid_rt, nthr_rt);
                        else stage_plain<ALG, H>(io.saved + save_off(k), B.I, io.rows, buf_in, B.CPi, row0, tid_rt, nthr_rt);
                        tile_sync<VAR>();
                        block_w1_grad<ALG, H, VAR>(B, gy, buf_in, mirror, mt, ge, det_goff);
                        tile_sync<VAR>();
                    }
                }
                // transposed MVLinear: gx[i] = sum_o W1[o][i][g] gy[o]   (A = gy tile in LDS)
                const WOff wo = wstore_offsets(B.O, B.CPi, B.CPo, G, ALG::P, B.w1_sub != 0);
                const WSrc sW1t{B.pbW1, wstore + B.lds_woff + wo.W1, B.O, B.CPi, B.w1_sub};
                if (k > 0) {
#pragma unroll
                    for (int d = 0; d < D; ++d) gout[d] = splat(0.f);
                    if (mt < B.NTi)
                        linear_from_tile<ALG, H, WLDS, true>(gout, buf_g, B.CPo, B.KKo, sW1t, mt, ge);
                    tile_sync<VAR>();
                    if (phased) {   // the rows go to the hand-over region; this wave reads them back in phase k - 1
                        const int ci = NW * mt + ge.cn;
                        if (mt < B.NTi && ci < B.I) {
                            float* hp = io.plw_g1 + save_off(k);
#pragma unroll
                            for (int v = 0; v < 4; ++v) {
                                const long grow = row0 + ge.r0 + v;
                                if (grow < io.rows) {
#pragma unroll
                                    for (int d4 = 0; d4 < D; d4 += 4)
                                        *reinterpret_cast<f4*>(hp + (grow * B.I + ci) * D + d4) =
                                            f4{gout[d4][v], gout[d4 + 1][v], gout[d4 + 2][v], gout[d4 + 3][v]};
                                }
                            }
                        }
                    }
                    if constexpr (PARK) { if (!phased) { park<ALG>(gout, buf_g, lane); tile_sync<VAR>(); } }
                    ge.stamp(17);
                } else {
                    float* stage = buf_in;   // free: block_backward ended with a tile sync
                    const int Cs0 = io.seg[0].ch;
                    for (int it = mt; it < B.NTi; it += MT) {
                        // skip input-channel tiles none of whose segments wants a gradient
                        // (attributes without grad): uniform per wave
                        bool wanted = false;
                        for (int t = 0; t < io.nseg; ++t) {
                            const bool overlaps = io.seg[t].off < NW * (it + 1) && io.seg[t].off + io.seg[t].ch > NW * it;
                            wanted |= overlaps && ((MODE == MODE_EDGE && t == 0) || io.gx[t] != nullptr);
                        }
                        if (!wanted) continue;
                        f4 gx[D];
#pragma unroll
                        for (int d = 0; d < D; ++d) gx[d] = splat(0.f);
                        linear_from_tile<ALG, H, WLDS, true>(gx, buf_g, B.CPo, B.KKo, sW1t, it, ge);
                        const int i = NW * it + ge.cn;
                        // which input segment does channel i belong to
                        int s = -1;
                        for (int t = 0; t < io.nseg; ++t)
                            if (i >= io.seg[t].off && i < io.seg[t].off + io.seg[t].ch) s = t;
                        if (MODE == MODE_EDGE && s == 0) {
                            store_dense<ALG, H>(gx, stage, Cs0, i, ge);
                        } else if (s >= 0 && io.gx[s]) {
                            const Seg& sg = io.seg[s];
                            const int ci = i - sg.off;
#pragma unroll
                            for (int v = 0; v < 4; ++v) {
                                const long grow = row0 + ge.r0 + v;
                                if (grow < io.rows) {
                                    long trow = grow;
                                    if (MODE == MODE_EDGE) trow = tidx[2 * R + ge.r0 + v];   // edge_attr lives in original order
                                    float scale = 1.0f;
                                    if (sg.deg) { const int dg = sg.deg[grow]; scale = 1.0f / float(dg > 1 ? dg : 1); }
                                    float* p = io.gx[s] + (trow * sg.ch + ci) * D;
#pragma unroll
                                    for (int d4 = 0; d4 < D; d4 += 4) {
                                        f4 val = f4{gx[d4][v], gx[d4 + 1][v], gx[d4 + 2][v], gx[d4 + 3][v]} * scale;
                                        if (MODE == MODE_NODE && s == 0 && io.resid_bwd)
                                            val += *reinterpret_cast<const f4*>(io.gy + (grow * OL + ci) * D + d4);
                                        *reinterpret_cast<f4*>(p + d4) = val;
                                    }
                                }
                            }
                        }
                    }
                    ge.stamp(17);
                    if constexpr (MODE == MODE_EDGE) {
                        tile_sync<VAR>();
                        if (io.gx[0] && io.row_store) {
                            store_rows_dense<ALG, H>(stage, Cs0 * D, row0, io.rows, io.gx[0], tid_rt, nthr_rt);
                        } else if (io.gx[0]) {
                            if constexpr (!MULTI) {
                                scatter_tile<ALG, H>(stage, Cs0 * D, tidx, tidx + R, io.gx[0], lane);
                            } else {
                                scatter_rows<ALG, H, false>(stage, Cs0 * D, tidx, io.gx[0], 1.0f, tid_rt, nthr_rt);
                                scatter_rows<ALG, H, false>(stage, Cs0 * D, tidx + R, io.gx[0], -1.0f, tid_rt, nthr_rt);
                            }
                        }
                    }
                    tile_sync<VAR>();
                    ge.stamp(18);
                }
            }
        }
    }

    if (phased) {
        // end of phase ph: its hand-over rows have left for L2, its gradient sums leave the mirror, which is zeroed for the
        // next block (all blocks share offset 0)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if constexpr (BWD && in_lds) {
            flush_mirror<ALG>(C.b[ph], mirror, threadIdx.x, blockDim.x, (size_t)blockIdx.x * (size_t)C.det_slice_floats);
            __syncthreads();
            for (int e = threadIdx.x; e < C.mirror_floats; e += blockDim.x) smem[e] = 0.f;
        }
        __syncthreads();
    }
  }   // phases

    if constexpr (BWD && in_lds) {
        __syncthreads();
        // deterministic mode: this workgroup's private copy of the accumulators (one row tile per workgroup: every word
        // has one writing wave, in tile order); det_reduce_kernel adds the copies in a fixed order
        if (!phased)
            for (int k = 0; k < C.nblk; ++k)
                flush_mirror<ALG>(C.b[k], mirror, threadIdx.x, blockDim.x, (size_t)blockIdx.x * (size_t)C.det_slice_floats);
    }
#ifdef CSMPN_STAMPS
    ge.stamp(19);
    if (io.stamps && lane == 0) {
        for (int i = 0; i < GE::kStampSlots; ++i) atomicAdd(io.stamps + i, ge.acc[i]);
        atomicAdd(io.stamps + GE::kStampSlots, 1ull);
    }
#endif
}

// ---------------------------------------------------------------------------------
// weight packing into MFMA B-fragment order (kernel in capi.hip)
struct PackSeg {
    const float* w;     // reference layout [O][I][G] or [O][I]
    f4* dst;
    int O, I, has_grades;
    int transposed;     // 0: frag(n = out, k = in); 1: frag(n = in, k = out)
    int NT, KK;         // N tiles (of 16/H channels), k-blocks of 16
    int count;          // f4 elements = NT*KK*H*G*64
};
struct PackDesc { int nseg; int G; int H; int total; PackSeg seg[24]; };

// ---------------------------------------------------------------------------------
// standalone geometric product (cliffordalgebra.py:44-54), one row per thread
template <class ALG, bool BWD>
__global__ void gp_kernel(const float* a, const float* b, const float* gout, float* out, float* ga, float* gb, long rows) {
    constexpr int D = ALG::D;
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    float x[D], y[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { x[d] = a[r * D + d]; y[d] = b[r * D + d]; }
    if constexpr (!BWD) {
        float o[D];
#pragma unroll
        for (int d = 0; d < D; ++d) o[d] = 0.f;
        static_for<0, D>([&](auto i) {
            static_for<0, D>([&](auto k) {
                constexpr int j = ALG::t.out[decltype(i)::value][decltype(k)::value];
                constexpr float sg = float(ALG::t.sign[decltype(i)::value][decltype(k)::value]);
                o[j] += sg * x[decltype(i)::value] * y[decltype(k)::value];
            });
        });
#pragma unroll
        for (int d = 0; d < D; ++d) out[r * D + d] = o[d];
    } else {
        float go[D], gx[D], gyv[D];
#pragma unroll
        for (int d = 0; d < D; ++d) { go[d] = gout[r * D + d]; gx[d] = 0.f; gyv[d] = 0.f; }
        static_for<0, D>([&](auto i) {
            static_for<0, D>([&](auto k) {
                constexpr int j = ALG::t.out[decltype(i)::value][decltype(k)::value];
                constexpr float sg = float(ALG::t.sign[decltype(i)::value][decltype(k)::value]);
                gx[decltype(i)::value] += sg * go[j] * y[decltype(k)::value];
                gyv[decltype(k)::value] += sg * go[j] * x[decltype(i)::value];
            });
        });
#pragma unroll
        for (int d = 0; d < D; ++d) { ga[r * D + d] += gx[d]; gb[r * D + d] += gyv[d]; }
    }
}

}  // namespace csmpn
