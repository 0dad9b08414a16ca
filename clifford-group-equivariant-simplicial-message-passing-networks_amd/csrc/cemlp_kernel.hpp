// Row-program kernel around the fused CEMLP block: gathers/concatenates the input
// rows of a 16-row tile into LDS, runs the CEMLP blocks, and writes / scatters the
// result. One template serves the three callers of the path:
//   MODE_PLAIN  CEMLP.forward on contiguous rows            (cegnn_utils.py:210-213)
//   MODE_EDGE   EGCL.message + PyG gather/scatter           (cegnn_utils.py:254-262,279)
//   MODE_NODE   EGCL.update (+ residual, + mean scale)      (cegnn_utils.py:264-275)
// BWD = recompute-forward + backward for the same three.
#pragma once
#include "cemlp_device.hpp"

namespace csmpn {

// cooperative gather of the concatenated input rows of one tile into LDS [16][D][CP]
template <class ALG>
__device__ void stage_input(const RowIO& io, float* tile, int RS, int CP, long row0, int tid, int nthreads) {
    constexpr int D = ALG::D;
    constexpr int DQ = D / 4;   // float4 chunks per channel
    int covered = 0;
    for (int s = 0; s < io.nseg; ++s) {
        const Seg& sg = io.seg[s];
        const int per_row = sg.ch * DQ;
        for (int e = tid; e < 16 * per_row; e += nthreads) {
            const int row = e / per_row, rem = e - row * per_row;
            const int ch = rem / DQ, dq = rem - ch * DQ;
            const long grow = row0 + row;
            f4 v = splat(0.f);
            if (grow < io.rows) {
                const long ra = sg.ia ? (long)sg.ia[grow] : grow;
                v = *reinterpret_cast<const f4*>(sg.a + (ra * sg.ch + ch) * D + dq * 4);
                if (sg.b) {
                    const long rb = sg.ib ? (long)sg.ib[grow] : grow;
                    v -= *reinterpret_cast<const f4*>(sg.b + (rb * sg.ch + ch) * D + dq * 4);
                }
                if (sg.deg) {
                    const int dg = sg.deg[ra];
                    v *= 1.0f / float(dg > 1 ? dg : 1);
                }
            }
            float* p = tile + row * RS + (dq * 4) * CP + sg.off + ch;
            p[0] = v.x; p[CP] = v.y; p[2 * CP] = v.z; p[3 * CP] = v.w;
        }
        covered = sg.off + sg.ch;
    }
    // zero the channel padding
    const int padc = CP - covered;
    if (padc > 0) {
        for (int e = tid; e < 16 * D * padc; e += nthreads) {
            const int row = e / (D * padc), rem = e - row * (D * padc);
            const int d = rem / padc, pc = rem - d * padc;
            tile[row * RS + d * CP + covered + pc] = 0.f;
        }
    }
}

// lane-layout tensor -> dense staging [16][nch*D] (row-major, channel, blade)
template <class ALG>
CSMPN_DEV void store_dense(const f4 (&t)[ALG::D], float* stage, int nch, int ch, int q) {
    constexpr int D = ALG::D;
    if (ch < nch) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            float* p = stage + ((4 * q + v) * nch + ch) * D;
#pragma unroll
            for (int d4 = 0; d4 < D; d4 += 4)
                *reinterpret_cast<f4*>(p + d4) = f4{t[d4][v], t[d4 + 1][v], t[d4 + 2][v], t[d4 + 3][v]};
        }
    }
}

// rows of a dense staged tile -> atomic adds into table rows selected by idx (sign * value).
// SEGMENTED (single-wave tiles, rows sorted by idx): equal consecutive targets are summed first.
template <class ALG, bool SEGMENTED>
__device__ void scatter_rows(const float* stage, int rowlen, const int* idx, long row0, long rows, float* table,
                             float sign, int tid, int nthreads) {
    constexpr int D = ALG::D;
    if constexpr (SEGMENTED) {
        constexpr int NPER = D / 4;   // rowlen <= 16*D  ->  <= D/4 elements per lane
        float acc[NPER];
#pragma unroll
        for (int j = 0; j < NPER; ++j) acc[j] = 0.f;
        int cur = -1;
        for (int row = 0; row < 16; ++row) {
            const long grow = row0 + row;
            if (grow >= rows) break;
            const int target = __builtin_amdgcn_readfirstlane(idx[grow]);
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
        for (int e = tid; e < 16 * rowlen; e += nthreads) {
            const int row = e / rowlen, f = e - row * rowlen;
            const long grow = row0 + row;
            if (grow < rows) atomicAdd(table + (long)idx[grow] * rowlen + f, sign * stage[e]);
        }
    }
}

// Forward: 512 threads (2 waves/SIMD at <=256 VGPRs). Backward keeps the whole forward
// state of a block live: 256 threads (1 wave/SIMD, up to 512 VGPRs) so that nothing spills
// (measured with 512-thread bounds: ~600 spilled VGPRs, >2 GB of scratch traffic per launch).
template <class ALG, int MODE, int VAR, bool BWD>
__global__ void __launch_bounds__(BWD ? 256 : 512) cemlp_kernel(const DevCemlp C, const RowIO io) {
    constexpr bool MULTI = kVarBarrier<VAR>;
    constexpr bool GT = VAR == VAR_GLOBAL;
    constexpr int D = ALG::D, G = ALG::G;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int MT = C.MT, RT = C.RT;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rt = wave / MT, mt = wave - rt * MT;
    const int tid_rt = mt * 64 + lane, nthr_rt = MT * 64;
    const int q = lane >> 4;
    float* mirror = smem;
    float* base;
    if constexpr (GT) base = C.gtiles + ((size_t)blockIdx.x * RT + rt) * C.tile_floats;
    else base = smem + C.mirror_floats + (size_t)rt * C.tile_floats;
    float* buf_in = base + C.off_in;
    auto buf_p = [&](int i) -> float* { return base + ((i & 1) ? C.off_p1 : C.off_p0); };
    float* buf_z = base + C.off_z;
    float* buf_g = base + C.off_g;
    float* red = base + C.off_red;
    constexpr bool in_lds = kVarMirror<VAR>;

    if constexpr (BWD) {
        if (in_lds) {
            for (int e = threadIdx.x; e < C.mirror_floats; e += blockDim.x) mirror[e] = 0.f;
        }
        __syncthreads();
    }

    const DevBlock& B0 = C.b[0];
    const DevBlock& BL = C.b[C.nblk - 1];
    const int RS0 = D * B0.CPi + 4;
    const long ntiles = (io.rows + 15) / 16;
    const long tiles_per_iter = (long)gridDim.x * RT;
    const long niter = (ntiles + tiles_per_iter - 1) / tiles_per_iter;

    for (long iter = 0; iter < niter; ++iter) {
        const long tile = iter * tiles_per_iter + (long)blockIdx.x * RT + rt;
        const long row0 = tile * 16;   // may be >= rows: fully masked tile
        stage_input<ALG>(io, buf_in, RS0, B0.CPi, row0, tid_rt, nthr_rt);
        tile_sync<VAR>();

        if constexpr (!BWD) {
            // ------------------------------------------------------------ forward
            const float* in = buf_in;
            f4 out[D];
            for (int k = 0; k < C.nblk; ++k) {
                const DevBlock& B = C.b[k];
                const LaneParams<ALG> lp = load_lane_params<ALG>(B, 16 * mt + (lane & 15));
                FwdState<ALG> S;
                block_forward<ALG, VAR>(B, lp, in, buf_z, red, MT, mt, lane, S, out);
                if (k + 1 < C.nblk) {
                    tile_sync<VAR>();
                    store_tile<ALG>(out, buf_p(k), D * B.CPo + 4, B.CPo, mt, lane);
                    tile_sync<VAR>();
                    in = buf_p(k);
                }
            }
            const int O = BL.O;
            const int c = 16 * mt + (lane & 15);
            if constexpr (MODE == MODE_EDGE) {
                tile_sync<VAR>();
                store_dense<ALG>(out, buf_g, O, c, q);
                tile_sync<VAR>();
                scatter_rows<ALG, !MULTI>(buf_g, O * D, io.dst, row0, io.rows, io.agg, 1.0f, tid_rt, nthr_rt);
                tile_sync<VAR>();
            } else {
                if (c < O) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const long grow = row0 + 4 * q + v;
                        if (grow < io.rows) {
                            float* p = io.y + (grow * O + c) * D;
#pragma unroll
                            for (int d4 = 0; d4 < D; d4 += 4) {
                                f4 val = f4{out[d4][v], out[d4 + 1][v], out[d4 + 2][v], out[d4 + 3][v]};
                                if (MODE == MODE_NODE && io.resid)
                                    val += *reinterpret_cast<const f4*>(io.resid + (grow * O + c) * D + d4);
                                *reinterpret_cast<f4*>(p + d4) = val;
                            }
                        }
                    }
                }
            }
        } else {
            // ------------------------------------------------------------ backward
            const int OL = BL.O;
            f4 gout[D];
            {
                const int c = 16 * mt + (lane & 15);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const long grow = row0 + 4 * q + v;
                    const bool ok = grow < io.rows && c < OL;
                    long srow = grow;
                    if (MODE == MODE_EDGE && ok) srow = io.dst[grow];
                    const float* p = io.gy + (srow * OL + c) * D;
#pragma unroll
                    for (int d4 = 0; d4 < D; d4 += 4) {
                        const f4 val = ok ? *reinterpret_cast<const f4*>(p + d4) : splat(0.f);
                        gout[d4][v] = val.x; gout[d4 + 1][v] = val.y; gout[d4 + 2][v] = val.z; gout[d4 + 3][v] = val.w;
                    }
                }
            }
            for (int k = C.nblk - 1; k >= 0; --k) {
                const DevBlock& B = C.b[k];
                // recompute the input tile of block k
                const float* in = buf_in;
                for (int j = 0; j < k; ++j) {
                    const DevBlock& Bj = C.b[j];
                    const LaneParams<ALG> lpj = load_lane_params<ALG>(Bj, 16 * mt + (lane & 15));
                    FwdState<ALG> Sj;
                    f4 oj[D];
                    block_forward<ALG, VAR>(Bj, lpj, in, buf_z, red, MT, mt, lane, Sj, oj);
                    tile_sync<VAR>();
                    store_tile<ALG>(oj, buf_p(j), D * Bj.CPo + 4, Bj.CPo, mt, lane);
                    tile_sync<VAR>();
                    in = buf_p(j);
                }
                const LaneParams<ALG> lp = load_lane_params<ALG>(B, 16 * mt + (lane & 15));
                f4 gy[D];
                {
                    FwdState<ALG> S;
                    f4 unused[D];
                    block_forward<ALG, VAR>(B, lp, in, buf_z, red, MT, mt, lane, S, unused);
                    block_backward<ALG, VAR>(B, lp, S, gout, in, buf_z, buf_g, red, mirror, MT, mt, lane, gy);
                }
                // transposed MVLinear: gx[i] = sum_o W1[o][i][g] gy[o]   (A = gy tile in LDS)
                const int RSo = D * B.CPo + 4;
                if (k > 0) {
#pragma unroll
                    for (int d = 0; d < D; ++d) gout[d] = splat(0.f);
                    if (mt < B.KKi)
                        linear_from_tile<ALG>(gout, buf_g, RSo, B.CPo, B.KKo, B.pbW1 + (size_t)mt * G * B.KKo * 64, lane);
                    tile_sync<VAR>();
                } else {
                    float* stage = buf_in;   // free: block_backward ended with a tile sync
                    const int Cs0 = io.seg[0].ch;
                    for (int it = mt; it < B.KKi; it += MT) {
                        f4 gx[D];
#pragma unroll
                        for (int d = 0; d < D; ++d) gx[d] = splat(0.f);
                        linear_from_tile<ALG>(gx, buf_g, RSo, B.CPo, B.KKo, B.pbW1 + (size_t)it * G * B.KKo * 64, lane);
                        const int i = 16 * it + (lane & 15);
                        // which input segment does channel i belong to
                        int s = -1;
                        for (int t = 0; t < io.nseg; ++t)
                            if (i >= io.seg[t].off && i < io.seg[t].off + io.seg[t].ch) s = t;
                        if (MODE == MODE_EDGE && s == 0) {
                            store_dense<ALG>(gx, stage, Cs0, i, q);
                        } else if (s >= 0 && io.gx[s]) {
                            const Seg& sg = io.seg[s];
                            const int ci = i - sg.off;
#pragma unroll
                            for (int v = 0; v < 4; ++v) {
                                const long grow = row0 + 4 * q + v;
                                if (grow < io.rows) {
                                    long trow = grow;
                                    if (MODE == MODE_EDGE) trow = io.perm[grow];   // edge_attr lives in original order
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
                    if constexpr (MODE == MODE_EDGE) {
                        tile_sync<VAR>();
                        if (io.gx[0]) {
                            scatter_rows<ALG, !MULTI>(stage, Cs0 * D, io.dst, row0, io.rows, io.gx[0], 1.0f, tid_rt, nthr_rt);
                            scatter_rows<ALG, false>(stage, Cs0 * D, io.src, row0, io.rows, io.gx[0], -1.0f, tid_rt, nthr_rt);
                        }
                    }
                    tile_sync<VAR>();
                }
            }
        }
    }

    if constexpr (BWD) {
        if (in_lds) {
            __syncthreads();
            for (int k = 0; k < C.nblk; ++k) flush_mirror<ALG>(C.b[k], mirror, threadIdx.x, blockDim.x);
        }
    }
}

// ---------------------------------------------------------------------------------
// weight packing into MFMA B-fragment order
struct PackSeg {
    const float* w;     // reference layout [O][I][G] or [O][I]
    f4* dst;
    int O, I, has_grades;
    int transposed;     // 0: frag(n = out, k = in); 1: frag(n = in, k = out)
    int NT, KK;         // tiles over n, k-blocks of 16
    int count;          // f4 elements = G*NT*KK*64
};
struct PackDesc { int nseg; int G; int total; int pad_; PackSeg seg[24]; };

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
