// Standalone forms of the four small Clifford layers (C-ABI in include/csmpn_hip.h):
//   csmpn_mvsilu_forward / _backward        MVSiLU, invariant "mag2"          cegnn_utils.py:53-83
//   csmpn_mvnorm_forward / _backward        NormalizationLayer                cegnn_utils.py:34-51
//   csmpn_mvlayernorm_forward / _backward   MVLayerNorm                       cegnn_utils.py:86-96
//   csmpn_wgp_forward / _backward           the path-weighted geometric product of
//                                           SteerableGeometricProductLayer    cegnn_utils.py:126-152
// Inside a CEMLP these steps are fused into the row programs (cemlp_*.hpp); no reference model calls
// the layers on their own, so these kernels are the plain HBM-bound form: one thread per (row,
// channel) with the channel's D blades in registers (compile-time sign tables, algebra.hpp), rows of a
// workgroup contiguous in memory. Parameter gradients: per-workgroup sums in LDS, then one float
// atomic per parameter and workgroup.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/csmpn_hip.h"
#include "algebra.hpp"
#include "capi_common.hpp"

namespace {
using namespace csmpn;

constexpr float kEps = 1e-6f;       // cegnn_utils.py:5
constexpr float kSmooth = 1e-16f;   // cliffordalgebra.py:148
constexpr int kThreads = 256;
constexpr int kMaxChannels = 256;   // per-workgroup LDS sums: C * (parameters per channel) floats

__device__ __forceinline__ float sigmoid_sat(float x) { return 1.0f / (1.0f + __expf(-fmaxf(x, -87.0f))); }
__device__ __forceinline__ float smooth_abs_sqrt(float q) { return sqrtf(sqrtf(q * q + kSmooth)); }

template <class ALG, int d>
constexpr float qsf = float(ALG::t.qsign[d]);

template <class ALG>
__device__ __forceinline__ void load_row(float (&x)[ALG::D], const float* p) {
    constexpr int D = ALG::D;
#pragma unroll
    for (int d = 0; d < D; d += 4) {
        const float4 v = *reinterpret_cast<const float4*>(p + d);
        x[d] = v.x; x[d + 1] = v.y; x[d + 2] = v.z; x[d + 3] = v.w;
    }
}
template <class ALG>
__device__ __forceinline__ void store_row(const float (&x)[ALG::D], float* p) {
    constexpr int D = ALG::D;
#pragma unroll
    for (int d = 0; d < D; d += 4) *reinterpret_cast<float4*>(p + d) = make_float4(x[d], x[d + 1], x[d + 2], x[d + 3]);
}
// q_g = sum over the blades of grade g of qsign[d] x_d^2 (cliffordalgebra.py:119-146)
template <class ALG>
__device__ __forceinline__ void grade_q(const float (&x)[ALG::D], float (&q)[ALG::G]) {
    static_for<0, ALG::G>([&](auto g) {
        float s = 0.f;
        static_for<ALG::gstart(g), ALG::gstart(g) + ALG::gsize(g)>([&](auto d) { s += qsf<ALG, d> * x[d] * x[d]; });
        q[g] = s;
    });
}

// per-workgroup parameter-gradient sums: acc[c * NP + k] in LDS, flushed with one atomic per entry
struct ParamSums {
    float* lds;
    __device__ void zero(int n) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = 0.f;
        __syncthreads();
    }
    __device__ void add(int i, float v) { atomicAdd(lds + i, v); }
};

// ---------------------------------------------------------------------------------- MVSiLU
template <class ALG, bool BWD>
__global__ void __launch_bounds__(kThreads) mvsilu_kernel(const float* __restrict__ x, const float* __restrict__ a,
                                                          const float* __restrict__ b, const float* __restrict__ gy, long rows,
                                                          int C, float* __restrict__ out, float* __restrict__ ga,
                                                          float* __restrict__ gb) {
    constexpr int D = ALG::D, G = ALG::G;
    extern __shared__ float smem[];
    ParamSums ps{smem};
    if constexpr (BWD) ps.zero(2 * G * C);
    const long t = (long)blockIdx.x * kThreads + threadIdx.x;
    if (t < rows * C) {
        const int c = (int)(t % C);
        float xv[D], u[G], gate[G];
        load_row<ALG>(xv, x + t * D);
        grade_q<ALG>(xv, u);
        u[0] = xv[0];
#pragma unroll
        for (int g = 0; g < G; ++g) gate[g] = sigmoid_sat(a[c * G + g] * u[g] + b[c * G + g]);
        if constexpr (!BWD) {
            float y[D];
            static_for<0, D>([&](auto d) { y[d] = gate[ALG::grade(d)] * xv[d]; });
            store_row<ALG>(y, out + t * D);
        } else {
            float gv[D], gx[D];
            load_row<ALG>(gv, gy + t * D);
            static_for<0, G>([&](auto g) {
                float gg = 0.f;
                static_for<ALG::gstart(g), ALG::gstart(g) + ALG::gsize(g)>([&](auto d) { gg += gv[d] * xv[d]; });
                const float gpre = gg * gate[g] * (1.0f - gate[g]);
                ps.add(c * 2 * G + g, gpre * u[g]);
                ps.add(c * 2 * G + G + g, gpre);
                const float gu = gpre * a[c * G + g];
                static_for<ALG::gstart(g), ALG::gstart(g) + ALG::gsize(g)>([&](auto d) {
                    gx[d] = gv[d] * gate[g] + (g == 0 ? gu : gu * (2.0f * qsf<ALG, d>) * xv[d]);
                });
            });
            store_row<ALG>(gx, out + t * D);
        }
    }
    if constexpr (BWD) {
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * G * C; i += kThreads) {
            const int c = i / (2 * G), k = i % (2 * G);
            const float v = smem[i];
            if (v != 0.f) atomicAdd((k < G ? ga : gb) + c * G + (k % G), v);
        }
    }
}

// ---------------------------------------------------------------------------------- NormalizationLayer
template <class ALG, bool BWD>
__global__ void __launch_bounds__(kThreads) mvnorm_kernel(const float* __restrict__ x, const float* __restrict__ a,
                                                          const float* __restrict__ gy, long rows, int C,
                                                          float* __restrict__ out, float* __restrict__ ga) {
    constexpr int D = ALG::D, G = ALG::G;
    extern __shared__ float smem[];
    ParamSums ps{smem};
    if constexpr (BWD) ps.zero(G * C);
    const long t = (long)blockIdx.x * kThreads + threadIdx.x;
    if (t < rows * C) {
        const int c = (int)(t % C);
        float xv[D], q[G], inv[G], sg[G], nu[G];
        load_row<ALG>(xv, x + t * D);
        grade_q<ALG>(xv, q);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            sg[g] = sigmoid_sat(a[c * G + g]);
            nu[g] = smooth_abs_sqrt(q[g]);
            inv[g] = 1.0f / (sg[g] * (nu[g] - 1.0f) + 1.0f + kEps);
        }
        if constexpr (!BWD) {
            float y[D];
            static_for<0, D>([&](auto d) { y[d] = xv[d] * inv[ALG::grade(d)]; });
            store_row<ALG>(y, out + t * D);
        } else {
            float gv[D], gx[D];
            load_row<ALG>(gv, gy + t * D);
            static_for<0, G>([&](auto g) {
                float dot = 0.f;
                static_for<ALG::gstart(g), ALG::gstart(g) + ALG::gsize(g)>([&](auto d) { dot += gv[d] * xv[d]; });
                const float gden = -dot * inv[g] * inv[g];
                ps.add(c * G + g, gden * (nu[g] - 1.0f) * sg[g] * (1.0f - sg[g]));
                const float inu = 1.0f / nu[g];
                const float gq = gden * sg[g] * (0.5f * q[g]) * (inu * inu * inu);   // d nu / d q = q / (2 nu^3)
                static_for<ALG::gstart(g), ALG::gstart(g) + ALG::gsize(g)>([&](auto d) {
                    gx[d] = gv[d] * inv[g] + gq * (2.0f * qsf<ALG, d>) * xv[d];
                });
            });
            store_row<ALG>(gx, out + t * D);
        }
    }
    if constexpr (BWD) {
        __syncthreads();
        for (int i = threadIdx.x; i < G * C; i += kThreads) {
            const float v = smem[i];
            if (v != 0.f) atomicAdd(ga + i, v);
        }
    }
}

// ---------------------------------------------------------------------------------- MVLayerNorm
// A workgroup covers kThreads / C whole rows (thread = (local row, channel)); the mean over the
// channels of a row goes through LDS.
template <class ALG, bool BWD>
__global__ void __launch_bounds__(kThreads) mvlayernorm_kernel(const float* __restrict__ x, const float* __restrict__ a,
                                                               const float* __restrict__ gy, long rows, int C,
                                                               float* __restrict__ out, float* __restrict__ ga) {
    constexpr int D = ALG::D;
    extern __shared__ float smem[];
    const int rpb = kThreads / C;                 // rows per workgroup
    float* rsum = smem;                           // [rpb] sum of the channel norms
    float* rdot = smem + rpb;                     // [rpb] sum_c a_c <gy_c, x_c>   (backward)
    float* psum = smem + 2 * rpb;                 // [C]   parameter-gradient sums (backward)
    for (int i = threadIdx.x; i < 2 * rpb + C; i += kThreads) smem[i] = 0.f;
    __syncthreads();
    const int lr = threadIdx.x / C, c = threadIdx.x % C;
    const long row = (long)blockIdx.x * rpb + lr;
    const bool act = lr < rpb && row < rows;
    float xv[D], gv[D];
    float q = 0.f, nl = 0.f, dot = 0.f, ac = 0.f;
    if (act) {
        ac = a[c];
        load_row<ALG>(xv, x + (row * C + c) * D);
        static_for<0, D>([&](auto d) { q += qsf<ALG, d> * xv[d] * xv[d]; });
        nl = smooth_abs_sqrt(q);
        atomicAdd(rsum + lr, nl);
        if constexpr (BWD) {
            load_row<ALG>(gv, gy + (row * C + c) * D);
            static_for<0, D>([&](auto d) { dot += gv[d] * xv[d]; });
            atomicAdd(rdot + lr, ac * dot);
        }
    }
    __syncthreads();
    if (act) {
        const float invM = 1.0f / (rsum[lr] / float(C) + kEps);
        if constexpr (!BWD) {
            float y[D];
            static_for<0, D>([&](auto d) { y[d] = ac * xv[d] * invM; });
            store_row<ALG>(y, out + (row * C + c) * D);
        } else {
            atomicAdd(psum + c, dot * invM);
            const float gM = -rdot[lr] * invM * invM / float(C);      // d/d(norm of one channel)
            const float inl = 1.0f / nl;
            const float gq = gM * (0.5f * q) * (inl * inl * inl);
            float gx[D];
            static_for<0, D>([&](auto d) { gx[d] = ac * gv[d] * invM + gq * (2.0f * qsf<ALG, d>) * xv[d]; });
            store_row<ALG>(gx, out + (row * C + c) * D);
        }
    }
    if constexpr (BWD) {
        __syncthreads();
        for (int i = threadIdx.x; i < C; i += kThreads) {
            const float v = psum[i];
            if (v != 0.f) atomicAdd(ga + i, v);
        }
    }
}

// ---------------------------------------------------------------------------------- weighted product
//   out[j] = sum_{(i,k) -> j} sign(i,k) w[c][path(grade i, grade j, grade k)] z[i] r[k]
template <class ALG, bool BWD>
__global__ void __launch_bounds__(kThreads) wgp_kernel(const float* __restrict__ z, const float* __restrict__ r,
                                                       const float* __restrict__ w, const float* __restrict__ gy, long rows,
                                                       int C, float* __restrict__ out, float* __restrict__ gz,
                                                       float* __restrict__ gr, float* __restrict__ gw) {
    constexpr int D = ALG::D, P = ALG::P;
    extern __shared__ float smem[];
    ParamSums ps{smem};
    if constexpr (BWD) ps.zero(P * C);
    const long t = (long)blockIdx.x * kThreads + threadIdx.x;
    if (t < rows * C) {
        const int c = (int)(t % C);
        float zv[D], rv[D];
        load_row<ALG>(zv, z + t * D);
        load_row<ALG>(rv, r + t * D);
        if constexpr (!BWD) {
            float y[D];
#pragma unroll
            for (int d = 0; d < D; ++d) y[d] = 0.f;
            static_for<0, P>([&](auto p) {
                constexpr int gi = ALG::t.path_g[p][0], gj = ALG::t.path_g[p][1], gk = ALG::t.path_g[p][2];
                constexpr int j0 = ALG::gstart(gj), nj = ALG::gsize(gj);
                float tmp[nj];
#pragma unroll
                for (int u = 0; u < nj; ++u) tmp[u] = 0.f;
                static_for<ALG::gstart(gi), ALG::gstart(gi) + ALG::gsize(gi)>([&](auto i) {
                    static_for<ALG::gstart(gk), ALG::gstart(gk) + ALG::gsize(gk)>([&](auto k) {
                        constexpr int j = ALG::t.out[i][k];
                        if constexpr (j >= j0 && j < j0 + nj) tmp[j - j0] += float(ALG::t.sign[i][k]) * zv[i] * rv[k];
                    });
                });
                const float wp = w[c * P + p];
#pragma unroll
                for (int u = 0; u < nj; ++u) y[j0 + u] += wp * tmp[u];
            });
            store_row<ALG>(y, out + t * D);
        } else {
            float gv[D], gzv[D], grv[D];
            load_row<ALG>(gv, gy + t * D);
#pragma unroll
            for (int d = 0; d < D; ++d) { gzv[d] = 0.f; grv[d] = 0.f; }
            static_for<0, P>([&](auto p) {
                constexpr int gi = ALG::t.path_g[p][0], gj = ALG::t.path_g[p][1], gk = ALG::t.path_g[p][2];
                constexpr int i0 = ALG::gstart(gi), ni = ALG::gsize(gi);
                constexpr int j0 = ALG::gstart(gj), nj = ALG::gsize(gj);
                constexpr int k0 = ALG::gstart(gk), nk = ALG::gsize(gk);
                float U[ni], V[nk];   // unweighted d/dz, d/dr of this path
#pragma unroll
                for (int u = 0; u < ni; ++u) U[u] = 0.f;
#pragma unroll
                for (int u = 0; u < nk; ++u) V[u] = 0.f;
                static_for<0, ni>([&](auto ii) {
                    static_for<0, nk>([&](auto kk) {
                        constexpr int i = i0 + ii, k = k0 + kk;
                        constexpr int j = ALG::t.out[i][k];
                        if constexpr (j >= j0 && j < j0 + nj) {
                            const float sg = float(ALG::t.sign[i][k]) * gv[j];
                            U[ii] += sg * rv[k];
                            V[kk] += sg * zv[i];
                        }
                    });
                });
                const float wp = w[c * P + p];
                float gwp = 0.f;
#pragma unroll
                for (int u = 0; u < ni; ++u) { gzv[i0 + u] += wp * U[u]; gwp += zv[i0 + u] * U[u]; }
#pragma unroll
                for (int u = 0; u < nk; ++u) grv[k0 + u] += wp * V[u];
                ps.add(c * P + p, gwp);
            });
            store_row<ALG>(gzv, gz + t * D);
            store_row<ALG>(grv, gr + t * D);
        }
    }
    if constexpr (BWD) {
        __syncthreads();
        for (int i = threadIdx.x; i < P * C; i += kThreads) {
            const float v = smem[i];
            if (v != 0.f) atomicAdd(gw + i, v);
        }
    }
}

// ---------------------------------------------------------------------------------- dispatch
enum Op { OP_SILU, OP_NORM, OP_LNORM, OP_WGP };

struct Args {
    const float *x, *r, *p0, *p1, *gy;
    long rows;
    int C;
    float *out, *out2, *g0, *g1;
};

template <class ALG, bool BWD>
hipError_t launch(Op op, const Args& A, hipStream_t st) {
    constexpr int G = ALG::G, P = ALG::P;
    const long n = A.rows * A.C;
    const unsigned grid = (unsigned)((n + kThreads - 1) / kThreads);
    switch (op) {
        case OP_SILU:
            hipLaunchKernelGGL((mvsilu_kernel<ALG, BWD>), dim3(grid), dim3(kThreads), BWD ? sizeof(float) * 2 * G * A.C : 0, st, A.x,
                               A.p0, A.p1, A.gy, A.rows, A.C, A.out, A.g0, A.g1);
            break;
        case OP_NORM:
            hipLaunchKernelGGL((mvnorm_kernel<ALG, BWD>), dim3(grid), dim3(kThreads), BWD ? sizeof(float) * G * A.C : 0, st, A.x, A.p0,
                               A.gy, A.rows, A.C, A.out, A.g0);
            break;
        case OP_LNORM: {
            const int rpb = kThreads / A.C;
            hipLaunchKernelGGL((mvlayernorm_kernel<ALG, BWD>), dim3((unsigned)((A.rows + rpb - 1) / rpb)), dim3(kThreads),
                               sizeof(float) * (2 * rpb + A.C), st, A.x, A.p0, A.gy, A.rows, A.C, A.out, A.g0);
            break;
        }
        case OP_WGP:
            hipLaunchKernelGGL((wgp_kernel<ALG, BWD>), dim3(grid), dim3(kThreads), BWD ? sizeof(float) * P * A.C : 0, st, A.x, A.r, A.p0,
                               A.gy, A.rows, A.C, A.out, A.out, A.out2, A.g0);
            break;
    }
    return hipGetLastError();
}

int run(const float* metric, int n, Op op, bool bwd, const Args& A, void* stream) {
    if (!metric || n < 2 || n > 5) return csmpn_fail(CSMPN_ERR_UNSUPPORTED, "standalone layers: n=%d not in 2..5", n);
    unsigned neg = 0;
    for (int i = 0; i < n; ++i) {
        if (metric[i] == -1.0f) neg |= 1u << i;
        else if (metric[i] != 1.0f) return csmpn_fail(CSMPN_ERR_UNSUPPORTED, "standalone layers: metric entries must be +-1");
    }
    if (A.rows < 0 || A.C < 1 || A.C > kMaxChannels) return csmpn_fail(CSMPN_ERR_INVALID, "standalone layers: bad rows / channels (1..%d)", kMaxChannels);
    if (A.rows == 0) return CSMPN_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = hipErrorInvalidValue;
#define CSMPN_LAYER_ALG(N_, NEG_)                                                              \
    if (n == N_ && neg == NEG_) e = bwd ? launch<Alg<N_, NEG_>, true>(op, A, st) : launch<Alg<N_, NEG_>, false>(op, A, st); else
    CSMPN_LAYER_ALG(2, 0u) CSMPN_LAYER_ALG(3, 0u) CSMPN_LAYER_ALG(4, 0u) CSMPN_LAYER_ALG(4, 0x8u) CSMPN_LAYER_ALG(5, 0u)
    CSMPN_LAYER_ALG(5, 0x10u)
        return csmpn_fail(CSMPN_ERR_UNSUPPORTED, "standalone layers: no kernels for this signature (n=%d, negative mask 0x%x)", n, neg);
#undef CSMPN_LAYER_ALG
    if (e != hipSuccess) return csmpn_fail(CSMPN_ERR_HIP, "standalone layer launch: %s", hipGetErrorString(e));
    return CSMPN_OK;
}
}  // namespace

extern "C" {

int csmpn_mvsilu_forward(const float* metric_host, int n, const float* x, const float* a, const float* b, int64_t rows,
                         int32_t channels, float* y, void* stream) {
    if (!x || !a || !b || !y) return csmpn_fail(CSMPN_ERR_INVALID, "csmpn_mvsilu_forward: null pointer");
    Args A{x, nullptr, a, b, nullptr, (long)rows, channels, y, nullptr, nullptr, nullptr};
    return run(metric_host, n, OP_SILU, false, A, stream);
}
int csmpn_mvsilu_backward(const float* metric_host, int n, const float* x, const float* a, const float* b, const float* gy,
                          int64_t rows, int32_t channels, float* gx, float* g_a, float* g_b, void* stream) {
    if (!x || !a || !b || !gy || !gx || !g_a || !g_b) return csmpn_fail(CSMPN_ERR_INVALID, "csmpn_mvsilu_backward: null pointer");
    Args A{x, nullptr, a, b, gy, (long)rows, channels, gx, nullptr, g_a, g_b};
    return run(metric_host, n, OP_SILU, true, A, stream);
}
int csmpn_mvnorm_forward(const float* metric_host, int n, const float* x, const float* a, int64_t rows, int32_t channels,
                         float* y, void* stream) {
    if (!x || !a || !y) return csmpn_fail(CSMPN_ERR_INVALID, "csmpn_mvnorm_forward: null pointer");
    Args A{x, nullptr, a, nullptr, nullptr, (long)rows, channels, y, nullptr, nullptr, nullptr};
    return run(metric_host, n, OP_NORM, false, A, stream);
}
int csmpn_mvnorm_backward(const float* metric_host, int n, const float* x, const float* a, const float* gy, int64_t rows,
                          int32_t channels, float* gx, float* g_a, void* stream) {
    if (!x || !a || !gy || !gx || !g_a) return csmpn_fail(CSMPN_ERR_INVALID, "csmpn_mvnorm_backward: null pointer");
    Args A{x, nullptr, a, nullptr, gy, (long)rows, channels, gx, nullptr, g_a, nullptr};
    return run(metric_host, n, OP_NORM, true, A, stream);
}
int csmpn_mvlayernorm_forward(const float* metric_host, int n, const float* x, const float* a, int64_t rows,
                              int32_t channels, float* y, void* stream) {
    if (!x || !a || !y) return csmpn_fail(CSMPN_ERR_INVALID, "csmpn_mvlayernorm_forward: null pointer");
    Args A{x, nullptr, a, nullptr, nullptr, (long)rows, channels, y, nullptr, nullptr, nullptr};
    return run(metric_host, n, OP_LNORM, false, A, stream);
}
int csmpn_mvlayernorm_backward(const float* metric_host, int n, const float* x, const float* a, const float* gy,
                               int64_t rows, int32_t channels, float* gx, float* g_a, void* stream) {
    if (!x || !a || !gy || !gx || !g_a) return csmpn_fail(CSMPN_ERR_INVALID, "csmpn_mvlayernorm_backward: null pointer");
    Args A{x, nullptr, a, nullptr, gy, (long)rows, channels, gx, nullptr, g_a, nullptr};
    return run(metric_host, n, OP_LNORM, true, A, stream);
}
int csmpn_wgp_forward(const float* metric_host, int n, const float* z, const float* r, const float* weight, int64_t rows,
                      int32_t channels, float* y, void* stream) {
    if (!z || !r || !weight || !y) return csmpn_fail(CSMPN_ERR_INVALID, "csmpn_wgp_forward: null pointer");
    Args A{z, r, weight, nullptr, nullptr, (long)rows, channels, y, nullptr, nullptr, nullptr};
    return run(metric_host, n, OP_WGP, false, A, stream);
}
int csmpn_wgp_backward(const float* metric_host, int n, const float* z, const float* r, const float* weight,
                       const float* gy, int64_t rows, int32_t channels, float* gz, float* gr, float* g_weight, void* stream) {
    if (!z || !r || !weight || !gy || !gz || !gr || !g_weight) return csmpn_fail(CSMPN_ERR_INVALID, "csmpn_wgp_backward: null pointer");
    Args A{z, r, weight, nullptr, gy, (long)rows, channels, gz, gr, g_weight, nullptr};
    return run(metric_host, n, OP_WGP, true, A, stream);
}

}  // extern "C"
