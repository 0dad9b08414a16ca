// Probe (round 3): issue rate of v_fmac_f32 vs v_fmac_f32_dpp (row_ror) vs mov_dpp + fmac at 1, 2, 4, 8 waves
// per SIMD, and of ds_read_b128 broadcast reads beside them. Decides the channel-mixing form of the
// (row, channel)-per-lane kernels (cemlp_cl.hpp).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

#define FMAC_DPP(acc, x, w, ROT) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_ror:" #ROT " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(w))

// role 0: plain fma, 8 chains; 1: asm v_fmac_f32_dpp, 8 chains; 2: update_dpp + fma (compiler's choice); 3: like 1 with one
// ds_read_b128 per 8 fmacs feeding the weights
template <int ROLE>
__global__ void __launch_bounds__(256) rate_kernel(float* out, int iters, float seed) {
    __shared__ f4 tab[64];
    const int l = threadIdx.x & 63;
    tab[l] = f4{1.0001f, 0.9999f, 1.0002f, 0.9998f};
    __syncthreads();
    float t[8], x[8];
    for (int i = 0; i < 8; ++i) { t[i] = seed + i + l * 1e-3f; x[i] = 1e-6f * (i + 1); }
    float w = 1.0001f;
    for (int it = 0; it < iters; ++it) {
        if constexpr (ROLE == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) t[k] = __builtin_fmaf(x[k], w, t[k]);
        } else if constexpr (ROLE == 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                FMAC_DPP(t[0], x[0], w, 2); FMAC_DPP(t[1], x[1], w, 2); FMAC_DPP(t[2], x[2], w, 2); FMAC_DPP(t[3], x[3], w, 2);
                FMAC_DPP(t[4], x[4], w, 2); FMAC_DPP(t[5], x[5], w, 2); FMAC_DPP(t[6], x[6], w, 2); FMAC_DPP(t[7], x[7], w, 2);
            }
        } else if constexpr (ROLE == 2) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int xi = __builtin_bit_cast(int, x[k]);
                    const float xr = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(xi, xi, 0x122, 0xF, 0xF, true));
                    t[k] = __builtin_fmaf(xr, w, t[k]);
                }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const f4 wv = tab[(l + u + it) & 63];
                FMAC_DPP(t[0], x[0], wv.x, 2); FMAC_DPP(t[1], x[1], wv.y, 2); FMAC_DPP(t[2], x[2], wv.y, 2); FMAC_DPP(t[3], x[3], wv.y, 2);
                FMAC_DPP(t[4], x[4], wv.z, 2); FMAC_DPP(t[5], x[5], wv.z, 2); FMAC_DPP(t[6], x[6], wv.z, 2); FMAC_DPP(t[7], x[7], wv.w, 2);
            }
        }
    }
    float r = 0.f;
    for (int i = 0; i < 8; ++i) r += t[i];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = r;
}

template <int ROLE>
static void run(float* d, const char* name) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int wps = 1; wps <= 8; wps *= 2) {
        hipLaunchKernelGGL(rate_kernel<ROLE>, dim3(256 * wps), dim3(256), 0, 0, d, 10, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(rate_kernel<ROLE>, dim3(256 * wps), dim3(256), 0, 0, d, iters, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s waves/SIMD=%d: %.3f ms -> %.2f ns per fma-instr per SIMD\n", name, wps, ms, ms * 1e6 / (iters * 64.0 * wps));
    }
}

int main() {
    float* d; hipMalloc(&d, sizeof(float) * 256 * 8 * 256);
    run<0>(d, "v_fma (plain)");
    run<1>(d, "v_fmac_f32_dpp (asm)");
    run<2>(d, "update_dpp + fma (compiler)");
    run<3>(d, "fmac_dpp + ds_read_b128/8");
    return 0;
}
