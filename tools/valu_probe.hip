// Probe (round 3): issue cost of the VALU forms the row programs are made of, at 1 / 2 / 4 waves per SIMD:
//  0: v_fmac_f32 (VOP2: d += a * b)               1: v_fma_f32 (VOP3, four distinct registers: d = a * b + c)
//  2: v_mul_f32                                   3: v_fma_f32 with a negated source (VOP3 modifier)
//  4: v_add_f32                                   5: chain of dependent v_fma (latency, one chain per wave)
//  6: v_pk_fma_f32 (two fp32 FMAs per lane)       7: v_pk_mul_f32       8: v_pk_add_f32   (ns per INSTRUCTION: 2 results each)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int ROLE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
    const int l = threadIdx.x;
    if constexpr (ROLE >= 6) {
        f2 a[16], t[16];
        for (int i = 0; i < 16; ++i) { a[i] = f2{seed + i * 0.01f + l * 1e-3f, seed - i * 0.01f}; t[i] = f2{1e-3f * i, 2e-3f * i}; }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if constexpr (ROLE == 6) t[i] = __builtin_elementwise_fma(a[i], a[(i + 5) & 15], t[i]);
                    else if constexpr (ROLE == 7) t[i] = a[i] * t[(i + 3) & 15];
                    else t[i] = a[i] + t[(i + 3) & 15];
                }
            }
        }
        f2 r = f2{0.f, 0.f};
        for (int i = 0; i < 16; ++i) r += t[i];
        out[(size_t)blockIdx.x * 256 + l] = r.x + r.y;
        return;
    }
    float a[16], t[16];
    for (int i = 0; i < 16; ++i) { a[i] = seed + i * 0.01f + l * 1e-3f; t[i] = 1e-3f * i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if constexpr (ROLE == 0) t[i] = __builtin_fmaf(a[i], a[(i + 5) & 15], t[i]);
                else if constexpr (ROLE == 1) t[i] = __builtin_fmaf(a[i], a[(i + 5) & 15], t[(i + 3) & 15]);
                else if constexpr (ROLE == 2) t[i] = a[i] * t[(i + 3) & 15];
                else if constexpr (ROLE == 3) t[i] = __builtin_fmaf(-a[i], a[(i + 5) & 15], t[(i + 3) & 15]);
                else if constexpr (ROLE == 4) t[i] = a[i] + t[(i + 3) & 15];
                else t[0] = __builtin_fmaf(t[0], a[i], a[(i + 5) & 15]);
            }
        }
    }
    float r = 0.f;
    for (int i = 0; i < 16; ++i) r += t[i];
    out[(size_t)blockIdx.x * 256 + l] = r;
}
template <int ROLE>
static void run(float* d, const char* name) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        hipLaunchKernelGGL(k<ROLE>, dim3(256 * wps), dim3(256), 0, 0, d, 10, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<ROLE>, dim3(256 * wps), dim3(256), 0, 0, d, iters, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s waves/SIMD=%d: %.3f ms -> %.2f ns per instr per SIMD\n", name, wps, ms, ms * 1e6 / (iters * 64.0 * wps));
    }
}
int main() {
    float* d; hipMalloc(&d, sizeof(float) * 256 * 8 * 256);
    run<0>(d, "v_fmac (d += a*b)");
    run<1>(d, "v_fma 4 distinct regs");
    run<2>(d, "v_mul");
    run<3>(d, "v_fma neg modifier");
    run<4>(d, "v_add");
    run<5>(d, "dependent v_fma chain");
    run<6>(d, "v_pk_fma_f32 (2 FMAs)");
    run<7>(d, "v_pk_mul_f32 (2 muls)");
    run<8>(d, "v_pk_add_f32 (2 adds)");
    return 0;
}
