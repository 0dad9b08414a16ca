// Do v_mfma_f32_16x16x4_f32 and fp32 VALU instructions of DIFFERENT waves of one SIMD run side by side (gfx950)?
// One workgroup of 8 waves per CU (two per SIMD). Modes: all waves MFMA | all waves VALU | waves 0-3 MFMA + waves 4-7 VALU (each doing
// the same per-wave work as in the pure runs). If the pipes are independent the mixed run takes max(t_mfma, t_valu) of a
// 4-wave run; if they share the ALUs it takes their sum.
// Second question (modes 5-7): the same with v_mfma_f32_16x16x32_bf16 - does the bf16 matrix pipe run beside the fp32 lanes?
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_overlap_probe.hip -o gpurun_out/overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

template <int MODE>   // 0: all mfma, 1: all valu, 2: waves 0-3 mfma / 4-7 valu, 3: only waves 0-3 mfma (4-7 exit), 4: only waves 4-7 valu
                      // 5: waves 0-3 bf16 mfma alone, 6: waves 0-3 bf16 mfma + 4-7 valu, 7: all 8 waves bf16 mfma
__global__ void __launch_bounds__(512) probe(float* out, int iters, float seed) {
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    const bool do_mfma = MODE == 0 || ((MODE == 2 || MODE == 3) && wave < 4);
    const bool do_bf16 = MODE == 7 || ((MODE == 5 || MODE == 6) && wave < 4);
    const bool do_valu = MODE == 1 || ((MODE == 2 || MODE == 4 || MODE == 6) && wave >= 4);
    f4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    float t[8];
    for (int i = 0; i < 8; ++i) t[i] = seed + l * 1e-3f + i;
    const float m = 1.0001f, b = 1e-6f;
    if (do_mfma) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(t[0], t[1], acc[k], 0, 0, 0);   // 32 MFMAs
        }
    } else if (do_bf16) {
        bf8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(l * 0.01f + i); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k], 0, 0, 0);   // 32 MFMAs
        }
    } else if (do_valu) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) t[k] = __builtin_fmaf(t[k], m, b);    // 256 VALU fma
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += t[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int MODE>
float run(float* d, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 0, 0, d, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 0, 0, d, iters, 1.0f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    float* d;
    hipMalloc(&d, 4096);
    const int iters = 20000;
    const float t0 = run<0>(d, iters), t1 = run<1>(d, iters), t2 = run<2>(d, iters), t3 = run<3>(d, iters), t4 = run<4>(d, iters);
    printf("per iteration and wave: 32 x v_mfma_f32_16x16x4_f32 | 256 x v_fma_f32\n");
    printf("all 8 waves MFMA           %8.3f ms  (%.1f cycles per MFMA and SIMD at 2.4 GHz)\n", t0, t0 * 1e-3 * 2.4e9 / (iters * 32.0 * 2));
    printf("all 8 waves VALU           %8.3f ms  (%.2f cycles per fma and SIMD)\n", t1, t1 * 1e-3 * 2.4e9 / (iters * 256.0 * 2));
    printf("waves 0-3 MFMA alone       %8.3f ms\n", t3);
    printf("waves 4-7 VALU alone       %8.3f ms\n", t4);
    printf("waves 0-3 MFMA + 4-7 VALU  %8.3f ms  (independent pipes: %.3f, shared ALUs: %.3f)\n", t2, t3 > t4 ? t3 : t4, t3 + t4);
    const float t5 = run<5>(d, iters), t6 = run<6>(d, iters), t7 = run<7>(d, iters);
    printf("per iteration and wave: 32 x v_mfma_f32_16x16x32_bf16\n");
    printf("all 8 waves bf16 MFMA      %8.3f ms  (%.1f cycles per MFMA and SIMD)\n", t7, t7 * 1e-3 * 2.4e9 / (iters * 32.0 * 2));
    printf("waves 0-3 bf16 MFMA alone  %8.3f ms\n", t5);
    printf("waves 0-3 bf16 + 4-7 VALU  %8.3f ms  (independent pipes: %.3f, shared: %.3f)\n", t6, t5 > t4 ? t5 : t4, t5 + t4);
    return 0;
}
