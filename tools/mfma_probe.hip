// Probe of v_mfma_f32_4x4x1_16b_f32 operand/result layout and of VALU/MFMA/trans issue rates (gfx950).
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_probe.hip -o gpurun_out/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void layout_kernel(float* out) {
    const int l = threadIdx.x;
    // a = 100 + lane, b = 1000*(lane+1): D_b[i][j] = a[4b+i] * b[4b+j] expected in lane 4b+j, reg i
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(float(100 + l), float(l + 1), c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out[l * 4 + i] = c[i];
    // cbsz = 4, abid = 0: every block takes A from block 0
    f4 d = {0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_4x4x1f32(float(100 + l), float(l + 1), d, 4, 0, 0);
    for (int i = 0; i < 4; ++i) out[256 + l * 4 + i] = d[i];
}

template <int MODE>
__global__ void rate_kernel(float* out, int iters, float seed) {
    const int l = threadIdx.x;
    f2 a0 = {seed + l, seed}, a1 = {seed * 2, 1.f}, a2 = {3.f, seed}, a3 = {seed, 4.f};
    f2 m = {1.0001f, 0.9999f}, b = {1e-6f, -1e-6f};
    f4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    float t0 = seed + l * 1e-3f, t1 = seed * 0.5f, t2 = 1.f + seed, t3 = 2.f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {          // packed fma
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                a0 = __builtin_elementwise_fma(a0, m, b); a1 = __builtin_elementwise_fma(a1, m, b);
                a2 = __builtin_elementwise_fma(a2, m, b); a3 = __builtin_elementwise_fma(a3, m, b);
            }
        } else if (MODE == 1) {   // scalar fma
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                t0 = __builtin_fmaf(t0, m.x, b.x); t1 = __builtin_fmaf(t1, m.x, b.x);
                t2 = __builtin_fmaf(t2, m.x, b.x); t3 = __builtin_fmaf(t3, m.x, b.x);
            }
        } else if (MODE == 2) {   // mfma 4x4x1, 8 independent accumulators
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(t0, t1, acc[k], 0, 0, 0);
        } else if (MODE == 3) {   // transcendental: rcp
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                t0 = __builtin_amdgcn_rcpf(t0); t1 = __builtin_amdgcn_rcpf(t1);
                t2 = __builtin_amdgcn_rcpf(t2); t3 = __builtin_amdgcn_rcpf(t3);
            }
        } else if (MODE == 4) {   // mfma 4x4x1 interleaved with packed fma (overlap test)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    acc[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(t0, t1, acc[k], 0, 0, 0);
                    a0 = __builtin_elementwise_fma(a0, m, b);
                }
            }
        } else if (MODE == 5) {   // mfma 16x16x4, 8 independent accumulators
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(t0, t1, acc[k], 0, 0, 0);
        } else if (MODE == 6) {   // dependent chain of mfma 4x4x1 on one accumulator
#pragma unroll
            for (int u = 0; u < 64; ++u) acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(t0, t1, acc[0], 0, 0, 0);
        }
    }
    float r = a0.x + a0.y + a1.x + a1.y + a2.x + a2.y + a3.x + a3.y + t0 + t1 + t2 + t3;
    for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + l] = r;
}

template <int MODE>
double time_mode(float* d, int waves_per_simd, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 4 * waves_per_simd;   // one wave per block
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(64), 0, 0, d, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3;
}

int main() {
    float* d;
    hipMalloc(&d, sizeof(float) * 4096 * 64);
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, d);
    std::vector<float> h(512);
    hipMemcpy(h.data(), d, sizeof(float) * 512, hipMemcpyDeviceToHost);
    int bad = 0, badb = 0;
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i) {
            const int b = l / 4, j = l % 4;
            const float want = float(100 + 4 * b + i) * float(4 * b + j + 1);
            if (h[l * 4 + i] != want) ++bad;
            const float wantb = float(100 + i) * float(4 * b + j + 1);
            if (h[256 + l * 4 + i] != wantb) ++badb;
        }
    printf("layout 4x4x1: D[lane 4b+j][reg i] == A[lane 4b+i]*B[lane 4b+j]: %s (%d mismatches)\n", bad ? "NO" : "YES", bad);
    printf("cbsz=4 abid=0 broadcast of block 0's A: %s (%d mismatches)\n", badb ? "NO" : "YES", badb);
    if (bad) { for (int l = 0; l < 8; ++l) printf("lane %d: %g %g %g %g\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]); }
    const int iters = 2000;
    const double clk = 2.4e9;
    for (int w = 1; w <= 2; ++w) {
        const double n = 1024.0 * w;   // waves
        double t;
        t = time_mode<0>(d, w, iters); printf("w/SIMD=%d pk_fma   : %.3f ms  -> %.2f cycles per wave-instr per SIMD\n", w, t * 1e3, t * clk / (iters * 64.0 * w));
        t = time_mode<1>(d, w, iters); printf("w/SIMD=%d fma      : %.3f ms  -> %.2f cycles per wave-instr per SIMD\n", w, t * 1e3, t * clk / (iters * 64.0 * w));
        t = time_mode<2>(d, w, iters); printf("w/SIMD=%d mfma4x4x1: %.3f ms  -> %.2f cycles per mfma per SIMD\n", w, t * 1e3, t * clk / (iters * 64.0 * w));
        t = time_mode<3>(d, w, iters); printf("w/SIMD=%d rcp      : %.3f ms  -> %.2f cycles per wave-instr per SIMD\n", w, t * 1e3, t * clk / (iters * 64.0 * w));
        t = time_mode<4>(d, w, iters); printf("w/SIMD=%d mfma+pk  : %.3f ms  -> %.2f cycles per (mfma+pk_fma) pair per SIMD\n", w, t * 1e3, t * clk / (iters * 64.0 * w));
        t = time_mode<5>(d, w, iters); printf("w/SIMD=%d mfma16x16x4: %.3f ms  -> %.2f cycles per mfma per SIMD\n", w, t * 1e3, t * clk / (iters * 64.0 * w));
        t = time_mode<6>(d, w, iters); printf("w/SIMD=%d mfma4x4x1 dependent chain: %.3f ms  -> %.2f cycles per mfma per SIMD\n", w, t * 1e3, t * clk / (iters * 64.0 * w));
        (void)n;
    }
    return bad != 0;
}
