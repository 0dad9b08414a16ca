// Second probe: do fp32 MFMA and VALU overlap (a) across two waves of one SIMD, (b) inside one wave;
// real shader clock; v_fma with SGPR operand; LDS-broadcast reads beside VALU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// role: 0 = VALU fma (8 independent chains), 1 = MFMA 4x4x1 (8 independent accs), 2 = MFMA 16x16x4
__device__ __forceinline__ float work(int role, int iters, float seed, int l) {
    float t[8];
    f4 acc[8];
    for (int i = 0; i < 8; ++i) { t[i] = seed + i + l * 1e-3f; acc[i] = f4{0.f, 0.f, 0.f, 0.f}; }
    const float m = 1.0001f, b = 1e-6f;
    if (role == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) t[k] = __builtin_fmaf(t[k], m, b);
        }
    } else if (role == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(t[0], t[1], acc[k], 0, 0, 0);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(t[0], t[1], acc[k], 0, 0, 0);
        }
    }
    float r = 0.f;
    for (int i = 0; i < 8; ++i) r += t[i] + acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    return r;
}

// 512 threads = 8 waves: waves 0-3 take roleA, waves 4-7 take roleB (-1 = idle)
__global__ void __launch_bounds__(512) pair_kernel(float* out, int itA, int itB, int roleA, int roleB, float seed, long long* clk) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int role = w < 4 ? roleA : roleB;
    const int iters = w < 4 ? itA : itB;
    long long c0 = clock64(), w0 = wall_clock64();
    float r = 0.f;
    if (role >= 0) r = work(role, iters, seed, l);
    long long c1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

// one wave: NV VALU fmas (independent chains) after every MFMA 4x4x1
template <int NV>
__global__ void __launch_bounds__(64) mix_kernel(float* out, int iters, float seed) {
    const int l = threadIdx.x;
    float t[8];
    f4 acc[8];
    for (int i = 0; i < 8; ++i) { t[i] = seed + i + l * 1e-3f; acc[i] = f4{0.f, 0.f, 0.f, 0.f}; }
    const float m = 1.0001f, b = 1e-6f;
    const float a0 = seed * 0.5f, b0 = seed * 0.25f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                acc[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, b0, acc[k], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < NV; ++v) t[(k * NV + v) & 7] = __builtin_fmaf(t[(k * NV + v) & 7], m, b);
            }
        }
    }
    float r = 0.f;
    for (int i = 0; i < 8; ++i) r += t[i] + acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 64 + l] = r;
}

// v_fma with an SGPR multiplicand streamed from constant memory (scalar loads), 64 accumulators:
// the VALU form of a linear layer with wave-uniform weights
__global__ void __launch_bounds__(64) sgpr_fma_kernel(float* out, const float* __restrict__ w, int iters, float seed) {
    const int l = threadIdx.x;
    float x[16], y[16];
    for (int i = 0; i < 16; ++i) { x[i] = seed + i + l; y[i] = 0.f; }
    for (int it = 0; it < iters; ++it) {
        const float* wp = w + (it & 7) * 256;
#pragma unroll
        for (int o = 0; o < 16; ++o)
#pragma unroll
            for (int i = 0; i < 16; ++i) y[o] = __builtin_fmaf(wp[o * 16 + i], x[i], y[o]);
    }
    float r = 0.f;
    for (int i = 0; i < 16; ++i) r += y[i];
    out[blockIdx.x * 64 + l] = r;
}

static double run_pair(float* d, long long* clk, int roleA, int roleB, int itA, int itB) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(pair_kernel, dim3(256), dim3(512), 0, 0, d, 10, 10, roleA, roleB, 1.0f, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(pair_kernel, dim3(256), dim3(512), 0, 0, d, itA, itB, roleA, roleB, 1.0f, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int NV> static void run_mix(float* d, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 1; w <= 2; ++w) {
        hipLaunchKernelGGL(mix_kernel<NV>, dim3(1024 * w), dim3(64), 0, 0, d, 10, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(mix_kernel<NV>, dim3(1024 * w), dim3(64), 0, 0, d, iters, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("mix NV=%d waves/SIMD=%d: %.3f ms -> %.1f ns per (mfma + %d fma) per SIMD\n", NV, w, ms, ms * 1e6 / (iters * 64.0 * w), NV);
    }
}

int main() {
    float* d; long long* clk; float* w;
    hipMalloc(&d, sizeof(float) * 2048 * 512); hipMalloc(&clk, 16); hipMalloc(&w, sizeof(float) * 4096);
    std::vector<float> hw(4096, 0.001f); hipMemcpy(w, hw.data(), sizeof(float) * 4096, hipMemcpyHostToDevice);
    const int it = 4000;
    long long h[2];
    const char* names[] = {"idle", "valu", "mfma4", "mfma16"};
    int roles[][2] = {{0, -1}, {1, -1}, {2, -1}, {0, 0}, {1, 1}, {0, 1}, {0, 2}, {2, 2}};
    for (auto& r : roles) {
        const double ms = run_pair(d, clk, r[0], r[1], it, it);
        hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        printf("waves0-3=%s waves4-7=%s: %.3f ms; clock64 %lld ticks, wall_clock64 %lld ticks (100 MHz?) -> clock64 rate %.1f MHz; ns per instr of one wave: %.2f\n",
               names[r[0] + 1], names[r[1] + 1], ms, h[0], h[1], h[1] ? h[0] * 100.0 / h[1] : 0.0, ms * 1e6 / (it * 64.0));
    }
    run_mix<0>(d, 2000); run_mix<1>(d, 2000); run_mix<2>(d, 2000); run_mix<3>(d, 2000); run_mix<4>(d, 2000);
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int wv = 1; wv <= 2; ++wv) {
            hipLaunchKernelGGL(sgpr_fma_kernel, dim3(1024 * wv), dim3(64), 0, 0, d, w, 10, 1.0f);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(sgpr_fma_kernel, dim3(1024 * wv), dim3(64), 0, 0, d, w, 4000, 1.0f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("sgpr-weight fma waves/SIMD=%d: %.3f ms -> %.2f ns per fma-instr per SIMD\n", wv, ms, ms * 1e6 / (4000 * 256.0 * wv));
        }
    }
    return 0;
}
