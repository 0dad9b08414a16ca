// Probe (round 3): cost of lane-private running sums in LDS: ds_add_f32 (no return) against a 16-byte
// read-modify-write, beside VALU work, at 2 waves per SIMD. Decides where the per-channel parameter-gradient sums of
// the (row, channel)-per-lane backward live.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

// MODE 0: VALU only (36 x 8 fma); 1: + 36 ds_add_f32 to [idx][thread]; 2: + 9 f4 read-modify-writes to [group][thread]
template <int MODE>
__global__ void __launch_bounds__(256, 2) k(float* out, int iters, float seed) {
    __shared__ float sums[36 * 256];
    const int t = threadIdx.x;
    for (int i = 0; i < 36; ++i) sums[i * 256 + t] = 0.f;
    __syncthreads();
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + i + t * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 36; ++i) {
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) a[k2] = __builtin_fmaf(a[k2], 1.0001f, 1e-6f);
            if constexpr (MODE == 1) {
                atomicAdd(&sums[i * 256 + t], a[i & 7]);
            } else if constexpr (MODE == 2) {
                if ((i & 3) == 3) {
                    f4* p = reinterpret_cast<f4*>(&sums[(i >> 2) * 1024 + 4 * t]);
                    f4 v = *p;
                    v += f4{a[0], a[1], a[2], a[3]};
                    *p = v;
                }
            }
        }
    }
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < 36; ++i) r += sums[i * 256 + t];
    for (int i = 0; i < 8; ++i) r += a[i];
    out[(size_t)blockIdx.x * 256 + t] = r;
}

template <int MODE>
static void run(float* d, const char* name) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256), 0, 0, d, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %.3f ms -> %.1f ns per iteration (36 values) per wave-pair\n", name, ms, ms * 1e6 / iters);
}
int main() {
    float* d; hipMalloc(&d, sizeof(float) * 512 * 256);
    run<0>(d, "VALU only (288 fma)");
    run<1>(d, "+ 36 ds_add_f32 lane-private");
    run<2>(d, "+ 9 x 16-byte read-modify-write");
    return 0;
}
