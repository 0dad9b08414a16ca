// One-time target sort of a complex's adjacency list (C-ABI: csmpn_csr_build, include/csmpn_hip.h).
// PyG / torch_scatter need no sort (they scatter with atomics); the segmented scatter of the edge
// kernels does. Stable LSD radix sort (hipCUB) of (target, edge id): inside one target's segment the
// edges keep ascending original id, so the order - and every summation order that follows from it -
// is deterministic, and a hub node with a huge in-degree costs no more than any other edge (round 1
// canonicalised the segments with a per-node insertion sort: O(deg^2)).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdint>

#include "../../include/csmpn_hip.h"
#include "capi_common.hpp"

namespace {

__global__ void csr_prepare_kernel(const int64_t* ei, long E, long N, int* keys, int* vals, int* flag) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int64_t s = ei[e], d = ei[E + e];
    const bool bad = s < 0 || s >= N || d < 0 || d >= N;
    if (bad) atomicOr(flag, 1);
    keys[e] = bad ? 0 : (int)d;
    vals[e] = (int)e;
}

__global__ void csr_gather_src_kernel(const int64_t* ei, long E, long N, const int* perm, int* src_s) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E) return;
    const int64_t s = ei[perm[i]];
    src_s[i] = (s < 0 || s >= N) ? 0 : (int)s;
}

// row_ptr[v] = first sorted position whose target is >= v (binary search); degree from differences
__global__ void csr_rowptr_kernel(const int* dst_s, long E, long N, int* row_ptr) {
    const long v = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v > N) return;
    long lo = 0, hi = E;
    while (lo < hi) {
        const long mid = (lo + hi) >> 1;
        if (dst_s[mid] < v) lo = mid + 1; else hi = mid;
    }
    row_ptr[v] = (int)lo;
}
__global__ void csr_degree_kernel(const int* row_ptr, long N, int* deg) {
    const long v = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v < N) deg[v] = row_ptr[v + 1] - row_ptr[v];
}

__global__ void csr_iota_kernel(long E, int* vals) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) vals[e] = (int)e;
}

// out[v] (+)= sum_{k in [add_ptr[v], add_ptr[v+1])} rows[add_order ? add_order[k] : k]
//            - sum_{k in [sub_ptr[v], sub_ptr[v+1])} rows[sub_order ? sub_order[k] : k]
// one thread per (v, 16-byte column); every sum runs in ascending k: a fixed order
__global__ void segment_reduce_kernel(const float* rows, int row_f4, long N, const int* add_ptr, const int* add_order,
                                      const int* sub_ptr, const int* sub_order, float* out, int accumulate) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long v = t / row_f4;
    if (v >= N) return;
    const int c = (int)(t - v * row_f4);
    const float4* R = reinterpret_cast<const float4*>(rows);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (add_ptr) {
        const int b = add_ptr[v], e = add_ptr[v + 1];
        for (int k = b; k < e; ++k) {
            const long r = add_order ? add_order[k] : k;
            const float4 x = R[r * row_f4 + c];
            acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
        }
    }
    if (sub_ptr) {
        const int b = sub_ptr[v], e = sub_ptr[v + 1];
        for (int k = b; k < e; ++k) {
            const long r = sub_order ? sub_order[k] : k;
            const float4 x = R[r * row_f4 + c];
            acc.x -= x.x; acc.y -= x.y; acc.z -= x.z; acc.w -= x.w;
        }
    }
    float4* o = reinterpret_cast<float4*>(out) + v * row_f4 + c;
    if (accumulate) {
        const float4 y = *o;
        acc.x += y.x; acc.y += y.y; acc.z += y.z; acc.w += y.w;
    }
    *o = acc;
}

int bits_for(int64_t n) {
    int b = 1;
    while ((int64_t(1) << b) < n && b < 31) ++b;
    return b;
}

size_t cub_temp_bytes(int64_t E, int64_t N) {
    size_t bytes = 0;
    hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const int*)nullptr, (int*)nullptr, (const int*)nullptr,
                                       (int*)nullptr, (int)E, 0, bits_for(N));
    return bytes;
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" {

size_t csmpn_csr_workspace_bytes(int64_t E, int64_t N) {
    if (E < 0 || N <= 0 || E >= (1ll << 31) || N >= (1ll << 31)) return 0;
    return 256 + 2 * align256(sizeof(int) * (size_t)(E > 0 ? E : 1)) + align256(cub_temp_bytes(E > 0 ? E : 1, N)) + 256;
}

int csmpn_csr_build(const int64_t* edge_index, int64_t E, int64_t N, int32_t* perm, int32_t* src_sorted,
                    int32_t* dst_sorted, int32_t* in_degree, int32_t* row_ptr, void* workspace, size_t workspace_bytes,
                    uint32_t flags, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (N <= 0 || E < 0 || N >= (1ll << 31) || E >= (1ll << 31))
        return csmpn_fail(CSMPN_ERR_INVALID, "bad sizes N=%lld E=%lld", (long long)N, (long long)E);
    if (!perm || !src_sorted || !dst_sorted || !in_degree || !row_ptr) return csmpn_fail(CSMPN_ERR_INVALID, "null output pointer");
    const size_t need = csmpn_csr_workspace_bytes(E, N);
    if (!workspace || workspace_bytes < need) return csmpn_fail(CSMPN_ERR_INVALID, "csr workspace too small: %zu < %zu", workspace_bytes, need);
    if (E > 0 && !edge_index) return csmpn_fail(CSMPN_ERR_INVALID, "edge_index is null");
    char* ws = static_cast<char*>(workspace);
    ws = reinterpret_cast<char*>(align256(reinterpret_cast<size_t>(ws)));
    int* flag = reinterpret_cast<int*>(ws);
    int* keys = reinterpret_cast<int*>(ws + 256);
    int* vals = reinterpret_cast<int*>(ws + 256 + align256(sizeof(int) * (size_t)(E > 0 ? E : 1)));
    void* temp = ws + 256 + 2 * align256(sizeof(int) * (size_t)(E > 0 ? E : 1));
    size_t temp_bytes = cub_temp_bytes(E > 0 ? E : 1, N);
    const unsigned block = 256;
    hipError_t e = hipMemsetAsync(flag, 0, sizeof(int), st);
    if (e != hipSuccess) return csmpn_fail(CSMPN_ERR_HIP, "hipMemsetAsync: %s", hipGetErrorString(e));
    if (E > 0) {
        const unsigned grid = (unsigned)((E + block - 1) / block);
        hipLaunchKernelGGL(csr_prepare_kernel, dim3(grid), dim3(block), 0, st, edge_index, (long)E, (long)N, keys, vals, flag);
        e = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, (const int*)keys, (int*)dst_sorted, (const int*)vals,
                                               (int*)perm, (int)E, 0, bits_for(N), st);
        if (e != hipSuccess) return csmpn_fail(CSMPN_ERR_HIP, "radix sort: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(csr_gather_src_kernel, dim3(grid), dim3(block), 0, st, edge_index, (long)E, (long)N,
                           (const int*)perm, (int*)src_sorted);
    }
    hipLaunchKernelGGL(csr_rowptr_kernel, dim3((unsigned)((N + 1 + block - 1) / block)), dim3(block), 0, st,
                       (const int*)dst_sorted, (long)E, (long)N, (int*)row_ptr);
    hipLaunchKernelGGL(csr_degree_kernel, dim3((unsigned)((N + block - 1) / block)), dim3(block), 0, st,
                       (const int*)row_ptr, (long)N, (int*)in_degree);
    e = hipGetLastError();
    if (e != hipSuccess) return csmpn_fail(CSMPN_ERR_HIP, "csr kernels: %s", hipGetErrorString(e));
    if (!(flags & CSMPN_FLAG_NO_VALIDATE)) {
        // one host round trip per complex (the result is cached by the caller): an out-of-range index
        // would otherwise drive every later gather / scatter out of bounds
        int host_flag = 0;
        e = hipMemcpyAsync(&host_flag, flag, sizeof(int), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) return csmpn_fail(CSMPN_ERR_HIP, "csr validation: %s", hipGetErrorString(e));
        if (host_flag) return csmpn_fail(CSMPN_ERR_INVALID, "edge_index has entries outside [0, %lld)", (long long)N);
    }
    return CSMPN_OK;
}

int csmpn_csr_source_order(const int32_t* src_sorted, int64_t E, int64_t N, int32_t* order, int32_t* row_ptr_src,
                           void* workspace, size_t workspace_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (N <= 0 || E < 0 || N >= (1ll << 31) || E >= (1ll << 31))
        return csmpn_fail(CSMPN_ERR_INVALID, "bad sizes N=%lld E=%lld", (long long)N, (long long)E);
    if (!order || !row_ptr_src || (E > 0 && !src_sorted)) return csmpn_fail(CSMPN_ERR_INVALID, "null pointer");
    const size_t need = csmpn_csr_workspace_bytes(E, N);
    if (!workspace || workspace_bytes < need) return csmpn_fail(CSMPN_ERR_INVALID, "csr workspace too small: %zu < %zu", workspace_bytes, need);
    char* ws = reinterpret_cast<char*>(align256(reinterpret_cast<size_t>(workspace)));
    int* keys_out = reinterpret_cast<int*>(ws + 256);
    int* vals = reinterpret_cast<int*>(ws + 256 + align256(sizeof(int) * (size_t)(E > 0 ? E : 1)));
    void* temp = ws + 256 + 2 * align256(sizeof(int) * (size_t)(E > 0 ? E : 1));
    size_t temp_bytes = cub_temp_bytes(E > 0 ? E : 1, N);
    const unsigned block = 256;
    if (E > 0) {
        hipLaunchKernelGGL(csr_iota_kernel, dim3((unsigned)((E + block - 1) / block)), dim3(block), 0, st, (long)E, vals);
        hipError_t e = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, (const int*)src_sorted, keys_out, (const int*)vals,
                                                          (int*)order, (int)E, 0, bits_for(N), st);
        if (e != hipSuccess) return csmpn_fail(CSMPN_ERR_HIP, "radix sort: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(csr_rowptr_kernel, dim3((unsigned)((N + 1 + block - 1) / block)), dim3(block), 0, st,
                       (const int*)keys_out, (long)E, (long)N, (int*)row_ptr_src);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return csmpn_fail(CSMPN_ERR_HIP, "source order kernels: %s", hipGetErrorString(e));
    return CSMPN_OK;
}

int csmpn_segment_reduce(const float* rows, int64_t row_floats, int64_t N, const int32_t* add_ptr, const int32_t* add_order,
                         const int32_t* sub_ptr, const int32_t* sub_order, float* out, int32_t accumulate, void* stream) {
    if (N <= 0) return CSMPN_OK;
    if (!rows || !out) return csmpn_fail(CSMPN_ERR_INVALID, "null pointer");
    if (row_floats <= 0 || row_floats % 4 || row_floats > (1 << 20)) return csmpn_fail(CSMPN_ERR_INVALID, "row length %lld is not a positive multiple of 4", (long long)row_floats);
    if ((reinterpret_cast<size_t>(rows) | reinterpret_cast<size_t>(out)) & 15) return csmpn_fail(CSMPN_ERR_INVALID, "rows/out must be 16-byte aligned");
    const int row_f4 = (int)(row_floats / 4);
    const long threads = (long)N * row_f4;
    const unsigned block = 256;
    hipLaunchKernelGGL(segment_reduce_kernel, dim3((unsigned)((threads + block - 1) / block)), dim3(block), 0, (hipStream_t)stream,
                       rows, row_f4, (long)N, (const int*)add_ptr, (const int*)add_order, (const int*)sub_ptr,
                       (const int*)sub_order, out, (int)accumulate);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return csmpn_fail(CSMPN_ERR_HIP, "segment reduce: %s", hipGetErrorString(e));
    return CSMPN_OK;
}

}  // extern "C"
