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

}  // extern "C"
