#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL>
__device__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <int H>
__device__ float channel_rows_sum(float v) {
    // lanes l and l^32: ds_bpermute (hipcc mis-selects the second result of
    // __builtin_amdgcn_permlane32_swap when both operands carry the same value)
    const float s = v + __shfl_xor(v, 32);
    float t = s + __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, s), 0x401F));
    if constexpr (H == 2) t += dpp_mov<0x128>(t);
    return t;
}
__global__ void k(float* o) {
  const int l = threadIdx.x;
  float v = float(1 << (l >> 4)) + 16.0f * (l & 15);   // row id 1,2,4,8 + 16*col
  o[l] = channel_rows_sum<1>(v);
  o[64 + l] = channel_rows_sum<2>(v);
}
int main() {
  float* o; hipMalloc(&o, 512);
  k<<<1, 64>>>(o);
  float h[128]; hipMemcpy(h, o, 512, hipMemcpyDeviceToHost);
  printf("expect H1: 15 + 64*col ; H2: 30 + 64*(col + col^8)\n");
  for (int l = 0; l < 64; l += 5) printf("lane %2d (col %2d): H1 %g (exp %g)  H2 %g (exp %g)\n", l, l & 15, h[l], 15.0 + 64.0 * (l & 15), h[64 + l], 30.0 + 64.0 * ((l & 15) + ((l & 15) ^ 8)));
  return 0;
}
