// Host-visible launchers of the channel-MFMA kernels (cemlp_cm.hpp), one set per compiled algebra.
#pragma once
#include <hip/hip_runtime.h>

#include "cemlp_device.hpp"

namespace csmpn {
constexpr int kCmMaxFwdGroups = 768;   // 4-wave workgroups of a forward launch: three per CU
constexpr int kCmMaxBwdGroups = 256;   // ... of a backward launch (one per CU: 512 registers); one slice of partial sums each
constexpr int kCmSliceCap = 512;       // slices the partial buffer is laid out for (block 1's start behind kCmSliceCap of block 0: = kClSliceCap)
#define CSMPN_DECLARE_CM(tag)                                                                                  \
    bool has_cemlp_cm_##tag(int mode, int nblk, int channels, int i0, bool bwd);                                \
    size_t cemlp_cm_partial_floats_##tag(int mode, int nblk, int channels, int i0);                             \
    hipError_t launch_cemlp_cm_##tag(int mode, int nblk, int channels, int i0, bool bwd, unsigned grid,         \
                                     hipStream_t st, const DevCemlp& C, const RowIO& io, bool* handled);
CSMPN_DECLARE_CM(n3)

}  // namespace csmpn
