// Host-visible launchers of the (row, channel)-per-lane kernels (cemlp_cl.hpp), one set per compiled algebra.
#pragma once
#include <hip/hip_runtime.h>

#include "cemlp_device.hpp"

namespace csmpn {
constexpr int kClMaxFwdGroups = 1024;   // 4-wave workgroups of a forward launch: four per CU (4 waves per SIMD)
constexpr int kClMaxBwdGroups = 512;    // ... of a backward launch: two per CU; one slice of partial sums each (= kClSliceCap of cemlp_cl.hpp)
#define CSMPN_DECLARE_CL(tag)                                                                                  \
    bool has_cemlp_cl_##tag(int mode, int nblk, int channels, int i0);                                          \
    size_t cemlp_cl_partial_floats_##tag(int mode, int nblk, int channels, int i0);                             \
    hipError_t launch_cemlp_cl_##tag(int mode, int nblk, int channels, int i0, bool bwd, unsigned grid,         \
                                     hipStream_t st, const DevCemlp& C, const RowIO& io, bool* handled);
CSMPN_DECLARE_CL(n3)

}  // namespace csmpn
