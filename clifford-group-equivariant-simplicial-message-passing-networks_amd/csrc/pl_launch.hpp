// Host-visible launchers of the parity-lane kernels (cemlp_pl.hpp), one set per compiled algebra.
#pragma once
#include <hip/hip_runtime.h>

#include "cemlp_device.hpp"

namespace csmpn {
constexpr int kPlMaxBwdGroups = 256;   // one 4-wave workgroup per CU in the backward
// parity-lane kernels: D = 32 algebras with an odd number of generators, every block 8 output channels
#define CSMPN_DECLARE_PL(tag)                                                                                 \
    bool has_cemlp_pl_##tag(int mode, int nblk, int channels, int i0);                                         \
    size_t cemlp_pl_slice_floats_##tag(int mode, int nblk, int channels, int i0);                              \
    hipError_t launch_cemlp_pl_##tag(int mode, int nblk, int channels, int i0, bool bwd, unsigned grid,        \
                                     hipStream_t st, const DevCemlp& C, const RowIO& io, bool* handled);
CSMPN_DECLARE_PL(n5)
CSMPN_DECLARE_PL(n5m)

}  // namespace csmpn
