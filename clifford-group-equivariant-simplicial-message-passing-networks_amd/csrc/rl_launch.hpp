// Host-visible launchers of the row-per-lane kernels (cemlp_rl.hpp), one set per compiled algebra.
#pragma once
#include <hip/hip_runtime.h>

#include "cemlp_device.hpp"

namespace csmpn {
constexpr int kRlMaxBwdGroups = 256;    // workgroups of a row-per-lane backward (one 4-wave group per CU)
constexpr int kRlPartialGroups = kRlMaxBwdGroups;       // slices of its partial buffer: one per workgroup
// row-per-lane kernels (cemlp_rl.hpp): narrow layers, every block 8 output channels
#define CSMPN_DECLARE_RL(tag)                                                                                  \
    bool has_cemlp_rl_##tag(int mode, int nblk, int channels, int i0);                                          \
    size_t cemlp_rl_partial_floats_##tag(int nblk, int channels, int i0);                                       \
    hipError_t launch_cemlp_rl_##tag(int mode, int nblk, int channels, int i0, bool bwd, unsigned grid,         \
                                     hipStream_t st, const DevCemlp& C, const RowIO& io, bool* handled);
CSMPN_DECLARE_RL(n3)

}  // namespace csmpn
