// Host-visible launchers of the 16-row-tile MFMA-mixing kernels for Cl(3,0) at 32 channels (cemlp_pq.hpp).
#pragma once
#include <hip/hip_runtime.h>

#include "cemlp_device.hpp"

namespace csmpn {
// workgroups a launch may have = gradient slices the workspace region holds (three 4-wave workgroups per CU)
constexpr unsigned kPqGridCap = 768;
// floats of the weight-fragment tables / of one workgroup's gradient slices (all blocks: each block's launch has its own region) for
// (mode, blocks, channels, attribute channels - MODE_PLAIN: input channels of block 0); 0: shape not served
size_t cemlp_pq_table_floats_n3(int mode, int nblk, int channels, int attr);
size_t cemlp_pq_slice_floats_n3(int mode, int nblk, int channels, int attr);
hipError_t launch_cemlp_pq_n3(int mode, int nblk, int channels, int attr, bool bwd, bool pack, unsigned grid, hipStream_t st, const DevCemlp& C,
                              const RowIO& io, float* tabs, bool* handled);
}  // namespace csmpn
