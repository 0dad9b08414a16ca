// Host-visible launchers of the wide parity-lane kernels (cemlp_plw.hpp), one set per compiled algebra.
#pragma once
#include <hip/hip_runtime.h>

#include "cemlp_device.hpp"

namespace csmpn {
constexpr int kPlwMaxGroups = 256;   // one workgroup per CU
// floats of the rotation tables for (mode, channels, attribute channels | input channels of MODE_PLAIN, blocks); 0: shape not served
#define CSMPN_DECLARE_PLW(tag)                                                                               \
    size_t cemlp_plw_table_floats_##tag(int mode, int channels, int attr, int nblk);                          \
    hipError_t launch_cemlp_plw_##tag(int mode, int channels, int attr, int nblk, bool bwd, unsigned grid,    \
                                      hipStream_t st, const DevCemlp& C, const RowIO& io, float* tabs, bool* handled);
CSMPN_DECLARE_PLW(n5)
CSMPN_DECLARE_PLW(n5m)

}  // namespace csmpn
