// Host-visible launchers of the 16-row-tile MFMA-mixing kernels for D = 32 (cemlp_pg.hpp), one set per compiled algebra.
#pragma once
#include <hip/hip_runtime.h>

#include "cemlp_device.hpp"

namespace csmpn {
// floats of the weight-fragment tables for (mode, channels, attribute channels); 0: shape not served
#define CSMPN_DECLARE_PG(tag)                                                                                       \
    size_t cemlp_pg_table_floats_##tag(int mode, int channels, int attr);                                            \
    size_t cemlp_pg_slice_floats_##tag(int mode, int channels, int attr);                                            \
    bool has_cemlp_pg_##tag(int mode, int channels, int attr, bool bwd);                                            \
    hipError_t launch_cemlp_pg_##tag(int mode, int channels, int attr, bool bwd, bool pack, unsigned grid, hipStream_t st,      \
                                     const DevCemlp& C, const RowIO& io, float* tabs, bool* handled);
CSMPN_DECLARE_PG(n5)
CSMPN_DECLARE_PG(n5m)

}  // namespace csmpn
