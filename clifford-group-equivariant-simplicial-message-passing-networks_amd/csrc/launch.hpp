// Host-visible launcher declarations, one set per compiled algebra.
#pragma once
#include <hip/hip_runtime.h>

#include "cemlp_device.hpp"

namespace csmpn {
constexpr int kMaxLdsBytes = 160 * 1024;

#define CSMPN_DECLARE_ALG(tag)                                                                                  \
    bool has_h2_##tag();                                                                                        \
    bool has_ps_##tag();                                                                                        \
    hipError_t launch_cemlp_ps_##tag(int mode, bool bwd, unsigned grid, unsigned block, size_t lds,            \
                                     hipStream_t st, const DevCemlp& C, const RowIO& io);                      \
    hipError_t launch_cemlp_##tag(int mode, int var, int h, bool bwd, unsigned grid, unsigned block, size_t lds,   \
                                  hipStream_t st, const DevCemlp& C, const RowIO& io);                          \
    hipError_t launch_gp_##tag(bool bwd, const float* a, const float* b, const float* gout, float* out,        \
                               float* ga, float* gb, long rows, hipStream_t st);

CSMPN_DECLARE_ALG(n2)      // Cl(2,0)
CSMPN_DECLARE_ALG(n3)      // Cl(3,0)
CSMPN_DECLARE_ALG(n4)      // Cl(4,0)
CSMPN_DECLARE_ALG(n5)      // Cl(5,0)
CSMPN_DECLARE_ALG(n5m)     // Cl(4,1): metric (1,1,1,1,-1)
CSMPN_DECLARE_ALG(n4m)     // Cl(3,1): metric (1,1,1,-1)
}  // namespace csmpn
