// 16-row-tile MFMA-mixing kernels for Cl(3,0), 32 channels (cemlp_pq.hpp): EGCL edge (6 attribute channels) and node (3) programs.
#include "cemlp_pq.hpp"
#include "pq_launch.hpp"

namespace csmpn {
namespace {
using ALG_T = Alg<3, 0u>;

template <int C, int MODE, int NA>
hipError_t pq_launch(bool bwd, bool pack, unsigned grid, hipStream_t st, const DevCemlp& Cd, const RowIO& io_in, float* tabs) {
    using CF = PqCfg<ALG_T, C, MODE, NA>;
    RowIO io = io_in;
    io.plw_tabs = tabs;
    if (pack) hipLaunchKernelGGL((pg_pack_kernel<CF, ALG_T>), dim3((CF::tab_floats + 255) / 256), dim3(256), 0, st, Cd, tabs);
    if (!bwd) {
        hipLaunchKernelGGL((cemlp_pq_fwd_kernel<ALG_T, CF>), dim3(grid), dim3(kPqThreads), sizeof(float) * CF::lds_floats, st, Cd, io);
        return hipGetLastError();
    }
    // one launch per block (last block first). Block 1's slices are summed by extra workgroups of the block-0 launch (PqAux),
    // block 0's by the fixed-order reduce launch behind it; each block has its own slice region.
    constexpr size_t lds = sizeof(float) * CF::bwd_lds_floats;
    float* part0 = io.plw_part;
    float* part1 = io.plw_part + (size_t)CF::slice_floats(0) * kPqMaxGroups;
    PqAux aux1{nullptr, (int)grid, 0};
    io.plw_part = part1;
    hipLaunchKernelGGL((cemlp_pq_bwd_kernel<ALG_T, CF, 1>), dim3(grid), dim3(kPqThreads), lds, st, Cd, io, aux1);
    constexpr unsigned nred1 = (CF::slice_floats(1) + 63) / 64;
    PqAux aux0{part1, (int)grid, (int)grid};
    io.plw_part = part0;
    hipLaunchKernelGGL((cemlp_pq_bwd_kernel<ALG_T, CF, 0>), dim3(grid + nred1), dim3(kPqThreads), lds, st, Cd, io, aux0);
    hipLaunchKernelGGL((pq_reduce_kernel<ALG_T, CF, 0>), dim3((CF::slice_floats(0) + 63) / 64), dim3(256), 0, st, Cd, (const float*)part0, (int)grid);
    return hipGetLastError();
}
}  // namespace

size_t cemlp_pq_table_floats_n3(int mode, int channels, int attr) {
    if (channels == 32 && mode == MODE_EDGE && attr == 6) return PqCfg<ALG_T, 32, MODE_EDGE, 6>::tab_floats;
    if (channels == 32 && mode == MODE_NODE && attr == 3) return PqCfg<ALG_T, 32, MODE_NODE, 3>::tab_floats;
    return 0;
}
size_t cemlp_pq_slice_floats_n3(int mode, int channels, int attr) {
    if (channels == 32 && mode == MODE_EDGE && attr == 6) return PqCfg<ALG_T, 32, MODE_EDGE, 6>::slice_floats(0) + PqCfg<ALG_T, 32, MODE_EDGE, 6>::slice_floats(1);
    if (channels == 32 && mode == MODE_NODE && attr == 3) return PqCfg<ALG_T, 32, MODE_NODE, 3>::slice_floats(0) + PqCfg<ALG_T, 32, MODE_NODE, 3>::slice_floats(1);
    return 0;
}
hipError_t launch_cemlp_pq_n3(int mode, int channels, int attr, bool bwd, bool pack, unsigned grid, hipStream_t st, const DevCemlp& C, const RowIO& io,
                              float* tabs, bool* handled) {
    *handled = true;
    if (channels == 32 && mode == MODE_EDGE && attr == 6) return pq_launch<32, MODE_EDGE, 6>(bwd, pack, grid, st, C, io, tabs);
    if (channels == 32 && mode == MODE_NODE && attr == 3) return pq_launch<32, MODE_NODE, 3>(bwd, pack, grid, st, C, io, tabs);
    *handled = false;
    return hipSuccess;
}
}  // namespace csmpn
