// 16-row-tile MFMA-mixing kernels for Cl(3,0), 32 channels (cemlp_pq.hpp): EGCL edge (6 attribute channels) and node (3) programs and
// the standalone CEMLPs of the md17 model (simplex embeddings 60 -> 32 and 90 -> 32 -> 32, head 32 -> 32: md17_cssmpnn.py:85-120,165-176).
#include "cemlp_pq.hpp"
#include "pq_launch.hpp"

namespace csmpn {
namespace {
using ALG_T = Alg<3, 0u>;
static_assert(kPqMaxGroups == (int)kPqGridCap, "slice regions are sized for the grid cap (capi.hip: pq_region_bytes)");

template <int MODE, int NA, int NBLK>
hipError_t pq_launch(bool bwd, bool pack, unsigned grid, hipStream_t st, const DevCemlp& Cd, const RowIO& io_in, float* tabs) {
    using CF = PqCfg<ALG_T, 32, MODE, NA, NBLK>;
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
    unsigned nred1 = 0;
    PqAux aux0{nullptr, (int)grid, 0};
    if constexpr (NBLK == 2) {
        float* part1 = io.plw_part + (size_t)CF::slice_floats(0) * kPqMaxGroups;
        PqAux aux1{nullptr, (int)grid, 0};
        io.plw_part = part1;
        hipLaunchKernelGGL((cemlp_pq_bwd_kernel<ALG_T, CF, 1>), dim3(grid), dim3(kPqThreads), lds, st, Cd, io, aux1);
        nred1 = (CF::slice_floats(1) + 63) / 64;
        aux0 = PqAux{part1, (int)grid, (int)grid};
        io.plw_part = part0;
    }
    hipLaunchKernelGGL((cemlp_pq_bwd_kernel<ALG_T, CF, 0>), dim3(grid + nred1), dim3(kPqThreads), lds, st, Cd, io, aux0);
    hipLaunchKernelGGL((pq_reduce_kernel<ALG_T, CF, 0>), dim3((CF::slice_floats(0) + 63) / 64), dim3(256), 0, st, Cd, (const float*)part0, (int)grid);
    return hipGetLastError();
}
}  // namespace

// served shapes: (mode, blocks, attribute channels / plain: input channels)
#define CSMPN_PQ_SHAPES(X) X(MODE_EDGE, 2, 6) X(MODE_NODE, 2, 3) X(MODE_PLAIN, 1, 60) X(MODE_PLAIN, 2, 90) X(MODE_PLAIN, 1, 32)

size_t cemlp_pq_table_floats_n3(int mode, int nblk, int channels, int attr) {
#define X(M, B, A) if (channels == 32 && mode == M && nblk == B && attr == A) return PqCfg<ALG_T, 32, M, A, B>::tab_floats;
    CSMPN_PQ_SHAPES(X)
#undef X
    return 0;
}
size_t cemlp_pq_slice_floats_n3(int mode, int nblk, int channels, int attr) {
#define X(M, B, A) if (channels == 32 && mode == M && nblk == B && attr == A) return PqCfg<ALG_T, 32, M, A, B>::slice_both;
    CSMPN_PQ_SHAPES(X)
#undef X
    return 0;
}
hipError_t launch_cemlp_pq_n3(int mode, int nblk, int channels, int attr, bool bwd, bool pack, unsigned grid, hipStream_t st, const DevCemlp& C,
                              const RowIO& io, float* tabs, bool* handled) {
    *handled = true;
#define X(M, B, A) if (channels == 32 && mode == M && nblk == B && attr == A) return pq_launch<M, A, B>(bwd, pack, grid, st, C, io, tabs);
    CSMPN_PQ_SHAPES(X)
#undef X
    *handled = false;
    return hipSuccess;
}
}  // namespace csmpn
