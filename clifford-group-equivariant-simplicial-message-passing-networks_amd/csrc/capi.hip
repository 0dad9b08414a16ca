// C-ABI of the library (declared in include/csmpn_hip.h): host-side table
// construction, launch planning, weight packing, CSR build and dispatch to the
// per-algebra kernel instantiations.
#include <atomic>
#include <mutex>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/csmpn_hip.h"
#include "capi_common.hpp"
#include "cemlp_kernel.hpp"
#include "launch.hpp"
#include "cl_launch.hpp"
#include "cm_launch.hpp"
#include "pl_launch.hpp"
#include "plw_launch.hpp"
#include "pg_launch.hpp"
#include "pq_launch.hpp"

using namespace csmpn;

thread_local char g_csmpn_err[512] = "";

int csmpn_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_csmpn_err, sizeof(g_csmpn_err), fmt, ap);
    va_end(ap);
    return code;
}

namespace {

// Every environment switch of the library, read ONCE per process. They select kernel families for A/B measurements and
// parity tests of the slower paths or bound a launch for experiments; none of them changes results beyond summation
// order. Documented in INTEGRATION.md ("Debug switches").
struct Switches {
    bool no_cl;          // CSMPN_NO_CL=1       Cl(3,0) 8-channel layers leave the (row, channel)-per-lane kernels (cemlp_cl.hpp) for the general ones
    bool no_cm;          // CSMPN_NO_CM=1       16 / 32-channel Cl(3,0) layers leave the channel-MFMA kernels (cemlp_cm*.hpp)
    bool no_cm_bwd;      // CSMPN_NO_CM_BWD=1   ... their backward only (forward stays)
    bool no_pl;          // CSMPN_NO_PL=1       8-channel Cl(5,0) / Cl(4,1) layers leave the parity-lane kernels (cemlp_pl.hpp)
    bool no_plw;         // CSMPN_NO_PLW=1      wide Cl(5,0) / Cl(4,1) layers leave the wide parity-lane kernels (cemlp_plw.hpp)
    bool plw8;           // CSMPN_PLW8=1        8 channels on the wide parity-lane kernels with one group
    bool no_pg;          // CSMPN_NO_PG=1       24 / 28 / 32-channel Cl(5,0) / Cl(4,1) layers leave the 16-row-tile MFMA-mixing kernels (cemlp_pg.hpp)
    bool no_pq;          // CSMPN_NO_PQ=1       32-channel Cl(3,0) layers leave the 16-row-tile MFMA-mixing kernels (cemlp_pq.hpp) for the channel-MFMA ones
    bool no_share;       // CSMPN_NO_SHARE=1    general kernels: z does not alias the input tile
    bool no_phased;      // CSMPN_NO_PHASED=1   general kernels: backward of all blocks per tile instead of block by block
    bool no_sliced;      // CSMPN_NO_SLICED_GRADS=1  general kernels: parameter-gradient atomics onto one copy
    bool debug;          // CSMPN_DEBUG         one line per launch on stderr: family, mode, shape, grid
    int force_ps;        // CSMPN_FORCE_PS=0|1  parity-split layout of the general kernels off / on (-1: by algebra)
    int force_h;         // CSMPN_FORCE_H=1|2   row halves per tile of the general kernels (0: by row count)
    int min_lds_tiles;   // CSMPN_MIN_LDS_TILES resident row tiles below which the general kernels leave the LDS variant
    long phased_min_rows;   // CSMPN_PHASED_MIN_ROWS  rows from which the phased backward is taken (default 4096)
    long cl_cap_fwd, cl_cap_bwd;   // CSMPN_CL_CAP_FWD / _BWD  fewer resident workgroups of the cl kernels (experiments)
};
// name of the kernel the calling thread dispatched last (csmpn_last_kernel)
thread_local char g_last_kernel[192] = "";
void note_kernel(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_kernel, sizeof(g_last_kernel), fmt, ap);
    va_end(ap);
}
const Switches& sw() {
    static const Switches s = [] {
        auto flag = [](const char* n) { const char* v = getenv(n); return v && atoi(v) != 0; };
        auto num = [](const char* n, long d) { const char* v = getenv(n); return v ? atol(v) : d; };
        Switches r;
        r.no_cl = flag("CSMPN_NO_CL"); r.no_cm = flag("CSMPN_NO_CM"); r.no_cm_bwd = flag("CSMPN_NO_CM_BWD");
        r.no_pl = flag("CSMPN_NO_PL"); r.no_plw = flag("CSMPN_NO_PLW"); r.plw8 = flag("CSMPN_PLW8");
        r.no_pg = flag("CSMPN_NO_PG"); r.no_pq = flag("CSMPN_NO_PQ");
        r.no_share = flag("CSMPN_NO_SHARE"); r.no_phased = flag("CSMPN_NO_PHASED"); r.no_sliced = flag("CSMPN_NO_SLICED_GRADS");
        r.debug = getenv("CSMPN_DEBUG") != nullptr;
        r.force_ps = getenv("CSMPN_FORCE_PS") ? (atoi(getenv("CSMPN_FORCE_PS")) != 0) : -1;
        r.force_h = getenv("CSMPN_FORCE_H") ? (atoi(getenv("CSMPN_FORCE_H")) == 2 ? 2 : 1) : 0;
        r.min_lds_tiles = (int)num("CSMPN_MIN_LDS_TILES", 1);
        r.phased_min_rows = num("CSMPN_PHASED_MIN_ROWS", 16L * 256);
        r.cl_cap_fwd = num("CSMPN_CL_CAP_FWD", 0); r.cl_cap_bwd = num("CSMPN_CL_CAP_BWD", 0);
        return r;
    }();
    return s;
}


#define fail csmpn_fail

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(CSMPN_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline int rup(int a, int b) { return cdiv(a, b) * b; }

// ----------------------------------------------------------------------------- algebra id
enum AlgId { ALG_NONE = -1, ALG_N2, ALG_N3, ALG_N4, ALG_N5, ALG_N5M, ALG_N4M };
const char* alg_name(AlgId id) {
    switch (id) {
        case ALG_N2: return "csmpn::Alg<2, 0u>";
        case ALG_N3: return "csmpn::Alg<3, 0u>";
        case ALG_N4: return "csmpn::Alg<4, 0u>";
        case ALG_N4M: return "csmpn::Alg<4, 8u>";
        case ALG_N5: return "csmpn::Alg<5, 0u>";
        case ALG_N5M: return "csmpn::Alg<5, 16u>";
        default: return "csmpn::Alg<?>";
    }
}

AlgId alg_id(const float* metric, int n) {
    if (!metric || n < 2 || n > 5) return ALG_NONE;
    unsigned neg = 0;
    for (int i = 0; i < n; ++i) {
        if (metric[i] == 1.0f) continue;
        if (metric[i] == -1.0f) { neg |= 1u << i; continue; }
        return ALG_NONE;
    }
    if (n == 2 && neg == 0) return ALG_N2;
    if (n == 3 && neg == 0) return ALG_N3;
    if (n == 4 && neg == 0) return ALG_N4;
    if (n == 5 && neg == 0) return ALG_N5;
    if (n == 5 && neg == 0x10u) return ALG_N5M;
    if (n == 4 && neg == 0x8u) return ALG_N4M;
    return ALG_NONE;
}

bool has_h2(AlgId id) {
    switch (id) {
        case ALG_N2: return has_h2_n2();
        case ALG_N3: return has_h2_n3();
        case ALG_N4: return has_h2_n4();
        case ALG_N5: return has_h2_n5();
        case ALG_N5M: return has_h2_n5m();
        case ALG_N4M: return has_h2_n4m();
        default: return false;
    }
}

bool has_ps(AlgId id) {
    switch (id) {
        case ALG_N3: return has_ps_n3();
        case ALG_N5: return has_ps_n5();
        case ALG_N5M: return has_ps_n5m();
        default: return false;
    }
}

hipError_t launch_cemlp_ps(AlgId id, int mode, bool bwd, unsigned grid, unsigned block, size_t lds, hipStream_t st,
                           const DevCemlp& C, const RowIO& io) {
    switch (id) {
        case ALG_N3: return launch_cemlp_ps_n3(mode, bwd, grid, block, lds, st, C, io);
        case ALG_N5: return launch_cemlp_ps_n5(mode, bwd, grid, block, lds, st, C, io);
        case ALG_N5M: return launch_cemlp_ps_n5m(mode, bwd, grid, block, lds, st, C, io);
        default: return hipErrorInvalidValue;
    }
}

int n_paths(AlgId id) {
    switch (id) {
        case ALG_N2: return Alg<2, 0u>::P;
        case ALG_N3: return Alg<3, 0u>::P;
        case ALG_N4: return Alg<4, 0u>::P;
        case ALG_N5: return Alg<5, 0u>::P;
        case ALG_N5M: return Alg<5, 0x10u>::P;
        case ALG_N4M: return Alg<4, 0x8u>::P;
        default: return 0;
    }
}

hipError_t launch_cemlp(AlgId id, int mode, int var, int h, bool bwd, unsigned grid, unsigned block, size_t lds,
                        hipStream_t st, const DevCemlp& C, const RowIO& io) {
    switch (id) {
        case ALG_N2: return launch_cemlp_n2(mode, var, h, bwd, grid, block, lds, st, C, io);
        case ALG_N3: return launch_cemlp_n3(mode, var, h, bwd, grid, block, lds, st, C, io);
        case ALG_N4: return launch_cemlp_n4(mode, var, h, bwd, grid, block, lds, st, C, io);
        case ALG_N5: return launch_cemlp_n5(mode, var, h, bwd, grid, block, lds, st, C, io);
        case ALG_N5M: return launch_cemlp_n5m(mode, var, h, bwd, grid, block, lds, st, C, io);
        case ALG_N4M: return launch_cemlp_n4m(mode, var, h, bwd, grid, block, lds, st, C, io);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_gp(AlgId id, bool bwd, const float* a, const float* b, const float* gout, float* out, float* ga,
                     float* gb, long rows, hipStream_t st) {
    switch (id) {
        case ALG_N2: return launch_gp_n2(bwd, a, b, gout, out, ga, gb, rows, st);
        case ALG_N3: return launch_gp_n3(bwd, a, b, gout, out, ga, gb, rows, st);
        case ALG_N4: return launch_gp_n4(bwd, a, b, gout, out, ga, gb, rows, st);
        case ALG_N5: return launch_gp_n5(bwd, a, b, gout, out, ga, gb, rows, st);
        case ALG_N5M: return launch_gp_n5m(bwd, a, b, gout, out, ga, gb, rows, st);
        case ALG_N4M: return launch_gp_n4m(bwd, a, b, gout, out, ga, gb, rows, st);
        default: return hipErrorInvalidValue;
    }
}

// ----------------------------------------------------------------------------- weight packing kernel
__global__ void pack_weights_kernel(const PackDesc P) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= P.total) return;
    int s = 0;
    while (e >= P.seg[s].count) { e -= P.seg[s].count; ++s; }
    const PackSeg& S = P.seg[s];
    const int H = P.H, NW = 16 / H;
    const int lane = e & 63;
    int rest = e >> 6;
    // fragment order [N tile][k-block][row half][grade][lane]
    const int g = rest % P.G; rest /= P.G;
    const int hp = rest % H; rest /= H;
    const int kk = rest % S.KK; rest /= S.KK;
    const int nt = rest;
    const int ncol = lane & 15;
    const int hcol = H == 1 ? 0 : (ncol >> 3);
    const int n = NW * nt + (H == 1 ? ncol : (ncol & 7));
    f4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = 16 * kk + 4 * r + (lane >> 4);   // k-slot (q, v = r) -> channel 16 kk + 4 v + q
        const int o = S.transposed ? k : n, i = S.transposed ? n : k;
        float val = 0.f;
        if (hcol == hp && o < S.O && i < S.I)
            val = S.has_grades ? S.w[((size_t)o * S.I + i) * P.G + g] : S.w[(size_t)o * S.I + i];
        v[r] = val;
    }
    S.dst[e] = v;
}

// ----------------------------------------------------------------------------- planning
struct Plan {
    DevCemlp C;
    PackDesc P;
    size_t pack_f4;       // f4 elements of packed weights
    unsigned threads;
    size_t lds_bytes;
    unsigned grid_cap;    // workgroups that fit on the chip at once
    int var;              // VAR_WAVE / VAR_GROUP / VAR_GROUP_NM / VAR_GLOBAL
    int H;                // row halves per tile
    bool ps;              // parity-split kernels (cemlp_ps.hpp): 16-row tiles, 8 channels x 2 blade parities
    bool det_general;     // deterministic mode on the general row-tile kernels: one row tile per workgroup, mirror slices
    void* workspace;      // the caller's workspace (the row-per-lane backward keeps its partial sums at its end)
    size_t workspace_bytes;
};

// floats of one row tile's buffers; tiles are [channel][D][R] with channel stride R*D + 4
struct TileLayout { int off_in, off_p0, off_p1, off_z, off_g, off_red, off_idx, total; };
TileLayout tile_layout(int D, int H, const csmpn_block_params* blocks, int nblk, bool bwd, int stage_rowlen,
                       bool use_saved = false, bool ps = false, bool share_inz = false) {
    int maxO = 0, maxCPo = 0;
    for (int k = 0; k < nblk; ++k) {
        maxO = blocks[k].out_features > maxO ? blocks[k].out_features : maxO;
        maxCPo = rup(blocks[k].out_features, 4) > maxCPo ? rup(blocks[k].out_features, 4) : maxCPo;
    }
    const int R = 16 * H, CS = R * D + 4, NW = 16 / H;
    const int MT = cdiv(maxO, NW);
    int sz_in = rup(blocks[0].in_features, 4) * CS;
    const int sz_o = maxCPo * CS;
    // backward with saved block inputs: ONE input buffer serves every block in turn
    const bool single_in = bwd && use_saved && nblk > 1;
    if ((single_in || share_inz) && sz_o > sz_in) sz_in = sz_o;
    TileLayout L;
    int off = 0;
    if (!bwd) {
        // forward: a block reads its input tile only in its first phase (MVLinear) and writes
        // its output after its last one, so the output of block k may overwrite the input of
        // block k (ping-pong degenerates to ONE input buffer); the dense scatter staging of
        // the edge forward reuses the z buffer, dead by then.
        const int sz_io = (nblk >= 2 && sz_o > sz_in) ? sz_o : sz_in;
        int sz_z = sz_o;
        if (stage_rowlen > 0 && R * stage_rowlen > sz_z) sz_z = rup(R * stage_rowlen, 4);
        if (MT == 1) {
            // single-wave tiles: the gated activations z are written after the block's only
            // read of its input (MVLinear) and the LDS executes a wave in order, so z (and the
            // scatter staging) share the input buffer too: ONE buffer per tile.
            const int sz_all = sz_io > sz_z ? sz_io : sz_z;
            L.off_in = off; L.off_p0 = off; L.off_p1 = off; L.off_z = off; L.off_g = off; off += sz_all;
        } else {
            L.off_in = off; L.off_p0 = off; L.off_p1 = off; off += sz_io;
            L.off_z = off; L.off_g = off; off += sz_z;
        }
    } else {
        L.off_in = off; off += sz_in;
        L.off_p0 = off; off += (nblk >= 2 && !single_in) ? sz_o : 0;
        L.off_p1 = off; off += (nblk >= 3 && !single_in) ? sz_o : 0;
        if (share_inz) L.off_z = L.off_in;   // z aliases the input buffer (sz_in >= sz_o, checked by the caller)
        else { L.off_z = off; off += sz_o; }
        L.off_g = off;
        int sz_g = sz_o;
        if (stage_rowlen > 0 && R * stage_rowlen > sz_g) sz_g = rup(R * stage_rowlen, 4);
        const int park = ps ? 128 * D : 256 * D;         // parking area of the incoming gradient
        if (MT == 1 && park > sz_g) sz_g = park;
        off += sz_g;
    }
    // cross-wave LayerNorm scratch of the barrier variants: reserved for single-wave tiles too (they
    // run a barrier variant when the weight store does not fit beside the tiles, or in global scratch)
    L.off_red = off; off += rup(MT * 16, 4);
    L.off_idx = off; off += rup(3 * R, 4);   // int copies of the tile's gathered row indices
    L.total = off;
    return L;
}
struct Choice { int var, rt, wgs; bool mirror; };
// backward kernels are built for 256 threads (512 VGPRs), forward for 512 threads
Choice choose_variant(int MT, size_t tile_bytes, size_t mirror_bytes, size_t wstore_bytes, bool bwd, bool ps = false) {
    // workgroups of at most 512 threads (forward, parity-split backward) / 256 threads (backward)
    // parity-split forward: 256-thread workgroups, three per CU (168 VGPRs: 3 waves per SIMD)
    const int waves = ps ? 4 : (bwd ? 4 : 8);
    const int max_wgs = (ps && !bwd) ? 3 : 2;
    const int max_rt = (waves / MT) > 0 ? waves / MT : 1;
    auto fit = [&](size_t fixed, int& rt_out, int& wgs_out) {
        int best_waves = 0;
        for (int wgs = 1; wgs <= max_wgs; ++wgs) {
            const size_t budget = (size_t)kMaxLdsBytes / wgs;
            if (budget <= fixed) continue;
            int rt = (int)((budget - fixed) / tile_bytes);
            if (rt > max_rt) rt = max_rt;
            if (rt < 1) continue;
            if (wgs * rt > best_waves) { best_waves = wgs * rt; rt_out = rt; wgs_out = wgs; }
        }
        return best_waves > 0;
    };
    Choice c{VAR_GLOBAL, 1, 1, false};
    int rt = 0, wgs = 1;
    const int min_lds_waves = sw().min_lds_tiles;
    // single-wave tiles with gradient mirror and weight store in LDS
    if (MT == 1 && fit(mirror_bytes + wstore_bytes, rt, wgs) && rt * wgs >= min_lds_waves) {
        c.var = VAR_WAVE; c.rt = rt; c.wgs = wgs; c.mirror = bwd && mirror_bytes > 0;
        return c;
    }
    if (fit(mirror_bytes, rt, wgs) && rt * wgs >= min_lds_waves) {
        c.var = VAR_GROUP; c.rt = rt; c.wgs = wgs; c.mirror = bwd && mirror_bytes > 0;
        return c;
    }
    if (fit(0, rt, wgs) && rt * wgs >= min_lds_waves) {   // tiles fit, the gradient mirror does not
        c.var = VAR_GROUP_NM; c.rt = rt; c.wgs = wgs; c.mirror = false;
        return c;
    }
    c.rt = (4 / MT) > 0 ? 4 / MT : 1;
    return c;
}
constexpr unsigned kGlobalTileGrid = 256;   // workgroups when the tiles live in global scratch

size_t packed_f4_count(int G, int H, const csmpn_block_params* blocks, int nblk) {
    size_t tot = 0;
    const int NW = 16 / H;
    for (int k = 0; k < nblk; ++k) {
        const int I = blocks[k].in_features, O = blocks[k].out_features;
        const size_t KKi = cdiv(I, 16), KKo = cdiv(O, 16), NTi = cdiv(I, NW), NTo = cdiv(O, NW);
        const size_t per = (size_t)H * G * 64;
        tot += per * (NTo * KKi + NTi * KKo);        // W1 forward + transposed
        tot += per * 4 * NTo * KKo;                  // WR, WL forward + transposed
    }
    return tot;
}

int mirror_floats_of(int I, int O, int G, int P, bool sub) {
    return (sub ? G : 1) * O * I + 2 * G * O * O + 3 * O + 3 * O * G + O * P;
}

// Tile height per launch. H = 2 (32-row tiles, 8 lane columns per half) needs every width
// <= 8 channels, an algebra with H = 2 kernels and the single-wave variant (which stages raw
// weights in LDS, so no packed fragments are shared between launches of different H).
int wstore_floats_of(int I, int O, int G, int P, bool sub) {
    return (sub ? G : 1) * O * rup(I, 4) + 2 * G * O * rup(O, 4) + 3 * O + 3 * O * G + O * P;
}
int wstore_total(int G, int P, const csmpn_block_params* blocks, int nblk) {
    int m = 0;
    for (int k = 0; k < nblk; ++k)
        m += rup(wstore_floats_of(blocks[k].in_features, blocks[k].out_features, G, P, blocks[k].lin_subspaces != 0), 4);
    return m;
}

int mirror_total(int G, int P, const csmpn_block_params* blocks, int nblk) {
    int m = 0;
    for (int k = 0; k < nblk; ++k)
        m += rup(mirror_floats_of(blocks[k].in_features, blocks[k].out_features, G, P, blocks[k].lin_subspaces != 0), 4);
    return m;
}

// Parity-split kernels: odd n, every block at most 8 output channels, tiles + weight store +
// gradient mirror resident in LDS. The decision does not depend on the direction or the row
// count, so a forward and the backward that reads its saved block inputs always agree.
bool decide_ps(AlgId id, int n, const csmpn_block_params* blocks, int nblk) {
    if (!has_ps(id)) return false;
    // Default: on for D = 32 (n = 5: half the registers per tensor and every lane column in use
    // instead of 8 of 16 - S3 runs 1.9x faster), off for Cl(3,0), where it measured 15-20 % slower
    // than the 32-row layout. CSMPN_FORCE_PS=0|1 overrides.
    if (sw().force_ps >= 0 ? sw().force_ps == 0 : n < 5) return false;
    for (int k = 0; k < nblk; ++k) if (blocks[k].out_features > 8) return false;
    const int D = 1 << n, G = n + 1;
    const size_t mirror = (size_t)mirror_total(G, n_paths(id), blocks, nblk) * 4;
    const size_t wst = (size_t)wstore_total(G, n_paths(id), blocks, nblk) * 4;
    // worst case: backward without saved inputs, forward with the widest staging row
    const TileLayout Lb = tile_layout(D, 1, blocks, nblk, true, 8 * D, false, true);
    const TileLayout Lf = tile_layout(D, 1, blocks, nblk, false, 8 * D, false, true);
    const Choice cb = choose_variant(1, (size_t)Lb.total * 4, mirror, wst, true, true);
    const Choice cf = choose_variant(1, (size_t)Lf.total * 4, 0, wst, false, true);
    const int need = n >= 5 ? 1 : 4;   // D = 32: one 16-row tile per CU is all the LDS holds in any layout
    return cb.var == VAR_WAVE && cf.var == VAR_WAVE && cb.rt * cb.wgs >= need && cf.rt * cf.wgs >= need;
}

int decide_h(AlgId id, int n, const csmpn_block_params* blocks, int nblk, bool bwd, int stage_rowlen,
             bool use_saved, long rows) {
    int maxO = 0;
    for (int k = 0; k < nblk; ++k) maxO = blocks[k].out_features > maxO ? blocks[k].out_features : maxO;
    if (maxO > 8 || !has_h2(id)) return 1;
    const int D = 1 << n, G = n + 1;
    const TileLayout L2 = tile_layout(D, 2, blocks, nblk, bwd, stage_rowlen, use_saved);
    const size_t mirror = bwd ? (size_t)mirror_total(G, n_paths(id), blocks, nblk) * 4 : 0;
    const size_t wst = (size_t)wstore_total(G, n_paths(id), blocks, nblk) * 4;
    const Choice c2 = choose_variant(1, (size_t)L2.total * 4, mirror, wst, bwd);
    if (c2.var != VAR_WAVE || c2.rt * c2.wgs < 2) return 1;
    // 32-row tiles only when there are enough of them to occupy every wave slot of the chip;
    // small row counts (e.g. the node update of a 10k-node complex) get 16-row tiles
    if (sw().force_h) return sw().force_h;   // debugging aid
    const long tiles2 = (rows + 31) / 32;
    if (tiles2 < 256L * c2.rt * c2.wgs) return 1;
    return 2;
}

// bwd / stage_rowlen decide the footprint. stage_rowlen: dense staging row length needed in
// buf_g (edge forward scatter).
bool general_phased_shape(int n, const csmpn_block_params* blocks, int nblk);   // below, beside the saved-region size

int make_plan(AlgId id, int n, const csmpn_block_params* blocks, const csmpn_block_grads* grads, int nblk,
              void* workspace, size_t workspace_bytes, bool bwd, int stage_rowlen, bool use_saved, long rows,
              Plan& plan, bool deterministic = false) {
    if (nblk < 1 || nblk > CSMPN_MAX_BLOCKS) return fail(CSMPN_ERR_INVALID, "n_blocks=%d not in 1..%d", nblk, CSMPN_MAX_BLOCKS);
    const int D = 1 << n, G = n + 1, P = n_paths(id);
    memset(&plan, 0, sizeof(plan));
    plan.workspace = workspace;
    plan.workspace_bytes = workspace_bytes;
    DevCemlp& C = plan.C;
    C.nblk = nblk;
    int maxO = 0;
    for (int k = 0; k < nblk; ++k) {
        const csmpn_block_params& b = blocks[k];
        if (b.in_features < 1 || b.out_features < 1) return fail(CSMPN_ERR_INVALID, "block %d: bad feature counts", k);
        if (k > 0 && b.in_features != blocks[k - 1].out_features)
            return fail(CSMPN_ERR_INVALID, "block %d: in_features %d != previous out_features %d", k, b.in_features,
                        blocks[k - 1].out_features);
        if (!b.lin_w || !b.silu_a || !b.silu_b || !b.gp_w || !b.norm_a || !b.right_w || !b.left_w || !b.left_b || !b.ln_a)
            return fail(CSMPN_ERR_INVALID, "block %d: null parameter pointer", k);
        maxO = b.out_features > maxO ? b.out_features : maxO;
    }
    const bool ps = decide_ps(id, n, blocks, nblk);
    const int H = ps ? 1 : decide_h(id, n, blocks, nblk, bwd, stage_rowlen, use_saved, rows);
    const int NW = ps ? 8 : 16 / H;
    const int MT = ps ? 1 : cdiv(maxO, NW);
    plan.ps = ps;
    if (MT > 4) return fail(CSMPN_ERR_UNSUPPORTED, "out_features %d > 64 not supported", maxO);
    C.MT = MT;
    C.H = H;
    plan.H = H;

    const size_t need = packed_f4_count(G, H, blocks, nblk) * sizeof(f4);
    if (workspace_bytes < need || !workspace) return fail(CSMPN_ERR_INVALID, "workspace too small: %zu < %zu", workspace_bytes, need);
    f4* ws = reinterpret_cast<f4*>(workspace);
    PackDesc& PD = plan.P;
    PD.G = G;
    PD.H = H;
    size_t cursor = 0;
    int mirror = 0, wstore = 0;
    auto add_seg = [&](const float* w, int O, int I, int has_grades, int transposed, int NT, int KK) -> const f4* {
        PackSeg& s = PD.seg[PD.nseg++];
        s.w = w; s.dst = ws + cursor; s.O = O; s.I = I; s.has_grades = has_grades; s.transposed = transposed;
        s.NT = NT; s.KK = KK; s.count = NT * KK * H * G * 64;
        PD.total += s.count;
        const f4* p = s.dst;
        cursor += (size_t)s.count;
        return p;
    };
    for (int k = 0; k < nblk; ++k) {
        const csmpn_block_params& b = blocks[k];
        DevBlock& B = C.b[k];
        B.I = b.in_features; B.O = b.out_features;
        B.KKi = cdiv(B.I, 16); B.KKo = cdiv(B.O, 16);
        B.NTi = cdiv(B.I, NW); B.NTo = cdiv(B.O, NW);
        B.CPi = rup(B.I, 4); B.CPo = rup(B.O, 4);
        B.has_b1 = b.lin_b != nullptr;
        B.w1_sub = b.lin_subspaces ? 1 : 0;
        B.b1 = b.lin_b; B.sa = b.silu_a; B.sb = b.silu_b; B.w = b.gp_w; B.an = b.norm_a; B.bL = b.left_b; B.la = b.ln_a;
        B.pfW1 = add_seg(b.lin_w, B.O, B.I, B.w1_sub, 0, B.NTo, B.KKi);
        B.pfWR = add_seg(b.right_w, B.O, B.O, 1, 0, B.NTo, B.KKo);
        B.pfWL = add_seg(b.left_w, B.O, B.O, 1, 0, B.NTo, B.KKo);
        B.pbW1 = add_seg(b.lin_w, B.O, B.I, B.w1_sub, 1, B.NTi, B.KKo);
        B.pbWR = add_seg(b.right_w, B.O, B.O, 1, 1, B.NTo, B.KKo);
        B.pbWL = add_seg(b.left_w, B.O, B.O, 1, 1, B.NTo, B.KKo);
        B.W1 = b.lin_w; B.WR = b.right_w; B.WL = b.left_w;
        B.lds_goff = mirror;
        B.lds_woff = wstore;
        wstore += rup(wstore_floats_of(B.I, B.O, G, P, B.w1_sub != 0), 4);
        mirror += rup(mirror_floats_of(B.I, B.O, G, P, B.w1_sub), 4);
        if (bwd) {
            if (!grads) return fail(CSMPN_ERR_INVALID, "grads is null");
            const csmpn_block_grads& g = grads[k];
            if (!g.lin_w || !g.silu_a || !g.silu_b || !g.gp_w || !g.norm_a || !g.right_w || !g.left_w || !g.left_b ||
                !g.ln_a || (B.has_b1 && !g.lin_b))
                return fail(CSMPN_ERR_INVALID, "block %d: null gradient pointer", k);
            B.gW1 = g.lin_w; B.gb1 = g.lin_b; B.gsa = g.silu_a; B.gsb = g.silu_b; B.gw = g.gp_w; B.gan = g.norm_a;
            B.gWR = g.right_w; B.gWL = g.left_w; B.gbL = g.left_b; B.gla = g.ln_a;
        }
    }
    plan.pack_f4 = cursor;

    // buffers of one row tile (floats)
    TileLayout L = tile_layout(D, H, blocks, nblk, bwd, stage_rowlen, use_saved, ps);
    // choose the storage variant, row tiles per workgroup and workgroups per CU
    Choice ch = choose_variant(MT, (size_t)L.total * 4, bwd ? (size_t)mirror * 4 : 0, (size_t)wstore * 4, bwd, ps);
    // Backward, when LDS (not registers) limits the resident waves: let z alias the input buffer and
    // stage the input tile a second time for the MVLinear weight gradient, if that buys a row tile
    // per CU (S2: 3 -> 4 waves per CU) or a better storage variant. Needs every block's input to be
    // re-stageable from memory: a single block, or saved block inputs.
    C.share_inz = 0;
    const bool allow_share = !sw().no_share;
    if (bwd && !ps && H == 1 && allow_share && (use_saved || nblk == 1)) {
        const TileLayout Ls = tile_layout(D, H, blocks, nblk, bwd, stage_rowlen, use_saved, ps, true);
        const Choice cs = choose_variant(MT, (size_t)Ls.total * 4, (size_t)mirror * 4, (size_t)wstore * 4, bwd, ps);
        // resident waves per CU: the backward kernels hold ~500 VGPRs, one wave per SIMD at most
        auto resident = [&](const Choice& c) { const int w = c.rt * c.wgs * MT; return w < 4 ? w : 4; };
        if (cs.var < ch.var || (cs.var == ch.var && resident(cs) > resident(ch))) {
            L = Ls; ch = cs; C.share_inz = 1;
        }
    }
    // Phased backward (round 3, cemlp_kernel.hpp): block by block, last first, each over all row tiles - the LDS mirror then
    // holds ONE block's gradient tensors. Taken when that puts more waves on the CU than the all-blocks mirror allows (md17's
    // 32-channel edge model: 110 KB of mirror left room for one 38 KB row tile = 2 waves per CU; 57 KB leave room for two).
    // Needs the saved block inputs and the hand-over region behind them (general_phased_shape: the same predicate sizes it).
    C.phased = 0;
    int mirror_used = mirror;
    const bool no_phased = sw().no_phased;
    // Only where the all-blocks form already keeps a mirror (never away from the no-mirror variant: switching the md17 task
    // model's small node stages to the mirror form cost 6 % of its step) and for launches of at least one row tile per CU
    // (measured on the md17 model, 11 266 adjacencies: step 4.62 ms with a 32 k-row threshold, 4.47 ms with 4 k or 8 k).
    const long phased_min_rows = sw().phased_min_rows;
    // From the no-mirror variant (per-tile float atomics onto the workgroup's copy) to the phased mirror form only for larger
    // launches (M32 node stage, 10 k rows: 0.59 -> 0.48 ms; the md17 model's 940-row node stages lose).
    const bool from_nm = ch.var == VAR_GROUP_NM && rows >= 2 * phased_min_rows;
    if (bwd && use_saved && nblk > 1 && !ps && H == 1 && !no_phased && rows >= phased_min_rows && (ch.var == VAR_GROUP || from_nm) &&
        general_phased_shape(n, blocks, nblk)) {
        int mirror_max = 0;
        for (int k = 0; k < nblk; ++k) {
            const int m = rup(mirror_floats_of(C.b[k].I, C.b[k].O, G, P, C.b[k].w1_sub), 4);
            mirror_max = m > mirror_max ? m : mirror_max;
        }
        auto resident = [&](const Choice& c) { const int w = c.rt * c.wgs * MT; return w < 4 ? w : 4; };
        for (int sh = 0; sh < (allow_share ? 2 : 1); ++sh) {
            const TileLayout Lp = tile_layout(D, H, blocks, nblk, bwd, stage_rowlen, use_saved, ps, sh != 0);
            const Choice cp = choose_variant(MT, (size_t)Lp.total * 4, (size_t)mirror_max * 4, (size_t)wstore * 4, bwd, ps);
            if (cp.var == VAR_GROUP && (resident(cp) > resident(ch) || (from_nm && !C.phased))) {
                L = Lp; ch = cp; C.share_inz = sh; C.phased = 1; mirror_used = mirror_max;
            }
        }
        if (C.phased)
            for (int k = 0; k < nblk; ++k) C.b[k].lds_goff = 0;
    }
    // Deterministic mode on these kernels (n <= 3: Cl(2,0), Cl(3,0) widths outside the lane kernels - the md17 / NBA layers):
    // ONE row tile per workgroup, so that every gradient word (LDS mirror or the workgroup's global copy) has one writing
    // wave - the MT waves of a tile own disjoint channels - and the order of its sums is the tile order.
    plan.det_general = false;
    if (deterministic && n <= 3 && !ps && ch.var != VAR_GLOBAL) {
        plan.det_general = true;
        if (bwd) ch.rt = 1;
    }
    C.off_in = L.off_in; C.off_p0 = L.off_p0; C.off_p1 = L.off_p1; C.off_z = L.off_z; C.off_g = L.off_g;
    C.off_red = L.off_red; C.off_idx = L.off_idx; C.tile_floats = L.total;
    const size_t tile_bytes = (size_t)L.total * 4;
    if (ps && ch.var != VAR_WAVE) return fail(CSMPN_ERR_INVALID, "internal: parity-split plan without the single-wave variant");
    if (H == 2 && ch.var != VAR_WAVE) return fail(CSMPN_ERR_INVALID, "internal: H=2 without the single-wave variant");
    C.RT = ch.rt;
    plan.var = ch.var;
    C.mirror_floats = ch.mirror ? mirror_used : 0;
    C.wstore_floats = ch.var == VAR_WAVE ? wstore : 0;
    if (ch.var != VAR_GLOBAL) {
        C.gtiles = nullptr;
        plan.lds_bytes = (size_t)(C.mirror_floats + C.wstore_floats) * 4 + (size_t)ch.rt * tile_bytes;
        plan.grid_cap = 256u * (unsigned)ch.wgs;
    } else {
        // tiles too large for the LDS: keep them in a global scratch behind the packed weights
        const size_t scratch = (size_t)kGlobalTileGrid * C.RT * tile_bytes;
        if (workspace_bytes < need + scratch)
            return fail(CSMPN_ERR_INVALID, "workspace too small: %zu < %zu", workspace_bytes, need + scratch);
        C.gtiles = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + need);
        plan.lds_bytes = 0;
        plan.grid_cap = kGlobalTileGrid;
    }
    plan.threads = (unsigned)(C.RT * MT * 64);
    return CSMPN_OK;
}

// Deterministic mode on the general kernels: every workgroup of a backward launch accumulates into its own zeroed copy of
// the gradient tensors ([kDetGroups, slice] floats at the end of the workspace, reference layouts back to back);
// det_reduce_kernel adds the copies in a fixed order.
constexpr int kDetGroups = 512;
int det_slice_floats_of(const csmpn_block_params* blocks, int nblk, int G, int P) {
    int m = 0;
    for (int k = 0; k < nblk; ++k)
        m += rup(mirror_floats_of(blocks[k].in_features, blocks[k].out_features, G, P, blocks[k].lin_subspaces != 0), 4);
    return m;
}
size_t det_slice_bytes(int n, const csmpn_block_params* blocks, int nblk) {
    if (n > 3 || nblk < 1) return 0;
    // upper bound over the algebras with n generators: paths <= (n + 1)^3 (Cl(3,0): 20 of 64, Cl(2,0): 10 of 27)
    const int G = n + 1, P = n == 3 ? 20 : (n == 2 ? 10 : (n + 1) * (n + 1) * (n + 1));
    return (size_t)det_slice_floats_of(blocks, nblk, G, P) * sizeof(float) * kDetGroups + 256;
}
struct DetMap {
    int n;
    int total;
    struct { float* dst; int off; int count; } t[40];
};
// grads += sum over the workgroups' copies, fixed order: one thread per word (consecutive threads read consecutive
// words of a copy), eight copies in flight
__global__ void __launch_bounds__(256) det_reduce_kernel(const DetMap M, const float* slices, int nslices) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= M.total) return;
    float s = 0.f;
    int w = 0;
    for (; w + 8 <= nslices; w += 8) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = slices[(size_t)(w + i) * M.total + e];
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    for (; w < nslices; ++w) s += slices[(size_t)w * M.total + e];
    for (int i = 0; i < M.n; ++i)
        if (e >= M.t[i].off && e < M.t[i].off + M.t[i].count) { M.t[i].dst[e - M.t[i].off] += s; return; }
}

int run_pack(const Plan& plan, hipStream_t st) {
    if (plan.P.total == 0 || plan.var == VAR_WAVE) return CSMPN_OK;   // VAR_WAVE stages raw weights in LDS
    const unsigned block = 256, grid = (unsigned)((plan.P.total + block - 1) / block);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(grid), dim3(block), 0, st, plan.P);
    HIP_TRY(hipGetLastError());
    return CSMPN_OK;
}

unsigned long long* g_stamps = nullptr;   // diagnostic builds: device buffer of cycle accumulators

// bytes of the lane kernels' partial buffer (one slice of parameter-gradient sums per workgroup of a backward launch),
// reserved at the END of the workspace; 0 when the shape is not served by those kernels
bool cm_bwd_enabled();
// 16-row-tile MFMA-mixing kernels for Cl(3,0) (cemlp_pq.hpp): weight-fragment tables + one gradient slice per workgroup
size_t pq_region_bytes(int nblk, int ch, int i0) {
    size_t best = 0;
    for (int mode : {MODE_EDGE, MODE_NODE, MODE_PLAIN}) {
        const int na = mode == MODE_EDGE ? i0 - ch : (mode == MODE_NODE ? i0 - 2 * ch : i0);
        const size_t tf = cemlp_pq_table_floats_n3(mode, nblk, ch, na);
        if (!tf) continue;
        const size_t b = (tf + cemlp_pq_slice_floats_n3(mode, nblk, ch, na) * kPqGridCap) * sizeof(float) + 1024;
        best = b > best ? b : best;
    }
    return best;
}
// standalone CEMLPs served by the same family (MODE_PLAIN of cemlp_pq.hpp: the md17 embeddings and head): their saved buffer holds,
// under CSMPN_FLAG_SAVE_STATE, the state regions of the family (one block: nothing else; two blocks: block-1 inputs + hand-over rows)
bool pq_plain_shape(int n, const csmpn_block_params* blocks, int nblk) {
    if (n != 3 || nblk < 1 || nblk > 2 || sw().no_pq || sw().no_cm || !cm_bwd_enabled()) return false;
    for (int k = 0; k < nblk; ++k)
        if (blocks[k].out_features != 32 || (k > 0 && blocks[k].in_features != 32)) return false;
    return cemlp_pq_table_floats_n3(MODE_PLAIN, nblk, 32, blocks[0].in_features) != 0;
}
size_t rl_partial_bytes(int n, const csmpn_block_params* blocks, int nblk) {
    if (n != 3 || nblk < 1 || nblk > 2) return 0;
    const int ch = blocks[0].out_features;
    for (int k = 0; k < nblk; ++k)
        if (blocks[k].out_features != ch || (k > 0 && blocks[k].in_features != ch)) return 0;
    // (row, channel)-per-lane backward (cemlp_cl.hpp): one slice per workgroup
    const int i0 = blocks[0].in_features;
    size_t clf = cemlp_cl_partial_floats_n3(MODE_EDGE, nblk, ch, i0);
    const size_t cln = cemlp_cl_partial_floats_n3(MODE_NODE, nblk, ch, i0);
    clf = cln > clf ? cln : clf;
    size_t cl = clf * sizeof(float) * kClMaxBwdGroups;
    // channel-MFMA backward (cemlp_cmb.hpp / cemlp_cmp.hpp): the same region and slice layout
    size_t cmf = cemlp_cm_partial_floats_n3(MODE_EDGE, nblk, ch, i0);
    const size_t cmn = cemlp_cm_partial_floats_n3(MODE_NODE, nblk, ch, i0);
    cmf = cmn > cmf ? cmn : cmf;
    const size_t cm = cmf * sizeof(float) * kCmSliceCap;
    const size_t lane = cm > cl ? cm : cl;
    const size_t pq = pq_region_bytes(nblk, ch, i0);   // the same region serves whichever family takes the launch
    return pq > lane ? pq : lane;
}
// The channel-MFMA backward (cemlp_cmb.hpp, round 4: two waves per SIMD, tensors parked in LDS) serves the 16-channel
// Cl(3,0) layers; CSMPN_NO_CM_BWD=1 leaves them to the row-per-lane backward (A/B measurements). Read ONCE per process: the
// size of the saved region the caller allocates (csmpn_cemlp_saved_floats_per_row) depends on it and must not change
// under a live plan.
bool cm_bwd_enabled() {
    return !sw().no_cm_bwd;
}
// the (row, channel)-per-lane backward hands d/d(block-1 input) from its block-1 launch to its block-0 launch through
// one more [rows, C, D] region behind the saved block inputs (as the wide parity-lane kernels do)
bool cl_shape(int n, const csmpn_block_params* blocks, int nblk) {
    if (n != 3 || nblk != 2) return false;
    const int ch = blocks[0].out_features, i0 = blocks[0].in_features;
    if (blocks[1].out_features != ch || blocks[1].in_features != ch) return false;
    if (has_cemlp_cl_n3(MODE_EDGE, nblk, ch, i0) || has_cemlp_cl_n3(MODE_NODE, nblk, ch, i0)) return true;
    return cm_bwd_enabled() && (has_cemlp_cm_n3(MODE_EDGE, nblk, ch, i0, true) || has_cemlp_cm_n3(MODE_NODE, nblk, ch, i0, true));
}

// Shapes whose backward may run block by block on the general kernels (cemlp_kernel.hpp, `phased`): small algebras, more
// than one block, no lane-kernel family of their own. They get a hand-over region as large as the saved inputs behind them.
bool general_phased_shape(int n, const csmpn_block_params* blocks, int nblk) {
    if (n > 3 || nblk < 2 || nblk > CSMPN_MAX_BLOCKS) return false;
    if (n == 3 && nblk == 2) {
        const int ch = blocks[0].out_features, i0 = blocks[0].in_features;
        if (blocks[1].out_features == ch && blocks[1].in_features == ch) {
            if (cl_shape(n, blocks, nblk)) return false;
        }
    }
    return true;
}

// bytes of the wide parity-lane kernels' rotation tables (cemlp_plw.hpp), also carved from the END of the workspace
// (never together with the row-per-lane region: different algebras). Upper bound over the entry points.
// ... and of the backward's partial buffer: one slice of weight-gradient MFMA tiles per workgroup (upper bound)
size_t plw_part_bytes(int ch) {
    const size_t NG = (ch + 7) / 8, nch0 = 2 * NG + 1;
    const size_t image = 8 * NG * (3 + 3 * 6 + 64) + 16;   // per-channel sums (CP x (3 + 3 G + P)), generous
    const size_t per_cu = 4 / NG > 0 ? 4 / NG : 1;
    size_t bytes = ((nch0 + 2 * NG) * 12 * 64 * NG + image) * sizeof(float) * kPlwMaxGroups * per_cu + 256;
    if (ch == 8) {   // the 8-channel parity-lane backward (cemlp_pl.hpp): one slice per wave, 4 waves x 256 workgroups
        const size_t pl = (size_t)(8 * 768 + 2 * 640) * sizeof(float) * 4 * kPlMaxBwdGroups + 256;
        bytes = pl > bytes ? pl : bytes;
    }
    return bytes;
}
size_t plw_table_bytes(int n, const csmpn_block_params* blocks, int nblk) {
    if (n != 5 || nblk < 1 || nblk > 2) return 0;
    const int ch = blocks[0].out_features;
    if (ch < 8 || ch > 32) return 0;
    if (nblk == 2 && (blocks[1].out_features != ch || blocks[1].in_features != ch)) return 0;
    const size_t NG = (ch + 7) / 8, nch0 = 2 * NG + 1;
    // (+ 64 KB: the weight-fragment tables of cemlp_pg.hpp, carved from the same region, are up to 368 KB at 28 / 32 channels)
    return ((2 * NG * nch0 + 4 * NG * NG) + (2 * NG * NG + 4 * NG * NG)) * 384 * sizeof(float) + 256 + plw_part_bytes(ch) + (ch > 16 ? 65536 : 0);
}

// (row, channel)-per-lane kernels (cemlp_cl.hpp): Cl(3,0), two blocks of 8 channels, the EGCL attribute widths of S1.
// CSMPN_NO_CL=1 leaves these shapes to the row-per-lane kernels (A/B measurements, parity tests of both paths).
bool cl_eligible(AlgId id, const Plan& plan, int mode, bool bwd, const RowIO& io, int* channels, int* i0) {
    if (sw().no_cl || id != ALG_N3) return false;
    const DevCemlp& C = plan.C;
    if (C.nblk != 2) return false;
    const int ch = C.b[0].O;
    for (int k = 0; k < C.nblk; ++k) {
        if (C.b[k].O != ch || !C.b[k].w1_sub) return false;
        if (k > 0 && C.b[k].I != ch) return false;
    }
    if (mode == MODE_EDGE && (io.seg[0].ch != ch || C.b[0].I != ch + (io.nseg > 1 ? io.seg[1].ch : 0))) return false;
    if (mode == MODE_NODE && (io.seg[0].ch != ch || io.seg[1].ch != ch || C.b[0].I != 2 * ch + (io.nseg > 2 ? io.seg[2].ch : 0))) return false;
    if (mode != MODE_EDGE && mode != MODE_NODE) return false;
    if (bwd && !io.saved) return false;
    *channels = ch;
    *i0 = C.b[0].I;
    if (!has_cemlp_cl_n3(mode, C.nblk, ch, C.b[0].I)) return false;
    if (bwd) {
        const size_t pb = cemlp_cl_partial_floats_n3(mode, C.nblk, ch, C.b[0].I) * sizeof(float) * kClMaxBwdGroups;
        if (!plan.workspace || plan.workspace_bytes < pb) return false;
    }
    return true;
}

// channel-MFMA kernels (cemlp_cm.hpp): Cl(3,0), two blocks of 16 channels (S2) or - forward only - 32 channels (md17), the EGCL
// attribute widths (6, 3).
// CSMPN_NO_CM=1 leaves these shapes to the row-per-lane kernels (A/B measurements, parity tests of both paths).
bool cm_eligible(AlgId id, const Plan& plan, int mode, bool bwd, const RowIO& io, int* channels, int* i0) {
    if (sw().no_cm || id != ALG_N3) return false;
    const DevCemlp& C = plan.C;
    if (C.nblk != 2) return false;
    const int ch = C.b[0].O;
    for (int k = 0; k < C.nblk; ++k) {
        if (C.b[k].O != ch || !C.b[k].w1_sub) return false;
        if (k > 0 && C.b[k].I != ch) return false;
    }
    if (mode == MODE_EDGE && (io.seg[0].ch != ch || C.b[0].I != ch + (io.nseg > 1 ? io.seg[1].ch : 0))) return false;
    if (mode == MODE_NODE && (io.seg[0].ch != ch || io.seg[1].ch != ch || C.b[0].I != 2 * ch + (io.nseg > 2 ? io.seg[2].ch : 0))) return false;
    if (mode != MODE_EDGE && mode != MODE_NODE) return false;
    if (bwd && !io.saved) return false;
    *channels = ch;
    *i0 = C.b[0].I;
    if (!has_cemlp_cm_n3(mode, C.nblk, ch, C.b[0].I, bwd)) return false;
    if (bwd) {
        if (!cm_bwd_enabled()) return false;
        const size_t pb = cemlp_cm_partial_floats_n3(mode, C.nblk, ch, C.b[0].I) * sizeof(float) * kCmSliceCap;
        if (!plan.workspace || plan.workspace_bytes < pb) return false;
    }
    return true;
}

// parity-lane kernels (cemlp_pl.hpp): Cl(5,0) / Cl(4,1), two blocks of 8 channels, the EGCL attribute widths of S3
bool pl_eligible(AlgId id, const Plan& plan, int mode, bool bwd, const RowIO& io, int* i0) {
    if (sw().no_pl || (id != ALG_N5 && id != ALG_N5M)) return false;
    const DevCemlp& C = plan.C;
    if (C.nblk != 2) return false;
    for (int k = 0; k < C.nblk; ++k) {
        if (C.b[k].O != 8 || !C.b[k].w1_sub) return false;
        if (k > 0 && C.b[k].I != 8) return false;
    }
    if (mode == MODE_EDGE && io.seg[0].ch != 8) return false;
    if (mode == MODE_NODE && (io.seg[0].ch != 8 || io.seg[1].ch != 8)) return false;
    if (mode != MODE_EDGE && mode != MODE_NODE) return false;
    if (bwd && !io.saved) return false;
    *i0 = C.b[0].I;
    return id == ALG_N5 ? has_cemlp_pl_n5(mode, C.nblk, 8, *i0) : has_cemlp_pl_n5m(mode, C.nblk, 8, *i0);
}

// wide parity-lane kernels (cemlp_plw.hpp): Cl(5,0) / Cl(4,1), two blocks of 16 / 24 / 28 / 32 channels, EGCL edge / node programs
bool plw_eligible(AlgId id, const Plan& plan, int mode, bool bwd, const RowIO& io, int* channels, int* attr) {
    if (sw().no_plw || (id != ALG_N5 && id != ALG_N5M)) return false;
    const DevCemlp& C = plan.C;
    if (C.nblk < 1 || C.nblk > 2) return false;
    const int ch = C.b[0].O;
    const bool plw8 = sw().plw8;   // 8 channels: wide kernels with one group
    // 8 channels belong to cemlp_pl.hpp; the one-group wide kernels take them only on request (round 2 measured them 4-5x
    // slower - compiled for four waves per SIMD by mistake, 1.3 KB of scratch; with the launch bounds repaired they are on a
    // par: S3 1.339 against 1.333 ms)
    if ((ch <= 8 && !(ch == 8 && plw8)) || ch > 32 || !C.b[0].w1_sub) return false;
    if (C.nblk == 2 && (C.b[1].O != ch || C.b[1].I != ch || !C.b[1].w1_sub)) return false;
    int na = 0;
    if (mode == MODE_EDGE) {
        if (io.seg[0].ch != ch) return false;
        na = io.nseg > 1 ? io.seg[1].ch : 0;
        if (C.b[0].I != ch + na) return false;
    } else if (mode == MODE_NODE) {
        if (io.seg[0].ch != ch || io.seg[1].ch != ch) return false;
        na = io.nseg > 2 ? io.seg[2].ch : 0;
        if (C.b[0].I != 2 * ch + na) return false;
    } else {
        na = C.b[0].I;                      // standalone CEMLP: its (<= 8) input channels are the one input chunk
        if (na < 1 || na > 8 || io.nseg != 1) return false;
    }
    const size_t tf = id == ALG_N5 ? cemlp_plw_table_floats_n5(mode, ch, na, C.nblk) : cemlp_plw_table_floats_n5m(mode, ch, na, C.nblk);
    if (tf == 0 || !plan.workspace || plan.workspace_bytes < tf * sizeof(float) + plw_part_bytes(ch) + 1024) return false;
    if (bwd && C.nblk > 1 && !io.saved) return false;
    *channels = ch;
    *attr = na;
    return true;
}

// 16-row-tile MFMA-mixing kernels (cemlp_pg.hpp): Cl(5,0) / Cl(4,1), two blocks of 24 / 28 / 32 channels, EGCL edge / node programs
bool pg_eligible(AlgId id, const Plan& plan, int mode, bool bwd, const RowIO& io, int* channels, int* attr) {
    if (sw().no_pg || (id != ALG_N5 && id != ALG_N5M)) return false;
    const DevCemlp& C = plan.C;
    if (C.nblk != 2) return false;
    const int ch = C.b[0].O;
    if (ch <= 16 || ch > 32 || !C.b[0].w1_sub || C.b[1].O != ch || C.b[1].I != ch || !C.b[1].w1_sub) return false;
    int na = 0;
    if (mode == MODE_EDGE) {
        if (io.seg[0].ch != ch) return false;
        na = io.nseg > 1 ? io.seg[1].ch : 0;
        if (C.b[0].I != ch + na) return false;
    } else if (mode == MODE_NODE) {
        if (io.seg[0].ch != ch || io.seg[1].ch != ch) return false;
        na = io.nseg > 2 ? io.seg[2].ch : 0;
        if (C.b[0].I != 2 * ch + na) return false;
    } else {
        return false;
    }
    if (!(id == ALG_N5 ? has_cemlp_pg_n5(mode, ch, na, bwd) : has_cemlp_pg_n5m(mode, ch, na, bwd))) return false;
    const size_t tf = id == ALG_N5 ? cemlp_pg_table_floats_n5(mode, ch, na) : cemlp_pg_table_floats_n5m(mode, ch, na);
    if (tf == 0 || !plan.workspace || plan.workspace_bytes < tf * sizeof(float) + plw_part_bytes(ch) + 1024) return false;
    // the backward of this family runs on the state its forward saved (CSMPN_FLAG_SAVE_STATE, in ITS lane order): without
    // the flag the forward still serves (it writes the row-major block-1 inputs every backward reads) and the wide
    // parity-lane backward recomputes from them
    if (bwd && !(io.saved && io.save_state)) return false;
    *channels = ch;
    *attr = na;
    return true;
}

// CSMPN_FLAG_WEIGHTS_PACKED on an EGCL backward entry point: the workspace is the one the stage's forward used with the same
// parameters - the weight-fragment tables of the 16-row-tile families (written by that forward's pack launch, both
// directions) are still there and the backward does not pack again. Set around run_rows by those entry points.
thread_local bool g_tables_ready = false;

// 16-row-tile MFMA-mixing kernels for Cl(3,0) (cemlp_pq.hpp): two blocks of 32 channels, EGCL edge / node programs
bool pq_eligible(AlgId id, const Plan& plan, int mode, bool bwd, const RowIO& io, int* channels, int* attr) {
    if (sw().no_pq || sw().no_cm || id != ALG_N3) return false;
    const DevCemlp& C = plan.C;
    if (C.nblk != 2 && !(C.nblk == 1 && mode == MODE_PLAIN)) return false;
    const int ch = C.b[0].O;
    if (ch != 32 || !C.b[0].w1_sub) return false;
    if (C.nblk == 2 && (C.b[1].O != ch || C.b[1].I != ch || !C.b[1].w1_sub)) return false;
    int na = 0;
    if (mode == MODE_EDGE) {
        if (io.seg[0].ch != ch) return false;
        na = io.nseg > 1 ? io.seg[1].ch : 0;
        if (C.b[0].I != ch + na) return false;
    } else if (mode == MODE_NODE) {
        if (io.seg[0].ch != ch || io.seg[1].ch != ch) return false;
        na = io.nseg > 2 ? io.seg[2].ch : 0;
        if (C.b[0].I != 2 * ch + na) return false;
    } else {
        // standalone CEMLP (the md17 embeddings and head): one contiguous input of I0 channels; not the fused embedding
        if (io.nseg != 1 || io.emb_nperm != 0 || io.seg[0].ch != C.b[0].I) return false;
        na = C.b[0].I;
    }
    const size_t tf = cemlp_pq_table_floats_n3(mode, C.nblk, ch, na);
    if (tf == 0 || !plan.workspace || plan.workspace_bytes < pq_region_bytes(C.nblk, ch, C.b[0].I)) return false;
    // the backward runs on the state its forward saved (CSMPN_FLAG_SAVE_STATE, in ITS lane order); without the flag the
    // forward still serves (it writes the row-major block-1 inputs) and the wave-pair backward (cemlp_cmp.hpp) recomputes
    if (bwd && !(io.saved && io.save_state && cm_bwd_enabled())) return false;
    *channels = ch;
    *attr = na;
    return true;
}

int run_rows(AlgId id, const Plan& plan, int mode, bool bwd, const RowIO& io_in, hipStream_t st, bool need_pack) {
    if (io_in.rows <= 0) return CSMPN_OK;
    RowIO io = io_in;
    io.stamps = g_stamps;
    {
        int channels = 0, attr = 0;
        if (pq_eligible(id, plan, mode, bwd, io, &channels, &attr)) {
            const long tiles = (io.rows + 15) / 16;          // one 16-row tile per workgroup iteration, three 4-wave workgroups per CU
            // (backward: a workgroup ends with one slice of weight-gradient tiles, 62-78 KB; measured with 1 / 2 / 3 tiles per
            // workgroup on launches below the cap: md17 step 2.11 / 2.28 / 2.43 ms, M32 1.024 / 1.008 / 1.012e8 edges/s: one tile)
            const long want = tiles;
            const unsigned grid = (unsigned)(want < (long)kPqGridCap ? (want > 0 ? want : 1) : kPqGridCap);
            const int nblk = plan.C.nblk;
            const size_t tb = cemlp_pq_table_floats_n3(mode, nblk, channels, attr) * sizeof(float);
            float* tabs = reinterpret_cast<float*>(static_cast<char*>(plan.workspace) + ((plan.workspace_bytes - tb - 16) & ~(size_t)255));
            io.plw_part = reinterpret_cast<float*>(reinterpret_cast<char*>(tabs) - cemlp_pq_slice_floats_n3(mode, nblk, channels, attr) * sizeof(float) * kPqGridCap);
            if (bwd) io.plw_g1 = const_cast<float*>(io.saved) + (size_t)io.rows * channels * 8;   // hand-over rows behind the saved block inputs
            if (!cm_bwd_enabled()) io.save_state = 0;   // no state regions in the saved buffer (state_channels())
            bool handled = false;
            if (sw().debug) fprintf(stderr, "[csmpn] pq mode=%d bwd=%d channels=%d attr=%d grid=%u rows=%ld\n", mode, (int)bwd, channels, attr, grid, io.rows);
            HIP_TRY(launch_cemlp_pq_n3(mode, nblk, channels, attr, bwd, !(bwd && g_tables_ready), grid, st, plan.C, io, tabs, &handled));
            if (handled) {
                note_kernel("csmpn::cemlp_pq_%s_kernel<%s, ...> (mode %d, %d channels, %d %s channels, %d block%s)", bwd ? "bwd" : "fwd",
                            alg_name(id), mode, channels, attr, mode == MODE_PLAIN ? "input" : "attribute", nblk, nblk > 1 ? "s" : "");
                return CSMPN_OK;
            }
        }
    }
    {
        int channels = 0, attr = 0;
        if (pg_eligible(id, plan, mode, bwd, io, &channels, &attr)) {
            const long tiles = (io.rows + 15) / 16;          // one 16-row tile per workgroup iteration, one 8-wave workgroup per CU
            const unsigned grid = (unsigned)(tiles < 256 ? tiles : 256);
            const size_t tb = (id == ALG_N5 ? cemlp_pg_table_floats_n5(mode, channels, attr) : cemlp_pg_table_floats_n5m(mode, channels, attr)) * sizeof(float);
            float* tabs = reinterpret_cast<float*>(static_cast<char*>(plan.workspace) + ((plan.workspace_bytes - tb - 16) & ~(size_t)255));
            io.plw_part = reinterpret_cast<float*>(reinterpret_cast<char*>(tabs) - plw_part_bytes(channels));
            if (bwd) io.plw_g1 = const_cast<float*>(io.saved) + (size_t)io.rows * channels * 32;   // hand-over rows behind the saved block inputs
            bool handled = false;
            if (sw().debug) fprintf(stderr, "[csmpn] pg mode=%d bwd=%d channels=%d attr=%d grid=%u rows=%ld\n", mode, (int)bwd, channels, attr, grid, io.rows);
            const bool pack = !(bwd && g_tables_ready);
            if (id == ALG_N5) HIP_TRY(launch_cemlp_pg_n5(mode, channels, attr, bwd, pack, grid, st, plan.C, io, tabs, &handled));
            else HIP_TRY(launch_cemlp_pg_n5m(mode, channels, attr, bwd, pack, grid, st, plan.C, io, tabs, &handled));
            if (handled) {
                note_kernel("csmpn::cemlp_pg_%s_kernel<%s, ...> (mode %d, %d channels, %d attribute channels)", bwd ? "bwd" : "fwd",
                            alg_name(id), mode, channels, attr);
                return CSMPN_OK;
            }
        }
    }
    {
        int channels = 0, attr = 0;
        if (plw_eligible(id, plan, mode, bwd, io, &channels, &attr)) {
            const long tiles = (io.rows + 3) / 4;          // one 4-row tile per workgroup iteration
            const long per_cu = 4 / ((channels + 7) / 8) > 0 ? 4 / ((channels + 7) / 8) : 1;   // workgroups of NG waves per CU at one wave per SIMD
            const long cap_plw = (bwd ? kPlwMaxGroups : 2 * kPlwMaxGroups) * per_cu;   // forward: twice that where LDS allows
            const unsigned grid = (unsigned)(tiles < cap_plw ? tiles : cap_plw);
            const size_t tb = (id == ALG_N5 ? cemlp_plw_table_floats_n5(mode, channels, attr, plan.C.nblk)
                                            : cemlp_plw_table_floats_n5m(mode, channels, attr, plan.C.nblk)) * sizeof(float);
            float* tabs = reinterpret_cast<float*>(static_cast<char*>(plan.workspace) + ((plan.workspace_bytes - tb - 16) & ~(size_t)255));
            io.plw_part = reinterpret_cast<float*>(reinterpret_cast<char*>(tabs) - plw_part_bytes(channels));
            if (bwd && plan.C.nblk > 1) io.plw_g1 = const_cast<float*>(io.saved) + (size_t)io.rows * channels * 32;   // see csmpn_cemlp_saved_floats_per_row
            bool handled = false;
            const bool debug_plw = sw().debug;
            if (debug_plw) fprintf(stderr, "[csmpn] plw mode=%d bwd=%d channels=%d attr=%d grid=%u rows=%ld\n", mode, (int)bwd, channels, attr, grid, io.rows);
            if (id == ALG_N5) HIP_TRY(launch_cemlp_plw_n5(mode, channels, attr, plan.C.nblk, bwd, grid, st, plan.C, io, tabs, &handled));
            else HIP_TRY(launch_cemlp_plw_n5m(mode, channels, attr, plan.C.nblk, bwd, grid, st, plan.C, io, tabs, &handled));
            if (handled) {   // the wide kernels' template arguments live in plw_inst.inc: family + shape
                note_kernel("csmpn::cemlp_plw_%s_kernel<%s, ...> (mode %d, %d channels, %d attribute channels, %d blocks)", bwd ? "bwd" : "fwd",
                            alg_name(id), mode, channels, attr, plan.C.nblk);
                return CSMPN_OK;
            }
        }
    }
    {
        int i0 = 0;
        if (pl_eligible(id, plan, mode, bwd, io, &i0)) {
            const long tiles = (io.rows + 3) / 4;          // 4 rows per wave tile
            const long cap = bwd ? kPlMaxBwdGroups : 512;  // one / two 4-wave workgroups per CU
            const long groups = (tiles + 3) / 4;
            const unsigned grid = (unsigned)(groups < cap ? groups : cap);
            if (bwd) {   // per-wave slices of parameter-gradient sums: at the end of the workspace (as the wide kernels' region)
                const size_t pb = plw_part_bytes(8);
                if (!plan.workspace || plan.workspace_bytes < pb + 1024) return fail(CSMPN_ERR_INVALID, "workspace too small for the parity-lane backward");
                io.plw_part = reinterpret_cast<float*>(static_cast<char*>(plan.workspace) + ((plan.workspace_bytes - pb - 16) & ~(size_t)255));
            }
            bool handled = false;
            const bool debug_pl = sw().debug;
            if (debug_pl) fprintf(stderr, "[csmpn] pl mode=%d bwd=%d i0=%d grid=%u rows=%ld\n", mode, (int)bwd, i0, grid, io.rows);
            if (id == ALG_N5) HIP_TRY(launch_cemlp_pl_n5(mode, plan.C.nblk, 8, i0, bwd, grid, st, plan.C, io, &handled));
            else HIP_TRY(launch_cemlp_pl_n5m(mode, plan.C.nblk, 8, i0, bwd, grid, st, plan.C, io, &handled));
            if (handled) {
                note_kernel("csmpn::cemlp_pl_kernel<%s, %d, %d, %d, %s, %s>", alg_name(id), mode, plan.C.nblk, i0, bwd ? "true" : "false",
                            bwd && io.save_state ? "true" : "false");
                return CSMPN_OK;
            }
        }
    }
    {
        int channels = 0, i0 = 0;
        if (cl_eligible(id, plan, mode, bwd, io, &channels, &i0)) {
            const long rows_per_wave = 64 / channels;
            const long tiles = (io.rows + rows_per_wave - 1) / rows_per_wave;
            // tile t belongs to wave t % (4 grid): four 4-wave workgroups per CU in the forward (~100 VGPRs), two in a
            // block backward (<= 256)
            long cap = bwd ? kClMaxBwdGroups : kClMaxFwdGroups;
            // experiments: fewer resident workgroups (never more: the partial buffer has kClMaxBwdGroups slices)
            const long cap_f = sw().cl_cap_fwd, cap_b = sw().cl_cap_bwd;
            if (!bwd && cap_f > 0 && cap_f < 4096) cap = cap_f;
            if (bwd && cap_b > 0 && cap_b < cap) cap = cap_b;
            const long groups = (tiles + 3) / 4;
            const unsigned grid = (unsigned)(groups < cap ? groups : cap);
            if (bwd) {
                const size_t pb = cemlp_cl_partial_floats_n3(mode, plan.C.nblk, channels, i0) * sizeof(float) * kClMaxBwdGroups;
                io.rl_partials = reinterpret_cast<float*>(static_cast<char*>(plan.workspace) + ((plan.workspace_bytes - pb) & ~(size_t)15));
                io.plw_g1 = const_cast<float*>(io.saved) + (size_t)io.rows * channels * 8;   // see csmpn_cemlp_saved_floats_per_row
            }
            bool handled = false;
            const bool debug_cl = sw().debug;
            if (debug_cl) fprintf(stderr, "[csmpn] cl mode=%d bwd=%d channels=%d i0=%d grid=%u rows=%ld\n", mode, (int)bwd, channels, i0, grid, io.rows);
            HIP_TRY(launch_cemlp_cl_n3(mode, plan.C.nblk, channels, i0, bwd, grid, st, plan.C, io, &handled));
            if (handled) {
                note_kernel("csmpn::cemlp_cl_%s_kernel<%s, %d, %d, %d, %d%s>", bwd ? "bwd" : "fwd", alg_name(id), channels, mode, plan.C.nblk,
                            i0 - (mode == MODE_EDGE ? 1 : 2) * channels, !bwd ? "" : (io.save_state ? ", true" : ", false"));
                return CSMPN_OK;
            }
        }
    }
    {
        int channels = 0, i0 = 0;
        if (cm_eligible(id, plan, mode, bwd, io, &channels, &i0)) {
            const long tiles = (io.rows + 15) / 16;   // tile t (16 rows) belongs to wave t % (4 grid)
            const long cap = bwd ? kCmMaxBwdGroups : (channels == 16 ? kCmMaxFwdGroups : 256);   // 32 channels: one workgroup per CU
            // tiles per workgroup and pass: 4 (one per wave), 8 in the 16-channel backward (8-wave workgroups), 2 in the
            // 32-channel backward (a wave PAIR per tile: with 4 the 59 tiles of an md17 batch's node launch went to 15
            // workgroups, two tiles after each other per pair, while 241 CUs idled)
            const long per_group = !bwd ? 4 : (channels == 32 ? 2 : 4);
            const long groups = (tiles + per_group - 1) / per_group;
            const unsigned grid = (unsigned)(groups < cap ? groups : cap);
            if (bwd) {
                const size_t pb = cemlp_cm_partial_floats_n3(mode, plan.C.nblk, channels, i0) * sizeof(float) * kCmSliceCap;
                io.rl_partials = reinterpret_cast<float*>(static_cast<char*>(plan.workspace) + ((plan.workspace_bytes - pb) & ~(size_t)15));
                io.plw_g1 = const_cast<float*>(io.saved) + (size_t)io.rows * channels * 8;   // see csmpn_cemlp_saved_floats_per_row
            }
            // the state regions of the 32-channel forward exist only while its pair backward is enabled (state_channels():
            // under CSMPN_NO_CM_BWD=1 the saved buffer holds block inputs + the general kernels' hand-over slots, nothing else)
            if (channels == 32 && !cm_bwd_enabled()) io.save_state = 0;
            bool handled = false;
            const bool debug_cm = sw().debug;
            if (debug_cm) fprintf(stderr, "[csmpn] cm mode=%d bwd=%d channels=%d i0=%d grid=%u rows=%ld\n", mode, (int)bwd, channels, i0, grid, io.rows);
            HIP_TRY(launch_cemlp_cm_n3(mode, plan.C.nblk, channels, i0, bwd, grid, st, plan.C, io, &handled));
            if (handled) {
                note_kernel("csmpn::cemlp_%s_kernel<%s, %d, %d, %d, %d%s>", !bwd ? "cm_fwd" : (channels == 32 ? "cmp" : "cmb"), alg_name(id), channels,
                            mode, plan.C.nblk, i0 - (mode == MODE_EDGE ? 1 : 2) * channels,
                            bwd && channels == 32 ? (io.save_state && plan.C.nblk > 1 ? ", true" : ", false") : "");
                return CSMPN_OK;
            }
        }
    }
    if ((id == ALG_N5 || id == ALG_N5M) && bwd && mode != MODE_PLAIN && io.rows >= 4096 && !getenv("CSMPN_QUIET")) {
        // a D = 32 EGCL stage outside the parity-lane widths: served, but by the general row-tile kernels whose backward
        // spills (4.9-5.8 KB of scratch per lane: DESIGN.md §4.6) - say so once instead of being silently slow
        static std::atomic<bool> warned{false};
        if (!warned.exchange(true))
            fprintf(stderr, "[csmpn] note: Cl(5,0) / Cl(4,1) layer with %d channels runs on the general row-tile kernels (slow path: "
                            "their backward spills registers). The parity-lane kernels serve two-block EGCL layers of 8, 16, 24, 28 "
                            "or 32 channels. (CSMPN_QUIET=1 silences this note.)\n", plan.C.b[0].O);
    }
    if (io.row_store && !plan.det_general)
        return fail(CSMPN_ERR_UNSUPPORTED,
                    "CSMPN_FLAG_DETERMINISTIC: this shape is served neither by the lane kernels (Cl(3,0) 8 / 16 channels, "
                    "Cl(5,0) / Cl(4,1) 8 / 16 / 24 / 28 / 32 channels; two blocks with saved block inputs) nor by the deterministic "
                    "form of the general kernels (n <= 3, tiles and gradient mirror resident in LDS)");
    size_t det_bytes = 0;
    float* det_slices = nullptr;
    DetMap det_map;
    int det_total = 0;
    // Per-workgroup copies of the gradient tensors at the end of the workspace: always in deterministic mode, and (round 3)
    // for every backward of the small algebras - the parameter-gradient atomics of ALL row tiles onto one copy were the
    // bulk of the md17-width backward (M32 node stage 1.04 -> 0.53 ms with private copies); CSMPN_NO_SLICED_GRADS=1: off.
    const bool no_sliced = sw().no_sliced;
    const bool sliced = bwd && (io.row_store || (!no_sliced && (id == ALG_N2 || id == ALG_N3) && !plan.ps && plan.var != VAR_GLOBAL));
    if (sliced) {
        const int G = (id == ALG_N2) ? 3 : 4, P = n_paths(id);   // det_general: n <= 3
        det_map.n = 0;
        for (int k = 0; k < plan.C.nblk; ++k) {
            const DevBlock& B = plan.C.b[k];
            auto add = [&](float* dst, int count) {
                if (dst) { det_map.t[det_map.n].dst = dst; det_map.t[det_map.n].off = det_total; det_map.t[det_map.n].count = count; ++det_map.n; }
                det_total += count;
            };
            add(B.gW1, (B.w1_sub ? G : 1) * B.O * B.I); add(B.gWR, G * B.O * B.O); add(B.gWL, G * B.O * B.O);
            add(B.has_b1 ? B.gb1 : nullptr, B.O); add(B.gsa, B.O * G); add(B.gsb, B.O * G); add(B.gw, B.O * P);
            add(B.gan, B.O * G); add(B.gbL, B.O); add(B.gla, B.O);
            det_total = rup(det_total, 4);
        }
        det_map.total = det_total;
        det_bytes = (size_t)det_total * sizeof(float) * kDetGroups + 256;
        if (!plan.workspace || plan.workspace_bytes < det_bytes) {
            if (io.row_store)
                return fail(CSMPN_ERR_INVALID, "workspace too small for the deterministic backward: %zu < %zu", plan.workspace_bytes, det_bytes);
            det_bytes = 0;   // a caller's smaller workspace: atomics onto the one copy
        } else {
            det_slices = reinterpret_cast<float*>(static_cast<char*>(plan.workspace) + ((plan.workspace_bytes - det_bytes) & ~(size_t)255));
        }
    }
    if (bwd && plan.C.phased) {   // hand-over region of the phased backward: behind the saved inputs, laid out like them
        size_t ch_saved = 0;
        for (int k = 0; k + 1 < plan.C.nblk; ++k) ch_saved += (size_t)plan.C.b[k].O;
        const int Dn = id == ALG_N2 ? 4 : (id == ALG_N3 ? 8 : 2);
        io.plw_g1 = const_cast<float*>(io.saved) + ch_saved * (size_t)io.rows * Dn;
    }
    // general row-tile kernels from here on: they read packed weight fragments (the lane kernels above do not)
    if (need_pack) {
        const int rcp = run_pack(plan, st);
        if (rcp) return rcp;
    }
    const long R = 16 * plan.H;
    const long ntiles = (io.rows + R - 1) / R;
    // few tiles (e.g. the node update of a 10k-node complex): fewer row tiles per workgroup,
    // so that the tiles spread over all CUs instead of filling a few of them
    DevCemlp Cd = plan.C;
    unsigned threads = plan.threads;
    size_t lds_bytes = plan.lds_bytes;
    if (plan.var != VAR_GLOBAL && Cd.RT > 1 && !bwd) {   // backward: per-workgroup mirror flush outweighs the spread (measured)
        long rt = (ntiles + 255) / 256;
        if (rt < 1) rt = 1;
        if (rt < Cd.RT) {
            lds_bytes -= (size_t)(Cd.RT - rt) * Cd.tile_floats * 4;
            Cd.RT = (int)rt;
            threads = (unsigned)(Cd.RT * Cd.MT * 64);
        }
    }
    long grid = (ntiles + Cd.RT - 1) / Cd.RT;
    if (grid > (long)plan.grid_cap) grid = plan.grid_cap;
    if (det_bytes && grid > kDetGroups) grid = kDetGroups;
    const bool debug = sw().debug;
    if (debug)
        fprintf(stderr, "[csmpn] mode=%d bwd=%d var=%d ps=%d share=%d phased=%d H=%d MT=%d RT=%d threads=%u lds=%zu grid=%ld tile_floats=%d mirror=%d rows=%ld\n",
                mode, (int)bwd, plan.var, (int)plan.ps, Cd.share_inz, (int)(bwd && Cd.phased), plan.H, Cd.MT, Cd.RT, threads, lds_bytes, grid,
                Cd.tile_floats, Cd.mirror_floats, io.rows);
    if (det_bytes) {
        // the kernels' accumulators = copy 0 of the zeroed region; workgroup b adds b * det_slice_floats
        HIP_TRY(hipMemsetAsync(det_slices, 0, (size_t)grid * det_total * sizeof(float), st));
        Cd.det_slice_floats = det_total;
        for (int i = 0, k = 0, j = 0; k < Cd.nblk; ++k) {
            DevBlock& B = Cd.b[k];
            float** ptrs[10] = {&B.gW1, &B.gWR, &B.gWL, &B.gb1, &B.gsa, &B.gsb, &B.gw, &B.gan, &B.gbL, &B.gla};
            const int G = (id == ALG_N2) ? 3 : 4, P = n_paths(id);
            const int counts[10] = {(B.w1_sub ? G : 1) * B.O * B.I, G * B.O * B.O, G * B.O * B.O, B.O, B.O * G, B.O * G, B.O * P, B.O * G, B.O, B.O};
            for (int q = 0; q < 10; ++q) {
                if (q == 3 && !B.has_b1) { j += counts[q]; continue; }
                *ptrs[q] = det_slices + j;
                j += counts[q];
            }
            j = rup(j, 4);
            (void)i;
        }
    }
    if (plan.ps) HIP_TRY(launch_cemlp_ps(id, mode, bwd, (unsigned)grid, threads, lds_bytes, st, Cd, io));
    else HIP_TRY(launch_cemlp(id, mode, plan.var, plan.H, bwd, (unsigned)grid, threads, lds_bytes, st, Cd, io));
    if (plan.ps) note_kernel("csmpn::cemlp_ps_kernel<%s, %d, %s>", alg_name(id), mode, bwd ? "true" : "false");
    else note_kernel("csmpn::cemlp_kernel<%s, %d, %d, %d, %s>", alg_name(id), mode, plan.var, plan.H, bwd ? "true" : "false");
    if (det_bytes) {
        hipLaunchKernelGGL(det_reduce_kernel, dim3((det_total + 255) / 256), dim3(256), 0, st, det_map, (const float*)det_slices, (int)grid);
        HIP_TRY(hipGetLastError());
    }
    return CSMPN_OK;
}

// ----------------------------------------------------------------------------- standalone MVLinear
// (cegnn_utils.py:287-338) for callers outside a CEMLP (projection heads, feature embeddings):
//   y[b,o,d] = sum_i W[o,i,grade(d)] x[b,i,d]  (+ bias[o] on blade 0);  W [O,I,G] or [O,I].
// HBM-bound ([rows, I, D] in, [rows, O, D] out); one thread per output element, blade index
// fastest so that a wave reads / writes whole rows; the grade table depends on n only.
struct MvLinDesc {
    const float *x, *w, *b, *gy;
    float *y, *gx, *gw, *gb;
    long rows;
    int I, O, D, G, sub;
    unsigned char grade[32];
};

__global__ void mvlinear_fwd_kernel(const MvLinDesc P) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = P.rows * P.O * P.D;
    if (e >= total) return;
    const int d = (int)(e % P.D);
    const int o = (int)((e / P.D) % P.O);
    const long r = e / ((long)P.D * P.O);
    const int g = P.sub ? P.grade[d] : 0, ws = P.sub ? P.G : 1;
    const float* xr = P.x + r * P.I * P.D + d;
    const float* wr = P.w + (long)o * P.I * ws + g;
    float acc = (P.b && d == 0) ? P.b[o] : 0.f;
    for (int i = 0; i < P.I; ++i) acc = fmaf(wr[i * ws], xr[(long)i * P.D], acc);
    P.y[e] = acc;
}

// d/dx[b,i,d] = sum_o gy[b,o,d] W[o,i,grade(d)]
__global__ void mvlinear_bwd_x_kernel(const MvLinDesc P) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = P.rows * P.I * P.D;
    if (e >= total) return;
    const int d = (int)(e % P.D);
    const int i = (int)((e / P.D) % P.I);
    const long r = e / ((long)P.D * P.I);
    const int g = P.sub ? P.grade[d] : 0, ws = P.sub ? P.G : 1;
    const float* gr = P.gy + r * P.O * P.D + d;
    const float* wc = P.w + (long)i * ws + g;
    float acc = 0.f;
    for (int o = 0; o < P.O; ++o) acc = fmaf(wc[(long)o * P.I * ws], gr[(long)o * P.D], acc);
    P.gx[e] = acc;
}

// d/dW[o,i,g] += sum_{rows, d in g} gy[b,o,d] x[b,i,d];  d/dbias[o] += sum_rows gy[b,o,0].
// Thread = one weight (or bias) element, blockIdx.y = a slab of rows it walks (x / gy rows of a slab are L1 / L2 hits:
// a wave's 64 elements share o or neighbour it); one atomic per element and slab. The slab is sized so that the launch
// has ~512 workgroups (8..64 rows): the first version gave every workgroup ALL elements of a 4-row slab - 235 workgroups x 1 152
// atomics onto the same 1 152 addresses took 28 us on the 940 rows of an md17 batch.
inline int mvlinear_slab(long rows, int elem_blocks) {
    long slabs = 512 / elem_blocks;
    if (slabs < 1) slabs = 1;
    long s = (rows + slabs - 1) / slabs;
    return (int)(s < 8 ? 8 : (s > 64 ? 64 : s));
}
__global__ void mvlinear_bwd_w_kernel(const MvLinDesc P, int slab) {
    const long r0 = (long)blockIdx.y * slab;
    const long r1 = r0 + slab < P.rows ? r0 + slab : P.rows;
    const int ws = P.sub ? P.G : 1;
    const int nw = P.O * P.I * ws;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nw + P.O) return;
    float acc = 0.f;
    if (e < nw) {
        if (!P.gw) return;
        const int g = e % ws, i = (e / ws) % P.I, o = e / (ws * P.I);
        // blades of grade g are contiguous (blade order: by grade): [d0, d1)
        int d0 = 0, d1 = P.D;
        if (P.sub) {
            while (P.grade[d0] != g) ++d0;
            d1 = d0;
            while (d1 < P.D && P.grade[d1] == g) ++d1;
        }
        for (long r = r0; r < r1; ++r) {
            const float* gr = P.gy + (r * P.O + o) * P.D;
            const float* xr = P.x + (r * P.I + i) * P.D;
            for (int d = d0; d < d1; ++d) acc = fmaf(gr[d], xr[d], acc);
        }
        atomicAdd(P.gw + e, acc);
    } else if (P.gb) {
        const int o = e - nw;
        for (long r = r0; r < r1; ++r) acc += P.gy[(r * P.O + o) * P.D];
        atomicAdd(P.gb + o, acc);
    }
}

int mvlinear_desc(int n, long rows, int I, int O, int sub, MvLinDesc& P) {
    if (n < 1 || n > 5) return fail(CSMPN_ERR_UNSUPPORTED, "MVLinear: n=%d not in 1..5", n);
    if (I < 1 || O < 1 || rows < 0) return fail(CSMPN_ERR_INVALID, "MVLinear: bad sizes rows=%ld I=%d O=%d", rows, I, O);
    memset(&P, 0, sizeof(P));
    P.rows = rows; P.I = I; P.O = O; P.D = 1 << n; P.G = n + 1; P.sub = sub ? 1 : 0;
    // blade order: by grade, then lexicographic (metric.py:18-29): the grade of index d is the
    // grade whose cumulative binomial range contains d
    int d = 0, c = 1;   // c = C(n, g)
    for (int g = 0; g <= n; ++g) {
        for (int k = 0; k < c; ++k) P.grade[d++] = (unsigned char)g;
        c = c * (n - g) / (g + 1);
    }
    return CSMPN_OK;
}

}  // namespace

// =============================================================================== C-ABI
extern "C" {

const char* csmpn_last_error(void) { return g_csmpn_err; }
const char* csmpn_last_kernel(void) { return g_last_kernel; }

#ifdef CSMPN_STAMPS
/* diagnostic library only (`make stamps`, never shipped): device buffer of 25 uint64 per-phase cycle sums + wave count */
void csmpn_debug_set_stamps(void* device_u64x25) { g_stamps = static_cast<unsigned long long*>(device_u64x25); }
#endif
int csmpn_abi_version(void) { return 2; }   // 2 (round 5): csmpn_embed_cemlp_* take n_vertex_rows, csmpn_cemlp_saved_floats takes flags;
                                            // CSMPN_FLAG_WEIGHTS_PACKED has a meaning on the EGCL / CEMLP backward entry points, CSMPN_FLAG_SAVE_STATE on csmpn_cemlp_*
const char* csmpn_build_target(void) { return "gfx950"; }

int csmpn_metric_supported(const float* metric_host, int n) { return alg_id(metric_host, n) != ALG_NONE ? 1 : 0; }

int csmpn_algebra_tables(const float* metric, int n, float* cayley, int64_t* index_to_bitmap, int64_t* bitmap_to_index,
                         int64_t* grades, int64_t* subspaces, uint8_t* paths) {
    if (!metric || n < 1 || n > 8) return fail(CSMPN_ERR_INVALID, "n=%d not in 1..8", n);
    const int D = 1 << n, G = n + 1;
    std::vector<int> bm(D), idx(D), gr(D);
    int pos = 0;
    for (int g = 0; g <= n; ++g) {
        // combinations of g generators in lexicographic order (metric.py:18-29)
        std::vector<int> comb(g);
        for (int t = 0; t < g; ++t) comb[t] = t;
        while (true) {
            int b = 0;
            for (int t = 0; t < g; ++t) b |= 1 << comb[t];
            bm[pos] = b; gr[pos] = g; idx[b] = pos; ++pos;
            int t = g - 1;
            while (t >= 0 && comb[t] == n - g + t) --t;
            if (t < 0) break;
            ++comb[t];
            for (int u = t + 1; u < g; ++u) comb[u] = comb[u - 1] + 1;
        }
    }
    if (index_to_bitmap) for (int i = 0; i < D; ++i) index_to_bitmap[i] = bm[i];
    if (bitmap_to_index) for (int i = 0; i < D; ++i) bitmap_to_index[i] = idx[i];
    if (grades) for (int i = 0; i < D; ++i) grades[i] = gr[i];
    if (subspaces) {
        for (int g = 0; g < G; ++g) subspaces[g] = 0;
        for (int i = 0; i < D; ++i) subspaces[gr[i]] += 1;
    }
    if (cayley) memset(cayley, 0, sizeof(float) * (size_t)D * D * D);
    if (paths) memset(paths, 0, (size_t)G * G * G);
    for (int i = 0; i < D; ++i)
        for (int k = 0; k < D; ++k) {
            const unsigned a = (unsigned)bm[i], b = (unsigned)bm[k];
            // metric.py:50-79: reordering sign times the metric of the shared generators
            int s = 0;
            for (unsigned t = a >> 1; t; t >>= 1) s += popcount_u(t & b);
            float coeff = (s & 1) ? -1.0f : 1.0f;
            for (int bit = 0; bit < n; ++bit)
                if ((a & b) >> bit & 1u) coeff *= metric[bit];
            const int j = idx[a ^ b];
            if (cayley) cayley[((size_t)i * D + j) * D + k] = coeff;
            if (paths && coeff != 0.0f) paths[(gr[i] * G + gr[j]) * G + gr[k]] = 1;
        }
    return CSMPN_OK;
}

int csmpn_geometric_product_forward(const float* metric, int n, const float* a, const float* b, float* out,
                                    int64_t rows, void* stream) {
    const AlgId id = alg_id(metric, n);
    if (id == ALG_NONE) return fail(CSMPN_ERR_UNSUPPORTED, "metric not supported by the HIP path");
    HIP_TRY(launch_gp(id, false, a, b, nullptr, out, nullptr, nullptr, rows, (hipStream_t)stream));
    return CSMPN_OK;
}

int csmpn_geometric_product_backward(const float* metric, int n, const float* a, const float* b, const float* gout,
                                     float* ga, float* gb, int64_t rows, void* stream) {
    const AlgId id = alg_id(metric, n);
    if (id == ALG_NONE) return fail(CSMPN_ERR_UNSUPPORTED, "metric not supported by the HIP path");
    HIP_TRY(launch_gp(id, true, a, b, gout, nullptr, ga, gb, rows, (hipStream_t)stream));
    return CSMPN_OK;
}

// channels (x D floats) per row of the CSMPN_FLAG_SAVE_STATE regions (cemlp_device.hpp: whole row tiles in the kernels' lane
// order, one region per tensor and block, rows rounded up to 16):
//   Cl(3,0) 8 channels (cemlp_cl.hpp)                    s of every block
//   Cl(3,0) 32 channels (cemlp_cm.hpp / cemlp_cmp.hpp)    s, y, R of every block
//   Cl(5,0) / Cl(4,1), 8 .. 32 channels (cemlp_pl.hpp / cemlp_plw.hpp)   s, y, R of every block, channels padded to groups of 8
static size_t state_channels(int n, const csmpn_block_params* blocks, int n_blocks) {
    if (pq_plain_shape(n, blocks, n_blocks)) return (size_t)3 * n_blocks * 32;   // cemlp_pq.hpp, MODE_PLAIN: s, y, R of every block
    if (n_blocks != 2) return 0;
    const size_t ch = (size_t)blocks[0].out_features;
    // (17 .. 32 channels: 32 - the 16-row-tile kernels of cemlp_pg.hpp keep 4 channels per wave, 8 waves per tile)
    if (plw_table_bytes(n, blocks, n_blocks)) return (size_t)3 * n_blocks * (ch > 16 ? 32 : (ch + 7) / 8 * 8);
    if (cl_shape(n, blocks, n_blocks)) {
        if (has_cemlp_cl_n3(MODE_EDGE, n_blocks, (int)ch, blocks[0].in_features) || has_cemlp_cl_n3(MODE_NODE, n_blocks, (int)ch, blocks[0].in_features))
            return (size_t)n_blocks * ch;
        if (ch == 32 && cm_bwd_enabled()) return (size_t)3 * n_blocks * 32;
    }
    return 0;
}
// channels per row in front of the state regions: the saved block inputs and the hand-over region(s)
static size_t base_channels(int n, const csmpn_block_params* blocks, int n_blocks) {
    size_t ch = 0;
    for (int k = 0; k + 1 < n_blocks; ++k) ch += (size_t)blocks[k].out_features;
    // wide parity-lane backward (cemlp_plw.hpp): one more [rows, O, D] region behind the saved inputs, the hand-over
    // of d/d(block-1 input) from its block-1 launch to its block-0 launch
    if (n_blocks == 2 && plw_table_bytes(n, blocks, n_blocks)) ch += (size_t)blocks[0].out_features;
    if (cl_shape(n, blocks, n_blocks)) ch += (size_t)blocks[0].out_features;   // the (row, channel)-per-lane / channel-MFMA backward likewise
    else if (general_phased_shape(n, blocks, n_blocks)) ch *= 2;               // the general kernels' phased backward: one hand-over slot per saved input
    return ch;
}

size_t csmpn_cemlp_saved_floats(int n, const csmpn_block_params* blocks, int n_blocks, int64_t rows, uint32_t flags) {
    if (rows <= 0 || !blocks || n_blocks < 1 || n_blocks > CSMPN_MAX_BLOCKS || n < 1 || n > 8) return 0;
    size_t base = base_channels(n, blocks, n_blocks);
    // the general kernels' hand-over slots (one per saved input: the per-row figure doubles for them) are used by the phased
    // backward only, and make_plan takes that form only from sw().phased_min_rows rows on (the same switch, read once)
    if (base && !cl_shape(n, blocks, n_blocks) && !(n_blocks == 2 && plw_table_bytes(n, blocks, n_blocks)) &&
        general_phased_shape(n, blocks, n_blocks) && rows < sw().phased_min_rows && !pq_plain_shape(n, blocks, n_blocks))
        base /= 2;   // (the 16-row-tile family's standalone CEMLPs always hand d/d(block-1 input) over through the second region)
    const size_t state_rows = (size_t)((rows + 15) & ~(int64_t)15);
    const size_t state = (flags & CSMPN_FLAG_SAVE_STATE) ? state_channels(n, blocks, n_blocks) : 0;
    return ((base * (size_t)rows) << n) + ((state * state_rows) << n);
}

// upper bound per row (the state regions hold up to 15 padding rows more: csmpn_cemlp_saved_floats is exact)
size_t csmpn_cemlp_saved_floats_per_row(int n, const csmpn_block_params* blocks, int n_blocks) {
    if (!blocks || n_blocks < 1 || n_blocks > CSMPN_MAX_BLOCKS || n < 1 || n > 8) return 0;
    return (base_channels(n, blocks, n_blocks) + state_channels(n, blocks, n_blocks)) << n;
}

size_t csmpn_cemlp_workspace_bytes(int n, const csmpn_block_params* blocks, int n_blocks) {
    if (!blocks || n_blocks < 1 || n_blocks > CSMPN_MAX_BLOCKS || n < 1 || n > 8) return 0;
    // H is not known without the metric: reserve for the larger packing (H = 2 when narrow)
    int maxO = 0;
    for (int k = 0; k < n_blocks; ++k) maxO = blocks[k].out_features > maxO ? blocks[k].out_features : maxO;
    size_t bytes = packed_f4_count(n + 1, 1, blocks, n_blocks) * sizeof(f4);
    if (maxO <= 8) {
        const size_t b2 = packed_f4_count(n + 1, 2, blocks, n_blocks) * sizeof(f4);
        bytes = b2 > bytes ? b2 : bytes;
    }
    const int D = 1 << n, MT = cdiv(maxO, 16);
    // worst case over the entry points (H = 1): backward layout / forward layout with the
    // edge-forward staging row
    const TileLayout Lb = tile_layout(D, 1, blocks, n_blocks, true, 0);
    const TileLayout Lf = tile_layout(D, 1, blocks, n_blocks, false, blocks[n_blocks - 1].out_features * D);
    const Choice cb = choose_variant(MT, (size_t)Lb.total * 4, 0, 0, true);
    const Choice cf = choose_variant(MT, (size_t)Lf.total * 4, 0, 0, false);
    // global tile scratch: reserved whenever a launch may choose it (the choice itself needs the
    // metric: path count -> mirror / weight-store size), i.e. for every tile too big to have a
    // few copies in LDS
    size_t scratch = 0;
    const int grt = (4 / MT) > 0 ? 4 / MT : 1;
    if (cb.var == VAR_GLOBAL || (size_t)Lb.total * 4 > 36 * 1024) scratch = (size_t)kGlobalTileGrid * grt * Lb.total * 4;
    if (cf.var == VAR_GLOBAL || (size_t)Lf.total * 4 > 36 * 1024) {
        const size_t s2 = (size_t)kGlobalTileGrid * grt * Lf.total * 4;
        scratch = s2 > scratch ? s2 : scratch;
    }
    const size_t tail = rl_partial_bytes(n, blocks, n_blocks) + plw_table_bytes(n, blocks, n_blocks);
    const size_t det = det_slice_bytes(n, blocks, n_blocks);   // never together with a lane-kernel region: the larger one
    return ((bytes + scratch + 15) & ~(size_t)15) + (tail > det ? tail : det) + 16 + 256;
}

int csmpn_cemlp_forward(const float* metric, int n, const csmpn_block_params* blocks, int n_blocks, const float* x,
                        int64_t rows, float* y, float* save_inputs, void* workspace, size_t workspace_bytes, uint32_t flags, void* stream) {
    const AlgId id = alg_id(metric, n);
    if (id == ALG_NONE) return fail(CSMPN_ERR_UNSUPPORTED, "metric not supported by the HIP path");
    Plan plan;
    int rc = make_plan(id, n, blocks, nullptr, n_blocks, workspace, workspace_bytes, false, 0, false, rows, plan,
                       (flags & CSMPN_FLAG_DETERMINISTIC) != 0);
    if (rc) return rc;
    const bool need_pack = !(flags & CSMPN_FLAG_WEIGHTS_PACKED);   // packed fragments are a matter of the general kernels: run_rows
    RowIO io;
    memset(&io, 0, sizeof(io));
    io.rows = rows; io.nseg = 1;
    io.seg[0].a = x; io.seg[0].ch = blocks[0].in_features; io.seg[0].off = 0;
    io.y = y; io.save = save_inputs;
    // CSMPN_FLAG_SAVE_STATE: honoured for the shapes whose saved buffer has state regions (pq_plain_shape), ignored otherwise
    io.save_state = ((flags & CSMPN_FLAG_SAVE_STATE) && pq_plain_shape(n, blocks, n_blocks)) ? 1 : 0;
    return run_rows(id, plan, MODE_PLAIN, false, io, (hipStream_t)stream, need_pack);
}

int csmpn_cemlp_backward(const float* metric, int n, const csmpn_block_params* blocks, const csmpn_block_grads* grads,
                         int n_blocks, const float* x, const float* gy, int64_t rows, float* gx, const float* saved_inputs, void* workspace,
                         size_t workspace_bytes, uint32_t flags, void* stream) {
    const AlgId id = alg_id(metric, n);
    if (id == ALG_NONE) return fail(CSMPN_ERR_UNSUPPORTED, "metric not supported by the HIP path");
    Plan plan;
    int rc = make_plan(id, n, blocks, grads, n_blocks, workspace, workspace_bytes, true, 0, saved_inputs != nullptr, rows, plan,
                       (flags & CSMPN_FLAG_DETERMINISTIC) != 0);
    if (rc) return rc;
    // fragments packed by the forward are only valid for the forward's own layout choice (a forward
    // with LDS-staged raw weights packs nothing): the backward packs for itself; no-op for VAR_WAVE
    const bool need_pack = true;
    RowIO io;
    memset(&io, 0, sizeof(io));
    io.rows = rows; io.nseg = 1;
    io.seg[0].a = x; io.seg[0].ch = blocks[0].in_features; io.seg[0].off = 0;
    io.gy = gy; io.gx[0] = gx; io.saved = saved_inputs;
    io.row_store = (flags & CSMPN_FLAG_DETERMINISTIC) ? 1 : 0;   // standalone CEMLP: no row table, atomic-free parameter sums only
    // CSMPN_FLAG_SAVE_STATE: honoured exactly where csmpn_cemlp_forward honours it (include/csmpn_hip.h: the shapes of the
    // 16-row-tile family, whose standalone forward writes the state regions); everywhere else the standalone forward writes
    // no state, so a backward that honoured the flag would read rows nobody wrote
    io.save_state = ((flags & CSMPN_FLAG_SAVE_STATE) && pq_plain_shape(n, blocks, n_blocks)) ? 1 : 0;
    g_tables_ready = io.save_state && (flags & CSMPN_FLAG_WEIGHTS_PACKED) != 0;
    rc = run_rows(id, plan, MODE_PLAIN, true, io, (hipStream_t)stream, need_pack);
    g_tables_ready = false;
    return rc;
}

// Fused simplex embedding (include/csmpn_hip.h): MODE_PLAIN of the wide parity-lane kernels with the embed descriptor
__global__ void index_range_kernel(const int* __restrict__ idx, long n, int hi, int* __restrict__ flag) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (idx[i] < 0 || idx[i] >= hi)) atomicOr(flag, 1);
}
__device__ int g_range_flag;            // written by index_range_kernel, read back under g_range_mutex
static std::mutex g_range_mutex;

// every entry of idx[0, n) in [0, hi)? One host round trip (callers that have validated their tables pass
// CSMPN_FLAG_NO_VALIDATE; the kernels clamp in any case)
static int check_index_range(const int32_t* idx, int64_t n, int64_t hi, hipStream_t st, const char* what) {
    if (n <= 0) return CSMPN_OK;
    std::lock_guard<std::mutex> lock(g_range_mutex);
    int* flag = nullptr;
    hipError_t e = hipGetSymbolAddress(reinterpret_cast<void**>(&flag), HIP_SYMBOL(g_range_flag));
    if (e == hipSuccess) e = hipMemsetAsync(flag, 0, sizeof(int), st);
    if (e != hipSuccess) return fail(CSMPN_ERR_HIP, "%s validation: %s", what, hipGetErrorString(e));
    hipLaunchKernelGGL(index_range_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, idx, (long)n,
                       (int)(hi > 0x7fffffff ? 0x7fffffff : hi), flag);
    int host_flag = 0;
    e = hipMemcpyAsync(&host_flag, flag, sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return fail(CSMPN_ERR_HIP, "%s validation: %s", what, hipGetErrorString(e));
    if (host_flag) return fail(CSMPN_ERR_INVALID, "%s has entries outside [0, %lld)", what, (long long)hi);
    return CSMPN_OK;
}

static int embed_io(const AlgId id, const csmpn_block_params* blocks, const float* vertex_feat, int64_t n_vertex_rows, int32_t kpv,
                    const int32_t* verts, int32_t nv, int32_t n_orders, int64_t n_rows, uint32_t flags, hipStream_t st, RowIO& io) {
    if (id != ALG_N5 && id != ALG_N5M) return fail(CSMPN_ERR_UNSUPPORTED, "fused embedding: Cl(5,0) / Cl(4,1) only");
    if (n_orders != 1 && n_orders != 2 && n_orders != 6) return fail(CSMPN_ERR_UNSUPPORTED, "fused embedding: 1, 2 or 6 vertex orders");
    if (!vertex_feat || !verts || kpv < 1 || nv < 1 || nv * kpv != blocks[0].in_features || nv * kpv > 8)
        return fail(CSMPN_ERR_UNSUPPORTED, "fused embedding: verts_per_row * channels_per_vertex must equal in_features (<= 8)");
    if (n_rows % n_orders) return fail(CSMPN_ERR_INVALID, "fused embedding: n_rows %ld is not a multiple of n_orders %d", (long)n_rows, n_orders);
    if (n_vertex_rows < 1 || n_vertex_rows > 0x7fffffff)
        return fail(CSMPN_ERR_INVALID, "fused embedding: n_vertex_rows %lld not in [1, 2^31)", (long long)n_vertex_rows);
    if (!(flags & CSMPN_FLAG_NO_VALIDATE)) {
        const int rc = check_index_range(verts, n_rows * nv, n_vertex_rows, st, "fused embedding: verts");
        if (rc) return rc;
    }
    memset(&io, 0, sizeof(io));
    io.rows = n_rows; io.nseg = 1;
    io.seg[0].a = vertex_feat; io.seg[0].ch = blocks[0].in_features; io.seg[0].off = 0;
    io.emb_verts = verts; io.emb_nperm = n_orders; io.emb_nv = nv; io.emb_k = kpv; io.emb_nrows = (int)n_vertex_rows;
    return CSMPN_OK;
}

int csmpn_embed_cemlp_forward(const float* metric, int n, const csmpn_block_params* blocks, int n_blocks, const float* vertex_feat,
                              int64_t n_vertex_rows, int32_t channels_per_vertex, const int32_t* verts, int32_t verts_per_row, int32_t n_orders,
                              int64_t n_rows, float* out, float* save_inputs, void* workspace, size_t workspace_bytes,
                              uint32_t flags, void* stream) {
    const AlgId id = alg_id(metric, n);
    if (id == ALG_NONE) return fail(CSMPN_ERR_UNSUPPORTED, "metric not supported by the HIP path");
    RowIO io;
    int rc = embed_io(id, blocks, vertex_feat, n_vertex_rows, channels_per_vertex, verts, verts_per_row, n_orders, n_rows, flags,
                      (hipStream_t)stream, io);
    if (rc) return rc;
    Plan plan;
    if ((rc = make_plan(id, n, blocks, nullptr, n_blocks, workspace, workspace_bytes, false, 0, false, n_rows, plan))) return rc;
    io.y = out; io.save = save_inputs;
    int channels = 0, attr = 0;
    if (!plw_eligible(id, plan, MODE_PLAIN, false, io, &channels, &attr))
        return fail(CSMPN_ERR_UNSUPPORTED, "fused embedding: shape not served by the wide parity-lane kernels");
    io.save_state = (flags & CSMPN_FLAG_SAVE_STATE) ? 1 : 0;   // two-block modules: y, R, s of the blocks (as the EGCL stages)
    return run_rows(id, plan, MODE_PLAIN, false, io, (hipStream_t)stream, false);
}

int csmpn_embed_cemlp_backward(const float* metric, int n, const csmpn_block_params* blocks, const csmpn_block_grads* grads,
                               int n_blocks, const float* vertex_feat, int64_t n_vertex_rows, int32_t channels_per_vertex,
                               const int32_t* verts, int32_t verts_per_row, int32_t n_orders, int64_t n_rows, const float* g_out,
                               const float* saved_inputs, void* workspace, size_t workspace_bytes, uint32_t flags, void* stream) {
    const AlgId id = alg_id(metric, n);
    if (id == ALG_NONE) return fail(CSMPN_ERR_UNSUPPORTED, "metric not supported by the HIP path");
    RowIO io;
    int rc = embed_io(id, blocks, vertex_feat, n_vertex_rows, channels_per_vertex, verts, verts_per_row, n_orders, n_rows, flags,
                      (hipStream_t)stream, io);
    if (rc) return rc;
    Plan plan;
    if ((rc = make_plan(id, n, blocks, grads, n_blocks, workspace, workspace_bytes, true, 0, saved_inputs != nullptr, n_rows, plan))) return rc;
    io.gy = g_out; io.saved = saved_inputs;
    int channels = 0, attr = 0;
    if (!plw_eligible(id, plan, MODE_PLAIN, true, io, &channels, &attr))
        return fail(CSMPN_ERR_UNSUPPORTED, "fused embedding: shape not served by the wide parity-lane kernels (two blocks need saved inputs)");
    io.save_state = (flags & CSMPN_FLAG_SAVE_STATE) ? 1 : 0;
    return run_rows(id, plan, MODE_PLAIN, true, io, (hipStream_t)stream, false);
}

int csmpn_mvlinear_forward(int n, const float* x, const float* weight, const float* bias, int64_t rows,
                           int32_t in_features, int32_t out_features, int32_t subspaces, float* y, void* stream) {
    MvLinDesc P;
    int rc = mvlinear_desc(n, (long)rows, in_features, out_features, subspaces, P);
    if (rc) return rc;
    if (rows == 0) return CSMPN_OK;
    if (!x || !weight || !y) return fail(CSMPN_ERR_INVALID, "MVLinear: null pointer");
    P.x = x; P.w = weight; P.b = bias; P.y = y;
    const long total = (long)rows * out_features * P.D;
    const unsigned block = 256, grid = (unsigned)((total + block - 1) / block);
    hipLaunchKernelGGL(mvlinear_fwd_kernel, dim3(grid), dim3(block), 0, (hipStream_t)stream, P);
    HIP_TRY(hipGetLastError());
    return CSMPN_OK;
}

int csmpn_mvlinear_backward(int n, const float* x, const float* weight, const float* gy, int64_t rows,
                            int32_t in_features, int32_t out_features, int32_t subspaces, float* gx, float* g_weight,
                            float* g_bias, void* stream) {
    MvLinDesc P;
    int rc = mvlinear_desc(n, (long)rows, in_features, out_features, subspaces, P);
    if (rc) return rc;
    if (rows == 0) return CSMPN_OK;
    if (!x || !weight || !gy) return fail(CSMPN_ERR_INVALID, "MVLinear: null pointer");
    P.x = x; P.w = weight; P.gy = gy; P.gx = gx; P.gw = g_weight; P.gb = g_bias;
    const unsigned block = 256;
    if (gx) {
        const long total = (long)rows * in_features * P.D;
        hipLaunchKernelGGL(mvlinear_bwd_x_kernel, dim3((unsigned)((total + block - 1) / block)), dim3(block), 0,
                           (hipStream_t)stream, P);
    }
    if (g_weight || g_bias) {   // the bias gradient comes from the same kernel: a frozen weight must not silence it
        const int nelem = out_features * in_features * (P.sub ? P.G : 1) + out_features;
        const unsigned eb = (unsigned)((nelem + block - 1) / block);
        const int slab = mvlinear_slab(rows, (int)eb);
        hipLaunchKernelGGL(mvlinear_bwd_w_kernel, dim3(eb, (unsigned)((rows + slab - 1) / slab)), dim3(block), 0,
                           (hipStream_t)stream, P, slab);
    }
    HIP_TRY(hipGetLastError());
    return CSMPN_OK;
}

int csmpn_egcl_edge_forward(const float* metric, int n, const csmpn_block_params* blocks, int n_blocks, const float* h,
                            int32_t channels, const float* edge_attr, int32_t attr_channels, const int32_t* perm,
                            const int32_t* src_sorted, const int32_t* dst_sorted, int64_t E, int64_t N, float* agg,
                            float* save_inputs, void* workspace, size_t workspace_bytes, uint32_t flags, void* stream) {
    const AlgId id = alg_id(metric, n);
    if (id == ALG_NONE) return fail(CSMPN_ERR_UNSUPPORTED, "metric not supported by the HIP path");
    if (attr_channels > 0 && !edge_attr && E > 0) return fail(CSMPN_ERR_INVALID, "edge_attr is null");   // an empty edge list carries no attribute rows
    if (channels + attr_channels != blocks[0].in_features)
        return fail(CSMPN_ERR_INVALID, "edge model in_features %d != %d + %d", blocks[0].in_features, channels, attr_channels);
    const int D = 1 << n;
    Plan plan;
    int rc = make_plan(id, n, blocks, nullptr, n_blocks, workspace, workspace_bytes, false,
                       blocks[n_blocks - 1].out_features * D, false, E, plan, (flags & CSMPN_FLAG_DETERMINISTIC) != 0);
    if (rc) return rc;
    const bool need_pack = !(flags & CSMPN_FLAG_WEIGHTS_PACKED);   // packed fragments are a matter of the general kernels: run_rows
    RowIO io;
    memset(&io, 0, sizeof(io));
    io.rows = E; io.nseg = attr_channels > 0 ? 2 : 1;
    io.seg[0].a = h; io.seg[0].ia = dst_sorted; io.seg[0].b = h; io.seg[0].ib = src_sorted; io.seg[0].ch = channels;
    io.seg[1].a = edge_attr; io.seg[1].ia = perm; io.seg[1].ch = attr_channels; io.seg[1].off = channels;
    io.agg = agg; io.dst = dst_sorted; io.src = src_sorted; io.perm = perm; io.save = save_inputs;
    io.row_store = (flags & CSMPN_FLAG_DETERMINISTIC) ? 1 : 0;   // agg is then the [E, O, D] message table
    io.save_state = (flags & CSMPN_FLAG_SAVE_STATE) ? 1 : 0;
    (void)N;
    return run_rows(id, plan, MODE_EDGE, false, io, (hipStream_t)stream, need_pack);
}

int csmpn_egcl_edge_backward(const float* metric, int n, const csmpn_block_params* blocks,
                             const csmpn_block_grads* grads, int n_blocks, const float* h, int32_t channels,
                             const float* edge_attr, int32_t attr_channels, const int32_t* perm,
                             const int32_t* src_sorted, const int32_t* dst_sorted, int64_t E, int64_t N,
                             const float* g_agg, float* gh, float* g_edge_attr, const float* saved_inputs, void* workspace,
                             size_t workspace_bytes, uint32_t flags, void* stream) {
    const AlgId id = alg_id(metric, n);
    if (id == ALG_NONE) return fail(CSMPN_ERR_UNSUPPORTED, "metric not supported by the HIP path");
    if (attr_channels > 0 && !edge_attr && E > 0) return fail(CSMPN_ERR_INVALID, "edge_attr is null");   // an empty edge list carries no attribute rows
    if (channels + attr_channels != blocks[0].in_features)
        return fail(CSMPN_ERR_INVALID, "edge model in_features %d != %d + %d", blocks[0].in_features, channels, attr_channels);
    Plan plan;
    int rc = make_plan(id, n, blocks, grads, n_blocks, workspace, workspace_bytes, true, 0, saved_inputs != nullptr, E, plan,
                       (flags & CSMPN_FLAG_DETERMINISTIC) != 0);
    if (rc) return rc;
    // fragments packed by the forward are only valid for the forward's own layout choice (a forward
    // with LDS-staged raw weights packs nothing): the backward packs for itself; no-op for VAR_WAVE
    const bool need_pack = true;
    RowIO io;
    memset(&io, 0, sizeof(io));
    io.rows = E; io.nseg = attr_channels > 0 ? 2 : 1;
    io.seg[0].a = h; io.seg[0].ia = dst_sorted; io.seg[0].b = h; io.seg[0].ib = src_sorted; io.seg[0].ch = channels;
    io.seg[1].a = edge_attr; io.seg[1].ia = perm; io.seg[1].ch = attr_channels; io.seg[1].off = channels;
    io.dst = dst_sorted; io.src = src_sorted; io.perm = perm;
    io.gy = g_agg; io.gx[0] = gh; io.gx[1] = g_edge_attr; io.saved = saved_inputs;
    io.row_store = (flags & CSMPN_FLAG_DETERMINISTIC) ? 1 : 0;   // gh is then the [E, C, D] per-edge gradient table
    io.save_state = (flags & CSMPN_FLAG_SAVE_STATE) ? 1 : 0;
    (void)N;
    g_tables_ready = (flags & CSMPN_FLAG_WEIGHTS_PACKED) != 0;
    rc = run_rows(id, plan, MODE_EDGE, true, io, (hipStream_t)stream, need_pack);
    g_tables_ready = false;
    return rc;
}

static int node_io(const csmpn_block_params* blocks, int n_blocks, const float* h, int channels, const float* agg,
                   int agg_channels, const float* node_attr, int attr_channels, const int32_t* in_degree,
                   int mean_aggr, int residual, int64_t N, RowIO& io) {
    if (attr_channels > 0 && !node_attr) return fail(CSMPN_ERR_INVALID, "node_attr is null");
    if (channels + agg_channels + attr_channels != blocks[0].in_features)
        return fail(CSMPN_ERR_INVALID, "node model in_features %d != %d + %d + %d", blocks[0].in_features, channels,
                    agg_channels, attr_channels);
    if (residual && blocks[n_blocks - 1].out_features != channels)
        return fail(CSMPN_ERR_INVALID, "residual needs out_features == channels");
    if (mean_aggr && !in_degree) return fail(CSMPN_ERR_INVALID, "in_degree is null");
    memset(&io, 0, sizeof(io));
    io.rows = N; io.nseg = attr_channels > 0 ? 3 : 2;
    io.seg[0].a = h; io.seg[0].ch = channels; io.seg[0].off = 0;
    io.seg[1].a = agg; io.seg[1].ch = agg_channels; io.seg[1].off = channels;
    io.seg[1].deg = mean_aggr ? in_degree : nullptr;
    io.seg[2].a = node_attr; io.seg[2].ch = attr_channels; io.seg[2].off = channels + agg_channels;
    return CSMPN_OK;
}

int csmpn_egcl_node_forward(const float* metric, int n, const csmpn_block_params* blocks, int n_blocks, const float* h,
                            int32_t channels, const float* agg, int32_t agg_channels, const float* node_attr,
                            int32_t attr_channels, const int32_t* in_degree, int32_t mean_aggr, int32_t residual,
                            int64_t N, float* out, float* save_inputs, void* workspace, size_t workspace_bytes, uint32_t flags, void* stream) {
    const AlgId id = alg_id(metric, n);
    if (id == ALG_NONE) return fail(CSMPN_ERR_UNSUPPORTED, "metric not supported by the HIP path");
    RowIO io;
    int rc = node_io(blocks, n_blocks, h, channels, agg, agg_channels, node_attr, attr_channels, in_degree, mean_aggr,
                     residual, N, io);
    if (rc) return rc;
    Plan plan;
    if ((rc = make_plan(id, n, blocks, nullptr, n_blocks, workspace, workspace_bytes, false, 0, false, N, plan,
                        (flags & CSMPN_FLAG_DETERMINISTIC) != 0))) return rc;
    const bool need_pack = !(flags & CSMPN_FLAG_WEIGHTS_PACKED);   // packed fragments are a matter of the general kernels: run_rows
    io.y = out; io.resid = residual ? h : nullptr; io.save = save_inputs;
    io.row_store = (flags & CSMPN_FLAG_DETERMINISTIC) ? 1 : 0;   // node stage: no row table, only atomic-free kernels qualify
    io.save_state = (flags & CSMPN_FLAG_SAVE_STATE) ? 1 : 0;
    return run_rows(id, plan, MODE_NODE, false, io, (hipStream_t)stream, need_pack);
}

int csmpn_egcl_node_backward(const float* metric, int n, const csmpn_block_params* blocks,
                             const csmpn_block_grads* grads, int n_blocks, const float* h, int32_t channels,
                             const float* agg, int32_t agg_channels, const float* node_attr, int32_t attr_channels,
                             const int32_t* in_degree, int32_t mean_aggr, int32_t residual, int64_t N,
                             const float* g_out, float* gh, float* g_agg, float* g_node_attr, const float* saved_inputs, void* workspace,
                             size_t workspace_bytes, uint32_t flags, void* stream) {
    const AlgId id = alg_id(metric, n);
    if (id == ALG_NONE) return fail(CSMPN_ERR_UNSUPPORTED, "metric not supported by the HIP path");
    RowIO io;
    int rc = node_io(blocks, n_blocks, h, channels, agg, agg_channels, node_attr, attr_channels, in_degree, mean_aggr,
                     residual, N, io);
    if (rc) return rc;
    Plan plan;
    if ((rc = make_plan(id, n, blocks, grads, n_blocks, workspace, workspace_bytes, true, 0, saved_inputs != nullptr, N, plan,
                        (flags & CSMPN_FLAG_DETERMINISTIC) != 0))) return rc;
    // fragments packed by the forward are only valid for the forward's own layout choice (a forward
    // with LDS-staged raw weights packs nothing): the backward packs for itself; no-op for VAR_WAVE
    const bool need_pack = true;
    io.gy = g_out; io.gx[0] = gh; io.gx[1] = g_agg; io.gx[2] = g_node_attr;
    io.resid_bwd = residual ? 1 : 0; io.saved = saved_inputs;
    io.row_store = (flags & CSMPN_FLAG_DETERMINISTIC) ? 1 : 0;
    io.save_state = (flags & CSMPN_FLAG_SAVE_STATE) ? 1 : 0;
    g_tables_ready = (flags & CSMPN_FLAG_WEIGHTS_PACKED) != 0;
    rc = run_rows(id, plan, MODE_NODE, true, io, (hipStream_t)stream, need_pack);
    g_tables_ready = false;
    return rc;
}

}  // extern "C"
