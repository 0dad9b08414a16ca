// Callers either side of the message-passing path (SURVEY.md §8(f)-1,2), C-ABI in include/csmpn_hip.h:
//   csmpn_simplex_rows          input rows of the simplex feature embedding: gather the vertices of every
//                               simplex in a given vertex order and embed their features at a grade
//                               (hulls_cssmpnn.py:96-125, md17_cssmpnn.py:85-120)
//   csmpn_readout_mse_forward / _backward
//                               scalar readout + loss of the hulls model: MVLinear -> blade 0 -> mean over
//                               the simplices of a graph -> squared error (hulls_cssmpnn.py:93,155-164)
//   csmpn_readout_traj_forward / _backward
//                               vector readout + loss of the trajectory models: MVLinear -> vector blades (+ loc) -> distance
//                               to the target -> per-graph MSE / ADE / FDE and per-vertex MSE (md17_cssmpnn.py:165-176,
//                               motion_cssmpnn.py:150-168, nba_cssmpnn.py:176-191)
//   csmpn_type_attr_forward / _backward
//                               node / edge attributes of the task models from a learned (or one-hot) table of simplex-type
//                               features (md17_cssmpnn.py:122-133, hulls_cssmpnn.py:127-140): one launch each way instead
//                               of ~20 small PyTorch launches per step
// HBM-bound gathers / reductions: one pass over the data, coalesced rows, fixed-order sums (no atomics; the 9-element
// table gradient of csmpn_type_attr_backward is the exception: block sums in LDS, then one float atomic per element and block).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/csmpn_hip.h"
#include "capi_common.hpp"

namespace {

constexpr int kMaxVertexBlocks = 4;

struct RowsDesc {
    const float* data[kMaxVertexBlocks];
    int channels[kMaxVertexBlocks];   // K_b
    int gstart[kMaxVertexBlocks];     // first blade of the grade
    int gsize[kMaxVertexBlocks];      // blades of the grade
    int choff[kMaxVertexBlocks + 1];  // first output channel of the block
    int nblocks;
    int vpr;                          // vertices per row
    int D;
};

// one thread per output element: out[r][ch][d]
__global__ void simplex_rows_kernel(RowsDesc ds, const int64_t* verts, long n_rows, long n_feat_rows, float* out) {
    const int ctot = ds.choff[ds.nblocks];
    const long rowlen = (long)ctot * ds.D;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_rows * rowlen) return;
    const long r = t / rowlen;
    const int e = (int)(t - r * rowlen);
    const int ch = e / ds.D, d = e - ch * ds.D;
    float v = 0.f;
#pragma unroll
    for (int b = 0; b < kMaxVertexBlocks; ++b) {
        if (b < ds.nblocks && ch >= ds.choff[b] && ch < ds.choff[b + 1]) {
            const int lc = ch - ds.choff[b];
            const int vert = lc / ds.channels[b], k = lc - vert * ds.channels[b];
            const int td = d - ds.gstart[b];
            if (td >= 0 && td < ds.gsize[b]) {
                long src = verts[r * ds.vpr + vert];
                if (src < 0 || src >= n_feat_rows) src = 0;   // out of range: validated once per batch on the host (SimplicialBatch.plan)
                v = ds.data[b][(src * ds.channels[b] + k) * ds.gsize[b] + td];
            }
        }
    }
    out[t] = v;
}

// one workgroup per graph: pred = mean_s (sum_c w[c] x[s][c][0] + b), loss = (pred - target)^2,
// xsum[g][c] = sum_s x[s][c][0] (for the weight gradient). Fixed-order tree sums.
__global__ void __launch_bounds__(256) readout_fwd_kernel(const float* x, const float* w, int wstride, const float* bias, int C,
                                                          int D, const int* ptr, const float* target, float* pred,
                                                          float* loss, float* xsum) {
    __shared__ float red[256];
    const int g = blockIdx.x;
    const int lo = ptr[g], hi = ptr[g + 1];
    const int cnt = hi - lo;
    float total = 0.f;
    for (int c = 0; c < C; ++c) {
        float a = 0.f;
        for (int s = lo + threadIdx.x; s < hi; s += 256) a += x[((long)s * C + c) * D];
        red[threadIdx.x] = a;
        __syncthreads();
        for (int k = 128; k > 0; k >>= 1) {
            if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
            __syncthreads();
        }
        const float sc = red[0];
        __syncthreads();
        if (threadIdx.x == 0) xsum[(long)g * C + c] = sc;
        total += w[(long)c * wstride] * sc;
    }
    if (threadIdx.x == 0) {
        const float n = float(cnt > 1 ? cnt : 1);
        const float p = total / n + (cnt > 0 && bias ? bias[0] : 0.f);
        pred[g] = p;
        const float e = p - target[g];
        loss[g] = e * e;
    }
}

// gx[s][c][d] = (d == 0) ? coef[graph(s)] * w[c] : 0, coef_g = g_loss_g * 2 (pred_g - target_g) / max(cnt_g, 1)
__global__ void readout_bwd_kernel(const float* w, int wstride, int C, int D, const int* ptr, int B, const float* coef,
                                   long S, float* gx) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long per = (long)C * D;
    if (t >= S * per) return;
    const long s = t / per;
    const int e = (int)(t - s * per);
    const int c = e / D, d = e - c * D;
    float v = 0.f;
    if (d == 0) {
        int lo = 0, hi = B;   // graph of row s: last g with ptr[g] <= s
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (ptr[mid] <= s) lo = mid; else hi = mid;
        }
        v = coef[lo] * w[(long)c * wstride];
    }
    gx[t] = v;
}

// one thread per (row, k): row < n_nodes -> node_attr[row][k][:], else edge e = row - n_nodes -> edge_attr[e][k][:] and
// edge_attr[e][K + k][:]; blade 0 = the table entry, the other blades 0
__global__ void type_attr_fwd_kernel(const float* table, int T, int K, const int* types, long n_nodes, const int* src, const int* dst,
                                     long n_edges, int D, float* node_attr, float* edge_attr) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long row = t / K;
    const int k = (int)(t % K);
    if (row >= n_nodes + n_edges) return;
    auto put = [&](float* p, float v) {
        p[0] = v;
        for (int d = 1; d < D; ++d) p[d] = 0.f;
    };
    // type ids clamped into [0, T): a bad id reads a wrong table row, never out of bounds (callers validate node_types once
    // per batch - the reference's nn.Embedding would raise)
    auto ty = [&](long r) { return min(max(types[r], 0), T - 1); };
    if (row < n_nodes) {
        put(node_attr + (row * K + k) * D, table[ty(row) * K + k]);
    } else {
        const long e = row - n_nodes;
        put(edge_attr + (e * 2 * K + k) * D, table[ty(src[e]) * K + k]);
        put(edge_attr + (e * 2 * K + K + k) * D, table[ty(dst[e]) * K + k]);
    }
}
// g_table[t][k] += sum of the blade-0 gradients of every attribute row that read table[t][k]
constexpr int kTypeBins = 64;   // n_types * K
__global__ void __launch_bounds__(256) type_attr_bwd_kernel(int K, int TK, const int* types, long n_nodes, const int* src, const int* dst,
                                                            long n_edges, int D, const float* g_node, const float* g_edge, float* g_table) {
    __shared__ float bins[kTypeBins];
    if (threadIdx.x < kTypeBins) bins[threadIdx.x] = 0.f;
    __syncthreads();
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long row = t / K;
    const int k = (int)(t % K);
    const int T = TK / K;
    auto ty = [&](long r) { return min(max(types[r], 0), T - 1); };   // as the forward: clamped into the bins
    if (row < n_nodes) {
        if (g_node) atomicAdd(&bins[ty(row) * K + k], g_node[(row * K + k) * D]);
    } else if (row < n_nodes + n_edges && g_edge) {
        const long e = row - n_nodes;
        atomicAdd(&bins[ty(src[e]) * K + k], g_edge[(e * 2 * K + k) * D]);
        atomicAdd(&bins[ty(dst[e]) * K + k], g_edge[(e * 2 * K + K + k) * D]);
    }
    __syncthreads();
    if (threadIdx.x < TK && bins[threadIdx.x] != 0.f) atomicAdd(g_table + threadIdx.x, bins[threadIdx.x]);
}

// ---- trajectory readout + loss (md17 / motion / NBA heads: md17_cssmpnn.py:165-176, motion_cssmpnn.py:150-168,
// nba_cssmpnn.py:176-191): final MVLinear restricted to the vector blades, + loc, distance to the target, per-graph sums.
constexpr int kTrajMaxOC = 4096;   // out_channels * channels staged in LDS
constexpr int kTrajItems = 1024;   // (vertex, out channel) items of a graph processed per pass

// one workgroup per graph; item = (vertex v of the graph, out channel o): p[a] = sum_c W[o][c][grade 1] x[row(v)][c][1 + a]
// (+ loc[v][o][a]); d = p - target[trow(v)][o][:]; fixed-order sums. per_graph[b] = (sum |d|^2 / (cnt O), sum |d| / (cnt O),
// sum_v |d[v][O-1]| / cnt) over the cnt scored vertices (trow >= 0); per_vertex[v] = sum_{o,a} d^2 / (O n) (0 if not scored).
__global__ void __launch_bounds__(256) readout_traj_fwd_kernel(const float* x, int C, int D, int n, const int* vrows, const float* w,
                                                               int O, int wstride, const float* loc, const float* target,
                                                               const int* trow, const int* vptr, float* pred, float* per_graph,
                                                               float* per_vertex) {
    __shared__ float W1[kTrajMaxOC];
    __shared__ float sq_item[kTrajItems];
    __shared__ float red[3][256];
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < O * C; i += 256) W1[i] = w[(long)i * wstride + 1];   // grade-1 weight of (o, c)
    __syncthreads();
    const int lo = vptr[b], hi = vptr[b + 1];
    const int vper = kTrajItems / O > 0 ? kTrajItems / O : 1;   // vertices per pass (O <= kTrajItems is checked on the host)
    float s_sq = 0.f, s_ade = 0.f, s_fde = 0.f;
    int cnt = 0;
    for (int v0 = lo; v0 < hi; v0 += vper) {
        const int nv = min(vper, hi - v0);
        for (int it = tid; it < nv * O; it += 256) {
            const int v = v0 + it / O, o = it % O;
            const long row = vrows ? vrows[v] : v;
            const int tr = trow ? trow[v] : v;
            float p[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
            const float* xr = x + row * (long)C * D + 1;
            for (int c = 0; c < C; ++c) {
                const float wv = W1[o * C + c];
                for (int a = 0; a < n; ++a) p[a] = fmaf(wv, xr[(long)c * D + a], p[a]);
            }
            float sq = 0.f;
            for (int a = 0; a < n; ++a) {
                const long e = ((long)v * O + o) * n + a;
                if (loc) p[a] += loc[e];
                pred[e] = p[a];
                if (tr >= 0) {
                    const float d = p[a] - target[((long)tr * O + o) * n + a];
                    sq = fmaf(d, d, sq);
                }
            }
            sq_item[it] = sq;
            if (tr >= 0) {
                const float nr = sqrtf(sq);
                s_sq += sq; s_ade += nr;
                if (o == O - 1) s_fde += nr;
            }
        }
        __syncthreads();
        for (int j = tid; j < nv; j += 256) {     // per-vertex sum over the out channels, in order
            float a = 0.f;
            for (int o = 0; o < O; ++o) a += sq_item[j * O + o];
            const int tr = trow ? trow[v0 + j] : v0 + j;
            per_vertex[v0 + j] = tr >= 0 ? a / float(O * n) : 0.f;
            cnt += tr >= 0 ? 1 : 0;
        }
        __syncthreads();
    }
    red[0][tid] = s_sq; red[1][tid] = s_ade; red[2][tid] = s_fde;
    __shared__ int cred[256];
    cred[tid] = cnt;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (tid < k) {
            red[0][tid] += red[0][tid + k]; red[1][tid] += red[1][tid + k]; red[2][tid] += red[2][tid + k];
            cred[tid] += cred[tid + k];
        }
        __syncthreads();
    }
    if (tid == 0) {
        const float cn = float(cred[0] > 1 ? cred[0] : 1);
        per_graph[b * 3 + 0] = red[0][0] / (cn * O);
        per_graph[b * 3 + 1] = red[1][0] / (cn * O);
        per_graph[b * 3 + 2] = red[2][0] / cn;
    }
}

// gpred[v][o][a] = d[a] * (g_graph[b][0] 2 / (cnt O) + g_graph[b][1] / (|d| cnt O) + [o == O-1] g_graph[b][2] / (|d| cnt)
//                          + g_vertex[v] 2 / (O n)),  d = pred - target; 0 for vertices that are not scored. One thread per (v, o).
__global__ void readout_traj_gpred_kernel(int V, int O, int n, const float* pred, const float* target, const int* trow,
                                          const int* graph_of_vertex, const int* vptr, const float* g_graph, const float* g_vertex,
                                          float* gpred) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)V * O) return;
    const int v = (int)(t / O), o = (int)(t % O);
    const int tr = trow ? trow[v] : v;
    float d[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, sq = 0.f;
    if (tr >= 0)
        for (int a = 0; a < n; ++a) {
            d[a] = pred[t * n + a] - target[((long)tr * O + o) * n + a];
            sq = fmaf(d[a], d[a], sq);
        }
    float coef = 0.f;
    if (tr >= 0) {
        const int b = graph_of_vertex[v];
        int cnt = 0;                                    // scored vertices of the graph (graphs are small: a short loop)
        for (int u = vptr[b]; u < vptr[b + 1]; ++u) cnt += (trow ? trow[u] : u) >= 0 ? 1 : 0;
        const float cn = float(cnt > 1 ? cnt : 1);
        const float nr = sqrtf(sq), inv = nr > 0.f ? 1.f / nr : 0.f;
        if (g_graph) {
            coef += g_graph[b * 3 + 0] * 2.f / (cn * O) + g_graph[b * 3 + 1] * inv / (cn * O);
            if (o == O - 1) coef += g_graph[b * 3 + 2] * inv / cn;
        }
        if (g_vertex) coef += g_vertex[v] * 2.f / float(O * n);
    }
    for (int a = 0; a < n; ++a) gpred[t * n + a] = coef * d[a];
}

// blocks [0, nbx): gx[s][c][1 + a] = sum_o W[o][c][1] gpred[vertex(s)][o][a], every other element of gx = 0 (one thread
// per (s, c)); blocks [nbx, nbx + O): g_w[o][c][grade 1] += sum_{v, a} gpred[v][o][a] x[row(v)][c][1 + a] - block = out
// channel, thread = (c, vertex group), fixed-order sums (no atomics).
__global__ void __launch_bounds__(256) readout_traj_bwd_kernel(const float* x, int C, int D, int n, long S, const int* vrows,
                                                               const int* vertex_of_row, int V, const float* w, int O, int wstride,
                                                               const float* gpred, int nbx, float* gx, float* g_w) {
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < nbx) {
        const long t = (long)blockIdx.x * 256 + tid;
        if (t >= S * C) return;
        const long s = t / C;
        const int c = (int)(t % C);
        const int v = vertex_of_row ? vertex_of_row[s] : (int)s;
        float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        if (v >= 0)
            for (int o = 0; o < O; ++o) {
                const float wv = w[((long)o * C + c) * wstride + 1];
                for (int a = 0; a < n; ++a) acc[a] = fmaf(wv, gpred[((long)v * O + o) * n + a], acc[a]);
            }
        float* g = gx + t * D;
        for (int d = 0; d < D; ++d) g[d] = (d >= 1 && d <= n) ? acc[d - 1] : 0.f;
        return;
    }
    __shared__ float red[256];
    const int o = (int)blockIdx.x - nbx;
    const int groups = 256 / C;                 // C <= 64: at least 4 vertex groups
    const int c = tid % C, vg = tid / C;
    float acc = 0.f;
    if (vg < groups)
        for (int v = vg; v < V; v += groups) {
            const long row = vrows ? vrows[v] : v;
            const float* xr = x + (row * C + c) * (long)D + 1;
            const float* gp = gpred + ((long)v * O + o) * n;
            for (int a = 0; a < n; ++a) acc = fmaf(gp[a], xr[a], acc);
        }
    red[tid] = acc;
    __syncthreads();
    if (tid < C) {
        float sum = 0.f;
        for (int k = 0; k < groups; ++k) sum += red[k * C + tid];
        g_w[((long)o * C + tid) * wstride + 1] += sum;
    }
}

int grade_start(int n, int g) {
    int s = 0, c = 1;
    for (int i = 0; i < g; ++i) { s += c; c = c * (n - i) / (i + 1); }
    return s;
}
int grade_size(int n, int g) {
    int c = 1;
    for (int i = 0; i < g; ++i) c = c * (n - i) / (i + 1);
    return c;
}

}  // namespace

extern "C" {

int csmpn_simplex_rows(int n, const csmpn_vertex_block* blocks, int n_blocks, const int64_t* verts, int64_t n_rows,
                       int32_t verts_per_row, int64_t n_feature_rows, float* out, void* stream) {
    if (n < 1 || n > 5) return csmpn_fail(CSMPN_ERR_UNSUPPORTED, "n = %d generators not supported", n);
    if (n_blocks < 1 || n_blocks > kMaxVertexBlocks) return csmpn_fail(CSMPN_ERR_INVALID, "1..%d vertex blocks", kMaxVertexBlocks);
    if (n_rows <= 0) return CSMPN_OK;
    if (!blocks || !verts || !out || verts_per_row < 1 || n_feature_rows < 1) return csmpn_fail(CSMPN_ERR_INVALID, "bad arguments");
    RowsDesc ds{};
    ds.nblocks = n_blocks; ds.vpr = verts_per_row; ds.D = 1 << n;
    ds.choff[0] = 0;
    for (int b = 0; b < n_blocks; ++b) {
        if (!blocks[b].data || blocks[b].channels < 1 || blocks[b].grade < 0 || blocks[b].grade > n)
            return csmpn_fail(CSMPN_ERR_INVALID, "vertex block %d: bad data / channels / grade", b);
        ds.data[b] = blocks[b].data;
        ds.channels[b] = blocks[b].channels;
        ds.gstart[b] = grade_start(n, blocks[b].grade);
        ds.gsize[b] = grade_size(n, blocks[b].grade);
        ds.choff[b + 1] = ds.choff[b] + verts_per_row * blocks[b].channels;
    }
    const long total = (long)n_rows * ds.choff[n_blocks] * ds.D;
    const unsigned block = 256;
    if ((total + block - 1) / block >= (1ll << 31)) return csmpn_fail(CSMPN_ERR_INVALID, "too many rows");
    hipLaunchKernelGGL(simplex_rows_kernel, dim3((unsigned)((total + block - 1) / block)), dim3(block), 0, (hipStream_t)stream,
                       ds, verts, (long)n_rows, (long)n_feature_rows, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return csmpn_fail(CSMPN_ERR_HIP, "simplex rows: %s", hipGetErrorString(e));
    return CSMPN_OK;
}

int csmpn_readout_mse_forward(int n, const float* x, const float* weight, int32_t weight_stride, const float* bias,
                              int64_t n_rows, int32_t channels, const int32_t* graph_ptr, int64_t n_graphs,
                              const float* target, float* pred, float* loss, float* channel_sums, void* stream) {
    if (n < 1 || n > 5) return csmpn_fail(CSMPN_ERR_UNSUPPORTED, "n = %d generators not supported", n);
    if (n_graphs <= 0) return CSMPN_OK;
    if (!x || !weight || !graph_ptr || !target || !pred || !loss || !channel_sums || channels < 1 || weight_stride < 1 || n_rows < 0)
        return csmpn_fail(CSMPN_ERR_INVALID, "bad arguments");
    hipLaunchKernelGGL(readout_fwd_kernel, dim3((unsigned)n_graphs), dim3(256), 0, (hipStream_t)stream, x, weight,
                       (int)weight_stride, bias, (int)channels, 1 << n, (const int*)graph_ptr, target, pred, loss, channel_sums);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return csmpn_fail(CSMPN_ERR_HIP, "readout forward: %s", hipGetErrorString(e));
    return CSMPN_OK;
}

int csmpn_readout_mse_backward(int n, const float* weight, int32_t weight_stride, int64_t n_rows, int32_t channels,
                               const int32_t* graph_ptr, int64_t n_graphs, const float* coef, float* gx, void* stream) {
    if (n < 1 || n > 5) return csmpn_fail(CSMPN_ERR_UNSUPPORTED, "n = %d generators not supported", n);
    if (n_rows <= 0) return CSMPN_OK;
    if (!weight || !graph_ptr || !coef || !gx || channels < 1 || n_graphs < 1) return csmpn_fail(CSMPN_ERR_INVALID, "bad arguments");
    const long total = (long)n_rows * channels * (1 << n);
    const unsigned block = 256;
    hipLaunchKernelGGL(readout_bwd_kernel, dim3((unsigned)((total + block - 1) / block)), dim3(block), 0, (hipStream_t)stream,
                       weight, (int)weight_stride, (int)channels, 1 << n, (const int*)graph_ptr, (int)n_graphs, coef,
                       (long)n_rows, gx);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return csmpn_fail(CSMPN_ERR_HIP, "readout backward: %s", hipGetErrorString(e));
    return CSMPN_OK;
}

int csmpn_readout_traj_forward(int n, const float* x, int32_t channels, const int32_t* vertex_rows, int64_t n_vertices,
                               const float* weight, int32_t out_channels, int32_t weight_stride, const float* loc, const float* target,
                               const int32_t* target_row, const int32_t* vertex_ptr, int64_t n_graphs, float* pred, float* per_graph,
                               float* per_vertex, void* stream) {
    if (n < 1 || n > 5) return csmpn_fail(CSMPN_ERR_UNSUPPORTED, "n = %d generators not supported", n);
    if (n_graphs <= 0 || n_vertices <= 0) return CSMPN_OK;
    if (!x || !weight || !target || !vertex_ptr || !pred || !per_graph || !per_vertex) return csmpn_fail(CSMPN_ERR_INVALID, "trajectory readout: null pointer");
    if (channels < 1 || channels > 64 || out_channels < 1 || out_channels > kTrajItems || (long)channels * out_channels > kTrajMaxOC || weight_stride < 2)
        return csmpn_fail(CSMPN_ERR_UNSUPPORTED, "trajectory readout: channels <= 64, out_channels * channels <= %d, weight [O, C, G >= 2]", kTrajMaxOC);
    hipLaunchKernelGGL(readout_traj_fwd_kernel, dim3((unsigned)n_graphs), dim3(256), 0, (hipStream_t)stream, x, (int)channels, 1 << n, n,
                       (const int*)vertex_rows, weight, (int)out_channels, (int)weight_stride, loc, target, (const int*)target_row,
                       (const int*)vertex_ptr, pred, per_graph, per_vertex);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return csmpn_fail(CSMPN_ERR_HIP, "trajectory readout forward: %s", hipGetErrorString(e));
    return CSMPN_OK;
}

int csmpn_readout_traj_backward(int n, const float* x, int32_t channels, int64_t n_rows, const int32_t* vertex_rows,
                                const int32_t* vertex_of_row, int64_t n_vertices, const float* weight, int32_t out_channels,
                                int32_t weight_stride, const float* pred, const float* target, const int32_t* target_row,
                                const int32_t* graph_of_vertex, const int32_t* vertex_ptr, const float* g_per_graph,
                                const float* g_per_vertex, float* gpred_scratch, float* gx, float* g_weight, void* stream) {
    if (n < 1 || n > 5) return csmpn_fail(CSMPN_ERR_UNSUPPORTED, "n = %d generators not supported", n);
    if (n_rows <= 0) return CSMPN_OK;
    if (!x || !weight || !pred || !target || !graph_of_vertex || !vertex_ptr || !gpred_scratch || !gx || !g_weight || n_vertices < 0)
        return csmpn_fail(CSMPN_ERR_INVALID, "trajectory readout: null pointer");
    if ((vertex_rows == nullptr) != (vertex_of_row == nullptr)) return csmpn_fail(CSMPN_ERR_INVALID, "trajectory readout: vertex_rows and vertex_of_row go together");
    if (!vertex_rows && n_rows != n_vertices) return csmpn_fail(CSMPN_ERR_INVALID, "trajectory readout: identity vertex rows need n_rows == n_vertices");
    if (channels < 1 || channels > 64 || out_channels < 1 || out_channels > kTrajItems || (long)channels * out_channels > kTrajMaxOC || weight_stride < 2)
        return csmpn_fail(CSMPN_ERR_UNSUPPORTED, "trajectory readout: channels <= 64, out_channels * channels <= %d, weight [O, C, G >= 2]", kTrajMaxOC);
    hipStream_t st = (hipStream_t)stream;
    const long items = (long)n_vertices * out_channels;
    if (items > 0)
        hipLaunchKernelGGL(readout_traj_gpred_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, (int)n_vertices,
                           (int)out_channels, n, pred, target, (const int*)target_row, (const int*)graph_of_vertex,
                           (const int*)vertex_ptr, g_per_graph, g_per_vertex, gpred_scratch);
    const long nbx = ((long)n_rows * channels + 255) / 256;
    if (nbx + out_channels >= (1ll << 31)) return csmpn_fail(CSMPN_ERR_INVALID, "trajectory readout: too many rows");
    hipLaunchKernelGGL(readout_traj_bwd_kernel, dim3((unsigned)(nbx + out_channels)), dim3(256), 0, st, x, (int)channels, 1 << n, n,
                       (long)n_rows, (const int*)vertex_rows, (const int*)vertex_of_row, (int)n_vertices, weight, (int)out_channels,
                       (int)weight_stride, (const float*)gpred_scratch, (int)nbx, gx, g_weight);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return csmpn_fail(CSMPN_ERR_HIP, "trajectory readout backward: %s", hipGetErrorString(e));
    return CSMPN_OK;
}

int csmpn_type_attr_forward(int n, const float* table, int32_t n_types, int32_t k, const int32_t* types, int64_t n_nodes,
                            const int32_t* src, const int32_t* dst, int64_t n_edges, float* node_attr, float* edge_attr, void* stream) {
    if (n < 1 || n > 5) return csmpn_fail(CSMPN_ERR_UNSUPPORTED, "n = %d generators not supported", n);
    if (n_nodes < 0 || n_edges < 0 || k < 1 || n_types < 1 || n_types * k > kTypeBins) return csmpn_fail(CSMPN_ERR_INVALID, "type attributes: bad sizes");
    if (n_nodes + n_edges == 0) return CSMPN_OK;
    if (!table || !types || (n_nodes && !node_attr) || (n_edges && (!src || !dst || !edge_attr)))
        return csmpn_fail(CSMPN_ERR_INVALID, "type attributes: null pointer");
    const long total = (long)(n_nodes + n_edges) * k;
    hipLaunchKernelGGL(type_attr_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, table, (int)n_types, (int)k,
                       (const int*)types, (long)n_nodes, (const int*)src, (const int*)dst, (long)n_edges, 1 << n, node_attr, edge_attr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return csmpn_fail(CSMPN_ERR_HIP, "type attributes forward: %s", hipGetErrorString(e));
    return CSMPN_OK;
}

int csmpn_type_attr_backward(int n, int32_t n_types, int32_t k, const int32_t* types, int64_t n_nodes, const int32_t* src,
                             const int32_t* dst, int64_t n_edges, const float* g_node_attr, const float* g_edge_attr, float* g_table,
                             void* stream) {
    if (n < 1 || n > 5) return csmpn_fail(CSMPN_ERR_UNSUPPORTED, "n = %d generators not supported", n);
    if (n_nodes < 0 || n_edges < 0 || k < 1 || n_types < 1 || n_types * k > kTypeBins) return csmpn_fail(CSMPN_ERR_INVALID, "type attributes: bad sizes");
    if (n_nodes + n_edges == 0 || (!g_node_attr && !g_edge_attr)) return CSMPN_OK;
    if (!types || !g_table || (n_edges && g_edge_attr && (!src || !dst))) return csmpn_fail(CSMPN_ERR_INVALID, "type attributes: null pointer");
    const long total = (long)(n_nodes + n_edges) * k;
    hipLaunchKernelGGL(type_attr_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (int)k,
                       (int)(n_types * k), (const int*)types, (long)n_nodes, (const int*)src, (const int*)dst, (long)n_edges, 1 << n,
                       g_node_attr, g_edge_attr, g_table);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return csmpn_fail(CSMPN_ERR_HIP, "type attributes backward: %s", hipGetErrorString(e));
    return CSMPN_OK;
}

}  // extern "C"
