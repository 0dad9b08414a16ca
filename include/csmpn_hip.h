/* csmpn_hip.h — C-ABI of the MI355X (gfx950) implementation of the Clifford
 * geometric-product / CEMLP / shared simplicial message-passing hot path.
 *
 * The reference exposes this path as a PyTorch nn.Module surface, not an FFI;
 * each entry point below cites the reference interface it replaces
 * (paths relative to the reference repository root):
 *
 *   csmpn_algebra_tables      CliffordAlgebra.__init__ / construct_gmt /
 *                             geometric_product_paths
 *                             (csmpn/algebra/cliffordalgebra.py:11-42,238-252,
 *                              csmpn/algebra/metric.py:18-120)
 *   csmpn_geometric_product_* CliffordAlgebra.geometric_product
 *                             (csmpn/algebra/cliffordalgebra.py:44-54), full-blade form
 *   csmpn_cemlp_*             CEMLP.forward and autograd through it: MVLinear,
 *                             MVSiLU, SteerableGeometricProductLayer (+ Normalization-
 *                             Layer), MVLayerNorm (csmpn/models/cegnn_utils.py:34-213,287-338)
 *   csmpn_egcl_*              EGCL.forward = PyG propagate: gather h_i/h_j, message
 *                             (edge CEMLP), scatter sum|mean over edge_index[1],
 *                             update (node CEMLP + residual)
 *                             (csmpn/models/cegnn_utils.py:216-284)
 *   csmpn_csr_build           the sort PyG/torch_scatter do not need but the
 *                             segmented scatter here does; one-time per complex
 *
 * Conventions: extern "C", plain pointers and sizes, int status return
 * (0 = ok, non-zero = error, message via csmpn_last_error()). All data
 * pointers are DEVICE pointers to contiguous fp32 (indices int32) unless a
 * parameter says "host". Every launch is asynchronous on the caller-supplied
 * hipStream_t (passed as void*). No entry point allocates device memory: the
 * caller provides the workspace (csmpn_*_workspace_bytes). Parameter tensors
 * are read in the reference's own state_dict layouts;
 * gradient tensors are ACCUMULATED (+=) in the same layouts and must be
 * zero-initialised by the caller when a fresh gradient is wanted.
 */
#ifndef CSMPN_HIP_H
#define CSMPN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CSMPN_MAX_BLOCKS 4
#define CSMPN_OK 0
#define CSMPN_ERR_UNSUPPORTED 1
#define CSMPN_ERR_INVALID 2
#define CSMPN_ERR_HIP 3

/* flags of the compute entry points */
#define CSMPN_FLAG_NO_VALIDATE 2u    /* csmpn_csr_build, csmpn_embed_cemlp_*: skip the (synchronous) range check of the index table */
#define CSMPN_FLAG_WEIGHTS_PACKED 1u /* forward entry points: the workspace already holds this
                                       CEMLP's packed weights (left there by an earlier forward with
                                       the same parameters): skip the pack kernel. csmpn_egcl_edge_backward /
                                       csmpn_egcl_node_backward: the workspace is the one the stage's FORWARD
                                       ran on, with the same parameters and untouched since - the 16-row-tile
                                       kernel families (Cl(5,0) / Cl(4,1) at 24-32 channels, Cl(3,0) at 32) then
                                       reuse the weight-fragment tables that forward packed; every other
                                       backward packs for its own tile layout and ignores the flag. */

#define CSMPN_FLAG_DETERMINISTIC 4u  /* csmpn_egcl_edge_forward/backward: no float atomics. The edge rows are
                                       not scattered: `agg` (forward) / `gh` (backward) is an [E, C, D] table in
                                       SORTED edge order, to be summed by csmpn_segment_reduce in a fixed order
                                       (the reference runs under torch.use_deterministic_algorithms(True),
                                       engineer/utils/seed.py:30). Honoured by the kernels whose parameter-gradient
                                       sums are atomic-free: Cl(3,0) with 8 or 16 channels, Cl(5,0) / Cl(4,1) with
                                       8 / 16 / 24 / 28 / 32 channels (two blocks, saved block inputs) and, since round 3,
                                       every Cl(2,0) / Cl(3,0) shape on the general kernels (one row tile per workgroup,
                                       per-workgroup copies of the gradient tensors + fixed-order sums; slower than the
                                       default). Every other shape returns CSMPN_ERR_UNSUPPORTED. Also accepted by
                                       csmpn_egcl_node_forward/backward and csmpn_cemlp_forward/backward, where it only
                                       selects the atomic-free parameter sums. */
#define CSMPN_FLAG_SAVE_STATE 8u     /* csmpn_egcl_{edge,node}_{forward,backward}, csmpn_embed_cemlp_{forward,backward} (two-block
                                      * modules), round 4; csmpn_cemlp_{forward,backward}, round 5, for the standalone Cl(3,0) CEMLPs
                                      * of 32 channels the 16-row-tile family serves (1 or 2 blocks; 32, 60 or 90 input channels:
                                      * csmpn_cemlp_saved_floats(..., CSMPN_FLAG_SAVE_STATE) > the size without the flag tells)
                                      * and ignored by them for every other shape: the forward ALSO stores, per block,
                                      * what the backward would otherwise recompute, in "state regions" behind the saved block
                                      * inputs and the hand-over region (csmpn_cemlp_saved_floats sizes them), and the backward
                                      * called with the same flag reads them:
                                      *   Cl(3,0), 8 channels: s = the block's output in front of its layer norm - no linear_left
                                      *     mix, no geometric product in the recompute (S1: edge forward +2.5 us, edge backward
                                      *     -6.5 us; y and R as well: measured a wash there);
                                      *   Cl(3,0), 32 channels and Cl(5,0) / Cl(4,1), 8 .. 32 channels: y (MVLinear output), R
                                      *     (linear_right output) and s - no channel mix and no geometric product at all in the
                                      *     recompute; these kernels run at 5-10 % of the HBM roofline, the extra rows travel
                                      *     under the arithmetic.
                                      * The state regions are private to the kernels (whole row tiles in the kernels' lane order).
                                      * Ignored by every other shape. Only for a saved buffer laid out for exactly the rows of the
                                      * call (csmpn_cemlp_saved_floats(.., rows) floats; not a slice of a larger one): the
                                      * regions are addressed by the call's row count. */

/* One CEMLP block = Sequential(MVLinear, MVSiLU, SteerableGeometricProductLayer,
 * MVLayerNorm) (cegnn_utils.py:177-207). Pointers in reference layouts. */
typedef struct csmpn_block_params {
    int32_t in_features;
    int32_t out_features;
    int32_t lin_subspaces;   /* 1: lin_w is [O,I,G]; 0: [O,I] (MVLinear subspaces=False) */
    int32_t reserved;
    const float* lin_w;      /* layers.k.0.weight */
    const float* lin_b;      /* layers.k.0.bias [1,O,1] or NULL */
    const float* silu_a;     /* layers.k.1.a [1,O,G] */
    const float* silu_b;     /* layers.k.1.b [1,O,G] */
    const float* gp_w;       /* layers.k.2.weight [O,P] */
    const float* norm_a;     /* layers.k.2.normalization.a [O,G] */
    const float* right_w;    /* layers.k.2.linear_right.weight [O,O,G] */
    const float* left_w;     /* layers.k.2.linear_left.weight [O,O,G] */
    const float* left_b;     /* layers.k.2.linear_left.bias [1,O,1] */
    const float* ln_a;       /* layers.k.3.a [1,O] */
} csmpn_block_params;

/* Gradient accumulators, same shapes as csmpn_block_params' tensors. */
typedef struct csmpn_block_grads {
    float* lin_w;
    float* lin_b;            /* NULL iff lin_b is NULL */
    float* silu_a;
    float* silu_b;
    float* gp_w;
    float* norm_a;
    float* right_w;
    float* left_w;
    float* left_b;
    float* ln_a;
} csmpn_block_grads;

/* Algebra: metric is a HOST array of n floats, every entry +1 or -1,
 * 2 <= n <= 5 for the device entry points (csmpn_metric_supported). */
int csmpn_metric_supported(const float* metric_host, int n);

/* Host-side table construction (any diagonal metric, n <= 8). All outputs are
 * HOST arrays: cayley [D*D*D] (left,out,right), index_to_bitmap [D],
 * bitmap_to_index [D], grades [D] (int64), subspaces [n+1] (int64),
 * paths [(n+1)^3] (uint8, (grade_left, grade_out, grade_right)). Any output
 * pointer may be NULL. */
int csmpn_algebra_tables(const float* metric_host, int n, float* cayley, int64_t* index_to_bitmap,
                         int64_t* bitmap_to_index, int64_t* grades, int64_t* subspaces, uint8_t* paths);

/* out[r, :] = a[r, :] * b[r, :] (geometric product), rows x D. Backward:
 * ga += d/da, gb += d/db given gout. */
int csmpn_geometric_product_forward(const float* metric_host, int n, const float* a, const float* b,
                                    float* out, int64_t rows, void* stream);
int csmpn_geometric_product_backward(const float* metric_host, int n, const float* a, const float* b,
                                     const float* gout, float* ga, float* gb, int64_t rows, void* stream);

/* Workspace (bytes) for a CEMLP of these blocks; covers packed weights and
 * per-launch scratch for any entry point below that takes this CEMLP. */
size_t csmpn_cemlp_workspace_bytes(int n, const csmpn_block_params* blocks, int n_blocks);

/* Saved block inputs (optional, every forward/backward pair below): the forward writes the
 * inputs of blocks 1..n_blocks-1 ([rows, O_{k-1}, D] each, back to back; rows = rows / n_edges /
 * n_nodes of the call, in the kernel's own row order) into save_inputs when it is non-NULL, and
 * the backward reads them from saved_inputs instead of recomputing the earlier blocks (one
 * block forward less per row, and a smaller LDS footprint). NULL on either side = recompute.
 * Floats per row: csmpn_cemlp_saved_floats_per_row(); for two-block Cl(5,0) CEMLPs of 9 .. 32 channels, for the
 * two-block 8-channel Cl(3,0) EGCL shapes (edge model 14 -> 8 -> 8, node model 19 -> 8 -> 8) and for multi-block CEMLPs of
 * the small algebras (n <= 3) outside the lane-kernel shapes this includes a second region of the same size behind the
 * saved inputs that the backward uses as scratch (a block's phase / launch hands d/d(its input) to the previous block's
 * there): the buffer is written by the backward although the pointer is const. */
size_t csmpn_cemlp_saved_floats_per_row(int n, const csmpn_block_params* blocks, int n_blocks);
/* Floats of the saved buffer of ONE launch over `rows` rows (round 4): rows * csmpn_cemlp_saved_floats_per_row() minus the
 * regions a launch of that size never touches - the general kernels' hand-over region exists only for launches that can
 * take the block-by-block backward (rows >= its threshold, default 4096; CSMPN_PHASED_MIN_ROWS): a 940-row node stage of a
 * multi-block CEMLP of the small algebras pays half. A buffer of this size is valid for forward and backward of that
 * launch. flags (ABI version 2): with CSMPN_FLAG_SAVE_STATE the state regions of that flag are included - pass the flag
 * here iff the forward / backward pair will be called with it (they are 3-4x the block inputs on the D = 32 and 32-channel
 * shapes: H28 248 instead of 56 channels x 32 floats per edge); without it the buffer holds block inputs + hand-over only.
 * The regions are counted with the rows rounded up to a multiple of 16 (whole row tiles): for those shapes the result may
 * exceed rows * per-row figure by up to 15 rows of state. */
size_t csmpn_cemlp_saved_floats(int n, const csmpn_block_params* blocks, int n_blocks, int64_t rows, uint32_t flags);

/* y[rows, O_last, D] = CEMLP(x[rows, I_0, D]). */
int csmpn_cemlp_forward(const float* metric_host, int n, const csmpn_block_params* blocks, int n_blocks,
                        const float* x, int64_t rows, float* y, float* save_inputs, void* workspace, size_t workspace_bytes,
                        uint32_t flags, void* stream);

/* gx[rows, I_0, D] = d<gy,y>/dx (overwritten; may be NULL), grads += d/dparams.
 * Recomputes the forward in-kernel from x. */
int csmpn_cemlp_backward(const float* metric_host, int n, const csmpn_block_params* blocks,
                         const csmpn_block_grads* grads, int n_blocks, const float* x, const float* gy,
                         int64_t rows, float* gx, const float* saved_inputs, void* workspace, size_t workspace_bytes, uint32_t flags, void* stream);

/* Standalone MVLinear (csmpn/models/cegnn_utils.py:287-338; the callers outside a CEMLP: the
 * projection heads hulls_cssmpnn.py:71-73 and the first stage of the simplex feature embeddings):
 *   y[b,o,d] = sum_i W[o,i,grade(d)] x[b,i,d] (+ bias[o] on blade 0)
 * weight is [O,I,G] (subspaces != 0) or [O,I]; bias [O] or NULL. The grade table depends on n only
 * (blade order metric.py:18-29), so every metric with n <= 5 generators is served. Backward: gx
 * [rows,I,D] overwritten (may be NULL), g_weight += d/dW (may be NULL), g_bias += d/dbias (may be NULL). */
int csmpn_mvlinear_forward(int n, const float* x, const float* weight, const float* bias, int64_t rows,
                           int32_t in_features, int32_t out_features, int32_t subspaces, float* y, void* stream);
int csmpn_mvlinear_backward(int n, const float* x, const float* weight, const float* gy, int64_t rows,
                            int32_t in_features, int32_t out_features, int32_t subspaces, float* gx, float* g_weight,
                            float* g_bias, void* stream);

/* The four small layers on their own (inside a CEMLP they are fused into the row programs; no
 * reference model calls them alone - these entry points give the nn.Modules a device forward /
 * backward). Rows are [rows, channels, D] float32, D = 2^n; metric entries +-1, n = 2..5 (the
 * signatures csmpn_metric_supported accepts); channels <= 256. Backward: gx overwritten, parameter
 * gradients ACCUMULATED (+=, float atomics: one per parameter and workgroup).
 *   MVSiLU, invariant "mag2" (cegnn_utils.py:53-83): a, b [channels, n+1]
 *     y_d = sigmoid(a[c,g] u_g + b[c,g]) x_d,  g = grade(d), u_0 = x_0, u_g = q_g(x) for g > 0
 *   NormalizationLayer (cegnn_utils.py:34-51): a [channels, n+1]
 *     y_d = x_d / (sigmoid(a[c,g]) (|x|_g - 1) + 1 + 1e-6),  |x|_g = (q_g(x)^2 + 1e-16)^(1/4)
 *   MVLayerNorm (cegnn_utils.py:86-96): a [channels]
 *     y[c] = a[c] x[c] / (mean_c |x[c]| + 1e-6),  |x| over all blades
 *   weighted geometric product of SteerableGeometricProductLayer (cegnn_utils.py:126-152):
 *     weight [channels, P] (P = number of grade paths, cliffordalgebra.py:238-252)
 *     y_j = sum_{(i,k)->j} sign(i,k) weight[c, path(grade i, grade j, grade k)] z_i r_k */
int csmpn_mvsilu_forward(const float* metric_host, int n, const float* x, const float* a, const float* b, int64_t rows,
                         int32_t channels, float* y, void* stream);
int csmpn_mvsilu_backward(const float* metric_host, int n, const float* x, const float* a, const float* b, const float* gy,
                          int64_t rows, int32_t channels, float* gx, float* g_a, float* g_b, void* stream);
int csmpn_mvnorm_forward(const float* metric_host, int n, const float* x, const float* a, int64_t rows, int32_t channels,
                         float* y, void* stream);
int csmpn_mvnorm_backward(const float* metric_host, int n, const float* x, const float* a, const float* gy, int64_t rows,
                          int32_t channels, float* gx, float* g_a, void* stream);
int csmpn_mvlayernorm_forward(const float* metric_host, int n, const float* x, const float* a, int64_t rows,
                              int32_t channels, float* y, void* stream);
int csmpn_mvlayernorm_backward(const float* metric_host, int n, const float* x, const float* a, const float* gy,
                               int64_t rows, int32_t channels, float* gx, float* g_a, void* stream);
int csmpn_wgp_forward(const float* metric_host, int n, const float* z, const float* r, const float* weight, int64_t rows,
                      int32_t channels, float* y, void* stream);
int csmpn_wgp_backward(const float* metric_host, int n, const float* z, const float* r, const float* weight,
                       const float* gy, int64_t rows, int32_t channels, float* gz, float* gr, float* g_weight, void* stream);

/* One-time per complex: sort the E directed adjacencies by target (stable radix sort).
 * edge_index is the reference's [2,E] int64 (row 0 = source j, row 1 = target i).
 * Outputs (device): perm[E] (sorted position -> original edge id), src_sorted[E],
 * dst_sorted[E] (int32), in_degree[N] (int32), row_ptr[N+1] (int32).
 * workspace: csmpn_csr_workspace_bytes(E, N) bytes of device memory. Inside one target's
 * segment the edges keep ascending original id, so the result (and the summation order
 * downstream) is deterministic. Every index is range-checked against [0, N): an entry outside
 * returns CSMPN_ERR_INVALID (PyG's scatter asserts in that case); the check costs one host
 * synchronisation per call - complexes are static and the result is cached by the caller -
 * and is skipped with CSMPN_FLAG_NO_VALIDATE (out-of-range entries are then mapped to node 0). */
size_t csmpn_csr_workspace_bytes(int64_t n_edges, int64_t n_nodes);
int csmpn_csr_build(const int64_t* edge_index, int64_t n_edges, int64_t n_nodes, int32_t* perm,
                    int32_t* src_sorted, int32_t* dst_sorted, int32_t* in_degree, int32_t* row_ptr,
                    void* workspace, size_t workspace_bytes, uint32_t flags, void* stream);

/* Deterministic mode, one-time per complex: the sorted edge positions ordered by SOURCE (stable, so
 * ascending sorted position inside one source's segment). order[E], row_ptr_src[N+1] (device, int32);
 * workspace as for csmpn_csr_build. */
int csmpn_csr_source_order(const int32_t* src_sorted, int64_t n_edges, int64_t n_nodes, int32_t* order,
                           int32_t* row_ptr_src, void* workspace, size_t workspace_bytes, void* stream);

/* Fixed-order segmented sum of a row table (the replacement of PyG's scatter under
 * deterministic algorithms):
 *   out[v] (+)= sum_{k in [add_ptr[v], add_ptr[v+1])} rows[add_order ? add_order[k] : k]
 *             - sum_{k in [sub_ptr[v], sub_ptr[v+1])} rows[sub_order ? sub_order[k] : k]
 * rows [*, row_floats], out [N, row_floats] (16-byte aligned, row_floats % 4 == 0); either
 * (ptr, order) pair may be NULL; accumulate != 0 adds to out instead of overwriting it. */
int csmpn_segment_reduce(const float* rows, int64_t row_floats, int64_t n_nodes, const int32_t* add_ptr,
                         const int32_t* add_order, const int32_t* sub_ptr, const int32_t* sub_order, float* out,
                         int32_t accumulate, void* stream);

/* EGCL message + aggregate (cegnn_utils.py:254-262 + PyG scatter):
 *   agg[v] += sum_{e: dst_e = v} EdgeCEMLP(cat_c[h[dst_e] - h[src_e], edge_attr[perm_e]])
 * agg [N, O, D] must be zeroed by the caller. edge_attr [E, A, D] in ORIGINAL edge
 * order (or NULL when A = 0). The mean's 1/max(deg,1) is applied by the node
 * entry points. */
int csmpn_egcl_edge_forward(const float* metric_host, int n, const csmpn_block_params* blocks, int n_blocks,
                            const float* h, int32_t channels, const float* edge_attr, int32_t attr_channels,
                            const int32_t* perm, const int32_t* src_sorted, const int32_t* dst_sorted,
                            int64_t n_edges, int64_t n_nodes, float* agg, float* save_inputs, void* workspace,
                            size_t workspace_bytes, uint32_t flags, void* stream);

/* Backward of the above. g_agg [N,O,D] is d/d(agg) (already divided by the degree
 * for aggr=mean). gh [N,C,D] += (scatter of +g to dst, -g to src);
 * g_edge_attr [E,A,D] (original order, overwritten) may be NULL. */
int csmpn_egcl_edge_backward(const float* metric_host, int n, const csmpn_block_params* blocks,
                             const csmpn_block_grads* grads, int n_blocks, const float* h, int32_t channels,
                             const float* edge_attr, int32_t attr_channels, const int32_t* perm,
                             const int32_t* src_sorted, const int32_t* dst_sorted, int64_t n_edges,
                             int64_t n_nodes, const float* g_agg, float* gh, float* g_edge_attr,
                             const float* saved_inputs, void* workspace, size_t workspace_bytes, uint32_t flags, void* stream);

/* EGCL update (cegnn_utils.py:264-275):
 *   out[v] = (residual ? h[v] : 0) + NodeCEMLP(cat_c[h[v], agg[v] * s_v, node_attr[v]])
 * s_v = 1/max(in_degree[v],1) if mean_aggr else 1. node_attr [N,T,D] or NULL. */
int csmpn_egcl_node_forward(const float* metric_host, int n, const csmpn_block_params* blocks, int n_blocks,
                            const float* h, int32_t channels, const float* agg, int32_t agg_channels,
                            const float* node_attr, int32_t attr_channels, const int32_t* in_degree,
                            int32_t mean_aggr, int32_t residual, int64_t n_nodes, float* out,
                            float* save_inputs, void* workspace, size_t workspace_bytes, uint32_t flags, void* stream);

/* Backward of the above: gh [N,C,D] (overwritten) = d/dh incl. residual,
 * g_agg [N,O,D] (overwritten) = d/d(agg) incl. the mean scale,
 * g_node_attr [N,T,D] (overwritten) may be NULL. */
int csmpn_egcl_node_backward(const float* metric_host, int n, const csmpn_block_params* blocks,
                             const csmpn_block_grads* grads, int n_blocks, const float* h, int32_t channels,
                             const float* agg, int32_t agg_channels, const float* node_attr,
                             int32_t attr_channels, const int32_t* in_degree, int32_t mean_aggr,
                             int32_t residual, int64_t n_nodes, const float* g_out, float* gh, float* g_agg,
                             float* g_node_attr, const float* saved_inputs, void* workspace, size_t workspace_bytes, uint32_t flags, void* stream);

/* Input rows of the simplex feature embedding (hulls_cssmpnn.py:96-125, md17_cssmpnn.py:85-120):
 * row r lists verts_per_row vertices (rows of the per-simplex feature tensors) in ONE vertex order;
 * block b contributes verts_per_row * channels_b channels (vertex by vertex), embedded at its grade:
 *   out[r][off_b + v * K_b + k][gstart(grade_b) + t] = blocks[b].data[verts[r][v]][k][t],  0 elsewhere.
 * out [n_rows, sum_b verts_per_row * K_b, 2^n]; blocks[b].data [n_feature_rows, K_b, C(n, grade_b)]. */
typedef struct csmpn_vertex_block {
    const float* data;
    int32_t channels;
    int32_t grade;
} csmpn_vertex_block;
int csmpn_simplex_rows(int n, const csmpn_vertex_block* blocks, int n_blocks, const int64_t* verts, int64_t n_rows,
                       int32_t verts_per_row, int64_t n_feature_rows, float* out, void* stream);

/* Fused simplex feature embedding (round 3; hulls_cssmpnn.py:96-125: vertex features of every d-simplex in all (d+1)!
 * vertex orders -> CEMLP -> sum over the orders), for the shapes the wide parity-lane kernels serve as standalone CEMLPs
 * (Cl(5,0) / Cl(4,1), 16 / 24 / 28 / 32 output channels, at most 8 input channels = verts_per_row * channels_per_vertex);
 * every other shape returns CSMPN_ERR_UNSUPPORTED (the caller composes csmpn_simplex_rows + csmpn_cemlp_* + a sum).
 *   vertex_feat [n_feature_rows, channels_per_vertex, D]  the embedded (full multivector) features of every batch row;
 *   verts [n_rows, verts_per_row] int32: row r = vertex order r % n_orders of simplex r / n_orders (n_rows = n_simplices * n_orders);
 *   out [n_simplices, O, D]: out[s] = sum over its n_orders rows of CEMLP(concat_v vertex_feat[verts[r][v]]).
 * Neither the [n_rows, I, D] input rows nor the [n_rows, O, D] per-order outputs exist in memory; the sum is taken in
 * registers in row order (no atomics). n_orders in {1, 2, 6}. save_inputs / saved_inputs as for csmpn_cemlp_* (rows = n_rows).
 * n_vertex_rows = n_feature_rows (ABI version 2). Every entry of verts must lie in [0, n_vertex_rows): both entry points
 * range-check the table (one synchronous host round trip on `stream`, as csmpn_csr_build) and return CSMPN_ERR_INVALID
 * otherwise; a caller that has validated its table once per batch passes CSMPN_FLAG_NO_VALIDATE (required inside a stream
 * capture). The kernels clamp the ids into [0, n_vertex_rows) in any case: a bad id reads a wrong row, never out of bounds.
 * backward: d/d(parameters) only (the features are data); g_out [n_simplices, O, D]. */
int csmpn_embed_cemlp_forward(const float* metric_host, int n, const csmpn_block_params* blocks, int n_blocks,
                              const float* vertex_feat, int64_t n_vertex_rows, int32_t channels_per_vertex, const int32_t* verts,
                              int32_t verts_per_row, int32_t n_orders, int64_t n_rows, float* out, float* save_inputs,
                              void* workspace, size_t workspace_bytes, uint32_t flags, void* stream);
int csmpn_embed_cemlp_backward(const float* metric_host, int n, const csmpn_block_params* blocks, const csmpn_block_grads* grads,
                               int n_blocks, const float* vertex_feat, int64_t n_vertex_rows, int32_t channels_per_vertex,
                               const int32_t* verts, int32_t verts_per_row, int32_t n_orders, int64_t n_rows, const float* g_out,
                               const float* saved_inputs, void* workspace, size_t workspace_bytes, uint32_t flags, void* stream);

/* Vector readout + loss of the trajectory task models (round 5; md17_cssmpnn.py:165-176, motion_cssmpnn.py:150-168,
 * nba_cssmpnn.py:176-191): the final MVLinear of the head restricted to the vector blades, the residual on the positions, the
 * distance to the target and the per-graph losses in one launch (fixed-order sums, no atomics).
 *   x [n_rows, C, D]; vertex_rows [V] int32 = row of x of every vertex (NULL: row v = vertex v, n_rows == V);
 *   weight [O, C, weight_stride] (layers' MVLinear.weight [O, C, G], subspaces = True: the grade-1 entry is used; the bias
 *   sits on blade 0 and does not reach the vector blades); loc [V, O, n] or NULL; target [V_t, O, n]; target_row [V] int32 =
 *   row of target of every vertex, -1 = not scored (NBA: the ball) (NULL: row v); vertex_ptr [B + 1] int32: the vertices of
 *   graph b are vertex_ptr[b] .. vertex_ptr[b + 1] - 1.
 *   pred[v][o][a]   = sum_c weight[o][c][1] x[row(v)][c][1 + a] (+ loc[v][o][a])                       [V, O, n]
 *   d               = pred - target over the cnt_b scored vertices of graph b
 *   per_graph[b]    = ( sum |d|^2 / (cnt_b O),  sum |d| / (cnt_b O),  sum_v |d[v][O - 1]| / cnt_b )     [B, 3]  (MSE, ADE, FDE)
 *   per_vertex[v]   = sum_{o, a} d^2 / (O n)  (0 if not scored)                                         [V]
 * backward: given g_per_graph [B, 3] and / or g_per_vertex [V] (either may be NULL): gx [n_rows, C, D] is WRITTEN (zero
 * outside the vector blades of the vertex rows; vertex_of_row [n_rows] int32 = vertex of a row or -1, NULL with vertex_rows
 * NULL), g_weight [O, C, weight_stride] is ACCUMULATED at the grade-1 entries; gpred_scratch [V, O, n] floats of scratch;
 * graph_of_vertex [V] int32. A zero distance contributes no ADE / FDE gradient (the reference's sqrt backward gives NaN there).
 * Limits: C <= 64, O <= 1024, O * C <= 4096. */
int csmpn_readout_traj_forward(int n, const float* x, int32_t channels, const int32_t* vertex_rows, int64_t n_vertices,
                               const float* weight, int32_t out_channels, int32_t weight_stride, const float* loc, const float* target,
                               const int32_t* target_row, const int32_t* vertex_ptr, int64_t n_graphs, float* pred, float* per_graph,
                               float* per_vertex, void* stream);
int csmpn_readout_traj_backward(int n, const float* x, int32_t channels, int64_t n_rows, const int32_t* vertex_rows,
                                const int32_t* vertex_of_row, int64_t n_vertices, const float* weight, int32_t out_channels,
                                int32_t weight_stride, const float* pred, const float* target, const int32_t* target_row,
                                const int32_t* graph_of_vertex, const int32_t* vertex_ptr, const float* g_per_graph,
                                const float* g_per_vertex, float* gpred_scratch, float* gx, float* g_weight, void* stream);

/* Node / edge attributes of the simplicial task models from per-type features (md17_cssmpnn.py:122-133 embed_simplex_types:
 * sim_type_embedding(node_types) embedded as scalars, edge attribute = (source, target) attributes side by side):
 *   node_attr[s][k][0]     = table[types[s]][k]                 [n_nodes, K, D], the other blades 0
 *   edge_attr[e][k][0]     = table[types[src[e]]][k]            [n_edges, 2 K, D]
 *   edge_attr[e][K + k][0] = table[types[dst[e]]][k]
 * table [n_types, K] (nn.Embedding.weight; n_types * K <= 64), types [n_nodes] in [0, n_types), src / dst = edge_index[0] /
 * edge_index[1] (original edge order), all int32. backward: g_table[t][k] += the blade-0 gradients of every row that read
 * table[t][k] (accumulated: the caller zeroes g_table; either gradient may be NULL; float atomics on n_types * K values). */
int csmpn_type_attr_forward(int n, const float* table, int32_t n_types, int32_t k, const int32_t* types, int64_t n_nodes,
                            const int32_t* src, const int32_t* dst, int64_t n_edges, float* node_attr, float* edge_attr, void* stream);
int csmpn_type_attr_backward(int n, int32_t n_types, int32_t k, const int32_t* types, int64_t n_nodes, const int32_t* src,
                             const int32_t* dst, int64_t n_edges, const float* g_node_attr, const float* g_edge_attr, float* g_table,
                             void* stream);

/* Scalar readout + loss of the convex-hulls model (hulls_cssmpnn.py:93,155-164):
 *   pred_g = mean_{s in graph g} (sum_c weight[c * weight_stride] * x[s][c][0]) + bias,  loss_g = (pred_g - target_g)^2
 * weight points at MVLinear.weight[0] (out_features = 1; weight_stride = n+1 with subspaces, else 1),
 * bias at MVLinear.bias or NULL. The simplices of graph g are rows [graph_ptr[g], graph_ptr[g+1]).
 * channel_sums [n_graphs, channels] receives sum_s x[s][c][0] (the weight gradient is coef^T channel_sums).
 * backward: gx[s][c][d] = (d == 0) ? coef[graph(s)] * weight[c * weight_stride] : 0 (overwritten), with
 * coef_g = dL/dloss_g * 2 (pred_g - target_g) / max(count_g, 1) computed by the caller. */
int csmpn_readout_mse_forward(int n, const float* x, const float* weight, int32_t weight_stride, const float* bias,
                              int64_t n_rows, int32_t channels, const int32_t* graph_ptr, int64_t n_graphs,
                              const float* target, float* pred, float* loss, float* channel_sums, void* stream);
int csmpn_readout_mse_backward(int n, const float* weight, int32_t weight_stride, int64_t n_rows, int32_t channels,
                               const int32_t* graph_ptr, int64_t n_graphs, const float* coef, float* gx, void* stream);

/* Last error message of the calling thread (never NULL). */
const char* csmpn_last_error(void);

/* Diagnostic: the kernel the calling thread's last CEMLP / EGCL-stage entry point dispatched ("" before the first
 * launch), spelled as rocprofv3 prints it where the family's template arguments are known at the dispatch site, e.g.
 * "csmpn::cemlp_cl_bwd_kernel<csmpn::Alg<3, 0u>, 8, 1, 2, 6>". bench.py reports it as roofline.kernel. */
const char* csmpn_last_kernel(void);

/* Library/ABI version and the gfx target it was built for. */
int csmpn_abi_version(void);
const char* csmpn_build_target(void);

#ifdef __cplusplus
}
#endif
#endif /* CSMPN_HIP_H */
