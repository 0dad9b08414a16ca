/* csmpn_cpu.h — C-ABI of the C++ CPU twin of the hot path (oracle/cpu_twin/csmpn_cpu.cpp).
 *
 * TEST INFRASTRUCTURE, not the product: a scalar, torch-free restatement of one EGCL layer
 * (csmpn/models/cegnn_utils.py:216-284 and everything under it) in the SPARSE sign-table
 * formulation the HIP kernels use (D^2 products per channel instead of the reference's dense D^3
 * einsum), OpenMP over edges / nodes. It serves as (a) a second, independent ground truth for the
 * kernels, pinned to the same golden fixtures as the PyTorch oracle, and (b) the "strong CPU
 * baseline" of bench.py. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it. Same conventions as csmpn_hip.h, but every pointer is a HOST pointer.
 */
#ifndef CSMPN_CPU_H
#define CSMPN_CPU_H

#include "csmpn_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* One EGCL layer, forward and (when gout != NULL) backward, float32.
 *   out[v] = (residual ? h[v] : 0) + NodeCEMLP(cat[h[v], agg[v] * s_v, node_attr[v]]),
 *   agg[v] = sum_{e: dst_e = v} EdgeCEMLP(cat[h[dst_e] - h[src_e], edge_attr[e]]),  s_v = 1/max(deg_v,1) if mean.
 * edge_index: the reference's [2,E] int64 (row 0 = source, row 1 = target), any order; the sums over
 * a node's incoming edges run in ascending edge id (deterministic). Gradient outputs are OVERWRITTEN
 * (gh [N,C,D]; g_edge_attr [E,A,D] / g_node_attr [N,T,D] may be NULL); parameter gradients are ADDED to
 * the csmpn_block_grads tensors (zero them first). metric: n floats, any diagonal metric, n <= 6.
 * threads <= 0: all hardware threads. Returns 0 or CSMPN_ERR_*. */
int csmpn_egcl_layer_cpu(const float* metric, int n,
                         const csmpn_block_params* edge_blocks, const csmpn_block_grads* edge_grads, int n_edge_blocks,
                         const csmpn_block_params* node_blocks, const csmpn_block_grads* node_grads, int n_node_blocks,
                         const float* h, int32_t channels, const int64_t* edge_index, int64_t n_edges, int64_t n_nodes,
                         const float* edge_attr, int32_t edge_attr_channels, const float* node_attr,
                         int32_t node_attr_channels, int32_t mean_aggr, int32_t residual, const float* gout,
                         float* out, float* gh, float* g_edge_attr, float* g_node_attr, int32_t threads);

/* y[rows, O_last, D] = CEMLP(x); backward when gy != NULL (gx overwritten, may be NULL; grads added). */
int csmpn_cemlp_cpu(const float* metric, int n, const csmpn_block_params* blocks, const csmpn_block_grads* grads,
                    int n_blocks, const float* x, int64_t rows, const float* gy, float* y, float* gx, int32_t threads);

const char* csmpn_cpu_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* CSMPN_CPU_H */
