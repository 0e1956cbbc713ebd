/*
 * egnn_amd.h -- C ABI of the MI355X (gfx950) EGNN denoiser library (libegnn_amd.so).
 *
 * The reference (Ren-Okubo/diffusion_model) is pure Python and has no FFI layer; its de-facto
 * operator interface for the eps_theta(x_t, h_t, t) path is two call signatures and a state-dict
 * layout.  Each entry point below names the reference interface it replaces.
 *
 * Conventions
 *   - every function returns 0 on success or a negative EGNN_E* code; nothing throws;
 *   - all pointers named d_* are DEVICE pointers owned by the caller; the library allocates
 *     device memory only inside the opaque egnn_ctx workspace (freed by egnn_destroy);
 *   - every launch goes to the hipStream_t passed in (as void*); no call synchronises the
 *     device except egnn_create/egnn_destroy/egnn_reserve_*, egnn_sampler_prepare (allocation),
 *     and the explicitly named *_sync helpers;
 *   - a context is re-entrant across streams only if calls are externally ordered; it is not
 *     thread-safe;
 *   - tensors are row-major fp32 unless noted; weights are torch.nn.Linear layout [out, in].
 */
#ifndef EGNN_AMD_H
#define EGNN_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libegnn_amd.so is built with -fvisibility=hidden: exactly the functions declared below are exported */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

typedef struct egnn_ctx egnn_ctx;

enum {
  EGNN_OK = 0,
  EGNN_EINVAL = -22,   /* bad argument / unsupported dimension           */
  EGNN_ENOMEM = -12,   /* device allocation failed                       */
  EGNN_ESTATE = -1,    /* call order violated (model/graph/weights unset) */
  EGNN_EHIP = -5       /* a HIP runtime call failed (see egnn_last_error) */
};

/* arithmetic of the per-edge MLP contractions */
enum { EGNN_PREC_F32 = 0,   /* v_mfma_f32_32x32x2_f32: exact fp32 (parity mode)          */
       EGNN_PREC_BF16 = 1,  /* v_mfma_f32_32x32x16_bf16, fp32 accumulate (throughput)    */
       EGNN_PREC_BF16X3 = 2,/* fp32-grade accuracy on the bf16 matrix cores: both operands of the per-edge second-layer
                               * products split into bf16 head + bf16 remainder, three MFMAs per tile (2^-17 relative);
                               * fp32 table, heads and node MLP.  Shapes outside the 128-edge-tile kernels run as F32. */
       EGNN_PREC_F16 = 3,   /* the BF16 path's kernels on fp16 MFMA operands (v_mfma_f32_16x16x32_f16 / 32x32x16_f16: same
                               * rate, 11 significant bits instead of 8; overflow saturates at +-65504); inference only
                               * (egcl_forward_save stays bf16).  Hidden widths other than 512 / 1024 run as F32. */
       EGNN_PREC_F16C8 = 4 };/* fp32-grade accuracy for TWO bf16-equivalents of matrix work: fp16 heads on
                               * v_mfma_f32_16x16x32_f16 + both remainder products on ONE block-scaled e4m3 instruction
                               * (v_mfma_scale_f32_16x16x128_f8f6f4, twice the f16 rate) with fixed power-of-two block scales;
                               * fp32 table, heads and split-operand node MLP as BF16X3 (csrc/edge_f16c8.hip); inference
                               * only.  Hidden widths other than 512 / 1024 run as F32. */

/* scope of the coordinate normaliser ||X_i - X_j||_F of EquivariantGraphNeuralNetwork.py:64 */
enum { EGNN_NORM_CALL = 0,  /* literal reference: one scalar over every edge of the call  */
       EGNN_NORM_GRAPH = 1 };/* one scalar per graph (== reference when called per graph) */

const char* egnn_last_error(void);
int egnn_version(void);

/* ---- workspace ------------------------------------------------------------------------ */
int egnn_create(egnn_ctx** out, int device);
int egnn_destroy(egnn_ctx* ctx);

/* Model dimensions.  Replaces the constructor arguments of
 * EquivariantGNN(L, m_input, m_hidden, m_output, x_input, x_hidden, x_output, h_input, h_hidden,
 * h_output) (EquivariantGraphNeuralNetwork.py:74-78) under the wiring of main.py:102-121:
 * m_input = x_input = 2*H+1, x_output = 1, h_input = H+M, h_output = H. */
int egnn_set_model(egnn_ctx* ctx, int L, int H, int M, int Wm, int Wx, int Wh);

/* Upload + repack the parameters of layer l (state-dict entries egcl_list.{l}.*; fp32 device
 * pointers in nn.Linear layout).  Must be called again whenever the parameters change. */
int egnn_pack_layer(egnn_ctx* ctx, void* stream, int l,
                    const float* d_m0_w, const float* d_m0_b,   /* mlp_m.0  [Wm, 2H+1], [Wm] */
                    const float* d_m2_w, const float* d_m2_b,   /* mlp_m.2  [M, Wm],    [M]  */
                    const float* d_x0_w, const float* d_x0_b,   /* mlp_x.0  [Wx, 2H+1], [Wx] */
                    const float* d_x2_w, const float* d_x2_b,   /* mlp_x.2  [Wx, Wx],   [Wx] */
                    const float* d_x4_w, const float* d_x4_b,   /* mlp_x.4  [1, Wx],    [1]  */
                    const float* d_h0_w, const float* d_h0_b,   /* mlp_h.0  [Wh, H+M],  [Wh] */
                    const float* d_h2_w, const float* d_h2_b,   /* mlp_h.2  [H, Wh],    [H]  */
                    const float* d_a_w, const float* d_a_b);    /* attention.0 [1, M],  [1]  */

/* Graph topology for subsequent forwards.  Replaces the edge_index argument of
 * EquivariantGNN.forward(edge_index, h, x) (:85) plus PyG's batch vector.
 *   d_edge_dst/d_edge_src : int32[E], edges sorted by destination (edge_index[0]) -- the node
 *                           that RECEIVES the message (flow='target_to_source', :10-11);
 *   d_row_ptr             : int32[N+1] CSR offsets of that ordering;
 *   d_graph_ptr           : int32[B+1] node ranges of the B graphs (graphs are contiguous);
 *   d_node_graph          : int32[N] graph id of every node.
 * The arrays are referenced, not copied: they must stay alive while the graph is set. */
int egnn_set_graph(egnn_ctx* ctx, int N, int E, int B,
                   const int32_t* d_edge_dst, const int32_t* d_edge_src, const int32_t* d_row_ptr,
                   const int32_t* d_graph_ptr, const int32_t* d_node_graph);

/* Optional second stream of the caller's for small graphs: when a layer's coordinate kernel leaves most CUs idle, the
 * message kernel of the same layer is launched on `stream` between a fork and a join event (graph edges under capture).
 * The library itself never creates a stream; NULL (default) = everything on the stream of the call. */
int egnn_set_side_stream(egnn_ctx* ctx, void* stream);

/* One EGCL layer: EGCL.forward(edge_index, h, coords) -> (h', x')
 * (EquivariantGraphNeuralNetwork.py:67-71).  h [N,H], x [N,3]; outputs must not alias inputs. */
int egcl_forward(egnn_ctx* ctx, void* stream, int layer, int prec, int norm_scope,
                 const float* d_h, const float* d_x, float* d_h_out, float* d_x_out);

/* The same layer in two stages, for one large graph whose receiving nodes are partitioned over ranks
 * (BASELINE configs[4]): each rank sets a graph that holds only the edges its nodes receive.
 * _begin runs node_pre + the fused edge pass and returns this rank's sums of d^2 (float[B], or [1] in 'call'
 * scope); the caller all-reduces them (the coordinate normaliser of :64 spans all edges) and passes the
 * totals to _end, which runs node_post; rows of nodes the rank does not own are meaningless and the caller
 * all-gathers the owned rows.  d_sq_sums may be NULL (single rank: use the local sums). */
int egcl_forward_begin(egnn_ctx* ctx, void* stream, int layer, int prec, int norm_scope,
                       const float* d_h, const float* d_x, float* d_sq_sums);
int egcl_forward_end(egnn_ctx* ctx, void* stream, int layer, int prec, int norm_scope,
                     const float* d_h, const float* d_x, const float* d_sq_sums, float* d_h_out, float* d_x_out);

/* ---- backward of one EGCL layer (training: EquivariantGNN.forward must be differentiable w.r.t. h, x and all
 * parameters; the reference gets this from torch autograd over :55-71) -------------------------
 * The layer's backward is   l1_act -> GEMM -> heads -> GEMM,GEMM -> l1_grad -> GEMM,GEMM   where the GEMMs are
 * the plain dgrad / wgrad products of the Linear layers (run by the caller's BLAS on the buffers below) and the
 * three stage functions fuse everything in between, one pass over HBM each.  prec selects the storage type of
 * the [edges, width] buffers (EGNN_PREC_F32: float, EGNN_PREC_BF16: bf16); arithmetic is fp32.  dst/src are
 * the int32 edge arrays of egnn_set_graph (any contiguous slice of edges); all pointers are device memory.
 *
 * egcl_read_aggregates: the three segment sums of the layer that was just run on ctx by egcl_forward
 *   (sum_m [N,M] = sum of gated messages :68, sum_x [N,3] = sum of (x_i-x_j)*s before the 1/(G+1) factor :64/:70,
 *   sq_sums [B] or [1] = sum of d^2 per graph / per call), so that the backward need not recompute them. */
int egcl_read_aggregates(egnn_ctx* ctx, void* stream, int norm_scope, float* d_sum_m, float* d_sum_x,
                         float* d_sq_sums);
/* s1[e][c] = SiLU(P[dst e][c] + Q[src e][c] + wd[c]*d2[e]) for C table columns: the first Linear+SiLU of mlp_x
 * and mlp_m (:13-14, :19-20) on in = [h_i | h_j | d2] (:56), with P = h.W1[:, :H]^T + b1 and Q = h.W1[:, H:2H]^T
 * tabulated per node by the caller and wd = W1[:, 2H]. */
int egcl_backward_l1_act(void* stream, int prec, int n_edges, int C, const int32_t* d_dst, const int32_t* d_src,
                         const float* d_P, const float* d_Q, const float* d_wd, const float* d_d2, void* d_s1_out);
/* In place: a2x [n_edges,W] (= s1x.W2x^T) -> dL/da2x and a2m [n_edges,M] (= s1m.W2m^T) -> dL/da2m, through
 * SiLU, the scalar head mlp_x.4 and xm = (x_i-x_j)*s (:62-65), and SiLU, the attention gate and m*gate (:57-60),
 * given the gradients of the segment sums g_sum_x [N,3] (already multiplied by 1/(G+1)) and g_sum_m [N,M].
 * Also writes g_diff [n_edges,3] = dL/d(x_i-x_j) through s, and ADDS the column sums to g_b2x [W], g_w3 [W],
 * g_b3 [1], g_b2m [M], g_wa [M], g_ba [1] (gradients of mlp_x.2.bias, mlp_x.4.weight/.bias, mlp_m.2.bias,
 * attention.0.weight/.bias).  b3 and ba are device pointers to the two scalar biases. */
int egcl_backward_heads(void* stream, int prec, int n_edges, int W, int M, const int32_t* d_dst, const int32_t* d_src,
                        const float* d_x, const float* d_g_sum_x, const float* d_g_sum_m, void* d_a2x_inout,
                        void* d_a2m_inout, const float* d_b2x, const float* d_w3, const float* d_b3,
                        const float* d_b2m, const float* d_wa, const float* d_ba, float* d_g_diff, float* d_g_b2x,
                        float* d_g_w3, float* d_g_b3, float* d_g_b2m, float* d_g_wa, float* d_g_ba);
/* In place: g_s1[e][c] *= SiLU'(P[dst e][c] + Q[src e][c] + wd[c]*d2[e])  (dL/ds1 -> dL/da1). */
int egcl_backward_l1_grad(void* stream, int prec, int n_edges, int C, const int32_t* d_dst, const int32_t* d_src,
                          const float* d_P, const float* d_Q, const float* d_wd, const float* d_d2,
                          void* d_g_s1_inout);

/* in[e] = [h_i | h_j | d2 | 1 | 0 ...] (:56; K1P >= 2H+2 columns, the ones column carries the bias gradient through
 * the wgrad GEMM) in the storage type of prec, and d2[e] = |x_i - x_j|^2 as float. */
int egcl_backward_gather_in(void* stream, int prec, int n_edges, int H, int K1P, const int32_t* d_dst,
                            const int32_t* d_src, const float* d_h, const float* d_x, void* d_in_out, float* d_d2_out);
/* Adjoint of the gather: g_in [n_edges,K1P] = dL/d[h_i | h_j | d2 | ...] (storage type of prec).
 * g_h[i] += g_in[:H], g_h[j] += g_in[H:2H]; dL/d(x_i-x_j) = g_diff[e] + 2 (g_in[2H] + g_sq_sums[segment of i]) (x_i-x_j)
 * is added to g_x[i] and subtracted from g_x[j] (fp32 atomic adds).  node_segment [N] maps a node to its entry of
 * g_sq_sums (the gradient of egcl_read_aggregates' sq_sums); NULL = one sum for the whole call. */
int egcl_backward_scatter(void* stream, int prec, int n_edges, int H, int K1P, const int32_t* d_dst,
                          const int32_t* d_src, const float* d_x, const void* d_g_in, const float* d_g_diff,
                          const float* d_g_sq_sums, const int32_t* d_node_segment, float* d_g_h, float* d_g_x);

/* Node MLP backward, activation stage (mlp_h = Linear -> SiLU -> Linear, :26-30 under autograd): from the recomputed first-layer
 * product z [N, ldz] fp32 (WITHOUT its bias b1 [W]) and dL/ds [N, ldg] fp32, one pass writes s = SiLU(z + b1) and
 * dL/dz = dL/ds * SiLU'(z + b1) as bf16 [N, ldo] (the operands of the weight-gradient / dgrad products) and ADDS the bias
 * gradient (column sums of dL/dz) to d_g_b1 [W] (fp32 atomics: zero it first).  W and the row strides multiples of 4. */
int egcl_backward_node_act(void* stream, int N, int W, const float* d_z, int ldz, const float* d_b1, const float* d_g_s, int ldg,
                           void* d_g_z_out, void* d_s_out, int ldo, float* d_g_b1);

/* First Linear layers of both edge MLPs, factorised as the forward factorises them (csrc/edge_bwd_first.hip): ONE pass over
 * dL/da1 (d_g1x [n_edges, Wx], d_g1m [n_edges, Wm], bf16; the outputs of egcl_backward_dgrad for the edges
 * [e_first, e_first + n_edges) of the plan) produces, for batches of graphs of at most 64 nodes, the per-node sums
 * Gd[n] = sum of g1 over the edges n receives, Gs[n] = over the edges n sends ([N, W] fp32, ACCUMULATED: zero them per layer),
 * cd[graph] = sum_e g1[e] d2_e ([B, W], accumulated) and the (Wx + Wm) / 256 shares of dL/d(d2_e) = g1[e] . W1[:, 2H]
 * (d_gd2_part [(Wx + Wm) / 256, n_edges], assigned).  The first-layer weight gradients and dL/dh are then node-level products
 * (dL/dW1[:, :H] = Gd^T h, dL/dW1[:, H:2H] = Gs^T h, dL/dW1[:, 2H] = sum of cd, dL/db1 = sum of Gd, dL/dh += Gd W1[:, :H] +
 * Gs W1[:, H:2H]) -- replaces egcl_backward_gather_in, two weight-gradient GEMMs over all edges, the row-streaming dgrad GEMM and
 * the feature half of egcl_backward_scatter (torch autograd over EquivariantGraphNeuralNetwork.py:13-25, :56). */
int egcl_backward_first_reduce(void* stream, int B, int max_graph_nodes, int e_first, int n_edges, const int32_t* d_graph_ptr,
                               const int32_t* d_row_ptr, const int32_t* d_edge_src, const float* d_x, const void* d_g1x, int Wx,
                               const void* d_g1m, int Wm, const float* d_wdx, const float* d_wdm, float* d_Gd_x, float* d_Gs_x,
                               float* d_Gd_m, float* d_Gs_m, float* d_cd_x, float* d_cd_m, float* d_gd2_part);
/* The geometry half of egcl_backward_scatter for that path: dL/d(x_i - x_j) = g_diff[e] + 2 (sum of the nparts shares of
 * dL/d(d2_e) + g_sq_sums[segment of i]) (x_i - x_j), added to g_x[i] and subtracted from g_x[j] (fp32 atomic adds). */
int egcl_backward_scatter_geom(void* stream, int n_edges, int nparts, const int32_t* d_dst, const int32_t* d_src, const float* d_x,
                               const float* d_gd2_part, const float* d_g_diff, const float* d_g_sq_sums,
                               const int32_t* d_node_segment, float* d_g_x);

/* Fused first half of the chain above for the bf16 fast path (reference widths: hidden 256 / 512 / 1024, m = 256, unpadded):
 * one pass of the forward's own MFMA edge kernels in "backward" mode replaces gather/l1_act -> GEMM -> heads.  For the
 * edges [e_first, e_first + n_edges) of the graph set on ctx it recomputes SiLU(P[dst] + Q[src] + wd d2) (fp16 table built
 * by egcl_backward_table from the layer's input h) and the second-layer products exactly as the forward did, and writes
 *   s1x [n_edges, Wx], s1m [n_edges, Wm]  bf16, the activations AS THE MFMA CONSUMED THEM: scaled by -log2(e) (multiply the
 *                                         wgrad g_a2^T . s1 by -ln 2),
 *   g_a2x [n_edges, Wx], g_a2m [n_edges, M]  bf16 = dL/d(second-layer pre-activation) (what egcl_backward_heads leaves),
 *   g_diff [n_edges, 3] and the column sums ADDED to g_b2x, g_w3, g_b3, g_b2m, g_wa, g_ba, as egcl_backward_heads does.
 * g_sum_x [N,3] must already carry the 1/(G+1) factor; g_sum_m is [N, M].  egcl_backward_fused_supported = 1 if the
 * context's model takes this path. */
int egcl_backward_fused_supported(egnn_ctx* ctx);
int egcl_backward_table(egnn_ctx* ctx, void* stream, int layer, const float* d_h);
int egcl_backward_edge_recompute(egnn_ctx* ctx, void* stream, int layer, const float* d_x, const float* d_g_sum_x,
                                 const float* d_g_sum_m, int e_first, int n_edges, void* d_s1x, void* d_s1m,
                                 void* d_g_a2x, void* d_g_a2m, float* d_g_diff, float* d_g_b2x, float* d_g_w3,
                                 float* d_g_b3, float* d_g_b2m, float* d_g_wa, float* d_g_ba);

/* The same first half WITHOUT the recompute pass, for a training process that can afford E x (2 Wx + Wm + M) bf16 per
 * layer (7 GB at 2^20 edges and the reference widths): egcl_forward_save is egcl_forward in bf16 mode whose edge kernels
 * also leave in HBM, for ALL E edges of the layer,
 *   s1x [E, Wx], s1m [E, Wm]  bf16 as above (scaled by -log2(e)),
 *   t2x [E, Wx], t2m [E, M]   bf16 = -log2(e) * (second-layer pre-activation incl. bias),
 *   s_shares [Wx / 512][E]    fp32 column-split shares of s_e = w3 . SiLU(a2) + b3 (their sum is s_e);
 * (h_out, x_out) are bitwise those of egcl_forward, and the per-graph sums of d^2 stay readable (egcl_read_aggregates).
 * egcl_backward_heads_saved then turns rows [e_first, e_first + n_edges) of t2x / t2m into dL/d(a2) IN PLACE (an
 * element-wise pass at HBM speed: loss.backward() of parts/train_per_iretation.py:172 for the two heads, :57-65) and adds
 * the column sums / writes g_diff exactly as egcl_backward_edge_recompute does.  d_t2x / d_t2m point at row e_first. */
int egcl_forward_save(egnn_ctx* ctx, void* stream, int layer, int norm_scope, const float* d_h, const float* d_x,
                      float* d_h_out, float* d_x_out, void* d_s1x, void* d_s1m, void* d_t2x, void* d_t2m,
                      float* d_s_shares);
int egcl_backward_heads_saved(egnn_ctx* ctx, void* stream, int layer, const float* d_x, const float* d_g_sum_x,
                              const float* d_g_sum_m, int e_first, int n_edges, void* d_t2x, void* d_t2m,
                              const float* d_s_shares, float* d_g_diff, float* d_g_b2x, float* d_g_w3, float* d_g_b3,
                              float* d_g_b2m, float* d_g_wa, float* d_g_ba);

/* Fused second half for the same path: g_a1 = (g_a2 . W2) * SiLU'(a1) for mlp_x ([n_edges, Wx] from [n_edges, Wx]) and
 * mlp_m ([n_edges, Wm] from [n_edges, M]) on MFMA, i.e. the dgrad GEMMs of mlp_x.2 / mlp_m.2 with egcl_backward_l1_grad
 * in their epilogue (the first-layer pre-activations come from the table egcl_backward_table left on ctx).  bf16 row-major
 * in and out. */
int egcl_backward_dgrad(egnn_ctx* ctx, void* stream, int layer, const float* d_x, int e_first, int n_edges,
                        const void* d_g_a2x, const void* d_g_a2m, void* d_g_a1x_out, void* d_g_a1m_out);

/* The same dgrad WITHOUT dL/da1 in memory (csrc/edge_bwd_dgrad_graph.hip; replaces egcl_backward_dgrad +
 * egcl_backward_first_reduce for batches of graphs of at most 64 nodes): one workgroup per (graph, 256 hidden units) multiplies
 * g_a2 . W2, applies SiLU'(a1) on the accumulator tile and reduces it there for the first Linear layers:
 *   d_G  bf16 [N, 2 Wx + 2 Wm] = [Gd_x | Gs_x | Gd_m | Gs_m], Gd[n] / Gs[n] = fp32 sums of g1 over the edges node n receives /
 *        sends, stored as the bf16 operands of the node-level products (dW1 = G^T [h | 1], dL/dh += G W1);
 *   d_cd_x / d_cd_m [B, W] fp32: cd[graph] = sum_e g1[e] d2_e;
 *   d_gd2_part [(Wx + Wm) / 256, n_edges]: the column-slice shares of dL/d(d2_e) for egcl_backward_scatter_geom.
 * All ASSIGNED for the graphs of the chunk (rows of graphs without edges are left as they are: zero them once).  The chunk
 * [e_first, e_first + n_edges) must consist of WHOLE graphs (egnn_set_graph's graph ranges); d_g_a2x / d_g_a2m are the chunk's
 * rows. */
int egcl_backward_dgrad_reduce(egnn_ctx* ctx, void* stream, int layer, const float* d_x, int e_first, int n_edges,
                               const void* d_g_a2x, const void* d_g_a2m, void* d_G, float* d_cd_x, float* d_cd_m,
                               float* d_gd2_part);

/* Weight gradients of the backward (loss.backward() of parts/train_per_iretation.py:172 through the Linear layers of
 * EquivariantGraphNeuralNetwork.py:13-30): a reduction over ALL edges (or nodes) on the matrix cores,
 *     C[m][n] (+)= scale * sum_e A[e][m] * B[e][n],   m < rows, n < cols,
 * A [E, lda] and B [E, ldb] row-major bf16 with the reduction index as ROWS (the operands as the other backward kernels
 * leave them: dL/da2 and the activations s1; dL/da1 and the gathered inputs), C fp32 with leading dimension ldc.  M and N are
 * the operand widths the kernel reads (M % 256 == 0, N % 128 == 0, M <= lda, N <= ldb; columns beyond rows / cols are
 * computed and dropped), split over slices of E whose fp32 partial tiles go through d_workspace
 * (egnn_gemm_tn_workspace_bytes) and are added in slice order: deterministic.  Replaces torch.mm / bmm (round 2). */
size_t egnn_gemm_tn_workspace_bytes(int E, int M, int N);
int egnn_gemm_tn_bf16(void* stream, int E, int M, int N, const void* d_A, int lda, const void* d_B, int ldb, float scale,
                      float* d_C, int ldc, int rows, int cols, int accumulate, void* d_workspace, size_t workspace_bytes);

/* First-layer dgrad of the backward (autograd through the first Linear of mlp_x / mlp_m, :13, :19): a row-streaming product
 *     out[e][n] = sum_c A0[e][c] W0[c][n] + sum_c A1[e][c] W1[c][n],   n < 128,
 * A0 / A1 row-major bf16 [E, lda] (dL/da1 of the two MLPs; A1 may be NULL), out bf16 [E, ldo] (128 columns written);
 * W0 / W1 are fragment packs made by egnn_gemm_rows_pack from fp32 [K, ldw] matrices whose first ncols columns are used:
 * K * 128 bf16 per chunk of 128 columns, chunk after chunk (ceil(ncols / 128) chunks).  K % 64 == 0.  n_chunks column chunks
 * run in ONE launch (out columns 128 j .. 128 j + 127 from chunk j of both packs; ldo >= 128 n_chunks): the node MLP's
 * backward products are 8 chunks wide over only N / 256 row blocks.  Replaces torch.mm + addmm_ (round 2). */
int egnn_gemm_rows_pack(void* stream, int K, int ncols, const float* d_W, int ldw, void* d_frags_out);
int egnn_gemm_rows_bf16(void* stream, int E, const void* d_A0, int lda0, int K0, const void* d_W0, const void* d_A1, int lda1,
                        int K1, const void* d_W1, void* d_out, int ldo, int out_f32 /* fp32 [E, ldo] output */, int n_chunks);

/* EquivariantGNN.forward(edge_index, h, x) -> (h_L, x_L) (:85-88): all L layers. */
int egnn_forward(egnn_ctx* ctx, void* stream, int prec, int norm_scope,
                 const float* d_h, const float* d_x, float* d_h_out, float* d_x_out);

/* epsilon extraction of the callers (parts/train_per_iretation.py:161-163, :367-369):
 * eps_x = remove_mean(x_L - x_in), eps_h = h_L[:, :A].  d_graph_ptr int32[B+1] gives per-graph
 * means (remove_mean(x, batch_index), diffusion_x_h.py:9-13); NULL = one mean over all N nodes
 * (:6-8). */
int egnn_eps(void* stream, int N, int H, int A, const int32_t* d_graph_ptr, int B, const float* d_h_out,
             const float* d_x_out, const float* d_x_in, float* d_eps_x, float* d_eps_h);

/* remove_mean(x, batch_index) (diffusion_x_h.py:5-14) on [N,D], out of place. */
int egnn_remove_mean(void* stream, int N, int D, const int32_t* d_graph_ptr, int B, const float* d_in,
                     float* d_out);

/* ---- diffusion process ----------------------------------------------------------------- */
/* polynomial_schedule + clip_noise_schedule (diffusion_x_h.py:92-106) evaluated on the host in
 * fp32 in the same operation order.  alpha/sigma: float[T+1]; table: float[(T+1)*4] rows
 * {1/alpha_ts, sigma2_ts/(alpha_ts*sigma_t), std, t/T}; row 0 = final decode constants
 * {1/alpha_0, sigma_0/alpha_0, sigma_0/alpha_0, 0} (train_per_iretation.py:416-426).
 * Any of the three outputs may be NULL. */
int schedule_table_build(int T, double s, double power, float* alpha, float* sigma, float* table);
/* same table from explicit alpha/sigma arrays (learned gamma schedule, diffusion_x_h.py:36-46) */
int schedule_table_from_alpha(int T, const float* alpha, const float* sigma, float* table);

/* reverse_diffuse_one_step(z, eps, t, mode) (diffusion_x_h.py:75-90) with the step constants
 * c = {1/alpha_ts, sigma2_ts/(alpha_ts sigma_t), std}:  z_out = z*c0 - eps*c1 + c2*noise, noise
 * mean-removed (per graph if d_graph_ptr, else over all nodes) when mode_pos != 0.  d_noise supplies
 * the N(0,1) draw (the reference uses torch's global RNG).  z is [N, D] with row stride ldz.
 * With (c0, c1, c2) = (alpha_t, 0, sigma_t) the same kernel is diffuse_zero_to_t (:51-59). */
int ddpm_reverse_step(void* stream, int N, int D, int mode_pos, const int32_t* d_graph_ptr, int B,
                      float c0, float c1, float c2, const float* d_z, int ldz, const float* d_eps,
                      const float* d_noise, float* d_z_out, int ldo);

/* ---- device-resident sampler (generate(), parts/train_per_iretation.py:264-444) ---------- */
/* Allocates sampler state for the current graph: pos [N,3], h [N,H] = [s*x_types | cond | t/T].
 * d_cond [N, H-A-1] is the constant conditioning block ([compressed spectrum | exO]); d_table is
 * the DEVICE copy of schedule_table_build's table.  Noise comes from a counter-based Philox
 * generator keyed by (seed, step, node, component) unless d_noise_* are given to
 * egnn_sampler_step. */
int egnn_sampler_prepare(egnn_ctx* ctx, int T, int A, float onehot_scale, const float* d_table,
                         const float* d_cond, uint64_t seed);
/* x_only = 1 selects the x-only reverse loop of the reference's test.py:253-279 (process of E3diffusion_new.py:63-98):
 * the atom types given to egnn_sampler_init stay FIXED, only the positions take reverse steps (mu in the x_hat form is
 * algebraically the mu of diffusion_x_h.py:61-73, so the same step table serves both), and the loop ends with the reverse
 * step at t = 1: there is no t = 0 decode (egnn_sampler_final refuses; read egnn_sampler_state).  0 (default) = x and h. */
int egnn_sampler_set_mode(egnn_ctx* ctx, int x_only);
/* x_T ~ N(0,I) mean-removed per graph, h_T ~ N(0,I) (:301-305); or copy from the given arrays */
int egnn_sampler_init(egnn_ctx* ctx, void* stream, const float* d_pos_init, const float* d_x_init);
/* run `nsteps` reverse steps starting at the sampler's current t (initially T), optionally
 * replaying a captured hipGraph of one step; explicit noise arrays [nsteps][N][3] / [nsteps][N][A]
 * (step-major, first entry = highest t) may be NULL. */
int egnn_sampler_run(egnn_ctx* ctx, void* stream, int prec, int norm_scope, int nsteps,
                     int use_graph, const float* d_noise_pos, const float* d_noise_h);
/* final t=0 decode (:391-428): pos_0, continuous h_0, argmax one-hot (int32 [N,A]) */
int egnn_sampler_final(egnn_ctx* ctx, void* stream, int prec, int norm_scope,
                       const float* d_noise_pos, const float* d_noise_h,
                       float* d_pos_out, float* d_hc_out, int32_t* d_onehot_out);
/* current state + sticky per-graph non-finite flags (int32[B]; :376-389, :431-434) */
int egnn_sampler_state(egnn_ctx* ctx, void* stream, float* d_pos, float* d_x_types,
                       int32_t* d_bad_flags, int* t_host);

/* The same three fused kernels (x_T/h_T draw, one reverse step :366-373, t = 0 decode :412-428) on CALLER-owned state,
 * for the sampler of ONE LARGE GRAPH whose receiving nodes are partitioned over ranks (BASELINE configs[4]): between
 * two steps the EGNN forward runs stage-wise (egcl_forward_begin / _end) with collectives in between, so the loop is
 * driven by the caller and every rank applies the step to its replicated copy of the state (Philox noise is keyed by
 * (seed, step, GLOBAL node id): identical on every rank).  pos [N,3] and h [N,H] = [s*x_types | cond | t/T] are updated
 * in place; bad int32[B] is the sticky non-finite flag; d_noise_* (this step's draws, [N,3] / [N,A]) may be NULL. */
int ddpm_sampler_init(void* stream, int N, int H, int A, int B, int T, const int32_t* d_graph_ptr, const float* d_table,
                      float onehot_scale, uint64_t seed, const float* d_cond, const float* d_pos_init,
                      const float* d_x_init, float* d_pos, float* d_h, int32_t* d_bad);
int ddpm_sampler_step(void* stream, int N, int H, int A, int B, int T, int t, const int32_t* d_graph_ptr,
                      const float* d_table, float onehot_scale, uint64_t seed, const float* d_h_out, const float* d_x_out,
                      const float* d_noise_pos, const float* d_noise_h, float* d_pos, float* d_h, int32_t* d_bad);
int ddpm_sampler_final(void* stream, int N, int H, int A, int B, int T, const int32_t* d_graph_ptr, const float* d_table,
                       float onehot_scale, uint64_t seed, const float* d_h_out, const float* d_x_out,
                       const float* d_noise_pos, const float* d_noise_h, float* d_pos, float* d_h, int32_t* d_bad,
                       float* d_pos_out, float* d_hc_out, int32_t* d_onehot_out);

/* ---- graph construction ----------------------------------------------------------
 * Fully connected graphs in the edge kernels' CSR layout: node i receives from every j != i of its graph,
 * j ascending -- the edge set and order of parts/train_per_iretation.py:308-313 /
 * split_to_train_and_test.py:88-92 with PyG collate offsets.  d_edge_base int64[B] = number of edges of
 * the graphs before g (host prefix sum of n(n-1)); edge arrays may be NULL to fill row_ptr only. */
int egnn_fc_graph_build(void* stream, int N, int B, const int32_t* d_graph_ptr, const int32_t* d_node_graph,
                        const int64_t* d_edge_base, int32_t* d_row_ptr, int32_t* d_edge_dst, int32_t* d_edge_src);
/* Radius graph (not a reference feature; BASELINE configs[4]): pass 1 counts neighbours with |x_i-x_j| < r
 * inside each graph, the caller prefix-sums deg into row_ptr, pass 2 fills the CSR edge list. */
int egnn_radius_graph_count(void* stream, int N, const float* d_x, const int32_t* d_graph_ptr,
                            const int32_t* d_node_graph, float radius, int32_t* d_deg);
int egnn_radius_graph_fill(void* stream, int N, const float* d_x, const int32_t* d_graph_ptr,
                           const int32_t* d_node_graph, float radius, const int32_t* d_row_ptr,
                           int32_t* d_edge_dst, int32_t* d_edge_src);

/* ---- evaluation statistics -------------------------------------------------------
 * RDF(position, sigma, R, dR, Normalize) about atom 0 of every graph (evaluate_RDF.py:39-60), out
 * float[B, nbins], nbins = number of entries of np.arange(dR, R + dR, dR).  R and dR are doubles: the bin edges are
 * numpy's float64 values (rounded to float32 only inside the reference's `r < d < r + dR` comparison). */
int egnn_rdf(void* stream, int B, const float* d_pos, const int32_t* d_graph_ptr, double R, double dR, float sigma,
             int normalize, int nbins, float* d_out);
/* Si-O-Si selection + CN2 angle / bond lengths (evaluate_Si-O-Si.py:23-53, CN2_evaluate.py:12-21):
 * out float[B,4] = {valid, angle in degrees, |r1-r0|, |r2-r0|}; onehot int32 [N, A], Si = [0,1]. */
int egnn_si_o_si(void* stream, int B, int A, const float* d_pos, const int32_t* d_onehot,
                 const int32_t* d_graph_ptr, float cutoff, float* d_out);

/* ---- the two small networks at the edge of the path ---------------------------------------------------
 * gamma_tilde(t_i) = l1(t_i) + l3(sigmoid(l2(l1(t_i)))) of GammaNetwork (SNR.py:50-52) with PositiveLinear's softplus weights
 * (:5-22) for n time points; d_l1_w [1], d_l2_w [hidden], d_l3_w [hidden] are the RAW parameters (l1.weight, l2.weight,
 * l3.weight).  The caller normalises with gamma_tilde(0), gamma_tilde(1) and rescales to [gamma_0, gamma_1] (:54-64). */
int egnn_gamma_tilde(void* stream, int n, int hidden, const float* d_t, const float* d_l1_w, const float* d_l2_w,
                     const float* d_l3_w, float* d_out);
/* One Linear (+ ReLU) over node rows: out [N, J] = act(in [N, K] . W^T + b), W in nn.Linear layout [J, K]
 * (SpectrumCompressor, DataPreprocessor.py:10-19, layer by layer). */
int egnn_dense_rows(void* stream, int N, int K, int J, const float* d_in, const float* d_W, const float* d_b, int relu,
                    float* d_out);

/* timing helper for bench.py: average duration (ms) of the fused edge kernel over the launches
 * recorded since the last reset, measured with HIP events on the launch stream. */
/* diagnostic builds only (-DEGNN_EXP_STAMP): s_memtime stamps [2][8][32][4] of one edge workgroup */
int egnn_debug_stamps(egnn_ctx* ctx, unsigned long long* host_out);
int egnn_profile_enable(egnn_ctx* ctx, int enable);
int egnn_profile_read(egnn_ctx* ctx, float* edge_ms_avg, int* edge_launches, float* node_ms_avg);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* EGNN_AMD_H */
