/* gae_hip_experimental.h -- the SEAMS of libgae_hip.so that its own host mirror (gae_dgl_amd/ops/, capture.py,
 * sparse.py, parallel.py) uses to fuse launches, build plans and defer reductions.  A maintainer who binds the library
 * behind the reference's call sites needs gae_hip.h only; nothing here changes a result -- every entry
 * point is a faster composition of, or a set-up step for, an entry point of gae_hip.h:
 *   gae_x_*                     launch-diet forms of a training step (producer epilogues, partial sums handed to the
 *                               optimiser launch, collate steps that read a device-side cursor)
 *   gae_spmm_plan_* / _ell_build / _csr_blockdiag / _csr_epilogue
 *                               plan construction and the specialised aggregation kernels gae_spmm_csr dispatches to
 *   gae_spx_*, gae_dense_to_csr_*   layer 1 from the non-zeros of constant input features (gae_dgl_amd.SparseFeatures)
 *   gae_linear2_*, gae_gcn2_*   the dense halves of a two-layer encoder on millions of rows (row-sharded RMAT path)
 * Same conventions as gae_hip.h: caller-owned buffers, 0 / negative / hipError_t return codes, asynchronous launches on
 * the stream passed last.  These signatures may change between versions without a GAE_VERSION major bump. */
#ifndef GAE_HIP_EXPERIMENTAL_H
#define GAE_HIP_EXPERIMENTAL_H

#include "gae_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* (from gae_hip.h: graph structure) */
/* gae_batch_select + gae_batch_plan in one launch (the step of a captured HIP graph): ids of batch *cursor_dev of the
 * epoch order -> out_ids, their prefix sums -> out_*_ptr, *cursor_dev += 1. */
int gae_x_batch_plan_next(const int64_t *graph_ptr, const int32_t *ds_indptr, const int32_t *ds_t_indptr,
                        const int64_t *order, int64_t n_order, int64_t *cursor_dev, int64_t n_graphs,
                        int64_t *out_ids, int64_t *out_node_ptr, int64_t *out_edge_ptr, int64_t *out_t_edge_ptr,
                        void *stream);

/* gae_x_batch_plan_next + gae_batch_gather (fixed-capacity form) in ONE launch, for batches of <= 1024 graphs: the ids
 * of batch *cursor_dev of the epoch order, their prefix sums (out_ids [n_graphs], out_node_ptr / out_edge_ptr
 * [n_graphs + 1]) and the gathered, capacity-padded batch; *cursor_dev += 1.  out_counts: int64[4], ZERO before the
 * first call -- [0..2] as gae_batch_gather, [3] is the launch's block ticket (left zero).  The CSR of the batch is
 * used for the transposed structure as well (symmetric datasets: every bond stored in both directions,
 * gae_dgl/prepare_data.py:61-64). */
int gae_x_batch_gather_next(const int64_t *graph_ptr, const int32_t *ds_indptr, const int32_t *ds_indices,
                          const void *ds_feat, int64_t ld_feat, int64_t F, int dtype,
                          const int64_t *order, int64_t n_order, int64_t *cursor_dev, int64_t n_graphs,
                          int64_t *out_ids, int64_t *out_node_ptr, int64_t *out_edge_ptr,
                          int64_t cap_nodes, int64_t cap_edges,
                          int32_t *out_indptr, int32_t *out_indices, void *out_feat, int64_t ld_out,
                          int32_t *out_ell, int32_t ell_width, int64_t *out_counts, void *stream);

/* plan construction on the device (csrc/plan_build.hip): classification of the rows, descriptors, compact tagged
 * ids of the mid rows, XCD-pinned regrouping of the very long rows -- integer work, deterministic (ascending rows, stable
 * partition, stable sort).  Every call synchronises `stream` once to hand counters to the host.
 *   sizes_host[0..6] = {light rows (1 .. threshold edges), mid rows (threshold < d <= pin_degree), their segments, their
 *                       edges, pinned rows (d > pin_degree), their edges, maximum degree};  scratch >= 64 bytes
 *   pin_degree = INT32_MAX: no pinned rows (every row above the threshold is a mid row). */
/* light_desc [n_light][4] (or NULL: no list); heavy_rows, heavy_seg_base [n_mid]; seg_heavy [segments]; seg_desc [segments][4]; mid_ids
 * [mid_edges] or NULL (no compact copy; hot_columns > 0 needs it: the ids of the hot_columns most gathered columns get the
 * tag); vh_rows [n_pinned]; vh_part_ptr [n_pinned + 1].  pinned_host_out[0] = virtual rows NV, [2] = tag threshold. */
/* pinned rows, two calls on the SAME scratch: vh_desc == NULL partitions the ids into vh_cols [pinned_edges], sorts the
 * chunks and reports pinned_host_out[1] = V; the second call fills vh_desc [V][4] and vh_part_pos [NV]. */
int gae_spmm_plan_sizes(const int32_t *indptr, int64_t n_rows, int32_t threshold, int32_t pin_degree, int32_t segment_edges,
                        int64_t *sizes_host, void *scratch, int64_t scratch_bytes, void *stream);

int64_t gae_spmm_plan_scratch_bytes(int64_t n_rows, int64_t n_cols, int64_t n_pinned, int64_t pinned_edges,
                                    int32_t segment_edges);

int gae_spmm_plan_build_rows(const int32_t *indptr, const int32_t *indices, int64_t n_rows, int64_t n_cols,
                             int32_t threshold, int32_t pin_degree, int32_t segment_edges, const int64_t *sizes,
                             int64_t hot_columns, int32_t *light_desc, int32_t *heavy_rows, int32_t *heavy_seg_base,
                             int32_t *seg_heavy, int32_t *seg_desc, int32_t *mid_ids, int32_t *vh_rows,
                             int32_t *vh_part_ptr, void *scratch, int64_t scratch_bytes, int64_t *pinned_host_out,
                             void *stream);

int gae_spmm_plan_build_pinned(const int32_t *indptr, const int32_t *indices, int64_t n_rows, int64_t n_cols,
                               int32_t segment_edges, const int64_t *sizes, const int32_t *vh_rows, int32_t *vh_cols,
                               int32_t *vh_desc, int32_t *vh_part_pos, void *scratch, int64_t scratch_bytes,
                               int64_t *pinned_host_out, void *stream);

/* width: 4, 8 or 16 slots per row */
int gae_spmm_ell_build(const int32_t *indptr, const int32_t *indices, int64_t n_rows, int32_t width,
                       int32_t skip_degree, int32_t *ell, void *stream);

/* Block-diagonal form of the same product (the batched molecule graphs of gae_dgl/train_inductive.py:31-35):
 * block_ptr[n_blocks + 1] (int32, device) cuts the rows into runs that are CLOSED under adjacency (whole member
 * graphs; every column id of a run's rows lies inside the run).  One thread block streams its slice of H into
 * LDS with coalesced 16-byte loads and gathers from there.  fp32, n_cols == n_rows, 16-byte aligned rows
 * (ld % 4 == 0).  max_block_rows bounds a run's row count, max_block_edges the index slice staged in LDS (edges
 * beyond it are read from global memory): gae_spmm_blockdiag_lds_bytes(...) <= 160 KiB.  Same CSR-order sums as
 * gae_spmm_csr (bit-identical results). */
int64_t gae_spmm_blockdiag_lds_bytes(int64_t max_block_rows, int64_t max_block_edges, int64_t ldh);

int gae_spmm_csr_blockdiag(const int32_t *indptr, const int32_t *indices, const int32_t *block_ptr,
                           const int32_t *block_eptr /* [n_blocks + 1] = indptr[block_ptr[.]] */,
                           int64_t n_blocks, int64_t max_block_rows, int64_t max_block_edges, int64_t n_rows,
                           const float *H, int64_t ldh, float *M, int64_t ldm, int64_t F,
                           const float *row_scale, const float *col_scale, int flags, void *stream);

/* (from gae_hip.h: transform-first GCN layer) */
int64_t gae_xw_fwd_splits(int64_t n, int64_t f_in, int64_t f_out, int dtype);

/* The first half of gae_xw_wgrad only: the per-(row partition, column slice) partial products stay in `workspace`
 * (gae_xw_wgrad_workspace_bytes) and layout_out describes them for gae_adam_step's deferred reduction:
 *   layout_out[0] = partials of dW, [1] = floats between two of them, [2] = row pitch (floats) of a partial's [f_out]
 *   rows (element (j, k) of partial q: workspace[q * [1] + j * [2] + k]); [3] = float offset of the db partials,
 *   [4] = their count, [5] = floats between two of them (element j of partial q: workspace[[3] + q * [5] + j]). */
int gae_x_xw_wgrad_partials(const void *X, int64_t ldx, int dtype, int64_t n, int64_t f_in,
                          const float *G, int64_t ldg, const float *Gmask, int64_t ldgm,
                          const float *D, int64_t ldd, const float *Dmask, int64_t lddm, int64_t f_out,
                          int want_dW, int want_db, void *workspace, int64_t workspace_bytes,
                          int64_t *layout_out, void *stream);

int gae_spmm_csr_epilogue(const int32_t *indptr, const int32_t *indices, int64_t n_rows, int64_t n_cols,
                          const float *H, int64_t ldh, const float *Hmask, float *Y, int64_t ldy, int64_t F,
                          const float *row_scale, const float *col_scale, const gae_spmm_plan *plan,
                          const float *bias, int act, int64_t n_splits, int64_t split_stride, void *stream);

/* Two GCN heads on one aggregate in ONE launch (VGAE's mu and log sigma heads, gae_dgl_amd/vgae.py; the reference has
 * a single head: gae_dgl/gae.py:36-45): gae_gcn_layer_fused with the weight given as two matrices stacked along their
 * STORED rows ([W; W2], w_split rows in W, same strides) and the bias as [bias; bias2].  Forward (w_transposed = 0):
 * Y = [act(M W^T + b) | act(M W2^T + b2)].  Backward of identity heads (w_transposed = 1, strides swapped as in
 * gae_gcn_layer_fused): dH = (A^T dY) [W; W2]. */
int gae_x_gcn_layer_fused2(const int32_t *indptr, const int32_t *indices, int64_t n_rows, int64_t n_cols,
                         const float *H, int64_t ldh, float *M, int64_t ldm, int64_t F,
                         const float *row_scale, const float *col_scale, const gae_spmm_plan *plan,
                         const float *W, const float *W2, int64_t w_split, int w_transposed,
                         int64_t w_stride_out, int64_t w_stride_in, const float *bias, const float *bias2,
                         int64_t J, int act, float *Y, int64_t ldy, void *stream);

/* The identity-activation BACKWARD of gae_gcn_layer_fused (gae_dgl/gae.py:26-31 under autograd) in one launch:
 *   dH [n, f_in] (lddh) = (A^T dY) W        the fused kernel on the CSR of A^T (plan_t: its plan), W [f_out, f_in] (ldw)
 *                                            as nn.Linear stores it;
 *   dW [f_out, f_in] = dY^T M,  db [f_out] = colsum(dY)     side work of the same thread blocks on their own 32 (16)
 *                                            rows: M [n, f_in] (ldm) is the aggregate the forward stored.
 * dY [n, f_out] (lddy: whole 16-byte vectors), f_out <= 32, f_in <= 32, square graph.  The weight gradient leaves the
 * launch as per-block partial sums in `workspace` (gae_x_gcn_layer_fused_wgrad_workspace_bytes(n, f_out, f_in)):
 * layout_out[0] = number of partials, [1] = floats between two partials, [2] = float offset of the db partials inside
 * one (dW partial: element o * f_in + i).  dW / db != NULL: a second, small launch adds them up (the library's one
 * order for partial lists); both NULL: the caller hands the list to gae_adam_step (gae_adam_tensor.partials) -- no
 * weight-gradient launch at all in a captured training step. */
int64_t gae_x_gcn_layer_fused_wgrad_workspace_bytes(int64_t n_rows, int64_t f_out, int64_t f_in);

int gae_x_gcn_layer_fused_wgrad(const int32_t *t_indptr, const int32_t *t_indices, int64_t n, const float *dY,
                              int64_t lddy, int64_t f_out, const float *row_scale, const float *col_scale,
                              const gae_spmm_plan *plan_t, const float *W, int64_t ldw, int64_t f_in, float *dH,
                              int64_t lddh, const float *M, int64_t ldm, float *dW, float *db, void *workspace,
                              int64_t workspace_bytes, int64_t *layout_out, void *stream);

/* ... and of gae_x_gcn_layer_fused2 (two identity heads on one aggregate): dY = [dY1 | dY2] ([n, f_out], f_out = d1 + d2),
 * the weight is the stack [W; W2] along its stored rows (w_split = d1 rows come from W; both ldw apart), dW [f_out, f_in]
 * is stacked alike (rows < w_split = dW1), db [f_out]. */
int gae_x_gcn_layer_fused2_wgrad(const int32_t *t_indptr, const int32_t *t_indices, int64_t n, const float *dY,
                               int64_t lddy, int64_t f_out, const float *row_scale, const float *col_scale,
                               const gae_spmm_plan *plan_t, const float *W, const float *W2, int64_t w_split,
                               int64_t ldw, int64_t f_in, float *dH, int64_t lddh, const float *M, int64_t ldm, float *dW,
                               float *db, void *workspace, int64_t workspace_bytes, int64_t *layout_out, void *stream);

/* (from gae_hip.h: dense halves of a two-layer encoder on very tall operands (csrc/tall.hip)) */
/* ---- dense halves of a two-layer encoder on very tall operands (csrc/tall.hip) ---------------------------------
 * gae_linear2_fwd:  Y1 = act1(A W1^T + b1) [n, f_mid],  T = Y1 W2^T [n, f_out]  in ONE pass over A [n, f_in]:
 * NodeApplyModule of layer 1 (gae_dgl/gae.py:13-16) and the dense half of layer 2 evaluated transform-first.
 * f_in <= 64, f_mid <= 32, f_out <= 32; rows of A, Y1, T whole 16-byte vectors; W1 [f_mid, f_in] (ldw1), W2 [f_out,
 * f_mid] (ldw2) as nn.Linear stores them; b1 may be NULL; Y1 may be NULL (inference: only T is wanted). */
/* a_dead [n] (or NULL): rows of A that ARE zero and were never written -- the rows without edges of an aggregate
 * produced with GAE_SPMM_SKIP_ROWS -- are not read (their outputs are act1(b1) and its image under W2).
 * rows [n_listed] (or NULL; replaces a_dead): LIST MODE -- only the listed rows (ascending ids < n) are read, computed
 * and written; the pass is bound by the fp32 matrix pipe, so on a power-law graph, where most rows of an aggregate
 * have no in-edges, it does a fraction of the work.  gae_linear2_fill_dead writes T for all the other rows (their
 * common value act1(b1) W2^T; `dead` [n] marks them). */
/* gae_gcn2_bwd_dense: every dense product of that encoder's backward pass in ONE pass over its four tall operands
 * (train_inductive.py:51 for the model of gae.py:36-45 with two layers), given G = A^T dZ [n, f_out]:
 *     dW2 = G^T Y1,  db2 = colsum(dZ),  dY1 = (G W2) (.) act1'(Y1),  dW1 = dY1^T M1,  db1 = colsum(dY1)
 * Y1 [n, f_mid] = the output of layer 1, M1 [n, f_in] = its stored aggregate A X; widths <= 32; rows of G and dZ whole
 * 16-byte vectors.  dY1 is never stored.  Per-block partial sums go to `workspace` (gae_gcn2_bwd_dense_workspace_bytes)
 * and are added in block order (deterministic).  layout_out != NULL: stop after the partials (dW1 .. db2 are not
 * written) and report {n_partials, floats per partial, offset of db1, of dW2, of db2} (dW1 at 0) for gae_adam_step's
 * deferred reduction. */
/* m1_dead / g_dead [n] (or NULL; recomputing form only): rows of M1 / G that ARE zero and were never written
 * (GAE_SPMM_SKIP_ROWS) are not read.
 * rows [n_listed] (or NULL): LIST MODE -- the rows that HAVE an M1 row, ascending; m1_dead must then mark exactly the
 * others.  The pass visits the listed rows only (g_dead_listed [n_listed]: the G mask by list entry, or NULL); the
 * others' share -- their H1 row is act1(b1), so it is a rank-one term of the column sums of their G and dZ rows --
 * is computed by two small launches and appended to the partial list as one more partial (workspace sized for it). */
/* (Y1 == NULL: the pass RECOMPUTES Y1 = act1(M1 W1^T + b1) from the tile of M1 it reads anyway -- W1 [f_mid, f_in] (ldw1),
 *  b1 [f_mid] or NULL -- with gae_linear2_fwd's products in its order, i.e. the same bits: the forward then need not
 *  store Y1 at all (gae_linear2_fwd with Y1 = NULL) and this pass reads 2 f_mid fewer floats per row.) */

int gae_linear2_fwd(const float *A, int64_t lda, int64_t n, int64_t f_in, const float *W1, int64_t ldw1,
                    const float *b1, int64_t f_mid, int act1, const float *W2, int64_t ldw2, int64_t f_out,
                    float *Y1, int64_t ldy1, float *T, int64_t ldt, const uint8_t *a_dead, const int32_t *rows,
                    int64_t n_listed, void *stream);

int gae_linear2_fill_dead(const float *b1, int64_t f_mid, int act1, const float *W2, int64_t ldw2, int64_t f_out,
                          const uint8_t *dead, int64_t n, float *T, int64_t ldt, void *stream);

int64_t gae_gcn2_bwd_dense_workspace_bytes(int64_t n, int64_t f_in, int64_t f_mid, int64_t f_out);

int gae_gcn2_bwd_dense(const float *G, int64_t ldg, const float *dZ, int64_t lddz, const float *Y1, int64_t ldy1,
                       int act1, const float *M1, int64_t ldm1, const float *W2, int64_t ldw2, int64_t n,
                       int64_t f_in, int64_t f_mid, int64_t f_out, float *dW1, float *db1, float *dW2, float *db2,
                       void *workspace, int64_t workspace_bytes, int64_t *layout_out, const float *W1, int64_t ldw1,
                       const float *b1, const uint8_t *m1_dead, const uint8_t *g_dead, const int32_t *rows,
                       int64_t n_listed, const uint8_t *g_dead_listed, void *stream);

/* (from gae_hip.h: layer 1 on SPARSE input features (opt-in; gae_dgl_amd.SparseFeatures)) */
/* ---- layer 1 on SPARSE input features (opt-in; gae_dgl_amd.SparseFeatures) -------
 * The citation features the reference loads as a dense FloatTensor (gae_dgl/train_transductive.py:37-38) are
 * bag-of-words rows with 1-10 % non-zeros.  Handed over in compressed form they give the same layer-1 values
 * (the skipped terms are exact zeros) from 8 bytes per non-zero instead of 4 bytes per entry:
 *   gae_dense_to_csr_count / _fill: compressed rows of a dense [n, K] matrix, columns ascending (count the non-zeros
 *       per row, prefix-sum them on the caller's side into rowptr[n + 1], fill col / val); used for X and for X^T.
 *   gae_spx_fwd:   P [n, f_out] = X W^T from the compressed rows of X (f_out <= 32), ascending-column order; the weight
 *       is first transposed into `workspace` (gae_spx_fwd_workspace_bytes(f_in): f_in rows of one 128-byte line), so that a
 *       non-zero gathers ONE line.  Measured (tools/r04/spx_bench.py, pair fwd + wgrad against gae_xw_fwd + gae_xw_wgrad):
 *       Citeseer 17.5 vs 34.7 us, Cora 15.0 vs 16.8 us, Pubmed 27.7 vs 26.4 us -- worth it for wide, very sparse X only
 *       (SparseFeatures.maybe_from_dense applies that rule).
 *   gae_spx_wgrad: dW [f_out, f_in] = G^T X from the compressed rows of X^T cut into SEGMENTS of <= 64 entries of
 *       one feature (seg_feat / seg_e0 / seg_slot [n_segments]: feature, first entry, index of the segment inside its
 *       feature; every feature has at least one -- possibly empty -- segment), and db = colsum(D (.) [Dmask > 0]).
 *       reduce = 1: dW / db are finished by a second launch; reduce = 0: the partial lists stay in `workspace`
 *       (gae_spx_wgrad_layout: [0] partials per element of dW, [1] floats between them (element (j, k) at j * f_in + k),
 *       [2] float offset of the db partials, [3] their count (32 floats apart), [4] workspace bytes) for
 *       gae_adam_step's deferred reduction. */

int gae_dense_to_csr_count(const float *X, int64_t ldx, int64_t n, int64_t K, int32_t *row_nnz, void *stream);

int gae_dense_to_csr_fill(const float *X, int64_t ldx, int64_t n, int64_t K, const int32_t *rowptr, int32_t *col,
                          float *val, void *stream);

int64_t gae_spx_fwd_workspace_bytes(int64_t f_in);

int gae_spx_fwd(const int32_t *rowptr, const int32_t *col, const float *val, int64_t n, int64_t f_in,
                const float *W, int64_t ldw, int64_t f_out, float *P, int64_t ldp, void *workspace,
                int64_t workspace_bytes, void *stream);

int gae_spx_wgrad_layout(int64_t n, int64_t f_in, int64_t max_segments_per_feature, int64_t *out);

int gae_spx_wgrad(const int32_t *t_rowptr, const int32_t *t_row, const float *t_val,
                  const int32_t *seg_feat, const int32_t *seg_e0, const int32_t *seg_slot, int64_t n_segments,
                  int64_t max_segments_per_feature, int64_t n, int64_t f_in,
                  const float *G, int64_t ldg, const float *D, int64_t ldd, const float *Dmask, int64_t lddm,
                  int64_t f_out, float *dW, int64_t lddw, float *db, int reduce,
                  void *workspace, int64_t workspace_bytes, void *stream);

/* (from gae_hip.h: K3-K5: node-apply (Linear + activation)) */
/* The first half of gae_linear_bwd's (dW, db): the per-row-slot partial products stay in `workspace` for
 * gae_adam_step's deferred reduction.  layout_out[0] = slots, [1] = floats between two slots (element e = o * f_in + i
 * of slot q: workspace[q * [1] + e]), [2] = float offset of a slot's f_out column sums (db) inside the slot. */
int gae_x_linear_bwd_partials(const float *dY, int64_t lddy, const float *Y, int64_t ldy, int act,
                            const float *M, int64_t ldm, int64_t n, int64_t f_in, int64_t f_out,
                            int want_dW, int want_db, void *workspace, int64_t workspace_bytes,
                            int64_t *layout_out, void *stream);

/* (from gae_hip.h: K7+K8+K9 fused: decoder + weighted BCE-with-logits, never materialising N x N) */
int gae_x_decoder_bce_prep_layout(int64_t n, int64_t d, void *workspace, int64_t workspace_bytes, gae_bce_prep *out);

/* gae_gcn_layer_fused (identity activation, square graph, J <= 16 outputs = the embedding Z [n, J], ldz) + the prepare
 * work of the loss that follows: mask [n, J] (ldmask) is the dropout multiplier -- drawn here (dropout_p > 0: the
 * Philox stream of gae_dropout_mask with *draw_dev as the draw index, written to `mask`) or given (dropout_p == 0, mask
 * may be NULL = all ones); counts_dev != NULL: fixed-capacity batch, rows >= counts_dev[0] are padding.
 * *n_prep_blocks_out = the number of column-sum partials written (pass it to gae_x_decoder_bce_prepared). */
int gae_x_gcn_layer_fused_prep(const int32_t *indptr, const int32_t *indices, int64_t n, const float *H, int64_t ldh,
                             float *M, int64_t ldm, int64_t F, const float *row_scale, const float *col_scale,
                             const gae_spmm_plan *plan, const float *W, int64_t w_stride_out, int64_t w_stride_in,
                             const float *bias, int64_t J, float *Z, int64_t ldz, const gae_bce_prep *prep, float *mask,
                             int64_t ldmask, float dropout_p, uint64_t seed, uint64_t offset, const uint64_t *draw_dev,
                             const int64_t *counts_dev, int64_t *n_prep_blocks_out, void *stream);

int gae_x_decoder_bce_prepared(float *mask, int64_t ldz, int64_t n, int64_t d, const int32_t *indptr,
                             const int32_t *indices, const int32_t *t_indptr, const int32_t *t_indices,
                             float pos_weight, const int64_t *counts_dev, float dropout_p, uint64_t *draw_dev,
                             int64_t n_prep_blocks, float *loss_out, float *dZ, int64_t lddz, void *workspace,
                             int64_t workspace_bytes, void *stream);

/* The VGAE head AND everything between it and the dense kernel of the loss in one launch (d = 16): eps of this draw
 * (draw_eps != 0: generated with gae_normal_noise's stream -- seed, offset, *draw_dev -- and WRITTEN to eps [n, d];
 * draw_eps == 0: eps is read), z = mu + eps exp(logstd) [n, d], the KL term as one partial per block of 64 rows in
 * kl_partial (capacity kl_capacity >= ceil(n / 64) doubles; scale -0.5 / n^2: put both into gae_bce_tail::kl_*), and
 * the prepare step of gae_decoder_bce on z without dropout (prep: gae_x_decoder_bce_prep_layout; then
 * gae_x_decoder_bce_prepared with *n_blocks_out).  mu / logstd: rows ldm floats apart (packed [mu | logstd]: ldm = 2 d). */
int gae_x_vgae_head_prep(const float *mu, const float *logstd, int64_t ldm, float *eps, int draw_eps, uint64_t seed,
                       uint64_t offset, const uint64_t *draw_dev, int64_t n, int64_t d, float *z, const gae_bce_prep *prep,
                       double *kl_partial, int64_t kl_capacity, int64_t *n_blocks_out, void *stream);

int gae_x_decoder_bce_defer_finalize(gae_bce_tail *tail_out);

int gae_x_decoder_bce_finalize(const gae_bce_tail *tail, void *stream);

/* (from gae_hip.h: K12: Adam) */
/* ... plus the deferred final reduction of the step's loss (gae_x_decoder_bce_defer_finalize; tail may be NULL) as one
 * more block of the same launch: one kernel node fewer in a captured training step. */
int gae_x_adam_step_tail(const gae_adam_tensor *tensors_host, int32_t n_tensors, float lr, float beta1, float beta2,
                       float eps, float weight_decay, uint64_t *state_dev, const gae_bce_tail *tail, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GAE_HIP_EXPERIMENTAL_H */
