/* gae_hip.h -- C ABI of libgae_hip.so: the MI355X (gfx950) implementation of
 * the GCN-encoder hot path of shionhonda/gae-dgl.
 *
 * The reference has no FFI of its own; its boundary is its Python module API
 * (gae_dgl/gae.py) plus the handful of calls it makes into the third-party
 * `dgl` package.  Each entry point below names the reference call it replaces
 * (paths relative to the upstream tree).  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions (all entry points):
 *   - plain pointers + sizes; every pointer is a DEVICE pointer unless the
 *     name ends in `_host`; the caller owns every buffer (inputs, outputs and
 *     workspace) -- the library never allocates, frees or retains memory;
 *   - `stream` is a hipStream_t passed as void*; every launch is asynchronous
 *     on it, there is no internal synchronisation;
 *   - return 0 = OK, negative = argument error detected before any launch
 *     (GAE_E_*), positive = hipError_t from the runtime; a thread-local
 *     message is available from gae_last_error(); nothing throws or exits;
 *   - row-major matrices with an explicit leading dimension (elements);
 *   - sparse structure is CSR with int32 `indptr[n_rows+1]`, int32 `indices[nnz]`;
 *     rows = destination nodes, columns = source nodes (in-edge aggregation).
 *
 * Two tiers of entry points:
 *   gae_*    THE SURFACE a maintainer binds: one entry point per operator of the reference's path --
 *            structure (gae_csr_from_coo, gae_batch_gather / _plan / _select, gae_degree_norm, gae_csr_to_dense,
 *            gae_rows_pack), aggregation (gae_spmm_csr / _ep / _epilogue / _blockdiag with their plan builders),
 *            node-apply (gae_linear_fwd / _bwd, gae_gcn_layer_fused, gae_xw_fwd / _wgrad, gae_linear2_fwd,
 *            gae_gcn2_bwd_dense), decoder + loss (gae_decoder_dense / _bwd, gae_bce_logits, gae_decoder_bce / _rows /
 *            _padded, gae_dropout_mask), the VGAE head, gae_segment_readout, gae_adam_step.  Stable names and
 *            argument meaning.
 *   gae_x_*  EXPERIMENTAL step fusions: the same arithmetic cut along the launch boundaries of one particular
 *            training step (a producer kernel that also runs the loss's prepare step, reductions that ride in the
 *            optimiser launch, two heads in one launch, collate + cursor in one launch).  They exist to take launches
 *            out of the captured step of gae_dgl_amd/capture.py, are paired in ways the comments spell out, and may
 *            change or disappear between rounds; every one of them has a gae_* sequence with the same result.
 *
 * Numerical contract of the dense products (gae_xw_fwd, the weight gradients, the fused loss): fp32 storage and fp32
 * accumulation; by default the multiplications run on the 16-bit matrix pipe from split operands (three bf16 pieces per
 * operand, six piece pairs; two fp16 pieces behind a range guard in the loss) -- within 1.5e-7 of fp64, as the fp32 MFMAs
 * they replace (knobs xw_p3 / atb_bf16 / bce_s_bf16 = 0 select exact fp32 MFMAs).  NON-FINITE INPUTS differ from an fp32
 * product: an operand entry that is +-Inf, or finite with |v| > 3.3895e38 (beyond the largest bf16), makes the affected
 * outputs NaN where an fp32 product would give +-Inf or a huge finite value (the residual v - bf16(v) is Inf - Inf).
 * Results are non-finite either way; NaN inputs give NaN in both forms.
 */
#ifndef GAE_HIP_H
#define GAE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GAE_VERSION 105 /* 0.1.1 */

enum {
    GAE_OK = 0,
    GAE_E_NULL = -1,      /* required pointer is NULL */
    GAE_E_SIZE = -2,      /* negative / overflowing / inconsistent size */
    GAE_E_ALIGN = -3,     /* pointer or leading dimension not aligned as required */
    GAE_E_DTYPE = -4,     /* unsupported dtype / activation code */
    GAE_E_WORKSPACE = -5, /* workspace too small */
    GAE_E_RANGE = -6      /* value out of the supported range */
};

enum { GAE_F32 = 0, GAE_BF16 = 1,       /* storage dtype; accumulation is always fp32 */
       GAE_U8 = 2 };                    /* 0/1 molecule features of a resident dataset (gae_batch_gather input only) */
enum { GAE_ACT_IDENTITY = 0, GAE_ACT_RELU = 1 };

typedef struct gae_device_info {
    int32_t compute_units;
    int32_t wavefront_size;
    int32_t lds_bytes_per_cu;
    int32_t l2_bytes;
    int64_t hbm_bytes;
    int32_t clock_khz;
    int32_t gfx_major_minor; /* e.g. 950 */
    char name[64];
} gae_device_info;

/* ---- library ------------------------------------------------------------ */
int gae_version(void);
const char *gae_last_error(void);
int gae_device_info_get(int device, gae_device_info *out_host);
/* Tuning / test knobs: ONE integer per process and knob (relaxed atomics, read at launch time).  Process-wide on
 * purpose: PyTorch runs the backward of an autograd Function on its engine's worker thread, so a per-thread value
 * would not reach the launches made in backward() (dW, A^T products).  Meant to be set once, before the launches
 * they should affect; every other piece of library state is per call.  Three kinds:
 *  - select among kernels with bit-identical results: "spmm_variant", "spmm_rpg", "spmm_nt", "spmm_tile_vecs",
 *    "spmm_ell", "spmm_ell_rpg", "spmm_hot", "spmm_desc", "spmm_parts" (bit mask of the parts of a skew-plan
 *    launch that run; experiments only), "bce_strip_store", "bce_fold_mirror";
 *  - change the ORDER in which partial sums are added (results agree within the fp32 tolerance of DESIGN.md
 *    section 6, not bit for bit): "atb_rows", "gemm_stream", "gemm_rows", "linear_wlds", "linear_f32x16", "linear_nw", "linear_depth", "bce_ri", "bce_sym", "bce_sym_ri",
 *    "bce_sym_grid", "bce_sym_tiles", "bce_grid"; a skew
 *    plan (gae_spmm_plan with heavy rows) and GAE_SPMM_ACCUMULATE do the same for the rows they touch;
 *    "xw" (0 = the layer-1 stream kernels of gae_xw_fwd are never used by gae_linear_fwd), "xw_rows" (rows per block
 *    of gae_xw_fwd, 0 = auto), "xw_parts" (row partitions of gae_xw_wgrad, 0 = auto) likewise; "xw_depth" / "xw_xcd" /
 *    "xw_glds" select pipeline depths / the block order of those kernels / whether gae_xw_wgrad stages its rows of G
 *    in LDS (same sums, bit for bit); "xw_dbg" (experiments: WRONG results --
 *    1 = no MFMAs, 2 = no loads of X);
 *  - select the ARITHMETIC of matrix-core products: "bce_s_bf16" / "bce_pv_bf16" / "atb_bf16" (1 = bf16 x 3 split
 *    products, default; 0 = exact fp32 MFMA) and "linear_bf16" (default 0 = exact fp32 forward Linear). */
int gae_tuning_set(const char *name, int64_t value);
int gae_tuning_get(const char *name, int64_t *value_out);

/* ---- graph structure -------------------------------------------------------
 * Replaces the DGL graph index built by DGLGraph.add_edges / dgl.batch
 * (gae_dgl/prepare_data.py:53,65; gae_dgl/train_inductive.py:34) that
 * g.update_all traverses (gae_dgl/gae.py:28). */

/* bytes of workspace gae_csr_from_coo needs for n_edges edges */
int64_t gae_csr_from_coo_workspace_bytes(int64_t n_edges, int64_t n_rows);

/* COO (row[e], col[e]) int64 -> CSR; rows ascending, columns ascending inside
 * a row, duplicates kept.  Pass (dst, src) for the aggregation matrix A and
 * (src, dst) for A^T (the backward structure).  Returns GAE_E_RANGE through a
 * device flag only for ids outside [0,n_rows) x [0,n_cols): out-of-range
 * edges set *status_dev (int32, may be NULL) to 1 and are clamped. */
int gae_csr_from_coo(const int64_t *row, const int64_t *col, int64_t n_edges,
                     int64_t n_rows, int64_t n_cols,
                     int32_t *indptr, int32_t *indices,
                     void *workspace, int64_t workspace_bytes,
                     int32_t *status_dev, void *stream);

/* Row pack of the row-sharded exchange (the reference has no multi-GPU code,
 * gae_dgl/train_inductive.py:29 is its only device logic): out[i] = H[idx[i]] for i < n_rows (idx NULL: H[i]),
 * zero rows from n_rows to n_out_rows.  idx (int64, device) entries must lie in [0, n_src_rows).  Lets the rows a
 * rank sends go straight into the all-to-all's send buffer and its own rows straight into the buffer the local CSR
 * indexes. */
int gae_rows_pack(const float *H, int64_t ldh, int64_t n_src_rows, const int64_t *idx, int64_t n_rows,
                  int64_t n_out_rows, int64_t F, float *out, int64_t ldo, void *stream);

/* g.in_degrees() + norm = deg^-1/2, inf -> 0 (gae_dgl/train_transductive.py:55-58).
 * deg_out (int32, may be NULL), norm_out (fp32, may be NULL). */
int gae_degree_norm(const int32_t *indptr, int64_t n_rows, int32_t *deg_out,
                    float *norm_out, void *stream);

/* g.adjacency_matrix().to_dense() (gae_dgl/train_inductive.py:44,
 * gae_dgl/train_transductive.py:59): out[r, c] = #edges (duplicates add).
 * Parity/debug path only; out is n_rows x n_cols fp32, ld in elements. */
int gae_csr_to_dense(const int32_t *indptr, const int32_t *indices, int64_t n_rows,
                     int64_t n_cols, float *out, int64_t ld, void *stream);

/* dgl.batch(samples) (gae_dgl/train_inductive.py:31-35) on a device-resident
 * dataset: the whole molecule set lives in HBM as ONE block-diagonal CSR (+ the CSR of A^T unless the set is
 * symmetric) and one feature matrix; a batch is assembled by two launches and no host <-> device copy.
 *   dataset: graph_ptr[n_total_graphs+1] (int64 node offsets), ds_indptr/ds_indices (CSR over all dataset nodes,
 *            column ids GLOBAL dataset node ids), ds_feat [n_total_nodes, F] (ld_feat) fp32 / bf16 / uint8
 *            (GAE_U8: the 0/1 atom one-hots of gae_dgl/prepare_data.py:31-36 stored in a byte each)
 *   graph_ids[n_graphs] (int64, device): the selected graphs, in batch order
 * gae_batch_plan  : out_node_ptr / out_edge_ptr / out_t_edge_ptr [n_graphs+1] (int64, device) = exclusive prefix
 *                   sums of the selected graphs' node / edge / transposed-edge counts ("node ids offset by the
 *                   prefix sum of node counts"); ds_t_indptr and out_t_edge_ptr may be NULL.
 * gae_batch_gather: out_indptr[N_b+1], out_indices[E_b] (ids rebased into the batch), out_feat [N_b, F] (ld_out;
 *                   same dtype as ds_feat, fp32 for GAE_U8; may be NULL: structure only), and -- optional, out_ell
 *                   != NULL -- the packed neighbour table of the batch CSR (gae_spmm_ell_build's format,
 *                   ell_width 4 / 8 / 16) written in the same pass.  N_b / E_b: the totals out_node_ptr[n_graphs] /
 *                   out_edge_ptr[n_graphs], which the caller also knows from its host copy of the per-graph sizes.
 *
 * Fixed-capacity batches (HIP-graph capture of the inductive step: every replay must launch the same shapes):
 * gae_batch_gather with cap_nodes > 0 pads the batch to cap_nodes rows -- the rows behind the last member graph
 * become isolated nodes (empty CSR rows, zero features, empty table rows) -- and writes the true sizes {nodes,
 * edges} to out_counts[0..1] (int64[3], device), which gae_decoder_bce_padded reads.  In that mode n_batch_nodes /
 * n_batch_edges are upper bounds (the capacity of the output arrays), not the exact totals, and the kernel guards
 * them ON THE DEVICE: of a batch whose prefix sums exceed cap_nodes rows or n_batch_edges edges only the longest
 * prefix of member graphs that fits is gathered (nothing is ever written behind the arrays; the sizes in
 * out_counts[0..1] are those of the kept prefix) and the number of graphs left out is ADDED to out_counts[2] --
 * zero it before an epoch and read it afterwards (gae_dgl_amd/capture.py raises when it is not 0).
 * gae_batch_select: out_ids[b] = order[*cursor_dev * batch_graphs + b], then *cursor_dev += 1 (device-side cursor:
 * a replayed graph walks an epoch order that was uploaded once). */
int gae_batch_select(const int64_t *order, int64_t n_order, int64_t *cursor_dev, int64_t batch_graphs,
                     int64_t *out_ids, void *stream);
int gae_batch_plan(const int64_t *graph_ptr, const int32_t *ds_indptr, const int32_t *ds_t_indptr,
                   const int64_t *graph_ids, int64_t n_graphs, int64_t *out_node_ptr, int64_t *out_edge_ptr,
                   int64_t *out_t_edge_ptr, void *stream);
int gae_batch_gather(const int64_t *graph_ptr, const int32_t *ds_indptr, const int32_t *ds_indices,
                     const void *ds_feat, int64_t ld_feat, int64_t F, int dtype,
                     const int64_t *graph_ids, int64_t n_graphs,
                     const int64_t *out_node_ptr, const int64_t *out_edge_ptr,
                     int64_t n_batch_nodes, int64_t n_batch_edges,
                     int32_t *out_indptr, int32_t *out_indices,
                     void *out_feat, int64_t ld_out, int32_t *out_ell, int32_t ell_width,
                     int64_t cap_nodes, int64_t *out_counts, void *stream);

/* ---- K1/K2: sparse aggregation ---------------------------------------------
 * M = diag(row_scale) * A * diag(col_scale) * H,  A given as CSR.
 * Forward  = g.update_all(copy_src('h','m'), sum('m','h'))   gae_dgl/gae.py:18-19,28
 * Backward = its autograd (gae_dgl/train_inductive.py:51): same call on the CSR of A^T.
 * row_scale / col_scale (fp32, length n_rows / n_cols) may be NULL (= the
 * reference's un-normalised sum); passing norm for both gives D^-1/2 A D^-1/2
 * (gae_dgl/train_transductive.py:55-57).
 * H: [n_cols, F] ld ldh, M: [n_rows, F] ld ldm, dtype GAE_F32 or GAE_BF16
 * (same dtype in and out, fp32 accumulate, no float atomics: deterministic).
 *
 * Degree-skew plan (optional, for power-law graphs): rows with more than
 * `threshold` edges are cut into segments of `segment_edges` edges that
 * separate waves gather; segment partial sums are added in segment order.
 * Build: gae_spmm_plan_sizes -> allocate -> gae_spmm_plan_build_rows (-> gae_spmm_plan_build_pinned), all on the
 * device; the plan is valid as long as indptr is.  plan = NULL (or n_heavy = 0): every row is summed by one lane
 * group in CSR order. */
typedef struct gae_spmm_plan {
    int32_t threshold;              /* rows with degree > threshold are heavy */
    int32_t segment_edges;          /* multiple of 64 */
    int64_t n_heavy;                /* heavy rows */
    int64_t n_segments;             /* sum over heavy rows of ceil(degree / segment_edges) */
    const int32_t *heavy_rows;      /* [n_heavy]    row ids                       (device) */
    const int32_t *heavy_seg_base;  /* [n_heavy]    first segment of the row      (device) */
    const int32_t *seg_heavy;       /* [n_segments] heavy-row slot of the segment (device) */
    const int32_t *ell;             /* [n_rows * ell_width] packed neighbour table (device) or NULL */
    int32_t ell_width;              /* 4, 8 or GAE_SPMM_ELL_WIDTH when `ell` is given, else 0 */
    int32_t reserved;
    const int32_t *hot_indices;     /* [n_edges] the CSR's column ids with the sign bit set on the most gathered
                                       columns, or NULL: the heavy-row kernel then loads the
                                       rows of all OTHER columns with the streaming hint, which keeps the hub
                                       rows of a power-law graph in L2.  A cache hint only: values unchanged. */
    /* XCD-pinned ("homed") part, optional (vh_n_virtual = 0: none).  The rows in vh_rows -- the very long rows of a
     * power-law graph; they must NOT appear in heavy_rows -- are evaluated from a virtual CSR: virtual row p holds
     * the <= segment_edges column ids (tagged like hot_indices, optional) of one (row, home, chunk) group, where
     * home(column) in 0..7 is any fixed hash, and p is ordered so that (p / 4) % 8 == home: thread block p / 4 runs
     * on XCD (p / 4) % 8, so every column is gathered through ONE of the eight private L2s and their capacities add
     * up (RMAT s24: 3.6 -> 2.6 ms for the rows with more than 256 edges).  Empty virtual rows pad the lists.  The
     * partials of row vh_rows[r] are the virtual rows vh_part_pos[vh_part_ptr[r] .. vh_part_ptr[r + 1]), added in
     * that order.  Changes the summation order of those rows (fp32 rounding), never the set of terms. */
    int64_t vh_n_rows, vh_n_virtual;
    const int32_t *vh_rows;         /* [vh_n_rows] */
    const int32_t *vh_indptr;       /* [vh_n_virtual + 1] */
    const int32_t *vh_indices;      /* [vh_indptr[vh_n_virtual]] column ids in virtual-row order */
    const int32_t *vh_hot_indices;  /* the same with hot tags, or NULL */
    const int32_t *vh_identity;     /* [vh_n_virtual] 0, 1, 2, ... */
    const int32_t *vh_part_ptr;     /* [vh_n_rows + 1] */
    const int32_t *vh_part_pos;     /* [vh_part_ptr[vh_n_rows]] */
    const int32_t *seg_desc;        /* [n_segments][4] {row, first edge, end edge, 1 = the row's only segment}
                                       (16-byte aligned) or NULL: one load in front of a
                                       segment's column ids instead of the chain seg_heavy -> heavy_rows /
                                       heavy_seg_base -> indptr (a wave of the heavy-row kernel lives for a handful
                                       of round trips: RMAT s24 launch 5.05 -> 4.85 ms).  Same sums. */
    /* Light-row list of a skew plan, optional (round 4): {row, first edge, end edge, 0} of every row with 1 ..
     * threshold in-edges, ascending rows (gae_spmm_plan_build_rows; 16-byte aligned).  On a power-law graph most rows are
     * EMPTY (R-MAT s24: 11.7 M of 16.8 M): with the list the light rows are produced by waves whose every lane group
     * has edges to gather (one descriptor load instead of two row-pointer loads in front of the column ids), and the
     * empty rows by a pure stream that writes act(bias).  Same sums, same bits. */
    const int32_t *light_desc;
    int64_t n_light;
    /* Device-built plans (gae_spmm_plan_build_rows / _pinned, round 4): */
    const int32_t *mid_indices;     /* compact copy of the column ids of the rows in heavy_rows, or NULL.  When given,
                                       seg_desc (mandatory then) indexes THIS array instead of the CSR's `indices`; */
    int32_t mid_tagged;             /* 1: its ids carry hot-column tags (sign bit), like hot_indices */
    int32_t reserved2;              /* 1: skip_rows marks EVERY row without edges (the empty-row stream is not launched) */
    const int32_t *vh_desc;         /* [vh_n_virtual][4] {p, first, end, 0} into vh_indices for virtual row p (empty
                                       positions: zeros), or NULL.  When given, vh_indptr / vh_identity are not read and
                                       vh_indices holds the pinned rows' ids in (row, home, column) order. */
    const uint8_t *skip_rows;       /* [n_rows] or NULL; with GAE_SPMM_SKIP_ROWS: rows r WITHOUT edges whose
                                       skip_rows[r] != 0 are not written at all (see the flag) */
} gae_spmm_plan;

/* A binder of THIS header alone passes plan == NULL everywhere (every product is then the plain CSR-order launch: correct
 * on any graph, the fastest form only for graphs without long rows).  Filling a gae_spmm_plan -- the packed neighbour table
 * `ell`, the heavy-row segments, the XCD-pinned regrouping -- takes the device-side builders gae_spmm_plan_sizes /
 * _build_rows / _build_pinned / gae_spmm_ell_build, which are declared in gae_hip_experimental.h: plans REQUIRE that header. */
/* Packed neighbour table (optional, for launches of a few 10 MB): slot k of row r at ell[r * width + k] holds the
 * row's k-th column id in CSR order; -1 = empty; a row with more than `width` ids keeps width - 1 of them and the
 * marker -2 in its last slot (the kernel continues from indptr / indices); rows with more than `skip_degree` ids
 * (the plan's threshold when it has heavy rows, else INT32_MAX) hold the marker -3 in slot 0.  With the table ONE
 * load replaces the dependent indptr -> indices chain in front of the gather; short launches are bounded by that
 * chain, not by bytes (Pubmed F = 500: 21 -> 15 us).  Rows that fit the table: same summation order, bit-identical
 * results.  A row that continues from indptr / indices (marker -2) is gathered by the whole wave, 64 ids per trip:
 * the same terms in another (fixed) order -- equal to the CSR-order sum up to fp32 rounding.  The table
 * costs width * 4 bytes per row of extra traffic: leave plan->ell NULL for graphs of millions of rows. */
#define GAE_SPMM_ELL_WIDTH 16
/* bytes of workspace gae_spmm_csr needs with this plan (0 without one) */
int64_t gae_spmm_workspace_bytes(const gae_spmm_plan *plan_host, int64_t F);

/* flags */
#define GAE_SPMM_STORE_PAD 1 /* M's rows are padded to whole 16-byte vectors (ldm >= roundup(F)) and the caller
                              * allows the pad columns [F, roundup(F)) to be overwritten: the tail vector of every
                              * row is stored whole (F = 39: 575 -> 443 us on the ZINC set; partially written
                              * 32-byte sectors are expensive) */
#define GAE_SPMM_TILE 2      /* the graph's gathers have poor locality (column ids far from the row id) and H is wider
                              * than one lane group: when H / M rows are whole 128-byte lines (ld * elem % 128 == 0,
                              * 128-byte aligned bases) XCD x sweeps feature tiles x, x + 8, ... so that a tile's
                              * slice of H is gathered from that XCD's own L2 instead of HBM (Pubmed F = 500: HBM
                              * traffic 170 -> 45 MB, 31 -> 19 us).  Hurts graphs with local neighbourhoods. */
#define GAE_SPMM_ACCUMULATE 4 /* M += diag(rs) A diag(cs) H instead of M = ...: lets a row-sharded product add the
                              * contribution of the REMOTE columns (after the exchange has landed) to the product
                              * of the rank's OWN columns, which ran while the exchange was in flight */
int gae_spmm_csr(const int32_t *indptr, const int32_t *indices, int64_t n_rows, int64_t n_cols,
                 const void *H, int64_t ldh, void *M, int64_t ldm, int64_t F, int dtype,
                 const float *row_scale, const float *col_scale,
                 const gae_spmm_plan *plan_host, void *workspace, int64_t workspace_bytes, int flags, void *stream);
#define GAE_SPMM_SKIP_ROWS 8  /* plans with a light-row list only: the stream that writes act(bias) to the rows without
                              * edges leaves out the rows marked in plan->skip_rows -- rows the caller KNOWS its
                              * consumers treat as zero without reading them (gae_linear2_fwd / gae_gcn2_bwd_dense take
                              * the same mask).  On R-MAT s24 70 % of the rows have no in-edges: their 1.5 GB of zeros are
                              * then neither written nor read back.  Ignored by every other kernel (they write all rows). */

/* GCN.forward (gae_dgl/gae.py:26-31) in ONE launch for narrow layers: the aggregation of gae_spmm_csr followed by
 * NodeApplyModule (gae.py:13-16), Y = act(M W^T + b), computed from the aggregated row while it is still in
 * registers.  F <= 64 input features, J <= 32 outputs, fp32, H rows of whole 16-byte vectors; the plan must carry a
 * packed neighbour table and no heavy rows.  W is addressed as W[o * w_stride_out + k * w_stride_in] (o < J, k < F):
 * (ld, 1) for nn.Linear's [J][F] weight, (1, ld) for its transpose -- the backward of an identity-activation layer
 * is the same launch on the CSR of A^T: dH = (A^T dY) W.  M (may be NULL) receives the aggregate the backward's
 * dW = dY^T M needs.  Sums run in CSR order per row (rows beyond the table: see the table's note); Y differs from the two-launch chain by fp32 rounding only
 * (the 4 + LPR-lane tree of the epilogue instead of the k-ordered chain of gae_linear_fwd). */
int gae_gcn_layer_fused(const int32_t *indptr, const int32_t *indices, int64_t n_rows, int64_t n_cols,
                        const float *H, int64_t ldh, float *M, int64_t ldm, int64_t F,
                        const float *row_scale, const float *col_scale, const gae_spmm_plan *plan,
                        const float *W, int64_t w_stride_out, int64_t w_stride_in, const float *bias,
                        int64_t J, int act, float *Y, int64_t ldy, void *stream);

/* ---- transform-first GCN layer ----------------------------------------------
 * GCN.forward (gae_dgl/gae.py:26-31) computes act((A H) W^T + b): update_all at the INPUT width, then the Linear.
 * For a layer that narrows the features (layer 1: 500 / 1433 / 3703 -> 32) the same value -- up to fp32 rounding,
 * measured 2e-7 of the scale -- is act(A (H W^T) + b): one dense pass over H at full HBM rate, the aggregation at
 * the OUTPUT width, nothing of width f_in written (the reference's order writes M = A H, 39 MB on Pubmed, and reads
 * it twice).  Three launches forward, three backward:
 *   forward   P = H W^T                         gae_xw_fwd        (W stationary in registers, H read once)
 *             Y = act(rs A cs P + b)            gae_spmm_csr_epilogue   (bias + activation at store time)
 *   backward  G = rs A^T cs (dY (.) [Y > 0])    gae_spmm_csr_epilogue on the CSR of A^T, Hmask = Y
 *             dW = G^T H, db = colsum(dY (.) [Y > 0])            gae_xw_wgrad   (H read once)
 * gae_xw_fwd / gae_xw_wgrad take fp32 or bf16-stored H (dtype GAE_F32 / GAE_BF16; bf16 rows feed
 * v_mfma_f32_16x16x32_bf16 directly in the forward, W split into hi + lo bf16 fragments, fp32 accumulation) with
 * f_in >= 193, f_out <= 32, rows of whole 16-byte vectors, H below 3.5 GiB (gae_xw_usable says whether a call is
 * accepted; other shapes: gae_linear_fwd / gae_linear_bwd, which use the same kernels where they apply).
 *   gae_xw_fwd:   P [n, f_out] (ldp) = act(X W^T + b); W [f_out, f_in] (ldw), b may be NULL.  workspace:
 *                 gae_xw_fwd_workspace_bytes (0 unless f_in is split over thread blocks: f_in > 1024 fp32 / 2048 bf16).
 *   gae_xw_wgrad: dW [f_out, f_in] (lddw) = (G (.) [Gmask > 0])^T X   (Gmask may be NULL; dW may be NULL)
 *                 db [f_out] = colsum(D (.) [Dmask > 0])              (db / D / Dmask may be NULL)
 *                 partial sums per (row partition, column slice) are added in partition order: deterministic.
 *   gae_spmm_csr_epilogue: Y = act(rs A cs (H (.) [Hmask > 0]) + bias); Hmask (may be NULL) has the layout of H;
 *                 F <= 64, fp32, a plan with a packed neighbour table and no heavy / XCD-pinned rows; CSR-order sums
 *                 (rows beyond the table: the same terms, whole-wave order).
 *                 n_splits > 1: H is the first of n_splits partial matrices split_stride floats apart -- what
 *                 gae_xw_fwd leaves with keep_splits = 1 when it splits a long f_in over thread blocks -- and a gathered
 *                 row is the sum of its partial rows in split order: the value the split reduction would have stored,
 *                 without that launch (Cora / Citeseer: one kernel node less per step).
 *   gae_xw_fwd_splits: the number of partial matrices gae_xw_fwd computes for these sizes (1: none). */
int gae_xw_usable(const void *X, int64_t ldx, int dtype, int64_t n, int64_t f_in, int64_t f_out);
int64_t gae_xw_fwd_workspace_bytes(int64_t n, int64_t f_in, int64_t f_out, int dtype);
/* keep_splits = 1 (b = NULL, act = identity, gae_xw_fwd_splits(...) > 1): the partial products [splits][n][f_out] stay
 * in `workspace` for a consumer that adds them itself (gae_spmm_csr_epilogue); P is not written. */
int gae_xw_fwd(const void *X, int64_t ldx, int dtype, int64_t n, int64_t f_in,
               const float *W, int64_t ldw, const float *b, int64_t f_out, int act,
               float *P, int64_t ldp, void *workspace, int64_t workspace_bytes, int keep_splits, void *stream);
int64_t gae_xw_wgrad_workspace_bytes(int64_t n, int64_t f_in, int dtype);
int gae_xw_wgrad(const void *X, int64_t ldx, int dtype, int64_t n, int64_t f_in,
                 const float *G, int64_t ldg, const float *Gmask, int64_t ldgm,
                 const float *D, int64_t ldd, const float *Dmask, int64_t lddm, int64_t f_out,
                 float *dW, int64_t lddw, float *db, void *workspace, int64_t workspace_bytes, void *stream);

/* gae_spmm_csr with a store-time epilogue, for ANY plan (degree-skew segments and XCD-pinned rows included):
 *     M = act(diag(rs) A diag(cs) H (+ M, flag GAE_SPMM_ACCUMULATE) + bias)          fp32; bias [F] may be NULL
 * The sparse half of a GCN layer evaluated as act(A (H W^T) + b) -- the value of gae_dgl/gae.py:26-31 up to fp32
 * rounding -- on power-law graphs, where gae_spmm_csr_epilogue (packed-table plans) does not apply: R-MAT's 32 -> 16
 * layer aggregates 16 instead of 32 floats per edge.  Same arguments, workspace and summation orders as gae_spmm_csr
 * (GAE_SPMM_TILE is not accepted). */
int gae_spmm_csr_ep(const int32_t *indptr, const int32_t *indices, int64_t n_rows, int64_t n_cols,
                    const float *H, int64_t ldh, float *M, int64_t ldm, int64_t F, const float *row_scale,
                    const float *col_scale, const gae_spmm_plan *plan, void *workspace, int64_t workspace_bytes,
                    int flags, const float *bias, int act, void *stream);

/* (dense halves of the two-layer encoder on very tall operands -- gae_linear2_fwd, gae_linear2_fill_dead,
 * gae_gcn2_bwd_dense -- and layer 1 on sparse input features -- gae_dense_to_csr_*, gae_spx_*: declared and documented in
 * gae_hip_experimental.h) */

/* ---- K3-K5: node-apply (Linear + activation) -------------------------------
 * Y = act(M W^T + b)      NodeApplyModule.forward, gae_dgl/gae.py:13-16
 * M [n, f_in] (ldm), W [f_out, f_in] row-major contiguous (nn.Linear.weight,
 * gae_dgl/gae.py:10), b [f_out] (may be NULL), Y [n, f_out] (ldy). fp32.
 * workspace (optional, gae_linear_fwd_workspace_bytes; 0 bytes for most shapes): operands with few rows and a
 * long f_in (Cora 2708 x 1433, Citeseer 3327 x 3703) are split along f_in over more thread blocks, whose partial
 * products meet in the workspace in fixed order; without it the product runs unsplit. */
int64_t gae_linear_fwd_workspace_bytes(int64_t n, int64_t f_in, int64_t f_out);
int gae_linear_fwd(const float *M, int64_t ldm, int64_t n, int64_t f_in,
                   const float *W, const float *b, int64_t f_out, int act,
                   float *Y, int64_t ldy, void *workspace, int64_t workspace_bytes, void *stream);

/* autograd of the above.  dYm = dY (.) [Y > 0] when act = RELU.
 *   dW [f_out, f_in] = dYm^T M      (NULL to skip)
 *   db [f_out]       = colsum(dYm)  (NULL to skip)
 *   dM [n, f_in]     = dYm W        (NULL to skip; layer 1 never needs it)
 * workspace: gae_linear_bwd_workspace_bytes(n, f_in, f_out). */
int64_t gae_linear_bwd_workspace_bytes(int64_t n, int64_t f_in, int64_t f_out);
int gae_linear_bwd(const float *dY, int64_t lddy, const float *Y, int64_t ldy, int act,
                   const float *M, int64_t ldm, const float *W,
                   int64_t n, int64_t f_in, int64_t f_out,
                   float *dW, float *db, float *dM, int64_t lddm,
                   void *workspace, int64_t workspace_bytes, void *stream);

/* ---- K6/K7: inner-product decoder ------------------------------------------
 * InnerProductDecoder.forward, gae_dgl/gae.py:69-72 with the identity
 * activation GAE passes (gae_dgl/gae.py:47).
 * gae_dropout_mask: inverted-dropout multiplier (0 or 1/(1-p)), Philox4x32-10
 * counter RNG keyed by (seed, offset + element index): reproducible for bwd.
 * draw_dev (device uint64, may be NULL): number of masks drawn so far.  The
 * Philox counter of element block q is (offset + q) in its low 64 bits and
 * *draw_dev in its high 64 bits: every draw is a stream of its own, whatever the
 * sizes of earlier draws, and a captured HIP graph draws a fresh mask on every
 * replay once the caller bumps the counter.
 * gae_decoder_dense: out = Zt Zt^T with Zt = Z (.) mask (mask may be NULL). */
int gae_dropout_mask(float *mask, int64_t n_elems, float p, uint64_t seed, uint64_t offset,
                     const uint64_t *draw_dev, void *stream);
int gae_decoder_dense(const float *Z, const float *mask, int64_t ldz, int64_t n, int64_t d,
                      float *out, int64_t ldo, void *stream);
/* dZ = ((G + G^T) Zt) (.) mask,  G = dL/dlogits [n, n]  (autograd of gae.py:70-71) */
int64_t gae_decoder_dense_bwd_workspace_bytes(int64_t n, int64_t d);
int gae_decoder_dense_bwd(const float *G, int64_t ldg, const float *Z, const float *mask,
                          int64_t ldz, int64_t n, int64_t d, float *dZ, int64_t lddz,
                          void *workspace, int64_t workspace_bytes, void *stream);

/* ---- VGAE head (BASELINE config 5; not in the reference: README.md:58 only cites Kipf & Welling 2016) ----
 * gae_normal_noise : eps ~ N(0,1), Philox4x32-10 + Box-Muller, same (seed, offset, draw_dev) contract as
 *                    gae_dropout_mask.
 * gae_vgae_head_fwd: z = mu + eps * exp(logstd);  kl_out = -(0.5/N) * mean_i sum_j (1 + 2 logstd - mu^2 -
 *                    exp(2 logstd))   (one fp32 on the device); eps / z contiguous [n, d]; the rows of mu and of
 *                    logstd (and of their gradients in gae_vgae_head_bwd) are ldm floats apart -- ldm = d for
 *                    separate matrices, 2 d when both heads come packed as [mu | logstd] from gae_x_gcn_layer_fused2.
 * gae_vgae_head_bwd: dmu = dz + gkl * dKL/dmu, dlogstd = dz * eps * exp(logstd) + gkl * dKL/dlogstd, with
 *                    gkl = *gkl_dev (upstream gradient of the KL scalar; NULL = 1), dz may be NULL (= 0). */
int gae_normal_noise(float *out, int64_t n_elems, uint64_t seed, uint64_t offset, const uint64_t *draw_dev,
                     void *stream);
int64_t gae_vgae_head_workspace_bytes(int64_t n_elems);
int gae_vgae_head_fwd(const float *mu, const float *logstd, int64_t ldm, const float *eps, int64_t n, int64_t d,
                      float *z, float *kl_out, void *workspace, int64_t workspace_bytes, void *stream);
int gae_vgae_head_bwd(const float *dz, const float *mu, const float *logstd, int64_t ldm, const float *eps,
                      const float *gkl_dev, int64_t n, int64_t d, float *dmu, float *dlogstd, void *stream);

/* ---- K7+K8+K9 fused: decoder + weighted BCE-with-logits, never materialising N x N
 * Replaces, for training, gae_dgl/train_inductive.py:44-51:
 *   adj = g.adjacency_matrix().to_dense(); pos_weight = (N^2 - sum(adj)) / sum(adj)
 *   loss = binary_cross_entropy_with_logits(model.forward(g), adj, pos_weight=pos_weight); loss.backward()
 * loss = mean_ij [(1-y) x + (1 + (pw-1) y) softplus(-x)],  x = (Zt Zt^T)_ij, Zt = Z (.) mask,
 * y_ij = #edges j->i read from the CSR (rows = destination; duplicates count).
 *   loss_out : 1 fp32 on the device
 *   dZ       : d loss / d Z  [n, d] (NULL = loss only, e.g. validation); needs the
 *              CSR of A^T (t_indptr, t_indices) for the G^T term
 *   d <= 64.  Ordered two-stage reductions: deterministic.
 * Dropout (gae_dgl/gae.py:70, always on in the reference):
 *   dropout_p == 0 : `mask` [n, d] (ld = ldz) is an optional INPUT (NULL = no dropout), e.g. one drawn by
 *                    gae_dropout_mask.
 *   dropout_p  > 0 : `mask` is an OUTPUT: this draw's multipliers -- the same Philox stream as
 *                    gae_dropout_mask(mask, n * d, p, seed, offset, draw_dev) -- are generated inside the call,
 *                    applied, and stored for the caller; at the end of the call *draw_dev (device counter, may be
 *                    NULL) is incremented by one, stream-ordered, so a replayed HIP graph draws a fresh mask
 *                    every time without a separate launch. */
int64_t gae_decoder_bce_workspace_bytes(int64_t n, int64_t n_local, int64_t d);
int gae_decoder_bce(const float *Z, float *mask, int64_t ldz, int64_t n, int64_t d,
                    const int32_t *indptr, const int32_t *indices,
                    const int32_t *t_indptr, const int32_t *t_indices, float pos_weight,
                    float dropout_p, uint64_t seed, uint64_t offset, uint64_t *draw_dev,
                    float *loss_out, float *dZ, int64_t lddz,
                    void *workspace, int64_t workspace_bytes, void *stream);

/* Row-sharded form (one rank of a 1-D row partition): rows
 * [row_begin, row_begin + n_local) of the N x N loss against ALL n columns.
 * Z / mask are the full (all-gathered) [n, d] arrays; indptr / t_indptr are the
 * rank's LOCAL row blocks of A and A^T (n_local + 1 entries, global column ids);
 * loss_out receives this block's share of the mean (sum over ranks = the loss);
 * dZ [n_local, d] is the gradient of the GLOBAL loss w.r.t. the local rows. */
int gae_decoder_bce_rows(const float *Z, float *mask, int64_t ldz, int64_t n, int64_t d,
                         int64_t row_begin, int64_t n_local,
                         const int32_t *indptr, const int32_t *indices,
                         const int32_t *t_indptr, const int32_t *t_indices, float pos_weight,
                         float dropout_p, uint64_t seed, uint64_t offset, uint64_t *draw_dev,
                         float *loss_out, float *dZ, int64_t lddz,
                         void *workspace, int64_t workspace_bytes, void *stream);

/* The same loss on a FIXED-CAPACITY batch (gae_batch_gather with cap_nodes > 0): Z / mask / dZ have n_cap rows, the
 * CSRs n_cap rows; counts_dev (int64[>= 2], device) holds the true {nodes, edges} of this batch.  Rows >= counts[0]
 * are padding: they take no part in the loss (pos_weight and the mean use the true N and E, read on the device)
 * and receive a zero gradient.  Lets the inductive training step of gae_dgl/train_inductive.py:92-95 run as ONE
 * captured HIP graph although every batch has a different size.  Workspace: gae_decoder_bce_workspace_bytes(n_cap,
 * n_cap, d). */
int gae_decoder_bce_padded(const float *Z, float *mask, int64_t ldz, int64_t n_cap, int64_t d,
                           const int32_t *indptr, const int32_t *indices,
                           const int32_t *t_indptr, const int32_t *t_indices, const int64_t *counts_dev,
                           float dropout_p, uint64_t seed, uint64_t offset, uint64_t *draw_dev,
                           float *loss_out, float *dZ, int64_t lddz,
                           void *workspace, int64_t workspace_bytes, void *stream);

/* Prepare step folded into the PRODUCER of Z.  gae_decoder_bce* start with a small launch that applies the dropout
 * mask to Z, pads it to 16 columns, splits it into bf16 hi / lo and adds up its columns.  When Z comes out of
 * gae_gcn_layer_fused (the last encoder layer of gae_dgl/gae.py:55-57 followed by the loss of
 * train_inductive.py:44-48), that launch can do the same work in its epilogue:
 *   gae_x_decoder_bce_prep_layout(n, d, ws, bytes, &prep)   where the pieces go inside the loss workspace `ws`
 *                                                         (gae_decoder_bce_workspace_bytes(n, n, d) bytes), d <= 16;
 *   gae_x_gcn_layer_fused_prep(..., &prep, mask, ...)       the layer + the prepare work (see there);
 *   gae_x_decoder_bce_prepared(..., n_prep_blocks, ..., ws) the loss from the dense kernel on: same arguments as
 *                                                         gae_decoder_bce / _padded (counts_dev != NULL: padded batch,
 *                                                         pos_weight ignored) minus Z / seed / offset.
 * One kernel node fewer per training step; the values are those of the three-launch form up to the order of the
 * column sums (fp64). */
typedef struct gae_bce_prep {
    float *Zt;                   /* [n][DP] fp32                                  (device, inside the workspace) */
    uint16_t *Zhi, *Zlo;         /* [n][DP] bf16 hi / lo */
    double *colsum_partial;      /* [blocks][2][DP] */
    double *scal;                /* [3]: pos_weight, 1 / N^2, pad pairs of a padded batch */
    double all_pairs;            /* pairs the dense kernel evaluates (for scal[2]) */
    int64_t max_blocks;          /* room in colsum_partial */
    int32_t DP, reserved;
} gae_bce_prep;

/* Deferred final reduction.  The last launch of gae_decoder_bce* adds the per-block partial sums to the scalar; the
 * backward pass does not read that scalar, so a training step may run the reduction later, next to other work:
 *   gae_x_decoder_bce_defer_finalize(&tail)  arms the calling thread: its NEXT gae_decoder_bce / _rows / _padded call
 *                                          launches everything but the reduction and describes it in `tail` (pointers
 *                                          into that call's workspace and outputs: keep them alive); NULL disarms.
 *   gae_x_adam_step_tail(..., &tail, stream) runs it as one extra block of the optimiser launch (same bits as below),
 *   gae_x_decoder_bce_finalize(&tail, stream) as a launch of its own (e.g. no optimiser step follows).
 * Until one of the two has run on the stream, loss_out is not written and draw_dev not advanced.  A tail with
 * loss_out == NULL (written for an empty row window) is a no-op. */
typedef struct gae_bce_tail {
    const double *dense_partial; int64_t n_dense;     /* {sum |x|, sum log2 t} per dense block          (device) */
    const double *edge_partial;  int64_t n_edge;      /* sparse terms per edge block                    (device) */
    const double *S;                                  /* column sums of Zt: all rows | row window [2][DP] (device) */
    int32_t DP, reserved;
    double pad_terms, inv_n2;
    float *loss_out;                                  /* (device) */
    uint64_t *bump_draw;                              /* draw counter to advance, or NULL               (device) */
    const double *scal;                               /* device-side {pos_weight, 1 / N^2, pad pairs} of a padded batch, or NULL */
    /* optional additive term (set by the caller after the loss call filled the rest; all zero = none): the block also
     * adds kl_partial[0 .. n_kl) (doubles), scales the sum, and writes loss_out = rec + kl, kl_out, rec_out -- the KL
     * term of a VGAE (gae_x_vgae_head_prep) without launches of its own */
    const double *kl_partial; int64_t n_kl; double kl_scale;
    float *kl_out, *rec_out;                          /* may be NULL */
} gae_bce_tail;

/* Reference-shaped loss on MATERIALISED logits: F.binary_cross_entropy_with_logits(adj_logits, adj,
 * pos_weight=pos_weight) with the default mean reduction (gae_dgl/train_inductive.py:48) and dLoss/dLogits.  Used
 * for embedding widths the fused kernel does not take (d > 64; gae_dgl/optuna_gae.py:29-34 samples hidden dims up
 * to 256) together with gae_decoder_dense / gae_csr_to_dense / gae_decoder_dense_bwd.  logits / labels / grad:
 * [n_rows, n_cols] fp32 with their leading dimensions; grad may be NULL (loss only) and may alias logits.
 * workspace: gae_bce_logits_workspace_bytes() bytes.  Ordered fp64 reduction: deterministic. */
int64_t gae_bce_logits_workspace_bytes(void);
int gae_bce_logits(const float *logits, int64_t ldx, const float *labels, int64_t ldy, int64_t n_rows, int64_t n_cols,
                   float pos_weight, float *loss_out, float *grad, int64_t ldg, void *workspace,
                   int64_t workspace_bytes, void *stream);

/* ---- graph-level readout ------------------------------------------------------
 * out[g] = [ mean | sum | max ] over the nodes graph_ptr[g] .. graph_ptr[g+1] of Z [n_nodes, d] (ldz): the 3 d
 * molecule feature the reference describes for its ESOL experiment (README.md:54; with DGL: mean_nodes /
 * sum_nodes / max_nodes over the batched graph of gae_dgl/train_inductive.py:34).  graph_ptr: int64 [n_graphs + 1]
 * node offsets (device), out [n_graphs, 3 d] (ldo), fp32.  Deterministic; an empty graph gives zeros. */
int gae_segment_readout(const float *Z, int64_t ldz, int64_t n_nodes, int64_t d, const int64_t *graph_ptr,
                        int64_t n_graphs, float *out, int64_t ldo, void *stream);

/* ---- K12: Adam ----------------------------------------------------------------
 * torch.optim.Adam(params, lr, betas, eps, weight_decay) of gae_dgl/train_inductive.py:40,50-52 and
 * train_transductive.py:43,66-68 (no amsgrad, no maximize) for up to GAE_ADAM_MAX_TENSORS contiguous fp32
 * tensors in ONE launch:
 *   g += weight_decay p;  m = lerp(m, g, 1 - beta1);  v = beta2 v + (1 - beta2) g^2
 *   p -= lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps),   t = state_dev[0] + 1
 * state_dev: GAE_ADAM_STATE_WORDS (6) x uint64 on the device, zeroed by the caller once: [0] = steps taken so far
 * (advanced by the call, stream-ordered, so a replayed HIP graph counts its own steps; may be preset to resume),
 * [1] = scratch ticket, [2..5] = the library's cache of beta1, beta1^steps, beta2, beta2^steps (doubles).
 * `tensors` is a HOST array (copied into the kernel arguments). */
#define GAE_ADAM_MAX_TENSORS 16
#define GAE_ADAM_STATE_WORDS 6
typedef struct gae_adam_tensor {
    float *param;          /* [n] updated in place            (device) */
    float *grad;           /* [n] read (n_partials = 0) or written (n_partials > 0)  (device) */
    float *exp_avg;        /* [n] first moment, in place       (device) */
    float *exp_avg_sq;     /* [n] second moment, in place      (device) */
    int64_t n;
    /* Deferred reduction (n_partials > 0): the gradient has not been added up yet -- it is the list of partial sums
     * gae_x_xw_wgrad_partials / gae_x_linear_bwd_partials left in their workspace,
     *     grad[e] = sum over q < n_partials, in order, of partials[q * partial_stride + (e / row_len) * row_pitch + e % row_len].
     * The kernel adds the list (deterministic order), WRITES the sum to grad[e] and applies the update: the separate
     * reduction launch of every weight gradient disappears from a training step (two kernel nodes of ~5 us each).
     * n_partials = 0: grad holds the gradient. */
    const float *partials;
    int64_t n_partials, partial_stride, row_len, row_pitch;
} gae_adam_tensor;
int gae_adam_step(const gae_adam_tensor *tensors_host, int32_t n_tensors, float lr, float beta1, float beta2,
                  float eps, float weight_decay, uint64_t *state_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GAE_HIP_H */
