// Layer 1 on SPARSE input features (opt-in: gae_dgl_amd.SparseFeatures).
//
// The citation datasets the reference trains on (gae_dgl/train_transductive.py:37-38: `torch.FloatTensor(data.features)`)
// are bag-of-words matrices: 1.3 % (Cora), 0.9 % (Citeseer), 10 % (Pubmed) of the entries are non-zero.  The reference
// stores and multiplies them densely; so does this library by default (gae_xw_fwd / gae_xw_wgrad stream the dense X
// once per direction).  A caller who hands the features over as SparseFeatures gets the same layer-1 VALUES --
// adding x * w for the non-zero x only; a skipped term is an exact +0 -- from the compressed rows:
//
//     forward   P [n, J]  = X W^T        P[i]     = sum over the non-zeros (k, x) of row i      of x * W[:, k]
//     backward  dW [J, K] = G^T X        dW[:, k] = sum over the non-zeros (i, x) of column k   of x * G[i]
//               db [J]    = colsum(D (.) [Dmask > 0])
//
// in the transform-first order of gae_xw_fwd's layer (Y = act(A P + b), G = A^T (dY (.) [Y > 0])): 8 bytes per
// non-zero instead of 4 bytes per entry move through HBM (Pubmed 7.9 MB instead of 39.4 MB per direction), the
// gathered operand (W^T: K x J floats; G: n x J floats) is small enough to live in LDS / L2.
//   * gae_dense_to_csr_count / _fill: the compressed rows of a dense matrix, columns ascending (= the order the
//     dense kernels would meet the same terms in), one wave per row, ballot + popcount; run once per feature matrix
//     for X and once for X^T.
//   * spx_fwd_kernel: a group of 8 lanes owns a row (lane l: outputs 4 l .. 4 l + 3); 8 (column, value) pairs per
//     coalesced load, broadcast by shuffles; W^T sits in LDS (staged once per block, K <= 1024 at J = 32) or, for
//     wider inputs, is read from global memory by output row (4 dwords per lane and non-zero; those matrices have
//     few non-zeros per row).
//   * spx_wgrad_kernel: the non-zeros of X^T are cut into segments of <= 256 entries of ONE feature column; a wave
//     per segment gathers rows of G (8 groups of 8 lanes, 4 segments' worth of rows in flight), the groups meet in a
//     fixed shuffle order, and the segment's 32 sums go to partial[segment slot][j][k] (a feature's last segment
//     zero-fills the slots behind its own, so every slot of every feature holds a number).  Extra blocks add D (.) mask
//     over row ranges (db partials).  Both are partial LISTS in the library's one summation order (gae::sum_partials):
//     gae_adam_step's deferred reduction or the stand-alone reduction launch finish them.
#include <string.h>

#include "common.h"

namespace {

using gae::v4f;
constexpr int kSeg = 64;             // non-zeros of X^T per lane group of the backward (sparse.SEGMENT)

// ---------------------------------------------------------------------------------------------------- builder
__global__ __launch_bounds__(256) void dense_row_nnz_kernel(const float *__restrict__ X, int64_t ldx, int64_t n, int64_t K,
                                                            int32_t *__restrict__ counts)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t(blockIdx.x) * 256 + threadIdx.x) >> 6;
    if (row >= n) return;
    int c = 0;
    for (int64_t k0 = 0; k0 < K; k0 += 64) {
        const int64_t k = k0 + lane;
        const bool nz = k < K && X[row * ldx + k] != 0.f;
        c += __popcll(__builtin_amdgcn_ballot_w64(nz));
    }
    if (lane == 0) counts[row] = c;
}

__global__ __launch_bounds__(256) void dense_row_fill_kernel(const float *__restrict__ X, int64_t ldx, int64_t n, int64_t K,
                                                             const int32_t *__restrict__ rowptr, int32_t *__restrict__ col,
                                                             float *__restrict__ val)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t(blockIdx.x) * 256 + threadIdx.x) >> 6;
    if (row >= n) return;
    int32_t at = rowptr[row];
    for (int64_t k0 = 0; k0 < K; k0 += 64) {
        const int64_t k = k0 + lane;
        const float x = k < K ? X[row * ldx + k] : 0.f;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(x != 0.f);
        if (x != 0.f) {
            const int pos = at + __popcll(m & ((1ull << lane) - 1ull));
            col[pos] = int32_t(k);
            val[pos] = x;
        }
        at += __popcll(m);
    }
}

// ---------------------------------------------------------------------------------------------------- forward
struct SpxFwdArgs {
    const int32_t *rowptr, *col;
    const float *val, *Wt;             // Wt [K][32]: the weight transposed (spx_transpose_kernel), rows of one 128-byte line
    float *P;
    int64_t n, ldp;
    int K, J;
};

// Wt[k][j] = W[j][k] (j < J; 0 for J <= j < 32): K x 32 floats -- the gather operand of the forward as rows of one line
__global__ __launch_bounds__(256) void spx_transpose_kernel(const float *__restrict__ W, int64_t ldw, int K, int J,
                                                            float *__restrict__ Wt)
{
    __shared__ float t[32][65];
    const int k0 = int(blockIdx.x) * 64;
    for (int idx = threadIdx.x; idx < 32 * 64; idx += 256) {          // coalesced over k
        const int j = idx >> 6, kk = idx & 63;
        t[j][kk] = (j < J && k0 + kk < K) ? W[int64_t(j) * ldw + k0 + kk] : 0.f;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 64 * 32; idx += 256) {          // coalesced over j
        const int kk = idx >> 5, j = idx & 31;
        if (k0 + kk < K) Wt[int64_t(k0 + kk) * 32 + j] = t[j][kk];
    }
}

// one group of 8 lanes per row of X (lane l: outputs 4 l .. 4 l + 3); 8 (column, value) pairs per coalesced load,
// broadcast by shuffles; the 8 rows of W^T they name are ALL requested (one 16-byte load per lane and non-zero, 8
// lanes = one line) before the first is used; terms are added in ascending column order
__global__ __launch_bounds__(256) void spx_fwd_kernel(const SpxFwdArgs a)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int lig = tid & 7, grp = tid >> 3;
    const int g0 = lane & ~7;
    const int64_t row = int64_t(blockIdx.x) * 32 + grp;
    if (row >= a.n) return;
    const int32_t e0 = a.rowptr[row], e1 = a.rowptr[row + 1];
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (int32_t e = e0; e < e1; e += 8) {
        const bool in = e + lig < e1;
        const int32_t c_mine = in ? a.col[e + lig] : 0;
        const float v_mine = in ? a.val[e + lig] : 0.f;              // entries behind the row: value 0 (an exact + 0)
        v4f w[8];
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = __shfl(c_mine, g0 + u, 64);
            v[u] = __shfl(v_mine, g0 + u, 64);
            w[u] = *reinterpret_cast<const v4f *>(a.Wt + int64_t(c) * 32 + 4 * lig);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = fmaf(v[u], w[u][q], acc[q]);
    }
    float *pp = a.P + row * a.ldp + 4 * lig;
    if (4 * lig + 4 <= a.J && (a.ldp & 3) == 0) *reinterpret_cast<v4f *>(pp) = acc;
    else
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (4 * lig + q < a.J) pp[q] = acc[q];
}

// ---------------------------------------------------------------------------------------------------- backward
struct SpxBwdArgs {
    const int32_t *t_rowptr, *t_row;   // compressed rows of X^T: feature k -> (node, value) pairs, nodes ascending
    const float *t_val;
    const int32_t *seg_feat, *seg_e0;  // segment s: feature seg_feat[s], entries [seg_e0[s], min(+kSeg, row end)), slot seg_slot
    const int32_t *seg_slot;
    const float *G, *D, *Dmask;
    float *part, *dbpart;              // part[slot][32][K], dbpart[block][32]
    int64_t n, ldg, ldd, lddm, rows_per_db_block;
    int K, J, n_segments, n_db_blocks, seg_blocks, n_slots;
};

__global__ __launch_bounds__(256) void spx_wgrad_kernel(const SpxBwdArgs a)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (int(blockIdx.x) >= a.seg_blocks) {
        // ---- db partial of a row range: thread (sub = tid / 32, j = tid % 32) adds rows r0 + sub + 8 i in order,
        //      the 8 subs meet in LDS in order
        __shared__ float dred[256];
        const int b = int(blockIdx.x) - a.seg_blocks;
        const int64_t r0 = int64_t(b) * a.rows_per_db_block, r1 = min(a.n, r0 + a.rows_per_db_block);
        const int sub = tid >> 5, j = tid & 31;
        float s = 0.f;
        for (int64_t rr = r0 + sub; rr < r1; rr += 8 * 8) {
            float dv[8], mv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t r = rr + 8 * u, rc = r < r1 ? r : r1 - 1;
                dv[u] = j < a.J ? a.D[rc * a.ldd + j] : 0.f;
                mv[u] = (a.Dmask != nullptr && j < a.J) ? a.Dmask[rc * a.lddm + j] : 1.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (rr + 8 * u < r1 && mv[u] > 0.f) ? dv[u] : 0.f;
        }
        dred[tid] = s;
        __syncthreads();
        if (tid < 32) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) t += dred[q * 32 + tid];
            a.dbpart[int64_t(b) * 32 + tid] = t;
        }
        return;
    }
    const int lig = lane & 7, grp = tid >> 3;                          // a group of 8 lanes per segment (lane l: outputs 4 l ..)
    const int g0 = lane & ~7;
    const int s = int(blockIdx.x) * 32 + grp;
    if (s >= a.n_segments) return;
    const int k = a.seg_feat[s];
    const int32_t e0 = a.seg_e0[s];
    const int32_t row_end = a.t_rowptr[k + 1];
    const int32_t e1 = min(e0 + kSeg, row_end);
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    // the segment's (node, value) pairs 8 at a time, the 8 rows of G they name all requested before the first is used;
    // terms are added in ascending node order
    for (int32_t e = e0; e < e1; e += 8) {
        const bool in = e + lig < e1;
        const int32_t r_mine = in ? a.t_row[e + lig] : 0;
        const float x_mine = in ? a.t_val[e + lig] : 0.f;
        v4f gv[8];
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = __shfl(r_mine, g0 + u, 64);
            x[u] = __shfl(x_mine, g0 + u, 64);
            const float *gp = a.G + int64_t(r) * a.ldg + 4 * lig;
            if (4 * lig + 4 <= a.J && (a.ldg & 3) == 0) gv[u] = *reinterpret_cast<const v4f *>(gp);
            else
#pragma unroll
                for (int q = 0; q < 4; ++q) gv[u][q] = 4 * lig + q < a.J ? gp[q] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = fmaf(x[u], gv[u][q], acc[q]);
    }
    (void)wave;
    {
        const int slot = a.seg_slot[s];
        float *pp = a.part + (int64_t(slot) * 32 + 4 * lig) * a.K + k;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (4 * lig + q < a.J) pp[int64_t(q) * a.K] = acc[q];
        // every (feature, slot) of the partial list must hold a number: the LAST segment of a feature with fewer
        // segments than the widest one zero-fills the slots behind it (the builder gives an empty feature one empty
        // segment, so slot 0 is always written)
        const int nnz = row_end - a.t_rowptr[k];
        int segs = (nnz + kSeg - 1) / kSeg;
        segs = segs < 1 ? 1 : segs;
        if (slot == segs - 1)
            for (int z = segs; z < a.n_slots; ++z)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (4 * lig + q < a.J) a.part[(int64_t(z) * 32 + 4 * lig + q) * a.K + k] = 0.f;
    }
}

} // namespace

// ---------------------------------------------------------------------------------------------------- C ABI
extern "C" int gae_dense_to_csr_count(const float *X, int64_t ldx, int64_t n, int64_t K, int32_t *row_nnz, void *stream)
{
    GAE_REQUIRE(n >= 0 && K >= 0 && ldx >= K, GAE_E_SIZE, "gae_dense_to_csr_count: bad sizes");
    if (n == 0) return GAE_OK;
    GAE_REQUIRE(X && row_nnz, GAE_E_NULL, "gae_dense_to_csr_count: NULL pointer");
    hipLaunchKernelGGL(dense_row_nnz_kernel, dim3(unsigned((n * 64 + 255) / 256)), dim3(256), 0, gae::as_stream(stream), X,
                       ldx, n, K, row_nnz);
    GAE_CHECK_LAUNCH("dense_row_nnz_kernel");
    return GAE_OK;
}

extern "C" int gae_dense_to_csr_fill(const float *X, int64_t ldx, int64_t n, int64_t K, const int32_t *rowptr, int32_t *col,
                                     float *val, void *stream)
{
    GAE_REQUIRE(n >= 0 && K >= 0 && ldx >= K, GAE_E_SIZE, "gae_dense_to_csr_fill: bad sizes");
    if (n == 0) return GAE_OK;
    GAE_REQUIRE(X && rowptr && (col && val), GAE_E_NULL, "gae_dense_to_csr_fill: NULL pointer");
    hipLaunchKernelGGL(dense_row_fill_kernel, dim3(unsigned((n * 64 + 255) / 256)), dim3(256), 0, gae::as_stream(stream), X,
                       ldx, n, K, rowptr, col, val);
    GAE_CHECK_LAUNCH("dense_row_fill_kernel");
    return GAE_OK;
}

extern "C" int64_t gae_spx_fwd_workspace_bytes(int64_t f_in) { return f_in < 1 ? GAE_E_SIZE : f_in * 32 * 4 + 256; }

extern "C" int gae_spx_fwd(const int32_t *rowptr, const int32_t *col, const float *val, int64_t n, int64_t f_in,
                           const float *W, int64_t ldw, int64_t f_out, float *P, int64_t ldp, void *workspace,
                           int64_t workspace_bytes, void *stream)
{
    GAE_REQUIRE(n >= 0 && f_in >= 1 && f_in < (1 << 24) && f_out >= 1 && f_out <= 32, GAE_E_RANGE,
                "gae_spx_fwd: needs f_in >= 1 and 1 <= f_out <= 32");
    GAE_REQUIRE(ldw >= f_in && ldp >= f_out, GAE_E_SIZE, "gae_spx_fwd: leading dimension too small");
    if (n == 0) return GAE_OK;
    GAE_REQUIRE(rowptr && W && P && workspace, GAE_E_NULL, "gae_spx_fwd: NULL pointer");
    GAE_REQUIRE(workspace_bytes >= gae_spx_fwd_workspace_bytes(f_in) && gae::aligned16(workspace), GAE_E_WORKSPACE,
                "gae_spx_fwd: workspace too small or misaligned");
    hipStream_t s = gae::as_stream(stream);
    float *Wt = static_cast<float *>(workspace);
    hipLaunchKernelGGL(spx_transpose_kernel, dim3(unsigned((f_in + 63) / 64)), dim3(256), 0, s, W, ldw, int(f_in),
                       int(f_out), Wt);
    GAE_CHECK_LAUNCH("spx_transpose_kernel");
    SpxFwdArgs a{};
    a.rowptr = rowptr; a.col = col; a.val = val; a.Wt = Wt; a.P = P; a.n = n; a.ldp = ldp;
    a.K = int(f_in); a.J = int(f_out);
    hipLaunchKernelGGL(spx_fwd_kernel, dim3(unsigned((n + 31) / 32)), dim3(256), 0, s, a);
    GAE_CHECK_LAUNCH("spx_fwd_kernel");
    return GAE_OK;
}

// host-side layout of the backward's partial lists, from the number of segments the caller built:
//   out[0] = slots (partials per element of dW), out[1] = floats between two slots, out[2] = float offset of the db
//   partials, out[3] = their count, out[4] = workspace bytes
extern "C" int gae_spx_wgrad_layout(int64_t n, int64_t f_in, int64_t max_segments_per_feature, int64_t *out)
{
    GAE_REQUIRE(n >= 0 && f_in >= 1 && max_segments_per_feature >= 0 && out, GAE_E_SIZE, "gae_spx_wgrad_layout: bad sizes");
    const int64_t slots = max_segments_per_feature < 1 ? 1 : max_segments_per_feature;
    int64_t dbb = (n + 127) / 128;          // <= 2 trips of 8 rows per thread (a partial per block: the list stays short)
    if (dbb > 256) dbb = 256;
    if (dbb < 1) dbb = 1;
    out[0] = slots; out[1] = 32 * f_in; out[2] = slots * 32 * f_in; out[3] = dbb;
    out[4] = (out[2] + dbb * 32) * 4 + 256;
    return GAE_OK;
}

extern "C" int gae_spx_wgrad(const int32_t *t_rowptr, const int32_t *t_row, const float *t_val, const int32_t *seg_feat,
                             const int32_t *seg_e0, const int32_t *seg_slot, int64_t n_segments,
                             int64_t max_segments_per_feature, int64_t n, int64_t f_in, const float *G, int64_t ldg,
                             const float *D, int64_t ldd, const float *Dmask, int64_t lddm, int64_t f_out, float *dW,
                             int64_t lddw, float *db, int reduce, void *workspace, int64_t workspace_bytes, void *stream)
{
    GAE_REQUIRE(n >= 1 && f_in >= 1 && f_in < (1 << 24) && f_out >= 1 && f_out <= 32 && n_segments >= 0, GAE_E_RANGE,
                "gae_spx_wgrad: needs n, f_in >= 1 and 1 <= f_out <= 32");
    int64_t lay[5];
    gae_spx_wgrad_layout(n, f_in, max_segments_per_feature, lay);
    GAE_REQUIRE(workspace && gae::aligned16(workspace) && workspace_bytes >= lay[4], GAE_E_WORKSPACE,
                "gae_spx_wgrad: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)lay[4]);
    GAE_REQUIRE(t_rowptr && G && ldg >= f_out, GAE_E_NULL, "gae_spx_wgrad: NULL pointer");
    GAE_REQUIRE(n_segments == 0 || (t_row && t_val && seg_feat && seg_e0 && seg_slot), GAE_E_NULL,
                "gae_spx_wgrad: NULL segment arrays");
    GAE_REQUIRE(!db || (D && ldd >= f_out && (!Dmask || lddm >= f_out)), GAE_E_NULL, "gae_spx_wgrad: db needs D");
    GAE_REQUIRE(!reduce || !dW || lddw >= f_in, GAE_E_SIZE, "gae_spx_wgrad: lddw < f_in");
    hipStream_t s = gae::as_stream(stream);
    SpxBwdArgs a{};
    a.t_rowptr = t_rowptr; a.t_row = t_row; a.t_val = t_val; a.seg_feat = seg_feat; a.seg_e0 = seg_e0; a.seg_slot = seg_slot;
    a.G = G; a.D = db ? D : nullptr; a.Dmask = Dmask;
    a.part = static_cast<float *>(workspace); a.dbpart = a.part + lay[2];
    a.n = n; a.ldg = ldg; a.ldd = ldd; a.lddm = lddm;
    a.K = int(f_in); a.J = int(f_out); a.n_segments = int(n_segments);
    a.n_db_blocks = db ? int(lay[3]) : 0;
    a.rows_per_db_block = (n + lay[3] - 1) / lay[3];
    a.seg_blocks = int((n_segments + 31) / 32);
    a.n_slots = int(lay[0]);
    if (a.seg_blocks + a.n_db_blocks > 0) {
        hipLaunchKernelGGL(spx_wgrad_kernel, dim3(unsigned(a.seg_blocks + a.n_db_blocks)), dim3(256), 0, s, a);
        GAE_CHECK_LAUNCH("spx_wgrad_kernel");
    }
    if (!reduce) return GAE_OK;
    gae::PartialList la{}, lb{};
    if (dW) la = gae::PartialList{a.part, dW, f_out * f_in, lay[0], 32 * f_in, f_in, f_in, lddw};
    if (db) lb = gae::PartialList{a.dbpart, db, f_out, lay[3], 32, f_out, f_out, f_out};
    return gae::launch_partials_reduce(la, lb, s);
}
