// K7+K8+K9 fused: inner-product decoder + weighted BCE-with-logits (mean) and
// its gradient, without ever materialising the N x N logits or labels.
//
// Replaces, for training, the chain of gae_dgl/train_inductive.py:44-51
//     adj = g.adjacency_matrix().to_dense(); pos_weight = (N^2 - sum adj)/sum adj
//     adj_logits = model.forward(g)        (gae_dgl/gae.py:70-71: Zt Zt^T)
//     loss = binary_cross_entropy_with_logits(adj_logits, adj, pos_weight=...)
//     loss.backward()
//
// Math.  x_ij = zt_i . zt_j,  y_ij = #edges j->i  (CSR rows = destination):
//   N^2 loss = sum_ij [(1 - y) x + (1 + (pw - 1) y) softplus(-x)]
//            = sum_ij softplus(x_ij)                       <- dense, label-free
//            + sum_{edges (i,j)} [-x_ij + (pw - 1) softplus(-x_ij)]   <- sparse
//   dL/dx_ij = [sigmoid(x_ij) + y_ij ((pw - 1) sigmoid(x_ij) - pw)] / N^2
//   dZt = (G + G^T) Zt   ->  dense part 2 sigmoid(X) Zt / N^2 (X symmetric)
//                            sparse part over in-edges (CSR) and out-edges (CSR of A^T).
//
// Dense kernel: flash-style.  Wave owns RI 16-row subtiles, streams column tiles
// through LDS, S^T = Zj Zi^T on v_mfma_f32_16x16x4_f32 (exact fp32), softplus /
// sigmoid on the VALU, O += sigmoid(S) Zj on the same MFMA shape: the S^T
// accumulator registers are directly the A fragments of the second product
// (lane = row i, reg r <-> j = 4 (lane >> 4) + r), so P never leaves registers.
// MFMA-bound (fp32 matrix rate), not HBM-bound: 4 KS (S) + 4 KS (PV) MFMAs per
// 256 logits with KS = ceil(d / 16).
// Reductions are two-stage and ordered: bit-stable run to run.
#include <string.h>
#include <type_traits>

#include "common.h"

namespace {

using gae::kWave;
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TJ = 64;   // columns staged per iteration
// RI = 16-row subtiles per wave (rows / block = 64 RI); tuning knobs "bce_ri", "bce_minw"
int g_bce_ri = 2;
int g_bce_minw = 0;

__device__ __forceinline__ void softplus_sigmoid(float x, float &sp, float &sg)
{
    // e = exp(-|x|) in (0, 1]; raw v_exp/v_log/v_rcp: an underflowing e flushes to 0, which is the
    // correctly rounded answer for both outputs (no denormal fix-up code needed)
    const float e = __builtin_amdgcn_exp2f(-1.44269504088896341f * fabsf(x));
    const float t = 1.0f + e;
    const float r = __builtin_amdgcn_rcpf(t);
    sp = fmaf(__builtin_amdgcn_logf(t), 0.69314718055994531f, fmaxf(x, 0.f));
    sg = (x >= 0.f ? 1.0f : e) * r;
}

// Zt[n][DP] = Z (.) mask, zero padded to DP = 16 KS columns: the dense kernel then reads
// clean 16-byte aligned rows without mask loads or feature bounds checks.
__global__ __launch_bounds__(256) void bce_prepare_kernel(const float *__restrict__ Z, const float *__restrict__ mask,
                                                          int64_t ldz, int64_t n, int d, int DP,
                                                          float *__restrict__ Zt)
{
    const int64_t total = n * DP, stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t i = e / DP;
        const int k = int(e - i * DP);
        float v = 0.f;
        if (k < d) {
            v = Z[i * ldz + k];
            if (mask) v *= mask[i * ldz + k];
        }
        Zt[e] = v;
    }
}

// ---------------------------------------------------------------------------
template <int KS, bool WITH_GRAD, int RI, int MINW>
__global__ __launch_bounds__(256, MINW) void bce_dense_kernel(
    const float *__restrict__ Zt /*[n][16 KS]*/, int64_t n, int64_t row_begin, int64_t n_local,
    int64_t cols_per_split, float *__restrict__ O_partial /*[splits][n_local][KS*16]*/,
    double *__restrict__ loss_partial /*[gridDim.x * gridDim.y]*/)
{
    constexpr int ROWS_PER_BLOCK = 4 * RI * 16;
    constexpr int DP = KS * 16;          // padded feature width
    constexpr int LDA = DP + 4;          // LDS row stride (floats): 16-byte aligned, breaks the power of two
    constexpr int V4 = TJ * DP / 4 / 256;  // float4 per thread and staged tile (1, 2, 4)
    __shared__ __attribute__((aligned(16))) float Zs[2][TJ * LDA];   // double-buffered column tile [j][k]
    __shared__ double red[4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    // rows are LOCAL ids (0 .. n_local) of the window [row_begin, row_begin + n_local) of Zt
    const int64_t row_base = int64_t(blockIdx.x) * ROWS_PER_BLOCK + wave * (RI * 16);
    const int64_t col_begin = int64_t(blockIdx.y) * cols_per_split;
    int64_t col_end = col_begin + cols_per_split;
    if (col_end > n) col_end = n;

    // B fragments of S^T = Zj Zi^T: lane (i = l15, g) holds Zt[i][16 c + 4 g + r]
    f32x4 bfrag[RI][KS];
#pragma unroll
    for (int ri = 0; ri < RI; ++ri) {
        const int64_t i = row_base + ri * 16 + l15;
#pragma unroll
        for (int c = 0; c < KS; ++c)
            bfrag[ri][c] = i < n_local ? *reinterpret_cast<const f32x4 *>(Zt + (row_begin + i) * DP + 16 * c + 4 * g)
                                       : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 oacc[RI][KS];
#pragma unroll
    for (int ri = 0; ri < RI; ++ri)
#pragma unroll
        for (int c = 0; c < KS; ++c) oacc[ri][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    double lsum[RI];
#pragma unroll
    for (int ri = 0; ri < RI; ++ri) lsum[ri] = 0.0;

    // staging: thread -> float4 #(tid + 256 q) of the tile; tile rows are contiguous in Zt
    auto load_tile = [&](int64_t j0, f32x4 (&reg)[V4]) {
#pragma unroll
        for (int q = 0; q < V4; ++q) {
            const int idx = tid + 256 * q;            // float4 index inside the tile
            const int jj = idx / (DP / 4);
            reg[q] = (j0 + jj < col_end) ? *reinterpret_cast<const f32x4 *>(Zt + (j0 + jj) * DP + (idx % (DP / 4)) * 4)
                                         : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_tile = [&](int buf, const f32x4 (&reg)[V4]) {
#pragma unroll
        for (int q = 0; q < V4; ++q) {
            const int idx = tid + 256 * q;
            *reinterpret_cast<f32x4 *>(&Zs[buf][(idx / (DP / 4)) * LDA + (idx % (DP / 4)) * 4]) = reg[q];
        }
    };

    auto compute_tile = [&](const float *zs, int64_t j0, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
        for (int jt = 0; jt < TJ / 16; ++jt) {
            // A fragments: lane (j = l15, g) -> zs[jt*16 + j][16 c + 4 g .. +3]
            f32x4 afrag[KS];
#pragma unroll
            for (int c = 0; c < KS; ++c)
                afrag[c] = *reinterpret_cast<const f32x4 *>(&zs[(jt * 16 + l15) * LDA + 16 * c + 4 * g]);
            f32x4 sacc[RI];
#pragma unroll
            for (int ri = 0; ri < RI; ++ri) sacc[ri] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < KS; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int ri = 0; ri < RI; ++ri)
                        sacc[ri] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[c][r], bfrag[ri][c][r], sacc[ri], 0, 0, 0);
            // sacc[ri][r] = x(i = l15 of subtile ri, j = jt*16 + 4 g + r)
            f32x4 p[RI];
#pragma unroll
            for (int ri = 0; ri < RI; ++ri) {
                float tsum = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float sp, sg;
                    softplus_sigmoid(sacc[ri][r], sp, sg);
                    if (!FULL) {
                        const bool jv = j0 + jt * 16 + 4 * g + r < col_end;
                        sp = jv ? sp : 0.f;
                        sg = jv ? sg : 0.f;
                    }
                    tsum += sp;
                    p[ri][r] = sg;
                }
                lsum[ri] += double(tsum);
            }
            if (WITH_GRAD) {
                // B fragments of O += P V: lane (nn = l15, g) -> V[jt*16 + 4 g + r][16 c + nn]
#pragma unroll
                for (int c = 0; c < KS; ++c) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = zs[(jt * 16 + 4 * g + r) * LDA + 16 * c + l15];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int ri = 0; ri < RI; ++ri)
                            oacc[ri][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(p[ri][r], v[r], oacc[ri][c], 0, 0, 0);
                }
            }
        }
    };

    f32x4 stage[V4];
    if (col_begin < col_end) {
        load_tile(col_begin, stage);
        store_tile(0, stage);
    }
    __syncthreads();
    int buf = 0;
    for (int64_t j0 = col_begin; j0 < col_end; j0 += TJ, buf ^= 1) {
        const bool more = j0 + TJ < col_end;
        if (more) load_tile(j0 + TJ, stage);        // in flight while this tile is consumed
        if (j0 + TJ <= col_end) compute_tile(Zs[buf], j0, std::true_type{});
        else compute_tile(Zs[buf], j0, std::false_type{});
        if (more) store_tile(buf ^ 1, stage);
        __syncthreads();
    }
    // ---- O partial: oacc[ri][c][r] = O(i = 4 g + r, nn = l15) of subtile ri, feature 16 c + nn
    if (WITH_GRAD) {
        float *op = O_partial + int64_t(blockIdx.y) * n_local * DP;
#pragma unroll
        for (int ri = 0; ri < RI; ++ri)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t i = row_base + ri * 16 + 4 * g + r;
                if (i < n_local) {
#pragma unroll
                    for (int c = 0; c < KS; ++c) op[i * DP + 16 * c + l15] = oacc[ri][c][r];
                }
            }
    }
    // ---- loss partial: drop rows >= n, wave reduce (fixed tree) -> block
    double ls = 0.0;
#pragma unroll
    for (int ri = 0; ri < RI; ++ri) ls += (row_base + ri * 16 + l15) < n_local ? lsum[ri] : 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ls += __shfl_down(ls, off, 64);
    if (lane == 0) red[wave] = ls;
    __syncthreads();
    if (tid == 0) loss_partial[int64_t(blockIdx.y) * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---------------------------------------------------------------------------
// Sparse part + assembly.  A group of LPR lanes owns node i (VEC features per
// lane): in-edges from the CSR give the loss terms and G_s Zt, out-edges from
// the CSR of A^T give G_s^T Zt; then
//   dZ[i] = mask[i] * ( 2 sum_splits O[s][i] + sparse ) / N^2.
// ---------------------------------------------------------------------------
template <int VEC, int LPR, bool WITH_GRAD>
__global__ __launch_bounds__(256) void bce_edges_kernel(
    const float *__restrict__ Zt /*[n][DP] = Z (.) mask, zero padded*/, const float *__restrict__ mask, int64_t ldz,
    int64_t row_begin, int64_t n_local, int d, const int32_t *__restrict__ indptr,
    const int32_t *__restrict__ indices,
    const int32_t *__restrict__ t_indptr, const int32_t *__restrict__ t_indices, float pw, float inv_n2,
    const float *__restrict__ O_partial, int n_splits, int DP, float *__restrict__ dZ, int64_t lddz,
    double *__restrict__ loss_partial)
{
    __shared__ double red[4];
    constexpr int RPB = 256 / LPR;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lig = tid % LPR;
    const int64_t i = int64_t(blockIdx.x) * RPB + tid / LPR;   // local row
    const int64_t gi = row_begin + i;                          // its row in Z / mask
    const int64_t n = n_local;
    const int f0 = lig * VEC;
    const bool rowv = i < n;

    static_assert(VEC == 4, "edge kernel reads the padded Zt rows as float4");
    const bool fv = f0 < DP;     // lanes beyond the padded width idle (LPR is a power of two >= DP / 4)
    float zi[VEC], acc[VEC];
    {
        const f32x4 t = (rowv && fv) ? *reinterpret_cast<const f32x4 *>(Zt + gi * DP + f0) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < VEC; ++q) { acc[q] = 0.f; zi[q] = t[q]; }
    }
    double lsum = 0.0;
    // In-edges (CSR: loss + G_s Zt) then out-edges (CSR of A^T: G_s^T Zt) of node i, 4 edges per batch:
    // the 4 neighbour ids come from one coalesced load, the 4 neighbour rows are all in flight before
    // the first dot product (one memory round trip per batch instead of one per edge).
    auto sweep = [&](const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, bool with_loss) {
        int32_t pos = rowv ? ptr[i] : 0;
        const int32_t end = rowv ? ptr[i + 1] : 0;
        const int glane0 = (lane / LPR) * LPR;
        while (pos < end) {
            const int32_t e = pos + (LPR >= 4 ? (lig & 3) : 0);
            const int32_t mine = e < end ? idx[e] : 0;
            float zj[4][VEC], dot[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int32_t j = LPR >= 4 ? __shfl(mine, glane0 + u, 64) : (pos + u < end ? idx[pos + u] : 0);
                const bool ev = pos + u < end;
                dot[u] = 0.f;
                const f32x4 t = (ev && fv) ? *reinterpret_cast<const f32x4 *>(Zt + int64_t(j) * DP + f0)
                                           : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < VEC; ++q) {
                    zj[u][q] = t[q];
                    dot[u] = fmaf(zi[q], t[q], dot[u]);
                }
            }
#pragma unroll
            for (int off = LPR / 2; off > 0; off >>= 1)
#pragma unroll
                for (int u = 0; u < 4; ++u) dot[u] += __shfl_xor(dot[u], off, 64);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (pos + u < end) {
                    const float x = dot[u];
                    float spn, sgn;               // softplus(-x), sigmoid(-x)
                    softplus_sigmoid(-x, spn, sgn);
                    const float sg = 1.0f - sgn;  // sigmoid(x)
                    if (with_loss && lig == 0) lsum += double(-x + (pw - 1.0f) * spn);
                    const float c = (pw - 1.0f) * sg - pw;
#pragma unroll
                    for (int q = 0; q < VEC; ++q) acc[q] = fmaf(c, zj[u][q], acc[q]);
                }
            }
            pos += 4;
        }
    };
    sweep(indptr, indices, true);
    if (WITH_GRAD) sweep(t_indptr, t_indices, false);
    if (WITH_GRAD && rowv) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            const int f = f0 + q;
            if (f < d) {
                float o = 0.f;
                for (int s = 0; s < n_splits; ++s) o += O_partial[(int64_t(s) * n + i) * DP + f];
                float v = (2.0f * o + acc[q]) * inv_n2;
                if (mask) v *= mask[gi * ldz + f];
                dZ[i * lddz + f] = v;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lsum += __shfl_down(lsum, off, 64);
    if (lane == 0) red[wave] = lsum;
    __syncthreads();
    if (tid == 0) loss_partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// single block: ordered sum of all partials -> mean
__global__ __launch_bounds__(256) void bce_finalize_kernel(const double *__restrict__ partial, int64_t count,
                                                           double inv_n2, float *__restrict__ loss_out)
{
    __shared__ double red[256];
    double s = 0.0;
    for (int64_t k = threadIdx.x; k < count; k += 256) s += partial[k];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (int(threadIdx.x) < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss_out = float(red[0] * inv_n2);
}

struct BcePlan {
    int64_t row_blocks, n_splits, cols_per_split, edge_blocks;
    int KS, DP, LPR, VEC, RI;
    int64_t o_bytes, zt_bytes, loss_count, total_bytes;
};

inline int64_t align256(int64_t x) { return (x + 255) / 256 * 256; }

bool bce_plan(int64_t n, int64_t n_local, int64_t d, bool vec_ok, BcePlan &p)
{
    const int ri = (g_bce_ri == 1 || g_bce_ri == 4) ? g_bce_ri : 2;
    const int64_t ROWS_PER_BLOCK = 64 * ri;
    p.RI = ri;
    if (d > 64) return false;
    p.KS = int((d + 15) / 16);
    if (p.KS == 3) p.KS = 4;
    if (p.KS < 1) p.KS = 1;
    p.DP = p.KS * 16;
    p.row_blocks = (n_local + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    if (p.row_blocks < 1) p.row_blocks = 1;
    int64_t col_tiles = (n + TJ - 1) / TJ;
    if (col_tiles < 1) col_tiles = 1;
    int64_t want = (2048 + p.row_blocks - 1) / p.row_blocks;  // ~8 blocks per CU
    if (want > col_tiles) want = col_tiles;
    if (want > 64) want = 64;
    if (want < 1) want = 1;
    const int64_t tiles_per_split = (col_tiles + want - 1) / want;
    p.cols_per_split = tiles_per_split * TJ;
    p.n_splits = (col_tiles + tiles_per_split - 1) / tiles_per_split;
    p.VEC = vec_ok ? 4 : 1;
    const int nvec = p.DP / 4;   // the edge kernel reads the padded Zt rows
    int lpr = 1;
    while (lpr < nvec) lpr <<= 1;
    p.LPR = lpr;
    p.edge_blocks = (n_local + (256 / lpr) - 1) / (256 / lpr);
    if (p.edge_blocks < 1) p.edge_blocks = 1;
    p.o_bytes = align256(p.n_splits * n_local * p.DP * 4);
    p.zt_bytes = align256(n * p.DP * 4);
    p.loss_count = p.row_blocks * p.n_splits + p.edge_blocks;
    p.total_bytes = p.o_bytes + p.zt_bytes + align256(p.loss_count * 8);
    return true;
}

template <bool WITH_GRAD>
int launch_dense(const BcePlan &p, const float *Zt, int64_t n, int64_t row_begin, int64_t n_local, float *O,
                 double *lp, hipStream_t s)
{
    const dim3 grid(unsigned(p.row_blocks), unsigned(p.n_splits));
#define GAE_BD(KS, RI, MW) hipLaunchKernelGGL((bce_dense_kernel<KS, WITH_GRAD, RI, MW>), grid, dim3(256), 0, s, Zt, n, row_begin, n_local, p.cols_per_split, O, lp)
    if (p.KS == 1) {
        if (p.RI == 1) { if (g_bce_minw >= 8) GAE_BD(1, 1, 8); else GAE_BD(1, 1, 1); }
        else if (p.RI == 4) GAE_BD(1, 4, 1);
        else if (g_bce_minw >= 8) GAE_BD(1, 2, 8);
        else if (g_bce_minw >= 6) GAE_BD(1, 2, 6);
        else if (g_bce_minw >= 5) GAE_BD(1, 2, 5);
        else GAE_BD(1, 2, 1);
    } else if (p.KS == 2) {
        if (p.RI == 1) GAE_BD(2, 1, 1); else GAE_BD(2, 2, 1);
    } else {
        if (p.RI == 1) GAE_BD(4, 1, 1); else GAE_BD(4, 2, 1);
    }
#undef GAE_BD
    GAE_CHECK_LAUNCH("bce_dense_kernel");
    return GAE_OK;
}

template <int VEC, bool WITH_GRAD>
int launch_edges(const BcePlan &p, const float *Zt, const float *mask, int64_t ldz, int64_t row_begin, int64_t n, int d,
                 const int32_t *ip, const int32_t *ix, const int32_t *tp, const int32_t *tx, float pw, float inv_n2,
                 const float *O, float *dZ, int64_t lddz, double *lp, hipStream_t s)
{
#define GAE_EDGE(LPR)                                                                                              \
    hipLaunchKernelGGL((bce_edges_kernel<VEC, LPR, WITH_GRAD>), dim3(unsigned(p.edge_blocks)), dim3(256), 0, s, Zt, \
                       mask, ldz, row_begin, n, d, ip, ix, tp, tx, pw, inv_n2, O, int(p.n_splits), p.DP, dZ, lddz, lp)
    switch (p.LPR) {
    case 1: GAE_EDGE(1); break;
    case 2: GAE_EDGE(2); break;
    case 4: GAE_EDGE(4); break;
    case 8: GAE_EDGE(8); break;
    case 16: GAE_EDGE(16); break;
    case 32: GAE_EDGE(32); break;
    default: GAE_EDGE(64); break;
    }
#undef GAE_EDGE
    GAE_CHECK_LAUNCH("bce_edges_kernel");
    return GAE_OK;
}

} // namespace

namespace gae {
int *bce_knob(const char *name)
{
    if (strcmp(name, "bce_ri") == 0) return &g_bce_ri;
    if (strcmp(name, "bce_minw") == 0) return &g_bce_minw;
    return nullptr;
}
} // namespace gae

extern "C" int64_t gae_decoder_bce_workspace_bytes(int64_t n, int64_t n_local, int64_t d)
{
    if (n < 0 || d < 0 || n_local < 0 || n_local > n) return GAE_E_SIZE;
    BcePlan p;
    if (!bce_plan(n, n_local, d, true, p)) return GAE_E_RANGE;
    return p.total_bytes + 256;
}

extern "C" int gae_decoder_bce_rows(const float *Z, const float *mask, int64_t ldz, int64_t n, int64_t d,
                                    int64_t row_begin, int64_t n_local, const int32_t *indptr,
                                    const int32_t *indices, const int32_t *t_indptr, const int32_t *t_indices,
                                    float pos_weight, float *loss_out, float *dZ, int64_t lddz, void *workspace,
                                    int64_t workspace_bytes, void *stream)
{
    GAE_REQUIRE(n > 0 && d > 0, GAE_E_SIZE, "gae_decoder_bce: n and d must be positive");
    GAE_REQUIRE(row_begin >= 0 && n_local >= 0 && row_begin + n_local <= n, GAE_E_SIZE,
                "gae_decoder_bce: row window outside [0, n)");
    GAE_REQUIRE(d <= 64, GAE_E_RANGE, "gae_decoder_bce: d = %lld > 64 not supported by the fused kernel",
                (long long)d);
    GAE_REQUIRE(n < (int64_t(1) << 31), GAE_E_SIZE, "gae_decoder_bce: n too large");
    GAE_REQUIRE(ldz >= d && (!dZ || lddz >= d), GAE_E_SIZE, "gae_decoder_bce: leading dimension too small");
    GAE_REQUIRE(Z && loss_out && workspace && (n_local == 0 || indptr), GAE_E_NULL, "gae_decoder_bce: NULL pointer");
    GAE_REQUIRE(!dZ || n_local == 0 || t_indptr, GAE_E_NULL, "gae_decoder_bce: the gradient needs the CSR of A^T");
    BcePlan p;
    bce_plan(n, n_local, d, true, p);
    GAE_REQUIRE(workspace_bytes >= p.total_bytes, GAE_E_WORKSPACE, "gae_decoder_bce: workspace %lld < %lld bytes",
                (long long)workspace_bytes, (long long)p.total_bytes);
    GAE_REQUIRE(gae::aligned16(workspace), GAE_E_ALIGN, "gae_decoder_bce: workspace not 16-byte aligned");
    hipStream_t s = gae::as_stream(stream);
    float *O = static_cast<float *>(workspace);
    float *Zt = reinterpret_cast<float *>(static_cast<char *>(workspace) + p.o_bytes);
    double *lp = reinterpret_cast<double *>(static_cast<char *>(workspace) + p.o_bytes + p.zt_bytes);
    const double inv_n2 = 1.0 / (double(n) * double(n));
    if (n_local == 0) {
        GAE_HIP(hipMemsetAsync(loss_out, 0, sizeof(float), s));
        return GAE_OK;
    }
    {
        int64_t gb = (n * p.DP + 255) / 256;
        if (gb > 2048) gb = 2048;
        hipLaunchKernelGGL(bce_prepare_kernel, dim3(unsigned(gb)), dim3(256), 0, s, Z, mask, ldz, n, int(d), p.DP, Zt);
        GAE_CHECK_LAUNCH("bce_prepare_kernel");
    }
    int rc = dZ ? launch_dense<true>(p, Zt, n, row_begin, n_local, O, lp, s)
                : launch_dense<false>(p, Zt, n, row_begin, n_local, O, lp, s);
    if (rc) return rc;
    double *lpe = lp + p.row_blocks * p.n_splits;
    rc = dZ ? launch_edges<4, true>(p, Zt, mask, ldz, row_begin, n_local, int(d), indptr, indices, t_indptr, t_indices,
                                    pos_weight, float(inv_n2), O, dZ, lddz, lpe, s)
            : launch_edges<4, false>(p, Zt, mask, ldz, row_begin, n_local, int(d), indptr, indices, t_indptr,
                                     t_indices, pos_weight, float(inv_n2), O, dZ, lddz, lpe, s);
    if (rc) return rc;
    hipLaunchKernelGGL(bce_finalize_kernel, dim3(1), dim3(256), 0, s, lp, p.loss_count, inv_n2, loss_out);
    GAE_CHECK_LAUNCH("bce_finalize_kernel");
    return GAE_OK;
}

extern "C" int gae_decoder_bce(const float *Z, const float *mask, int64_t ldz, int64_t n, int64_t d,
                               const int32_t *indptr, const int32_t *indices, const int32_t *t_indptr,
                               const int32_t *t_indices, float pos_weight, float *loss_out, float *dZ, int64_t lddz,
                               void *workspace, int64_t workspace_bytes, void *stream)
{
    return gae_decoder_bce_rows(Z, mask, ldz, n, d, 0, n, indptr, indices, t_indptr, t_indices, pos_weight, loss_out,
                                dZ, lddz, workspace, workspace_bytes, stream);
}
