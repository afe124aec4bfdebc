// K1/K2: sparse aggregation  M = diag(rs) A diag(cs) H  on gfx950.
//
// Replaces DGL's fused copy_src+sum message passing that the reference invokes
// with g.update_all(gcn_msg, gcn_reduce) (gae_dgl/gae.py:18-19,28) and its
// autograd (gae_dgl/train_inductive.py:51; call it on the CSR of A^T).
//
// HBM-bound gather: algorithmic bytes per launch
//     B_alg = 4 (n_rows + 1) + 4 nnz + s F n_cols + s F n_rows
// Layout: CSR rows = destination nodes; H / M row-major, one node per row.
//
// Kernel family "rowgroup": a group of LPR lanes (power of two, <= 64) owns one
// output row and VEC contiguous features per lane and chunk (16-byte vectors
// when the layout allows); a wave64 therefore streams 64/LPR rows at once and a
// neighbour row is read as one contiguous LPR*VEC*4-byte segment (F=32 fp32:
// 8 lanes x float4 = one 128-B line).  Sums run in CSR order in fp32 without
// atomics: results are bit-stable run to run.  Block ids are remapped so that
// each XCD (private L2) owns a contiguous row range.
#include <string.h>

#include "common.h"

namespace {

using gae::kWave;

template <typename T, int VEC>
struct VecIO;

template <>
struct VecIO<float, 4> {
    static __device__ __forceinline__ void load(const float *p, float (&v)[4])
    {
        const float4 t = *reinterpret_cast<const float4 *>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void store(float *p, const float (&v)[4])
    {
        *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
    static __device__ __forceinline__ void store_nt(float *p, const float (&v)[4])
    {
        typedef float f4 __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(f4{v[0], v[1], v[2], v[3]}, reinterpret_cast<f4 *>(p));
    }
};
template <>
struct VecIO<float, 1> {
    static __device__ __forceinline__ void load(const float *p, float (&v)[1]) { v[0] = *p; }
    static __device__ __forceinline__ void store(float *p, const float (&v)[1]) { *p = v[0]; }
    static __device__ __forceinline__ void store_nt(float *p, const float (&v)[1]) { __builtin_nontemporal_store(v[0], p); }
};
template <>
struct VecIO<unsigned short, 8> {
    static __device__ __forceinline__ void load(const unsigned short *p, float (&v)[8])
    {
        const uint4 t = *reinterpret_cast<const uint4 *>(p);
        const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ void store(unsigned short *p, const float (&v)[8])
    {
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            w[i] = unsigned(gae::f32_to_bf16(v[2 * i])) | (unsigned(gae::f32_to_bf16(v[2 * i + 1])) << 16);
        *reinterpret_cast<uint4 *>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    static __device__ __forceinline__ void store_nt(unsigned short *p, const float (&v)[8]) { store(p, v); }
};
template <>
struct VecIO<unsigned short, 1> {
    static __device__ __forceinline__ void load(const unsigned short *p, float (&v)[1])
    {
        v[0] = gae::bf16_to_f32(*p);
    }
    static __device__ __forceinline__ void store(unsigned short *p, const float (&v)[1])
    {
        *p = gae::f32_to_bf16(v[0]);
    }
    static __device__ __forceinline__ void store_nt(unsigned short *p, const float (&v)[1]) { store(p, v); }
};

__device__ __forceinline__ void store_scalar(float *p, float v) { *p = v; }
__device__ __forceinline__ void store_scalar(unsigned short *p, float v) { *p = gae::f32_to_bf16(v); }

template <typename T, int VEC, int LPR, int CH, bool SCALED>
__global__ __launch_bounds__(256) void spmm_rowgroup_kernel(
    const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices, int64_t n_rows,
    const T *__restrict__ H, int64_t ldh, T *__restrict__ M, int64_t ldm, int F,
    const float *__restrict__ row_scale, const float *__restrict__ col_scale)
{
    constexpr int RPB = 256 / LPR;      // rows per block
    constexpr int TILE = LPR * VEC;     // features per chunk
    constexpr int UNR = 4;              // neighbour rows in flight per group
    const unsigned blk = gae::xcd_remap(blockIdx.x, gridDim.x);
    const int lig = threadIdx.x % LPR;
    const int64_t row = int64_t(blk) * RPB + threadIdx.x / LPR;
    if (row >= n_rows) return;
    const int f0 = blockIdx.y * (CH * TILE) + lig * VEC;

    float acc[CH][VEC];
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[c][i] = 0.f;

    bool live[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) live[c] = (f0 + c * TILE) < F;

    const int32_t start = indptr[row], end = indptr[row + 1];
    int32_t e = start;
    for (; e + UNR <= end; e += UNR) {
        int32_t j[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) j[u] = indices[e + u];
        float v[UNR][CH][VEC];
        float cs[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (SCALED) cs[u] = col_scale[j[u]];
            const T *hp = H + int64_t(j[u]) * ldh + f0;
#pragma unroll
            for (int c = 0; c < CH; ++c)
                if (live[c]) VecIO<T, VEC>::load(hp + c * TILE, v[u][c]);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c)
                if (live[c]) {
#pragma unroll
                    for (int i = 0; i < VEC; ++i)
                        acc[c][i] = SCALED ? fmaf(cs[u], v[u][c][i], acc[c][i]) : acc[c][i] + v[u][c][i];
                }
    }
    for (; e < end; ++e) {
        const int32_t j = indices[e];
        const float cs = SCALED ? col_scale[j] : 1.f;
        const T *hp = H + int64_t(j) * ldh + f0;
#pragma unroll
        for (int c = 0; c < CH; ++c)
            if (live[c]) {
                float v[VEC];
                VecIO<T, VEC>::load(hp + c * TILE, v);
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[c][i] = SCALED ? fmaf(cs, v[i], acc[c][i]) : acc[c][i] + v[i];
            }
    }
    const float rs = SCALED ? row_scale[row] : 1.f;
    T *mp = M + row * ldm + f0;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        if (!live[c]) continue;
        if (SCALED) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[c][i] *= rs;
        }
        const int f = f0 + c * TILE;
        if (VEC == 1 || f + VEC <= F) {
            VecIO<T, VEC>::store(mp + c * TILE, acc[c]);
        } else {
#pragma unroll
            for (int i = 0; i < VEC; ++i)
                if (f + i < F) store_scalar(mp + c * TILE + i, acc[c][i]);
        }
    }
}


// ---------------------------------------------------------------------------
// v2 "rowgroup2": same ownership (LPR lanes own RPG rows), but
//   * neighbour ids are fetched by ONE coalesced load per group and batch (lane
//     u of the group loads indices[pos + u]) and broadcast with ds_bpermute /
//     readlane instead of every lane re-loading them;
//   * the neighbour rows of a batch (up to NB per owned row, RPG rows) are all
//     issued before the first is consumed, with predication instead of a
//     serial tail loop: one HBM round trip per batch instead of one per edge;
//   * for LPR == 64 the row, its edge range and the neighbour base addresses
//     are wave-uniform (SGPR) values.
// Summation order is unchanged (CSR order), so results are bit-identical to v1.
// ---------------------------------------------------------------------------
template <typename T, int VEC, int LPR, int CH, int RPG, bool SCALED, bool NT_STORE>
__global__ __launch_bounds__(256) void spmm_rowgroup2_kernel(
    const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices, int64_t n_rows,
    const T *__restrict__ H, int64_t ldh, T *__restrict__ M, int64_t ldm, int F,
    const float *__restrict__ row_scale, const float *__restrict__ col_scale, unsigned n_row_blocks,
    unsigned n_ftiles, int xcd_tiled)
{
    constexpr int GPB = 256 / LPR;            // groups per block
    constexpr int RPB = GPB * RPG;            // rows per block
    constexpr int TILE = LPR * VEC;
    constexpr int NB = LPR >= 4 ? 4 : LPR;    // neighbours per batch and row
    unsigned blk, ftile;
    if (xcd_tiled) {
        // Feature-tiled XCD mapping: XCD x (= blockIdx % 8) owns feature tiles x, x+8, ... and sweeps all
        // row blocks of one tile before the next, so the tile's slice of H (n_cols * TILE * 4 bytes) stays
        // resident in that XCD's private 4 MiB L2 while it is gathered.
        const unsigned xcd = blockIdx.x % gae::kNumXcd, k = blockIdx.x / gae::kNumXcd;
        const unsigned tl = k / n_row_blocks;
        blk = k - tl * n_row_blocks;
        ftile = xcd + gae::kNumXcd * tl;
        if (ftile >= n_ftiles) return;
    } else {
        blk = gae::xcd_remap(blockIdx.x, gridDim.x);
        ftile = blockIdx.y;
    }
    const int lig = threadIdx.x % LPR;
    int grp = threadIdx.x / LPR;
    if (LPR == 64) grp = __builtin_amdgcn_readfirstlane(grp);
    const int glane0 = (threadIdx.x & 63) - lig;  // first lane of this group inside the wave
    const int f0 = ftile * (CH * TILE) + lig * VEC;

    bool live[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) live[c] = (f0 + c * TILE) < F;

    int64_t row[RPG];
    int32_t pos[RPG], end[RPG];
    float acc[RPG][CH][VEC];
#pragma unroll
    for (int r = 0; r < RPG; ++r) {
        row[r] = int64_t(blk) * RPB + r * GPB + grp;
        const bool rv = row[r] < n_rows;
        pos[r] = rv ? indptr[row[r]] : 0;
        end[r] = rv ? indptr[row[r] + 1] : 0;
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[r][c][i] = 0.f;
    }

    for (;;) {
        int rem_max = 0;
#pragma unroll
        for (int r = 0; r < RPG; ++r) rem_max = max(rem_max, end[r] - pos[r]);
        if (rem_max <= 0) break;
        // ---- one coalesced index load per owned row
        int32_t myidx[RPG];
#pragma unroll
        for (int r = 0; r < RPG; ++r) {
            const int32_t e = pos[r] + (LPR >= 4 ? (lig & 3) : lig);
            myidx[r] = e < end[r] ? indices[e] : 0;
        }
        float v[RPG][NB][CH][VEC];
        float cs[RPG][NB];
#pragma unroll
        for (int r = 0; r < RPG; ++r)
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                int32_t j;
                if (LPR == 64) j = __builtin_amdgcn_readlane(myidx[r], u);
                else j = __shfl(myidx[r], glane0 + u, 64);
                const bool ev = pos[r] + u < end[r];
                if (SCALED) cs[r][u] = ev ? col_scale[j] : 0.f;
                const T *hp = H + int64_t(j) * ldh + f0;
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    if (ev && live[c]) {
                        VecIO<T, VEC>::load(hp + c * TILE, v[r][u][c]);
                    } else {
#pragma unroll
                        for (int i = 0; i < VEC; ++i) v[r][u][c][i] = 0.f;
                    }
                }
            }
#pragma unroll
        for (int r = 0; r < RPG; ++r) {
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const bool ev = pos[r] + u < end[r];
                if (ev) {
#pragma unroll
                    for (int c = 0; c < CH; ++c)
#pragma unroll
                        for (int i = 0; i < VEC; ++i)
                            acc[r][c][i] = SCALED ? fmaf(cs[r][u], v[r][u][c][i], acc[r][c][i])
                                                  : acc[r][c][i] + v[r][u][c][i];
                }
            }
            pos[r] = min(pos[r] + NB, end[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < RPG; ++r) {
        if (row[r] >= n_rows) continue;
        const float rs = SCALED ? row_scale[row[r]] : 1.f;
        T *mp = M + row[r] * ldm + f0;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (!live[c]) continue;
            if (SCALED) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[r][c][i] *= rs;
            }
            const int f = f0 + c * TILE;
            if (VEC == 1 || f + VEC <= F) {
                if (NT_STORE) VecIO<T, VEC>::store_nt(mp + c * TILE, acc[r][c]);
                else VecIO<T, VEC>::store(mp + c * TILE, acc[r][c]);
            } else {
#pragma unroll
                for (int i = 0; i < VEC; ++i)
                    if (f + i < F) store_scalar(mp + c * TILE + i, acc[r][c][i]);
            }
        }
    }
}

// tuning knobs (gae_tuning_set): read-mostly process-wide integers
int g_spmm_variant = 2;   // 1 = v1 rowgroup, 2 = v2 rowgroup2
int g_spmm_rpg = 2;       // rows per group (v2)
int g_spmm_nt = 1;        // non-temporal stores of M (v2)
int g_spmm_tile_vecs = 0; // 16-byte vectors per XCD feature tile (0 auto, -1 off)

template <typename T, int VEC, int LPR, int CH, int RPG>
int launch_rowgroup2(const int32_t *indptr, const int32_t *indices, int64_t n_rows, const T *H, int64_t ldh, T *M,
                     int64_t ldm, int F, const float *rs, const float *cs, bool nt, bool tiled, hipStream_t s)
{
    constexpr int RPB = (256 / LPR) * RPG;
    const int nvec = (F + VEC - 1) / VEC;
    const unsigned nrb = unsigned((n_rows + RPB - 1) / RPB), nft = unsigned((nvec + LPR * CH - 1) / (LPR * CH));
    const dim3 grid = tiled ? dim3(gae::kNumXcd * ((nft + gae::kNumXcd - 1) / gae::kNumXcd) * nrb) : dim3(nrb, nft);
    const int xt = tiled ? 1 : 0;
#define GAE_L2(SC, NT)                                                                                              \
    hipLaunchKernelGGL((spmm_rowgroup2_kernel<T, VEC, LPR, CH, RPG, SC, NT>), grid, dim3(256), 0, s, indptr, indices, \
                       n_rows, H, ldh, M, ldm, F, rs, cs, nrb, nft, xt)
    if (rs || cs) { if (nt) GAE_L2(true, true); else GAE_L2(true, false); }
    else { if (nt) GAE_L2(false, true); else GAE_L2(false, false); }
#undef GAE_L2
    GAE_CHECK_LAUNCH("spmm_rowgroup2_kernel");
    return GAE_OK;
}

template <typename T, int VEC>
int dispatch_rowgroup2(const int32_t *indptr, const int32_t *indices, int64_t n_rows, const T *H, int64_t ldh, T *M,
                       int64_t ldm, int F, const float *rs, const float *cs, int rpg, bool nt, int64_t n_cols,
                       hipStream_t s)
{
    const int nvec = (F + VEC - 1) / VEC;
    bool tiled = false;
#define GAE_RG2(LPR, CH)                                                                                          \
    do {                                                                                                          \
        if (rpg >= 2 && CH == 1)                                                                                  \
            return launch_rowgroup2<T, VEC, LPR, CH, 2>(indptr, indices, n_rows, H, ldh, M, ldm, F, rs, cs, nt,   \
                                                        tiled, s);                                                \
        return launch_rowgroup2<T, VEC, LPR, CH, 1>(indptr, indices, n_rows, H, ldh, M, ldm, F, rs, cs, nt, tiled, \
                                                    s);                                                           \
    } while (0)
    // Wide rows whose per-tile slice of H fits one XCD's L2: feature-tiled XCD mapping.
    {
        int tv = g_spmm_tile_vecs;  // vectors (16 B) per feature tile; 0 = auto, < 0 = off
        if (tv == 0 && nvec > 16) {
            const int64_t budget = int64_t(5) << 19;  // 2.5 MiB of a 4 MiB L2
            for (int cand : {32, 16, 8, 4})
                if (n_cols * cand * 16 <= budget && (nvec + cand - 1) / cand >= 8) { tv = cand; break; }
        }
        if (tv > 0 && (nvec + tv - 1) / tv >= 2) {
            tiled = true;
            if (tv <= 4) GAE_RG2(4, 1);
            if (tv <= 8) GAE_RG2(8, 1);
            if (tv <= 16) GAE_RG2(16, 1);
            if (tv <= 32) GAE_RG2(32, 1);
            GAE_RG2(64, 1);
        }
    }
    if (nvec <= 4) GAE_RG2(4, 1);
    if (nvec <= 8) GAE_RG2(8, 1);
    if (nvec <= 16) GAE_RG2(16, 1);
    if (nvec <= 32) GAE_RG2(32, 1);
    if (nvec <= 64) GAE_RG2(64, 1);
    if (nvec <= 128) GAE_RG2(64, 2);
    GAE_RG2(64, 4);
#undef GAE_RG2
}


template <typename T, int VEC, int LPR, int CH>
int launch_rowgroup(const int32_t *indptr, const int32_t *indices, int64_t n_rows, const T *H, int64_t ldh, T *M,
                    int64_t ldm, int F, const float *rs, const float *cs, hipStream_t s)
{
    constexpr int RPB = 256 / LPR;
    const int nvec = (F + VEC - 1) / VEC;
    const unsigned gx = unsigned((n_rows + RPB - 1) / RPB);
    const unsigned gy = unsigned((nvec + LPR * CH - 1) / (LPR * CH));
    if (rs || cs)
        hipLaunchKernelGGL((spmm_rowgroup_kernel<T, VEC, LPR, CH, true>), dim3(gx, gy), dim3(256), 0, s, indptr,
                           indices, n_rows, H, ldh, M, ldm, F, rs, cs);
    else
        hipLaunchKernelGGL((spmm_rowgroup_kernel<T, VEC, LPR, CH, false>), dim3(gx, gy), dim3(256), 0, s, indptr,
                           indices, n_rows, H, ldh, M, ldm, F, rs, cs);
    GAE_CHECK_LAUNCH("spmm_rowgroup_kernel");
    return GAE_OK;
}

template <typename T, int VEC>
int dispatch_rowgroup(const int32_t *indptr, const int32_t *indices, int64_t n_rows, const T *H, int64_t ldh, T *M,
                      int64_t ldm, int F, const float *rs, const float *cs, hipStream_t s)
{
    const int nvec = (F + VEC - 1) / VEC;
#define GAE_RG(LPR, CH) return launch_rowgroup<T, VEC, LPR, CH>(indptr, indices, n_rows, H, ldh, M, ldm, F, rs, cs, s)
    if (nvec <= 1) GAE_RG(1, 1);
    if (nvec <= 2) GAE_RG(2, 1);
    if (nvec <= 4) GAE_RG(4, 1);
    if (nvec <= 8) GAE_RG(8, 1);
    if (nvec <= 16) GAE_RG(16, 1);
    if (nvec <= 32) GAE_RG(32, 1);
    if (nvec <= 64) GAE_RG(64, 1);
    if (nvec <= 128) GAE_RG(64, 2);
    GAE_RG(64, 4);
#undef GAE_RG
}

} // namespace

extern "C" int gae_spmm_csr(const int32_t *indptr, const int32_t *indices, int64_t n_rows, int64_t n_cols,
                            const void *H, int64_t ldh, void *M, int64_t ldm, int64_t F, int dtype,
                            const float *row_scale, const float *col_scale, void *stream)
{
    GAE_REQUIRE(n_rows >= 0 && n_cols >= 0 && F >= 0, GAE_E_SIZE, "gae_spmm_csr: negative size");
    GAE_REQUIRE(F < (int64_t(1) << 24), GAE_E_SIZE, "gae_spmm_csr: F too large");
    GAE_REQUIRE(ldh >= F && ldm >= F, GAE_E_SIZE, "gae_spmm_csr: leading dimension smaller than F");
    GAE_REQUIRE(dtype == GAE_F32 || dtype == GAE_BF16, GAE_E_DTYPE, "gae_spmm_csr: unsupported dtype %d", dtype);
    GAE_REQUIRE((row_scale == nullptr) == (col_scale == nullptr), GAE_E_NULL,
                "gae_spmm_csr: row_scale and col_scale must both be given or both be NULL");
    if (n_rows == 0 || F == 0) return GAE_OK;
    GAE_REQUIRE(indptr && M, GAE_E_NULL, "gae_spmm_csr: NULL pointer");
    // `indices` may be NULL only for an edge-less graph (indptr all zero): it is never dereferenced then
    GAE_REQUIRE(n_cols == 0 || H, GAE_E_NULL, "gae_spmm_csr: H is NULL with n_cols > 0");
    GAE_REQUIRE((n_rows + 3) / 4 < (int64_t(1) << 31), GAE_E_SIZE, "gae_spmm_csr: too many rows for one launch");
    hipStream_t s = gae::as_stream(stream);
    const int f = int(F);
    if (dtype == GAE_F32) {
        const float *h = static_cast<const float *>(H);
        float *m = static_cast<float *>(M);
        const bool vec = (ldh % 4 == 0) && (ldm % 4 == 0) && gae::aligned16(H) && gae::aligned16(M);
        if (vec) {
            if (g_spmm_variant == 2 && f > 12)
                return dispatch_rowgroup2<float, 4>(indptr, indices, n_rows, h, ldh, m, ldm, f, row_scale, col_scale,
                                                    g_spmm_rpg, g_spmm_nt != 0, n_cols, s);
            return dispatch_rowgroup<float, 4>(indptr, indices, n_rows, h, ldh, m, ldm, f, row_scale, col_scale, s);
        }
        if (g_spmm_variant == 2 && f > 3)
            return dispatch_rowgroup2<float, 1>(indptr, indices, n_rows, h, ldh, m, ldm, f, row_scale, col_scale,
                                                g_spmm_rpg, g_spmm_nt != 0, n_cols, s);
        return dispatch_rowgroup<float, 1>(indptr, indices, n_rows, h, ldh, m, ldm, f, row_scale, col_scale, s);
    }
    const unsigned short *h = static_cast<const unsigned short *>(H);
    unsigned short *m = static_cast<unsigned short *>(M);
    const bool vec = (ldh % 8 == 0) && (ldm % 8 == 0) && gae::aligned16(H) && gae::aligned16(M);
    if (vec) {
        if (g_spmm_variant == 2 && f > 24)
            return dispatch_rowgroup2<unsigned short, 8>(indptr, indices, n_rows, h, ldh, m, ldm, f, row_scale,
                                                         col_scale, g_spmm_rpg, false, n_cols, s);
        return dispatch_rowgroup<unsigned short, 8>(indptr, indices, n_rows, h, ldh, m, ldm, f, row_scale, col_scale, s);
    }
    if (g_spmm_variant == 2 && f > 3)
        return dispatch_rowgroup2<unsigned short, 1>(indptr, indices, n_rows, h, ldh, m, ldm, f, row_scale, col_scale,
                                                     g_spmm_rpg, false, n_cols, s);
    return dispatch_rowgroup<unsigned short, 1>(indptr, indices, n_rows, h, ldh, m, ldm, f, row_scale, col_scale, s);
}

namespace gae { int *dense_knob(const char *name); }

extern "C" int gae_tuning_set(const char *name, int64_t value)
{
    GAE_REQUIRE(name != nullptr, GAE_E_NULL, "gae_tuning_set: name is NULL");
    const struct { const char *k; int *v; } knobs[] = {
        {"spmm_variant", &g_spmm_variant}, {"spmm_rpg", &g_spmm_rpg}, {"spmm_nt", &g_spmm_nt},
        {"spmm_tile_vecs", &g_spmm_tile_vecs}};
    for (const auto &kv : knobs)
        if (strcmp(kv.k, name) == 0) {
            *kv.v = int(value);
            return GAE_OK;
        }
    if (int *k = gae::dense_knob(name)) {
        *k = int(value);
        return GAE_OK;
    }
    gae::set_error("gae_tuning_set: unknown knob '%s'", name);
    return GAE_E_RANGE;
}
