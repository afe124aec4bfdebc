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
};
template <>
struct VecIO<float, 1> {
    static __device__ __forceinline__ void load(const float *p, float (&v)[1]) { v[0] = *p; }
    static __device__ __forceinline__ void store(float *p, const float (&v)[1]) { *p = v[0]; }
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
        if (vec) return dispatch_rowgroup<float, 4>(indptr, indices, n_rows, h, ldh, m, ldm, f, row_scale, col_scale, s);
        return dispatch_rowgroup<float, 1>(indptr, indices, n_rows, h, ldh, m, ldm, f, row_scale, col_scale, s);
    }
    const unsigned short *h = static_cast<const unsigned short *>(H);
    unsigned short *m = static_cast<unsigned short *>(M);
    const bool vec = (ldh % 8 == 0) && (ldm % 8 == 0) && gae::aligned16(H) && gae::aligned16(M);
    if (vec)
        return dispatch_rowgroup<unsigned short, 8>(indptr, indices, n_rows, h, ldh, m, ldm, f, row_scale, col_scale, s);
    return dispatch_rowgroup<unsigned short, 1>(indptr, indices, n_rows, h, ldh, m, ldm, f, row_scale, col_scale, s);
}
