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

namespace gae {
// spmm_ell.hip: the packed-neighbour-table kernels (lean instruction stream; see that file)
bool spmm_ell_usable(int64_t n_cols, int64_t ldh, int elem, int ell_width, int tile_vecs);
int spmm_ell_launch(const int32_t *indptr, const int32_t *indices, const int32_t *ell, int ell_width, int64_t n_rows,
                    int64_t n_cols, const void *H, int64_t ldh, void *M, int64_t ldm, int F, int dtype,
                    const float *rs, const float *cs, int tile_vecs, int xcd_tiled, int store_pad, int store_mode,
                    hipStream_t s);
Knob *spmm_ell_knob(const char *name);
}

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
    static __device__ __forceinline__ void load_nt(const float *p, float (&v)[4])      // streaming hint
    {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p));
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
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
    // write-through store that does NOT keep the line in the XCD's L2 (MI355X_MICROARCH.md, "stores of each
    // flavour"): the output stream of an XCD-tiled launch must not compete with the L2-resident slice of H
    static __device__ __forceinline__ void store_sc1(float *p, const float (&v)[4])
    {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 x = {v[0], v[1], v[2], v[3]};
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
    }
};
template <>
struct VecIO<float, 1> {
    static __device__ __forceinline__ void load(const float *p, float (&v)[1]) { v[0] = *p; }
    static __device__ __forceinline__ void load_nt(const float *p, float (&v)[1]) { v[0] = __builtin_nontemporal_load(p); }
    static __device__ __forceinline__ void store(float *p, const float (&v)[1]) { *p = v[0]; }
    static __device__ __forceinline__ void store_nt(float *p, const float (&v)[1]) { __builtin_nontemporal_store(v[0], p); }
    static __device__ __forceinline__ void store_sc1(float *p, const float (&v)[1]) { __builtin_nontemporal_store(v[0], p); }
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
    static __device__ __forceinline__ void load_nt(const unsigned short *p, float (&v)[8])
    {
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        const u4 t = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(p));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(t[i] << 16);
            v[2 * i + 1] = __uint_as_float(t[i] & 0xffff0000u);
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
    static __device__ __forceinline__ void store_sc1(unsigned short *p, const float (&v)[8]) { store(p, v); }
};
template <>
struct VecIO<unsigned short, 1> {
    static __device__ __forceinline__ void load(const unsigned short *p, float (&v)[1])
    {
        v[0] = gae::bf16_to_f32(*p);
    }
    static __device__ __forceinline__ void load_nt(const unsigned short *p, float (&v)[1])
    {
        v[0] = gae::bf16_to_f32(__builtin_nontemporal_load(p));
    }
    static __device__ __forceinline__ void store(unsigned short *p, const float (&v)[1])
    {
        *p = gae::f32_to_bf16(v[0]);
    }
    static __device__ __forceinline__ void store_nt(unsigned short *p, const float (&v)[1]) { store(p, v); }
    static __device__ __forceinline__ void store_sc1(unsigned short *p, const float (&v)[1]) { store(p, v); }
};

// GAE_SPMM_ACCUMULATE: v += the values already stored at p (first `valid` elements of the vector)
template <typename T, int VEC>
__device__ __forceinline__ void add_old(const T *p, float (&v)[VEC], int valid)
{
#pragma unroll
    for (int i = 0; i < VEC; ++i)
        if (i < valid) {
            float t[1];
            VecIO<T, 1>::load(p + i, t);
            v[i] += t[0];
        }
}

__device__ __forceinline__ void store_scalar(float *p, float v) { *p = v; }
__device__ __forceinline__ void store_scalar(unsigned short *p, float v) { *p = gae::f32_to_bf16(v); }

// store-time epilogue of gae_spmm_csr_ep: v = act(v + bias[f .. f + VEC)) (bias may be NULL; features >= F untouched).
// The test on (bias, act) is wave-uniform: plain launches skip it.
template <int VEC>
__device__ __forceinline__ void epilogue(float (&v)[VEC], const float *__restrict__ bias, int act, int f, int F)
{
    if (bias == nullptr && act == GAE_ACT_IDENTITY) return;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        float y = v[i] + ((bias != nullptr && f + i < F) ? bias[f + i] : 0.f);
        if (act == GAE_ACT_RELU) y = fmaxf(y, 0.f);
        v[i] = y;
    }
}

template <typename T, int VEC, int LPR, int CH, bool SCALED>
__global__ __launch_bounds__(256) void spmm_rowgroup_kernel(
    const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices, int64_t n_rows,
    const T *__restrict__ H, int64_t ldh, T *__restrict__ M, int64_t ldm, int F,
    const float *__restrict__ row_scale, const float *__restrict__ col_scale, int accumulate)
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
        if (accumulate) add_old<T, VEC>(mp + c * TILE, acc[c], F - f);      // GAE_SPMM_ACCUMULATE: M += result
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
// One batch of the gather: NB neighbour rows per owned row, all issued before the first is consumed (predicated,
// no serial tail), then added in slot order.
template <typename T, int VEC, int LPR, int CH, int RPG, int NB, bool SCALED>
__device__ __forceinline__ void gather_batch(const T *__restrict__ H, int64_t ldh, int f0, const bool (&live)[CH],
                                             const int32_t (&j)[RPG][NB], const bool (&ev)[RPG][NB],
                                             const float *__restrict__ col_scale, float (&acc)[RPG][CH][VEC])
{
    constexpr int TILE = LPR * VEC;
    float v[RPG][NB][CH][VEC];
    float cs[RPG][NB];
#pragma unroll
    for (int r = 0; r < RPG; ++r)
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            if (SCALED) cs[r][u] = ev[r][u] ? col_scale[j[r][u]] : 0.f;
            const T *hp = H + int64_t(j[r][u]) * ldh + f0;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (ev[r][u] && live[c]) {
                    VecIO<T, VEC>::load(hp + c * TILE, v[r][u][c]);
                } else {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) v[r][u][c][i] = 0.f;
                }
            }
        }
#pragma unroll
    for (int r = 0; r < RPG; ++r)
#pragma unroll
        for (int u = 0; u < NB; ++u)
            if (ev[r][u]) {
#pragma unroll
                for (int c = 0; c < CH; ++c)
#pragma unroll
                    for (int i = 0; i < VEC; ++i)
                        acc[r][c][i] = SCALED ? fmaf(cs[r][u], v[r][u][c][i], acc[r][c][i]) : acc[r][c][i] + v[r][u][c][i];
            }
}

// Markers of the packed neighbour table (gae_spmm_ell_build)
constexpr int32_t kEllEmpty = -1;      // no neighbour in this slot
constexpr int32_t kEllOverflow = -2;   // last slot: the row has more neighbours, continue from the CSR arrays
constexpr int32_t kEllSkip = -3;       // slot 0: heavy row of the skew plan, produced by the segment kernels

// ---------------------------------------------------------------------------
// v2 "rowgroup2": same ownership (LPR lanes own RPG rows), but
//   * neighbour ids are fetched by ONE coalesced load per group and batch (lane
//     u of the group loads indices[pos + u]) and broadcast with ds_bpermute /
//     readlane instead of every lane re-loading them;
//   * the neighbour rows of a batch (up to NB per owned row, RPG rows) are all
//     issued before the first is consumed, with predication instead of a
//     serial tail loop: one HBM round trip per batch instead of one per edge;
//   * for LPR == 64 the row, its edge range and the neighbour base addresses
//     are wave-uniform (SGPR) values;
//   * ELLW > 0: the first ELLW neighbours of every row come from the packed table
//     ell[row][ELLW] -- ONE load replaces the dependent indptr -> indices chain
//     (launches of a few 10 MB are bounded by that chain, not by bytes: Pubmed
//     F = 500 21 -> 15 us); longer rows continue from the CSR arrays.
// Summation order is unchanged (CSR order), so results are bit-identical to v1.
// ---------------------------------------------------------------------------
template <typename T, int VEC, int LPR, int CH, int RPG, bool SCALED, int ELLW = 0>
__global__ __launch_bounds__(256) void spmm_rowgroup2_kernel(
    const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices, int64_t n_rows,
    const T *__restrict__ H, int64_t ldh, T *__restrict__ M, int64_t ldm, int F,
    const float *__restrict__ row_scale, const float *__restrict__ col_scale, unsigned n_row_blocks,
    unsigned n_ftiles, int xcd_tiled, int tile_w, int skip_deg, int store_pad, int store_mode,
    const int32_t *__restrict__ ell, int accumulate, const float *__restrict__ ep_bias, int ep_act,
    const int32_t *__restrict__ light_desc, int64_t n_light)
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
    const int f0 = ftile * tile_w + lig * VEC;    // tile_w = features per tile (= CH * TILE unless XCD-tiled)

    bool live[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) live[c] = (lig * VEC + c * TILE) < tile_w && (f0 + c * TILE) < F;

    int64_t row[RPG];
    int32_t pos[RPG], end[RPG];
    float acc[RPG][CH][VEC];
#pragma unroll
    for (int r = 0; r < RPG; ++r) {
        row[r] = int64_t(blk) * RPB + r * GPB + grp;
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[r][c][i] = 0.f;
    }

    if (ELLW == 0 && light_desc != nullptr) {
        // ---- light-row list of a skew plan (gae_spmm_plan::light_desc): item -> {row, first edge, end edge}; every
        //      lane group of the wave has edges to gather, no row-pointer loads
#pragma unroll
        for (int r = 0; r < RPG; ++r) {
            const int64_t item = row[r];
            const bool iv = item < n_light;
            const int4 d = *reinterpret_cast<const int4 *>(light_desc + (iv ? item : 0) * 4);
            row[r] = iv ? int64_t(d.x) : n_rows;
            pos[r] = iv ? d.y : 0;
            end[r] = iv ? d.z : 0;
        }
    } else if (ELLW > 0) {
        // ---- packed-table phase: slot k of a row sits in lane k % LPR, register k / LPR of its group
        constexpr int KI = ELLW > LPR ? ELLW / LPR : 1;
        int32_t slot[RPG][KI];
#pragma unroll
        for (int r = 0; r < RPG; ++r)
#pragma unroll
            for (int q = 0; q < KI; ++q) {
                const int k = q * LPR + lig;
                slot[r][q] = (row[r] < n_rows && k < ELLW) ? ell[row[r] * ELLW + k] : kEllEmpty;
            }
#pragma unroll
        for (int b = 0; b < ELLW / NB; ++b) {
            int32_t j[RPG][NB];
            bool ev[RPG][NB];
            bool any = false;
#pragma unroll
            for (int r = 0; r < RPG; ++r)
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int k = b * NB + u;
                    if (LPR == 64) j[r][u] = __builtin_amdgcn_readlane(slot[r][k / LPR], k % LPR);
                    else j[r][u] = __shfl(slot[r][k / LPR], glane0 + k % LPR, 64);
                    ev[r][u] = j[r][u] >= 0;
                    if (u == 0) any = any || ev[r][u];
                }
            // slots fill from the left: an empty first slot of the batch for every row of the wave ends the phase
            if (b > 0 && __builtin_amdgcn_ballot_w64(any) == 0) break;
            gather_batch<T, VEC, LPR, CH, RPG, NB, SCALED>(H, ldh, f0, live, j, ev, col_scale, acc);
        }
#pragma unroll
        for (int r = 0; r < RPG; ++r) {
            int32_t first, last;
            if (LPR == 64) {
                first = __builtin_amdgcn_readlane(slot[r][0], 0);
                last = __builtin_amdgcn_readlane(slot[r][(ELLW - 1) / LPR], (ELLW - 1) % LPR);
            } else {
                first = __shfl(slot[r][0], glane0, 64);
                last = __shfl(slot[r][(ELLW - 1) / LPR], glane0 + (ELLW - 1) % LPR, 64);
            }
            pos[r] = end[r] = 0;
            if (first == kEllSkip) row[r] = n_rows;        // heavy row: produced by the segment kernels (plan)
            if (last == kEllOverflow) {                    // rare: the rest of the row from the CSR arrays
                pos[r] = indptr[row[r]] + (ELLW - 1);
                end[r] = indptr[row[r] + 1];
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < RPG; ++r) {
            const bool rv = row[r] < n_rows;
            pos[r] = rv ? indptr[row[r]] : 0;
            end[r] = rv ? indptr[row[r] + 1] : 0;
            if (end[r] - pos[r] > skip_deg) {  // heavy row: produced by the segment kernels (plan)
                row[r] = n_rows;
                end[r] = pos[r];
            }
        }
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
            const int32_t e = pos[r] + (LPR >= 4 ? (lig & (NB - 1)) : lig);
            myidx[r] = e < end[r] ? indices[e] : 0;
        }
        int32_t j[RPG][NB];
        bool ev[RPG][NB];
#pragma unroll
        for (int r = 0; r < RPG; ++r)
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                if (LPR == 64) j[r][u] = __builtin_amdgcn_readlane(myidx[r], u);
                else j[r][u] = __shfl(myidx[r], glane0 + u, 64);
                ev[r][u] = pos[r] + u < end[r];
            }
        gather_batch<T, VEC, LPR, CH, RPG, NB, SCALED>(H, ldh, f0, live, j, ev, col_scale, acc);
#pragma unroll
        for (int r = 0; r < RPG; ++r) pos[r] = min(pos[r] + NB, end[r]);
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
            if (accumulate) add_old<T, VEC>(mp + c * TILE, acc[r][c], F - f);   // GAE_SPMM_ACCUMULATE: M += result
            epilogue<VEC>(acc[r][c], ep_bias, ep_act, f, F);                     // gae_spmm_csr_ep: + bias, activation
            // store_pad: M's rows are padded to a whole vector and the caller allows the pad columns to be
            // overwritten -> the tail vector is written whole (full 16-byte stores, no partially written sectors)
            if (VEC == 1 || f + VEC <= F || store_pad) {
                // store_mode is a kernel argument (wave-uniform): 0 = plain, 1 = non-temporal, 2 = write-through sc1
                if (store_mode == 2) VecIO<T, VEC>::store_sc1(mp + c * TILE, acc[r][c]);
                else if (store_mode == 1) VecIO<T, VEC>::store_nt(mp + c * TILE, acc[r][c]);
                else VecIO<T, VEC>::store(mp + c * TILE, acc[r][c]);
            } else {
#pragma unroll
                for (int i = 0; i < VEC; ++i)
                    if (f + i < F) store_scalar(mp + c * TILE + i, acc[r][c][i]);
            }
        }
    }
}

// packed neighbour table: slot k of row r at ell[r * W + k]
__global__ __launch_bounds__(256) void ell_build_kernel(const int32_t *__restrict__ indptr,
                                                        const int32_t *__restrict__ indices, int64_t n_rows, int W,
                                                        int skip_deg, int32_t *__restrict__ ell)
{
    const int64_t idx = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (idx >= n_rows * W) return;
    const int64_t row = idx / W;
    const int k = int(idx - row * W);
    const int32_t p = indptr[row], deg = indptr[row + 1] - p;
    int32_t v;
    if (deg > skip_deg) v = k == 0 ? kEllSkip : kEllEmpty;
    else if (deg > W && k == W - 1) v = kEllOverflow;
    else v = k < deg ? indices[p + k] : kEllEmpty;
    ell[idx] = v;
}

// Empty rows of a skew plan with a light-row list: M[row] = act(0 (+ M[row]) + bias) for every row WITHOUT in-edges --
// a pure stream (one lane per 16-byte vector; the row pointers come from L1).  Rows with edges are written by the
// list / segment / combine kernels.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void spmm_fill_empty_kernel(const int32_t *__restrict__ indptr, int64_t n_rows,
                                                              T *__restrict__ M, int64_t ldm, int F, int nvec,
                                                              int store_pad, int accumulate,
                                                              const float *__restrict__ ep_bias, int ep_act,
                                                              const uint8_t *__restrict__ skip_rows)
{
    // one lane group of G = 2^k >= nvec lanes per row (G <= 16; wider rows: the group walks them): the row's two
    // pointers are read once per group, the stores of a group are one contiguous piece of the row
    int G = 1;
    while (G < nvec && G < 16) G <<= 1;
    const int lig = int(threadIdx.x) % G;
    const int64_t stride = int64_t(gridDim.x) * (256 / G);
    for (int64_t row = int64_t(blockIdx.x) * (256 / G) + threadIdx.x / G; row < n_rows; row += stride) {
        if (indptr[row + 1] != indptr[row]) continue;
        if (skip_rows != nullptr && skip_rows[row]) continue;        // GAE_SPMM_SKIP_ROWS: nobody reads this row
      for (int fv = lig; fv < nvec; fv += G) {
        const int f = fv * VEC;
        T *mp = M + row * ldm + f;
        float v[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i] = 0.f;
        if (accumulate) add_old<T, VEC>(mp, v, F - f);
        epilogue<VEC>(v, ep_bias, ep_act, f, F);
        if (VEC == 1 || f + VEC <= F || store_pad) {
            VecIO<T, VEC>::store_nt(mp, v);
        } else {
#pragma unroll
            for (int i = 0; i < VEC; ++i)
                if (f + i < F) store_scalar(mp + i, v[i]);
        }
      }
    }
}

// tuning knobs (gae_tuning_set): read-mostly process-wide integers
gae::Knob g_spmm_variant{2};   // 1 = v1 rowgroup, 2 = v2 rowgroup2
gae::Knob g_spmm_rpg{0};       // rows per lane group (v2): 0 = auto (2 for launches of >= 32768 waves, else 1), 1, 2
gae::Knob g_spmm_parts{7};     // "spmm_parts" (experiments): which parts of a skew-plan launch run: 1 light rows | 2 segmented rows | 4 pinned rows
gae::Knob g_spmm_light{1};     // "spmm_light": use the plan's light-row list (1) or sweep all rows with the row-group kernel (0); bit-identical
gae::Knob g_spmm_desc{1};      // "spmm_desc": segment descriptors / identity segments (1) or the plan's index chain (0); bit-identical
gae::Knob g_spmm_hot{1};       // "spmm_hot": use the plan's hot-column tags (streaming loads of cold rows); 0 = plain loads
gae::Knob g_spmm_nt{-1};       // store policy of M (v2): -1 = auto (sc1 under feature tiles, else nt), 0 plain, 1 non-temporal, 2 write-through sc1
gae::Knob g_spmm_tile_vecs{0}; // 16-byte vectors per XCD feature tile: 0 = auto when GAE_SPMM_TILE is set, -1 = never, > 0 = forced
gae::Knob g_spmm_ell{1};       // the plan's packed neighbour table: 0 = ignore it, 1 = spmm_ell.hip kernels (row-group
                          // kernel when they cannot run the launch), 2 = row-group kernel only

constexpr int kEllWidth = 16;

template <typename T, int VEC, int LPR, int CH, int RPG>
int launch_rowgroup2(const int32_t *indptr, const int32_t *indices, int64_t n_rows, const T *H, int64_t ldh, T *M,
                     int64_t ldm, int F, const float *rs, const float *cs, int st, int tile_vecs, int skip_deg,
                     int flags, const int32_t *ell, hipStream_t s, const float *ep_bias = nullptr,
                     int ep_act = GAE_ACT_IDENTITY, const int32_t *light_desc = nullptr, int64_t n_light = 0)
{
    const int store_pad = ((flags & GAE_SPMM_STORE_PAD) && (F + VEC - 1) / VEC * VEC <= ldm) ? 1 : 0;
    constexpr int RPB = (256 / LPR) * RPG;
    const int nvec = (F + VEC - 1) / VEC;
    const bool tiled = tile_vecs > 0;
    const int tw = tiled ? tile_vecs : LPR * CH;   // 16-byte vectors per feature tile
    const int64_t n_items = light_desc ? n_light : n_rows;
    if (n_items == 0) return GAE_OK;
    const unsigned nrb = unsigned((n_items + RPB - 1) / RPB), nft = unsigned((nvec + tw - 1) / tw);
    const dim3 grid = tiled ? dim3(gae::kNumXcd * ((nft + gae::kNumXcd - 1) / gae::kNumXcd) * nrb) : dim3(nrb, nft);
    const int xt = tiled ? 1 : 0;
#define GAE_L2(SC, EW)                                                                                              \
    hipLaunchKernelGGL((spmm_rowgroup2_kernel<T, VEC, LPR, CH, RPG, SC, EW>), grid, dim3(256), 0, s, indptr, indices,  \
                       n_rows, H, ldh, M, ldm, F, rs, cs, nrb, nft, xt, tw * VEC, skip_deg, store_pad, store_mode, ell,        \
                       (flags & GAE_SPMM_ACCUMULATE) ? 1 : 0, ep_bias, ep_act, light_desc, n_light)
    constexpr int EW = VEC > 1 ? kEllWidth : 0;
    // store policy: 0 plain, 1 non-temporal, 2 write-through sc1; auto (-1) = sc1 under XCD feature tiles (the
    // output stream must not evict the tile's L2-resident slice of H), non-temporal otherwise
    const int store_mode = sizeof(T) != 4 ? 0 : st >= 0 ? st : (tiled ? 2 : 1);
    if (VEC > 1 && ell) {
        if (rs || cs) GAE_L2(true, EW); else GAE_L2(false, EW);
    } else {
        if (rs || cs) GAE_L2(true, 0); else GAE_L2(false, 0);
    }
#undef GAE_L2
    GAE_CHECK_LAUNCH("spmm_rowgroup2_kernel");
    return GAE_OK;
}

// XCD feature tiling (GAE_SPMM_TILE): tile width in 16-byte vectors, 0 = do not tile.  8k tiles of whole 128-byte
// lines so that every XCD sweeps the same number of tiles; the widest tile whose slice of H stays L2-resident.
inline int auto_tile_vecs(int nvec, int64_t n_cols)
{
    const int64_t budget = int64_t(21) << 18;   // 5.25 MiB: a 4 MiB L2 keeps most of a slice slightly above its size
    for (int k = 1; k <= 16; ++k) {
        int tv = (nvec + 8 * k - 1) / (8 * k);
        tv = (tv + 7) / 8 * 8;
        if (tv < 8) return 0;
        if (tv <= 256 && n_cols * tv * 16 <= budget) return tv;
    }
    return 0;
}

template <typename T, int VEC>
int dispatch_rowgroup2(const int32_t *indptr, const int32_t *indices, int64_t n_rows, const T *H, int64_t ldh, T *M,
                       int64_t ldm, int F, const float *rs, const float *cs, int rpg, int st, int64_t n_cols,
                       int skip_deg, int flags, const int32_t *ell, int ell_width, hipStream_t s,
                       const float *ep_bias = nullptr, int ep_act = GAE_ACT_IDENTITY,
                       const int32_t *light_desc = nullptr, int64_t n_light = 0)
{
    const int nvec = (F + VEC - 1) / VEC;
    int tile_vecs = 0;
    if (ep_bias != nullptr || ep_act != GAE_ACT_IDENTITY) ell = nullptr;   // the epilogue lives in the row-group kernel
    if (light_desc != nullptr) ell = nullptr;
    // XCD feature tiles (GAE_SPMM_TILE): wide rows, poor gather locality, rows made of whole 128-byte lines
    if (VEC > 1 && nvec > 16 && g_spmm_tile_vecs >= 0) {
        const bool lines = (ldh * sizeof(T)) % 128 == 0 && (ldm * sizeof(T)) % 128 == 0 &&
                           reinterpret_cast<uintptr_t>(H) % 128 == 0 && reinterpret_cast<uintptr_t>(M) % 128 == 0;
        const int tv = g_spmm_tile_vecs > 0 ? g_spmm_tile_vecs
                                            : ((flags & GAE_SPMM_TILE) && lines ? auto_tile_vecs(nvec, n_cols) : 0);
        if (tv > 0 && (nvec + tv - 1) / tv >= 2) tile_vecs = tv;
    }
    // packed-table kernels (spmm_ell.hip) whenever the plan carries a table they can use
    if (VEC > 1 && ell && g_spmm_ell == 1 && !(flags & GAE_SPMM_ACCUMULATE)) {
        const int tv = tile_vecs > 0 ? tile_vecs : (nvec < 64 ? nvec : 64);
        if (tv <= 64 && gae::spmm_ell_usable(n_cols, ldh, int(sizeof(T)), ell_width, tv)) {
            const int store_pad = ((flags & GAE_SPMM_STORE_PAD) && int64_t(nvec) * VEC <= ldm) ? 1 : 0;
            // write-through (sc1) under feature tiles: the launch itself is 0.4 us faster on Pubmed and the Linear that
            // reads M next finds it in the memory-side cache (non-temporal stores: 22.6 us instead of 17 us for that
            // launch inside a Pubmed step); untiled launches (molecule batches) are faster with non-temporal stores
            const int store_mode = sizeof(T) != 4 ? 0 : st >= 0 ? st : (tile_vecs > 0 ? 2 : 1);
            return gae::spmm_ell_launch(indptr, indices, ell, ell_width, n_rows, n_cols, H, ldh, M, ldm, F,
                                        sizeof(T) == 4 ? GAE_F32 : GAE_BF16, rs, cs, tv, tile_vecs > 0 ? 1 : 0,
                                        store_pad, store_mode, s);
        }
    }
    if (ell_width != kEllWidth) ell = nullptr;     // the row-group kernel reads 16-wide tables only
#define GAE_RG2(LPR, CH)                                                                                          \
    do {                                                                                                          \
        /* two rows per lane group halve the wave count: pays once the launch is many occupancy rounds long   \
         * (ZINC set 362 -> 288 us), costs parallelism on short ones (Pubmed F = 32: 4.0 -> 5.2 us) */           \
        const int64_t tw_ = tile_vecs > 0 ? tile_vecs : LPR * CH;      /* vectors per feature tile */           \
        const int64_t waves_ = (light_desc ? n_light : n_rows) * LPR / 64 * ((nvec + tw_ - 1) / tw_);            \
        const int rpg_ = rpg > 0 ? rpg : (waves_ >= 32768 ? 2 : 1);                                               \
        if (rpg_ >= 2 && CH == 1)                                                                                 \
            return launch_rowgroup2<T, VEC, LPR, CH, 2>(indptr, indices, n_rows, H, ldh, M, ldm, F, rs, cs, st,   \
                                                        tile_vecs, skip_deg, flags, ell, s, ep_bias, ep_act,      \
                                                        light_desc, n_light);                                     \
        return launch_rowgroup2<T, VEC, LPR, CH, 1>(indptr, indices, n_rows, H, ldh, M, ldm, F, rs, cs, st,       \
                                                    tile_vecs, skip_deg, flags, ell, s, ep_bias, ep_act,          \
                                                    light_desc, n_light);                                         \
    } while (0)
    if (tile_vecs > 0) {
        const int tv = tile_vecs;
        if (tv <= 4) GAE_RG2(4, 1);
        if (tv <= 8) GAE_RG2(8, 1);
        if (tv <= 16) GAE_RG2(16, 1);
        if (tv <= 32) GAE_RG2(32, 1);
        if (tv <= 64) GAE_RG2(64, 1);
        if (tv <= 128) GAE_RG2(64, 2);
        if (tv <= 256) GAE_RG2(64, 4);
        tile_vecs = 0;
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
                    int64_t ldm, int F, const float *rs, const float *cs, int acc, hipStream_t s)
{
    constexpr int RPB = 256 / LPR;
    const int nvec = (F + VEC - 1) / VEC;
    const unsigned gx = unsigned((n_rows + RPB - 1) / RPB);
    const unsigned gy = unsigned((nvec + LPR * CH - 1) / (LPR * CH));
    if (rs || cs)
        hipLaunchKernelGGL((spmm_rowgroup_kernel<T, VEC, LPR, CH, true>), dim3(gx, gy), dim3(256), 0, s, indptr,
                           indices, n_rows, H, ldh, M, ldm, F, rs, cs, acc);
    else
        hipLaunchKernelGGL((spmm_rowgroup_kernel<T, VEC, LPR, CH, false>), dim3(gx, gy), dim3(256), 0, s, indptr,
                           indices, n_rows, H, ldh, M, ldm, F, rs, cs, acc);
    GAE_CHECK_LAUNCH("spmm_rowgroup_kernel");
    return GAE_OK;
}

template <typename T, int VEC>
int dispatch_rowgroup(const int32_t *indptr, const int32_t *indices, int64_t n_rows, const T *H, int64_t ldh, T *M,
                      int64_t ldm, int F, const float *rs, const float *cs, int acc, hipStream_t s)
{
    const int nvec = (F + VEC - 1) / VEC;
#define GAE_RG(LPR, CH) return launch_rowgroup<T, VEC, LPR, CH>(indptr, indices, n_rows, H, ldh, M, ldm, F, rs, cs, acc, s)
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

// ---------------------------------------------------------------------------
// Block-diagonal graphs (batched molecules, dgl.batch of train_inductive.py:34): every thread block owns a run of
// consecutive rows that is CLOSED under adjacency (whole member graphs), so all neighbour rows of its rows live in
// the same contiguous slice of H.  The slice is streamed into LDS with perfectly coalesced 16-byte loads (every
// HBM byte is read exactly once, whatever the row width: F = 39 / ld = 40 rows straddle 128-B lines and are
// expensive to gather from L1/L2), the block's index slice is staged next to it, and the gathers run at LDS
// speed.  Memory round trips per block: one for (H slice, indptr), one for the indices -- instead of three
// dependent trips per row.  Sums stay in CSR order: bit-identical to the row-group kernel.
// ---------------------------------------------------------------------------
template <int LPR, bool SCALED, int TI>
__global__ __launch_bounds__(256) void spmm_blockdiag_kernel(
    const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices, const int32_t *__restrict__ block_ptr,
    const int32_t *__restrict__ block_eptr, const float *__restrict__ H, int64_t ldh, float *__restrict__ M,
    int64_t ldm, int F, const float *__restrict__ row_scale, const float *__restrict__ col_scale, int max_rows,
    int max_edges, int store_pad)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *tile = lds;                                              // [max_rows][ldh]
    int32_t *iptr = reinterpret_cast<int32_t *>(lds + int64_t(max_rows) * ldh);   // [max_rows + 1]
    int32_t *idx = iptr + ((max_rows + 1 + 3) & ~3);               // [max_edges]
    constexpr int GPB = 256 / LPR;
    const int tid = threadIdx.x, lig = tid % LPR, grp = tid / LPR;
    // block descriptor (row run and its edge run: block_eptr[b] = indptr[block_ptr[b]], precomputed so that the
    // index slice does not have to wait for the row pointers) -- wave-uniform scalar loads
    const int r0 = block_ptr[blockIdx.x], r1 = block_ptr[blockIdx.x + 1];
    const int32_t e0 = block_eptr[blockIdx.x], e1 = block_eptr[blockIdx.x + 1];
    const int nr = r1 - r0;
    if (nr <= 0) return;          // empty run (block-uniform): nothing to stage, nothing to write
    const int ne = min(e1 - e0, max_edges);
    // ---- one bulk round trip: H slice (contiguous nr * ldh floats), row pointers, neighbour ids (as LOCAL row
    //      offsets).  Every load is issued (branch-free, clamped index) into registers BEFORE the first LDS
    //      write: a plain `for (i = tid; i < n; i += 256) lds[i] = g[i]` is compiled to load / vmcnt(0) /
    //      ds_write per iteration, i.e. one serialised HBM round trip per 4 KiB.
    {
        constexpr int EI = 4, PI = 2;                 // idx pieces (max_edges <= 1024), iptr pieces (rows <= 511)
        const float4 *src = reinterpret_cast<const float4 *>(H + int64_t(r0) * ldh);
        const int n4 = int(int64_t(nr) * ldh / 4);
        float4 t4[TI];
        int32_t pv[PI], iv[EI];
#pragma unroll
        for (int q = 0; q < TI; ++q) t4[q] = src[min(tid + 256 * q, n4 - 1)];
#pragma unroll
        for (int q = 0; q < PI; ++q) pv[q] = indptr[r0 + min(tid + 256 * q, nr)];
#pragma unroll
        for (int q = 0; q < EI; ++q) iv[q] = indices[e0 + min(tid + 256 * q, max(e1 - e0 - 1, 0))];
        float4 *dst = reinterpret_cast<float4 *>(tile);
#pragma unroll
        for (int q = 0; q < TI; ++q)
            if (tid + 256 * q < n4) dst[tid + 256 * q] = t4[q];
#pragma unroll
        for (int q = 0; q < PI; ++q)
            if (tid + 256 * q <= nr) iptr[tid + 256 * q] = pv[q];
#pragma unroll
        for (int q = 0; q < EI; ++q)
            if (tid + 256 * q < ne) idx[tid + 256 * q] = iv[q] - r0;
    }
    __syncthreads();
    // ---- phase C: gather from LDS; group `grp` owns rows grp, grp + GPB, ...
    const int f0 = lig * 4;
    const bool lv = f0 < F;
    for (int r = grp; r < nr; r += GPB) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const int32_t s = iptr[r] - e0, e = iptr[r + 1] - e0;
        for (int32_t k = s; k < e; ++k) {
            const int j = k < ne ? idx[k] : indices[e0 + k] - r0;      // overflow edges: straight from global
            if (lv) {
                const float4 v = *reinterpret_cast<const float4 *>(tile + int64_t(j) * ldh + f0);
                const float c = SCALED ? col_scale[r0 + j] : 1.f;
                acc[0] = SCALED ? fmaf(c, v.x, acc[0]) : acc[0] + v.x;
                acc[1] = SCALED ? fmaf(c, v.y, acc[1]) : acc[1] + v.y;
                acc[2] = SCALED ? fmaf(c, v.z, acc[2]) : acc[2] + v.z;
                acc[3] = SCALED ? fmaf(c, v.w, acc[3]) : acc[3] + v.w;
            }
        }
        if (lv) {
            if (SCALED) {
                const float rs = row_scale[r0 + r];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] *= rs;
            }
            float *mp = M + int64_t(r0 + r) * ldm + f0;
            if (f0 + 4 <= F || store_pad) {
                typedef float f4 __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(f4{acc[0], acc[1], acc[2], acc[3]}, reinterpret_cast<f4 *>(mp));
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (f0 + q < F) mp[q] = acc[q];
            }
        }
    }
}

// one wave per segment; partial[s][0..F)
template <typename T, int VEC, int LPR, int CH, bool SCALED>
__global__ __launch_bounds__(256) void spmm_segment_kernel(
    const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices, const T *__restrict__ H, int64_t ldh,
    int F, const float *__restrict__ col_scale, const int32_t *__restrict__ heavy_rows,
    const int32_t *__restrict__ heavy_seg_base, const int32_t *__restrict__ seg_heavy, int64_t n_segments, int seg,
    float *__restrict__ partial, int ldp, const int32_t *__restrict__ hot_indices,
    const float *__restrict__ row_scale, T *__restrict__ M, int64_t ldm, int accumulate, int direct,
    const int32_t *__restrict__ seg_desc, const float *__restrict__ ep_bias, int ep_act)
{
    // hot_indices (plan, optional): the column ids again, with the sign bit set on the columns that are gathered
    // most often.  Rows of the other columns are loaded with the non-temporal hint, so the few thousand hub rows
    // of a power-law graph stay in the XCD's L2 instead of being evicted by rows that are touched once
    // (RMAT s24, F = 32: L2 hit rate of this kernel 14 % -> ~30 %, 8.27 -> 7.40 ms; values are unaffected).
    constexpr int G = 64 / LPR;       // lane groups per wave = edges gathered per load instruction
    constexpr int TILE = LPR * VEC;
    constexpr int NB = 8;             // edges in flight per group
    const int lane = threadIdx.x & 63, lig = lane % LPR, g = lane / LPR;
    const int64_t sidx = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (sidx >= n_segments) return;
    // A wave of this kernel lives for a handful of memory round trips, one of which is the gather itself: the chain
    // segment -> heavy slot -> row -> edge range -> column ids in front of it set the launch time of the 9..256-edge
    // rows of RMAT s24 (2.2 M waves of ~35 edges).  Two shorter forms: the plan's 16-byte segment descriptor {row,
    // first edge, end edge, only segment of its row} (gae_spmm_plan_build_rows), or -- heavy_rows == NULL -- segment s IS
    // row s of the CSR handed in (the virtual rows of the XCD-pinned part).
    int64_t row;
    int32_t e0, e1;
    bool single;
    if (seg_desc) {
        const int4 d = *reinterpret_cast<const int4 *>(seg_desc + sidx * 4);
        row = d.x; e0 = d.y; e1 = d.z;
        single = direct && d.w;
    } else if (!heavy_rows) {
        row = sidx; e0 = indptr[sidx]; e1 = indptr[sidx + 1];
        single = direct != 0;
    } else {
        const int h = seg_heavy[sidx];
        row = heavy_rows[h];
        const int k = int(sidx - heavy_seg_base[h]);
        const int32_t r_end = indptr[row + 1];
        e0 = indptr[row] + k * seg;
        e1 = min(e0 + seg, r_end);
        single = direct && k == 0 && e1 == r_end;   // the row's only segment: its sum goes straight to M (most heavy
        //                                             rows of a power-law graph; the combine kernel skips them)
    }
    const int f0 = blockIdx.y * (CH * TILE) + lig * VEC;
    bool live[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) live[c] = (f0 + c * TILE) < F;
    float acc[CH][VEC];
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[c][i] = 0.f;

    for (int32_t base = e0; base < e1; base += 64) {
        const int32_t myidx = base + lane < e1 ? (hot_indices ? hot_indices[base + lane] : indices[base + lane])
                                               : 0;                            // 64 neighbour ids, coalesced
#pragma unroll
        for (int ub = 0; ub < LPR; ub += NB) {        // 64 / G = LPR edges per group and index batch
            if (base + ub * G >= e1) break;
            float v[NB][CH][VEC];
            float cs[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int slot = (ub + u) * G + g;    // edge handled by this group
                const bool ev = (ub + u) < LPR && base + slot < e1;
                const int32_t jt = __shfl(myidx, slot & 63, 64);
                const bool cold = hot_indices != nullptr && jt >= 0;
                const int32_t j = jt & 0x7fffffff;
                if (SCALED) cs[u] = ev ? col_scale[j] : 0.f;
                const T *hp = H + int64_t(j) * ldh + f0;
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    if (ev && live[c]) {
                        if (cold) VecIO<T, VEC>::load_nt(hp + c * TILE, v[u][c]);
                        else VecIO<T, VEC>::load(hp + c * TILE, v[u][c]);
                    } else {
#pragma unroll
                        for (int i = 0; i < VEC; ++i) v[u][c][i] = 0.f;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < NB; ++u)
#pragma unroll
                for (int c = 0; c < CH; ++c)
#pragma unroll
                    for (int i = 0; i < VEC; ++i)
                        acc[c][i] = SCALED ? fmaf(cs[u], v[u][c][i], acc[c][i]) : acc[c][i] + v[u][c][i];
        }
    }
    // fixed butterfly across the G groups (lanes with equal lig)
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[c][i] += __shfl_xor(acc[c][i], off, 64);
    if (g == 0 && single) {                 // same arithmetic as spmm_combine_kernel with one partial: 0 + sum, scale, (+ old)
        const float rs = row_scale ? row_scale[row] : 1.f;
        T *mp = M + row * ldm + f0;
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int i = 0; i < VEC; ++i)
                if (f0 + c * TILE + i < F) {
                    float sv[1] = {(0.f + acc[c][i]) * rs};
                    if (accumulate) {
                        float t[1];
                        VecIO<T, 1>::load(mp + c * TILE + i, t);
                        sv[0] += t[0];
                    }
                    epilogue<1>(sv, ep_bias, ep_act, f0 + c * TILE + i, F);
                    store_scalar(mp + c * TILE + i, sv[0]);
                }
    } else if (g == 0) {
        float *pp = partial + sidx * ldp + f0;
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int i = 0; i < VEC; ++i)
                if (f0 + c * TILE + i < F) pp[c * TILE + i] = acc[c][i];
    }
}

// M[row] = rs[row] * sum_k partial[base + k]   (segment order), for heavy rows of MORE than one segment (the
// segment kernel writes single-segment rows itself).  One lane per (heavy row, 4 features) -- a whole wave per row,
// most of which return at once, made this launch 0.7 ms on RMAT s24.  (Plan slots are handed out by two independent
// atomic counters: heavy_seg_base is not monotonic in h, so the segment count has to come from indptr.)
template <typename T>
__global__ __launch_bounds__(256) void spmm_combine_kernel(const int32_t *__restrict__ indptr,
                                                           const int32_t *__restrict__ heavy_rows,
                                                           const int32_t *__restrict__ heavy_seg_base,
                                                           int64_t n_heavy, int seg,
                                                           const float *__restrict__ partial, int ldp, int F,
                                                           const float *__restrict__ row_scale, T *__restrict__ M,
                                                           int64_t ldm, int accumulate, int lanes_per_row,
                                                           const float *__restrict__ ep_bias, int ep_act)
{
    const int64_t gt = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const int64_t h = gt / lanes_per_row;
    const int f0 = int(gt - h * lanes_per_row) * 4;
    if (h >= n_heavy || f0 >= F) return;
    const int64_t base = heavy_seg_base[h];
    const int64_t row = heavy_rows[h];
    const int ns = (indptr[row + 1] - indptr[row] + seg - 1) / seg;
    if (ns <= 1) return;                          // written by the segment kernel itself
    const float rs = row_scale ? row_scale[row] : 1.f;
    const float *pp = partial + base * ldp + f0;   // ldp is a multiple of 4: aligned float4 pieces
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 4 <= ns; k += 4) {                  // 4 independent loads per trip, added in segment order
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4 *>(pp + int64_t(k + u) * ldp);
#pragma unroll
        for (int u = 0; u < 4; ++u) { s[0] += v[u].x; s[1] += v[u].y; s[2] += v[u].z; s[3] += v[u].w; }
    }
    for (; k < ns; ++k) {
        const float4 v = *reinterpret_cast<const float4 *>(pp + int64_t(k) * ldp);
        s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (f0 + i >= F) break;
        float sv[1] = {s[i] * rs};
        if (accumulate) {
            float t[1];
            VecIO<T, 1>::load(M + row * ldm + f0 + i, t);
            sv[0] += t[0];
        }
        epilogue<1>(sv, ep_bias, ep_act, f0 + i, F);
        store_scalar(M + row * ldm + f0 + i, sv[0]);
    }
}

// XCD-pinned part of a plan ("homed" rows, gae_spmm_plan::vh_*): M[row] = rs[row] * sum of the row's virtual-row
// partials in a fixed order.  One WAVE per row: lane = (slice, 4 features); slice s of the 64 / LPRC adds the
// partials s, s + slices, ... of the plan's (home, chunk) order with 4 loads in flight, then the slices meet in a
// fixed butterfly.  (One lane group per row made the launch as long as the longest row's serial walk: RMAT s24's
// largest row has ~1000 partials, 0.15 ms for a 0.2 GB stream.)
template <typename T, int LPRC>
__global__ __launch_bounds__(256) void spmm_vh_combine_kernel(const int32_t *__restrict__ vh_rows,
                                                              const int32_t *__restrict__ part_ptr,
                                                              const int32_t *__restrict__ part_pos, int64_t n_vh,
                                                              const float *__restrict__ partial, int ldp, int F,
                                                              const float *__restrict__ row_scale,
                                                              T *__restrict__ M, int64_t ldm, int accumulate,
                                                              const float *__restrict__ ep_bias, int ep_act)
{
    constexpr int SLICES = 64 / LPRC;
    const int lane = threadIdx.x & 63;
    const int64_t r = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (r >= n_vh) return;
    const int sl = lane / LPRC, f0 = (int(blockIdx.y) * LPRC + lane % LPRC) * 4;     // blockIdx.y: rows wider than 256 features
    const bool fv = f0 < F;
    const int32_t p0 = part_ptr[r], p1 = part_ptr[r + 1];
    const int64_t row = vh_rows[r];
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (fv) {
        int32_t k = p0 + sl;
        for (; k + 3 * SLICES < p1; k += 4 * SLICES) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                v[u] = *reinterpret_cast<const float4 *>(partial + int64_t(part_pos[k + u * SLICES]) * ldp + f0);
#pragma unroll
            for (int u = 0; u < 4; ++u) { s[0] += v[u].x; s[1] += v[u].y; s[2] += v[u].z; s[3] += v[u].w; }
        }
        for (; k < p1; k += SLICES) {
            const float4 v = *reinterpret_cast<const float4 *>(partial + int64_t(part_pos[k]) * ldp + f0);
            s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
        }
    }
#pragma unroll
    for (int off = LPRC; off < 64; off <<= 1)
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i] += __shfl_xor(s[i], off, 64);
    if (sl != 0 || !fv) return;
    const float rs = row_scale ? row_scale[row] : 1.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (f0 + i >= F) break;
        float sv[1] = {s[i] * rs};
        if (accumulate) {
            float t[1];
            VecIO<T, 1>::load(M + row * ldm + f0 + i, t);
            sv[0] += t[0];
        }
        epilogue<1>(sv, ep_bias, ep_act, f0 + i, F);
        store_scalar(M + row * ldm + f0 + i, sv[0]);
    }
}

template <typename T, int VEC, int LPR, int CH>
int launch_segments(const int32_t *indptr, const int32_t *indices, const T *H, int64_t ldh, int F, const float *cs,
                    const gae_spmm_plan *plan, float *partial, int ldp, const float *rs, T *M, int64_t ldm,
                    int accumulate, int direct, hipStream_t s, const float *ep_bias, int ep_act)
{
    const int nvec = (F + VEC - 1) / VEC;
    const dim3 grid(unsigned((plan->n_segments + 3) / 4), unsigned((nvec + LPR * CH - 1) / (LPR * CH)));
    // device-built plans: the segments read a compact copy of their ids (seg_desc indexes it), tagged or not
    const int32_t *hot = g_spmm_hot ? plan->hot_indices : nullptr;
    const int32_t *desc = g_spmm_desc ? plan->seg_desc : nullptr;
    if (plan->mid_indices) {
        indices = plan->mid_indices;
        hot = (plan->mid_tagged && g_spmm_hot) ? plan->mid_indices : nullptr;
        desc = plan->seg_desc;
    }
    if (cs)
        hipLaunchKernelGGL((spmm_segment_kernel<T, VEC, LPR, CH, true>), grid, dim3(256), 0, s, indptr, indices, H, ldh,
                           F, cs, plan->heavy_rows, plan->heavy_seg_base, plan->seg_heavy, plan->n_segments,
                           plan->segment_edges, partial, ldp, hot, rs, M, ldm,
                           accumulate, direct, desc, ep_bias, ep_act);
    else
        hipLaunchKernelGGL((spmm_segment_kernel<T, VEC, LPR, CH, false>), grid, dim3(256), 0, s, indptr, indices, H,
                           ldh, F, cs, plan->heavy_rows, plan->heavy_seg_base, plan->seg_heavy, plan->n_segments,
                           plan->segment_edges, partial, ldp, hot, rs, M, ldm,
                           accumulate, direct, desc, ep_bias, ep_act);
    GAE_CHECK_LAUNCH("spmm_segment_kernel");
    return GAE_OK;
}

template <typename T, int VEC>
int dispatch_segments(const int32_t *indptr, const int32_t *indices, const T *H, int64_t ldh, int F, const float *cs,
                      const gae_spmm_plan *plan, float *partial, int ldp, const float *rs, T *M, int64_t ldm,
                      int accumulate, int direct, hipStream_t s, const float *ep_bias = nullptr,
                      int ep_act = GAE_ACT_IDENTITY)
{
    const int nvec = (F + VEC - 1) / VEC;
#define GAE_SEG(LPR, CH)                                                                                              \
    return launch_segments<T, VEC, LPR, CH>(indptr, indices, H, ldh, F, cs, plan, partial, ldp, rs, M, ldm, accumulate, \
                                            direct, s, ep_bias, ep_act)
    if (nvec <= 4) GAE_SEG(4, 1);
    if (nvec <= 8) GAE_SEG(8, 1);
    if (nvec <= 16) GAE_SEG(16, 1);
    if (nvec <= 32) GAE_SEG(32, 1);
    if (nvec <= 64) GAE_SEG(64, 1);
    GAE_SEG(64, 2);
#undef GAE_SEG
}

inline int64_t align256(int64_t x) { return (x + 255) / 256 * 256; }
inline int plan_ldp(int64_t F) { return int((F + 3) / 4 * 4); }

template <typename T, int VEC>
int run_spmm(const int32_t *indptr, const int32_t *indices, int64_t n_rows, int64_t n_cols, const T *h, int64_t ldh,
             T *m, int64_t ldm, int f, const float *rs, const float *cs, bool vec, const gae_spmm_plan *plan,
             void *workspace, int flags, hipStream_t s, const float *ep_bias = nullptr, int ep_act = GAE_ACT_IDENTITY)
{
    const bool epi = ep_bias != nullptr || ep_act != GAE_ACT_IDENTITY;
    const bool heavy = plan && plan->n_heavy > 0;
    const bool homed = plan && plan->vh_n_virtual > 0;
    const int skip = (heavy || homed) ? plan->threshold : 0x7fffffff;
    const int min_f = vec ? (sizeof(T) == 4 ? 12 : 24) : 3;
    int rc = GAE_OK;
    const bool listed = (heavy || homed) && plan->light_desc != nullptr && g_spmm_light && f > min_f;
    if ((heavy || homed) && !(g_spmm_parts & 1)) {
    } else if (listed) {
        // empty rows: a pure stream; rows with 1 .. threshold edges: the plan's list (every lane group has work)
        const int nvec = (f + VEC - 1) / VEC;
        const int store_pad = ((flags & GAE_SPMM_STORE_PAD) && int64_t(nvec) * VEC <= ldm) ? 1 : 0;
        const int acc_f = (flags & GAE_SPMM_ACCUMULATE) ? 1 : 0;
        // (skip_rows covering EVERY row without edges -- plan->reserved2 = 1, the caller's word -- leaves the stream nothing)
        const bool nothing_to_fill = (flags & GAE_SPMM_SKIP_ROWS) && plan->skip_rows != nullptr && plan->reserved2 == 1;
        if ((!acc_f || epi) && !nothing_to_fill) {
            int fg = 1;
            while (fg < nvec && fg < 16) fg <<= 1;
            const int64_t want = (n_rows + 256 / fg - 1) / (256 / fg);
            hipLaunchKernelGGL((spmm_fill_empty_kernel<T, VEC>), dim3(unsigned(want < 16384 ? want : 16384)), dim3(256), 0, s,
                               indptr, n_rows, m, ldm, f, nvec, store_pad, acc_f, ep_bias, ep_act,
                               (flags & GAE_SPMM_SKIP_ROWS) ? plan->skip_rows : nullptr);
            GAE_CHECK_LAUNCH("spmm_fill_empty_kernel");
        }
        rc = dispatch_rowgroup2<T, VEC>(indptr, indices, n_rows, h, ldh, m, ldm, f, rs, cs, g_spmm_rpg, g_spmm_nt, n_cols,
                                        skip, flags, nullptr, 0, s, ep_bias, ep_act, plan->light_desc, plan->n_light);
    } else if ((g_spmm_variant == 2 || epi) && f > min_f)
        rc = dispatch_rowgroup2<T, VEC>(indptr, indices, n_rows, h, ldh, m, ldm, f, rs, cs, g_spmm_rpg,
                                        g_spmm_nt, n_cols, skip, flags,
                                        (plan && g_spmm_ell) ? plan->ell : nullptr, plan ? plan->ell_width : 0, s,
                                        ep_bias, ep_act);
    else {
        GAE_REQUIRE(!epi, GAE_E_RANGE, "gae_spmm_csr_ep: needs F > %d for this layout", min_f);
        GAE_REQUIRE(!heavy && !homed, GAE_E_RANGE, "gae_spmm_csr: a skew plan needs F > %d for this layout", min_f);
        rc = dispatch_rowgroup<T, VEC>(indptr, indices, n_rows, h, ldh, m, ldm, f, rs, cs,
                                       (flags & GAE_SPMM_ACCUMULATE) ? 1 : 0, s);
    }
    if (rc || (!heavy && !homed)) return rc;
    float *partial = static_cast<float *>(workspace);
    const int ldp = plan_ldp(f);
    const int acc_flag = (flags & GAE_SPMM_ACCUMULATE) ? 1 : 0;
    if (heavy && (g_spmm_parts & 2)) {
        rc = dispatch_segments<T, VEC>(indptr, indices, h, ldh, f, cs, plan, partial, ldp, rs, m, ldm, acc_flag, 1, s,
                                       ep_bias, ep_act);
        if (rc) return rc;
        const int lanes_per_row = (f + 3) / 4;
        const int64_t combine_threads = plan->n_heavy * lanes_per_row;
        hipLaunchKernelGGL((spmm_combine_kernel<T>), dim3(unsigned((combine_threads + 255) / 256)), dim3(256), 0, s,
                           indptr, plan->heavy_rows, plan->heavy_seg_base, plan->n_heavy, plan->segment_edges, partial, ldp, f, rs,
                           m, ldm, acc_flag, lanes_per_row, ep_bias, ep_act);
        GAE_CHECK_LAUNCH("spmm_combine_kernel");
    }
    if (homed && (g_spmm_parts & 4)) {
        // XCD-pinned rows: the plan's virtual CSR (one virtual row = one segment = one (row, home, chunk) group of
        // column ids; position p is gathered by block p / 4, i.e. on XCD (p / 4) % 8 = the home of its columns) goes
        // through the same segment kernel into float partials, which are then added per real row in plan order
        float *pv = partial + (heavy ? align256(plan->n_segments * ldp * 4) / 4 : 0);
        gae_spmm_plan v = *plan;
        v.n_heavy = v.n_segments = plan->vh_n_virtual;
        v.heavy_rows = v.heavy_seg_base = v.seg_heavy = g_spmm_desc ? nullptr : plan->vh_identity;   // NULL: segment s = row s
        v.seg_desc = nullptr;
        v.hot_indices = plan->vh_hot_indices;
        v.mid_indices = nullptr;
        if (plan->vh_desc) {                // device-built: {p, first, end} into the (row, home, column)-ordered ids
            v.mid_indices = plan->vh_indices; v.mid_tagged = 0; v.seg_desc = plan->vh_desc; v.hot_indices = nullptr;
            v.heavy_rows = v.heavy_seg_base = v.seg_heavy = nullptr;
        }
        rc = dispatch_segments<T, VEC>(plan->vh_indptr, plan->vh_indices, h, ldh, f, cs, &v, pv, ldp, nullptr, m, ldm,
                                       0, 0, s);
        if (rc) return rc;
        const int lanes_per_row = (f + 3) / 4;
        const dim3 cgrid(unsigned((plan->vh_n_rows + 3) / 4), unsigned((lanes_per_row + 63) / 64));
#define GAE_VHC(L)                                                                                                   \
    hipLaunchKernelGGL((spmm_vh_combine_kernel<T, L>), cgrid, dim3(256), 0, s, plan->vh_rows, plan->vh_part_ptr,      \
                       plan->vh_part_pos, plan->vh_n_rows, pv, ldp, f, rs, m, ldm, acc_flag, ep_bias, ep_act)
        if (lanes_per_row <= 4) GAE_VHC(4);
        else if (lanes_per_row <= 8) GAE_VHC(8);
        else if (lanes_per_row <= 16) GAE_VHC(16);
        else if (lanes_per_row <= 32) GAE_VHC(32);
        else GAE_VHC(64);
#undef GAE_VHC
        GAE_CHECK_LAUNCH("spmm_vh_combine_kernel");
    }
    return GAE_OK;
}

} // namespace

extern "C" int gae_spmm_ell_build(const int32_t *indptr, const int32_t *indices, int64_t n_rows, int32_t width,
                                  int32_t skip_degree, int32_t *ell, void *stream)
{
    GAE_REQUIRE(n_rows >= 0, GAE_E_SIZE, "gae_spmm_ell_build: negative n_rows");
    GAE_REQUIRE(width == 4 || width == 8 || width == GAE_SPMM_ELL_WIDTH, GAE_E_RANGE,
                "gae_spmm_ell_build: width must be 4, 8 or %d", GAE_SPMM_ELL_WIDTH);
    GAE_REQUIRE(skip_degree >= 1, GAE_E_RANGE, "gae_spmm_ell_build: skip_degree >= 1 required");
    if (n_rows == 0) return GAE_OK;
    GAE_REQUIRE(indptr && ell, GAE_E_NULL, "gae_spmm_ell_build: NULL pointer");
    GAE_REQUIRE(n_rows * width < (int64_t(1) << 39), GAE_E_SIZE, "gae_spmm_ell_build: table too large");
    hipStream_t s = gae::as_stream(stream);
    hipLaunchKernelGGL(ell_build_kernel, dim3(unsigned((n_rows * width + 255) / 256)), dim3(256), 0, s, indptr, indices,
                       n_rows, int(width), int(skip_degree), ell);
    GAE_CHECK_LAUNCH("ell_build_kernel");
    return GAE_OK;
}

extern "C" int64_t gae_spmm_workspace_bytes(const gae_spmm_plan *plan, int64_t F)
{
    if (F < 0) return GAE_E_SIZE;
    if (!plan) return 0;
    int64_t b = 0;
    if (plan->n_heavy > 0) b += align256(plan->n_segments * plan_ldp(F) * 4);
    if (plan->vh_n_virtual > 0) b += align256(plan->vh_n_virtual * plan_ldp(F) * 4);
    return b;
}

static int spmm_csr_impl(const int32_t *indptr, const int32_t *indices, int64_t n_rows, int64_t n_cols,
                         const void *H, int64_t ldh, void *M, int64_t ldm, int64_t F, int dtype,
                         const float *row_scale, const float *col_scale, const gae_spmm_plan *plan,
                         void *workspace, int64_t workspace_bytes, int flags, void *stream, const float *ep_bias,
                         int ep_act)
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
    if (plan && plan->vh_n_virtual > 0)
        GAE_REQUIRE(plan->vh_n_rows > 0 && plan->vh_rows && plan->vh_indices && (plan->vh_desc || (plan->vh_indptr && plan->vh_identity)) &&
                        plan->vh_part_ptr && plan->vh_part_pos && plan->threshold >= 1 && plan->segment_edges >= 64 &&
                        plan->segment_edges % 64 == 0,
                    GAE_E_RANGE, "gae_spmm_csr: malformed plan (XCD-pinned part)");
    if (plan && (plan->n_heavy > 0 || plan->vh_n_virtual > 0)) {
        GAE_REQUIRE(!plan->mid_indices || plan->seg_desc, GAE_E_RANGE, "gae_spmm_csr: a plan with mid_indices needs seg_desc");
        GAE_REQUIRE(plan->n_heavy == 0 ||
                        (plan->heavy_rows && plan->heavy_seg_base && plan->seg_heavy && plan->n_segments >= plan->n_heavy &&
                         plan->threshold >= 1 && plan->segment_edges >= 64 && plan->segment_edges % 64 == 0),
                    GAE_E_RANGE, "gae_spmm_csr: malformed plan");
        const int64_t need = gae_spmm_workspace_bytes(plan, F);
        GAE_REQUIRE(workspace && workspace_bytes >= need, GAE_E_WORKSPACE, "gae_spmm_csr: workspace %lld < %lld bytes",
                    (long long)workspace_bytes, (long long)need);
        GAE_REQUIRE(gae::aligned16(workspace), GAE_E_ALIGN, "gae_spmm_csr: workspace not 16-byte aligned");
    }
    if (plan && plan->ell)
        GAE_REQUIRE(plan->ell_width == 4 || plan->ell_width == 8 || plan->ell_width == GAE_SPMM_ELL_WIDTH, GAE_E_RANGE,
                    "gae_spmm_csr: plan->ell_width must be 4, 8 or %d", GAE_SPMM_ELL_WIDTH);
    hipStream_t s = gae::as_stream(stream);
    const int f = int(F);
    if (dtype == GAE_F32) {
        const float *h = static_cast<const float *>(H);
        float *m = static_cast<float *>(M);
        const bool vec = (ldh % 4 == 0) && (ldm % 4 == 0) && gae::aligned16(H) && gae::aligned16(M);
        if (vec) return run_spmm<float, 4>(indptr, indices, n_rows, n_cols, h, ldh, m, ldm, f, row_scale, col_scale, true, plan, workspace, flags, s, ep_bias, ep_act);
        return run_spmm<float, 1>(indptr, indices, n_rows, n_cols, h, ldh, m, ldm, f, row_scale, col_scale, false, plan, workspace, flags, s, ep_bias, ep_act);
    }
    GAE_REQUIRE(ep_bias == nullptr && ep_act == GAE_ACT_IDENTITY, GAE_E_DTYPE, "gae_spmm_csr_ep: fp32 operands only");
    const unsigned short *h = static_cast<const unsigned short *>(H);
    unsigned short *m = static_cast<unsigned short *>(M);
    const bool vec = (ldh % 8 == 0) && (ldm % 8 == 0) && gae::aligned16(H) && gae::aligned16(M);
    if (vec) return run_spmm<unsigned short, 8>(indptr, indices, n_rows, n_cols, h, ldh, m, ldm, f, row_scale, col_scale, true, plan, workspace, flags, s);
    return run_spmm<unsigned short, 1>(indptr, indices, n_rows, n_cols, h, ldh, m, ldm, f, row_scale, col_scale, false, plan, workspace, flags, s);
}

extern "C" int gae_spmm_csr(const int32_t *indptr, const int32_t *indices, int64_t n_rows, int64_t n_cols,
                            const void *H, int64_t ldh, void *M, int64_t ldm, int64_t F, int dtype,
                            const float *row_scale, const float *col_scale, const gae_spmm_plan *plan,
                            void *workspace, int64_t workspace_bytes, int flags, void *stream)
{
    return spmm_csr_impl(indptr, indices, n_rows, n_cols, H, ldh, M, ldm, F, dtype, row_scale, col_scale, plan, workspace,
                         workspace_bytes, flags, stream, nullptr, GAE_ACT_IDENTITY);
}

// gae_spmm_csr with a store-time epilogue: M = act(diag(rs) A diag(cs) H (+ M if GAE_SPMM_ACCUMULATE) + bias), fp32,
// ANY plan (degree-skew segments and XCD-pinned rows included: every kernel that stores a finished row applies it).
// The sparse half of a GCN layer evaluated as act(A (H W^T) + b) on a power-law graph -- gae.py:26-31 up to fp32
// rounding -- where gae_spmm_csr_epilogue (packed-table plans only) does not apply.
extern "C" int gae_spmm_csr_ep(const int32_t *indptr, const int32_t *indices, int64_t n_rows, int64_t n_cols,
                               const float *H, int64_t ldh, float *M, int64_t ldm, int64_t F, const float *row_scale,
                               const float *col_scale, const gae_spmm_plan *plan, void *workspace,
                               int64_t workspace_bytes, int flags, const float *bias, int act, void *stream)
{
    GAE_REQUIRE(act == GAE_ACT_IDENTITY || act == GAE_ACT_RELU, GAE_E_RANGE, "gae_spmm_csr_ep: act %d", act);
    GAE_REQUIRE(!(flags & GAE_SPMM_TILE), GAE_E_RANGE, "gae_spmm_csr_ep: XCD feature tiles are not available with an epilogue");
    return spmm_csr_impl(indptr, indices, n_rows, n_cols, H, ldh, M, ldm, F, GAE_F32, row_scale, col_scale, plan,
                         workspace, workspace_bytes, flags, stream, bias, act);
}

extern "C" int64_t gae_spmm_blockdiag_lds_bytes(int64_t max_block_rows, int64_t max_block_edges, int64_t ldh)
{
    if (max_block_rows < 0 || max_block_edges < 0 || ldh < 0) return GAE_E_SIZE;
    return max_block_rows * ldh * 4 + ((max_block_rows + 1 + 3) & ~int64_t(3)) * 4 + max_block_edges * 4;
}

extern "C" int gae_spmm_csr_blockdiag(const int32_t *indptr, const int32_t *indices, const int32_t *block_ptr,
                                      const int32_t *block_eptr, int64_t n_blocks, int64_t max_block_rows,
                                      int64_t max_block_edges,
                                      int64_t n_rows, const float *H, int64_t ldh, float *M, int64_t ldm, int64_t F,
                                      const float *row_scale, const float *col_scale, int flags, void *stream)
{
    GAE_REQUIRE(n_blocks >= 0 && max_block_rows >= 0 && max_block_edges >= 0 && n_rows >= 0 && F >= 0, GAE_E_SIZE,
                "gae_spmm_csr_blockdiag: negative size");
    GAE_REQUIRE(F <= 256, GAE_E_RANGE, "gae_spmm_csr_blockdiag: F = %lld > 256 (use gae_spmm_csr)", (long long)F);
    GAE_REQUIRE(ldh >= F && ldm >= F, GAE_E_SIZE, "gae_spmm_csr_blockdiag: leading dimension smaller than F");
    GAE_REQUIRE((row_scale == nullptr) == (col_scale == nullptr), GAE_E_NULL,
                "gae_spmm_csr_blockdiag: row_scale and col_scale must both be given or both be NULL");
    if (n_blocks == 0 || n_rows == 0 || F == 0) return GAE_OK;
    GAE_REQUIRE(indptr && block_ptr && block_eptr && H && M, GAE_E_NULL, "gae_spmm_csr_blockdiag: NULL pointer");
    GAE_REQUIRE(ldh % 4 == 0 && ldm % 4 == 0 && gae::aligned16(H) && gae::aligned16(M), GAE_E_ALIGN,
                "gae_spmm_csr_blockdiag: H / M rows must be 16-byte aligned (pad the leading dimension)");
    const int64_t lds = gae_spmm_blockdiag_lds_bytes(max_block_rows, max_block_edges, ldh);
    GAE_REQUIRE(lds <= 160 * 1024, GAE_E_RANGE, "gae_spmm_csr_blockdiag: block slice needs %lld B of LDS (> 160 KiB)",
                (long long)lds);
    hipStream_t s = gae::as_stream(stream);
    const int nvec = int((F + 3) / 4);
    const bool scaled = row_scale != nullptr;
    const int store_pad = ((flags & GAE_SPMM_STORE_PAD) && (F + 3) / 4 * 4 <= ldm) ? 1 : 0;
    // TI = 16-byte pieces of the H slice per thread (compile-time so that all loads are issued up front)
    const int64_t n4max = max_block_rows * ldh / 4;
    const int ti = n4max <= 2 * 256 ? 2 : n4max <= 4 * 256 ? 4 : n4max <= 8 * 256 ? 8 : n4max <= 16 * 256 ? 16 : 0;
    GAE_REQUIRE(ti != 0 && max_block_rows <= 511 && max_block_edges <= 1024, GAE_E_RANGE,
                "gae_spmm_csr_blockdiag: a block may hold at most 511 rows / 64 KiB of H / 1024 staged edges");
#define GAE_BDL(LPR, SC, TI)                                                                                          \
    do {                                                                                                               \
        /* the attribute is per kernel and device: set it once per (instantiation, device), and again only when a   \
         * launch asks for more LDS than any before (host microseconds on a launch-bound path otherwise) */          \
        static int configured[16] = {0};                                                                               \
        int dev_ = 0;                                                                                                  \
        GAE_HIP(hipGetDevice(&dev_));                                                                                  \
        if (dev_ < 0 || dev_ >= 16 || configured[dev_] < int(lds)) {                                                   \
            GAE_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&spmm_blockdiag_kernel<LPR, SC, TI>),           \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));                        \
            if (dev_ >= 0 && dev_ < 16) configured[dev_] = int(lds);                                                   \
        }                                                                                                              \
        hipLaunchKernelGGL((spmm_blockdiag_kernel<LPR, SC, TI>), dim3(unsigned(n_blocks)), dim3(256), size_t(lds), s,  \
                           indptr, indices, block_ptr, block_eptr, H, ldh, M, ldm, int(F), row_scale, col_scale,       \
                           int(max_block_rows), int(max_block_edges), store_pad);                                      \
    } while (0)
#define GAE_BDT(LPR, SC)                                                                                              \
    do {                                                                                                               \
        if (ti == 2) GAE_BDL(LPR, SC, 2);                                                                              \
        else if (ti == 4) GAE_BDL(LPR, SC, 4);                                                                         \
        else if (ti == 8) GAE_BDL(LPR, SC, 8);                                                                         \
        else GAE_BDL(LPR, SC, 16);                                                                                     \
    } while (0)
#define GAE_BDG(LPR)                                                                                                  \
    do {                                                                                                               \
        if (scaled) GAE_BDT(LPR, true);                                                                                \
        else GAE_BDT(LPR, false);                                                                                      \
    } while (0)
    if (nvec <= 4) GAE_BDG(4);
    else if (nvec <= 8) GAE_BDG(8);
    else if (nvec <= 16) GAE_BDG(16);
    else if (nvec <= 32) GAE_BDG(32);
    else GAE_BDG(64);
#undef GAE_BDG
#undef GAE_BDT
#undef GAE_BDL
    GAE_CHECK_LAUNCH("spmm_blockdiag_kernel");
    return GAE_OK;
}

// ---- hot-column tags of a plan: how often every column is gathered, and the ids with the sign bit on the hot ones
namespace {
} // namespace

namespace gae { Knob *dense_knob(const char *name); Knob *bce_knob(const char *name); Knob *xw_knob(const char *name); Knob *optim_knob(const char *name); }

namespace {
gae::Knob *find_knob(const char *name)
{
    const struct { const char *k; gae::Knob *v; } knobs[] = {
        {"spmm_variant", &g_spmm_variant}, {"spmm_rpg", &g_spmm_rpg}, {"spmm_nt", &g_spmm_nt},
        {"spmm_tile_vecs", &g_spmm_tile_vecs}, {"spmm_ell", &g_spmm_ell}, {"spmm_hot", &g_spmm_hot},
        {"spmm_desc", &g_spmm_desc}, {"spmm_parts", &g_spmm_parts}, {"spmm_light", &g_spmm_light}};
    for (const auto &kv : knobs)
        if (strcmp(kv.k, name) == 0) return kv.v;
    if (gae::Knob *k = gae::spmm_ell_knob(name)) return k;
    if (gae::Knob *k = gae::dense_knob(name)) return k;
    if (gae::Knob *k = gae::xw_knob(name)) return k;
    if (gae::Knob *k = gae::optim_knob(name)) return k;
    return gae::bce_knob(name);
}
} // namespace

extern "C" int gae_tuning_set(const char *name, int64_t value)
{
    GAE_REQUIRE(name != nullptr, GAE_E_NULL, "gae_tuning_set: name is NULL");
    gae::Knob *k = find_knob(name);
    GAE_REQUIRE(k != nullptr, GAE_E_RANGE, "gae_tuning_set: unknown knob '%s'", name);
    *k = int(value);
    return GAE_OK;
}

extern "C" int gae_tuning_get(const char *name, int64_t *value_out)
{
    GAE_REQUIRE(name != nullptr && value_out != nullptr, GAE_E_NULL, "gae_tuning_get: NULL argument");
    const gae::Knob *k = find_knob(name);
    GAE_REQUIRE(k != nullptr, GAE_E_RANGE, "gae_tuning_get: unknown knob '%s'", name);
    *value_out = int(*k);
    return GAE_OK;
}
