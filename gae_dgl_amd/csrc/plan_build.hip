// Construction of a gae_spmm_plan for a power-law CSR entirely on the device (round 4; replaces the torch sorts /
// searchsorted / repeat_interleave of the round-2 builder: 0.3 - 0.7 s and 5.1 GB per R-MAT s24 rank).
//
// The plan serves gae_spmm_csr = DGL's update_all(copy_src, sum) of gae_dgl/gae.py:18-19,28 on graphs whose row lengths
// span five orders of magnitude.  Rows are classified by in-degree d (T = threshold, T2 = pin_degree):
//   light   1 <= d <= T      -> light_desc {row, e0, e1}: the row-group kernel's work list (empty rows: fill stream)
//   mid     T < d <= T2      -> heavy_rows / heavy_seg_base / seg_heavy / seg_desc: one wave per <= seg edges; optionally a
//                               COMPACT copy of their column ids with hot-column tags (seg_desc then indexes the copy)
//   pinned  d > T2           -> the edges regrouped by home(column) in 0..7 into chunks of <= seg ids ("virtual rows"),
//                               laid out so that virtual row p is gathered on XCD (p / 4) % 8 = its home:
//                               vh_cols (ids in (row, home, column) order), vh_desc {p, e0, e1}, vh_part_ptr / vh_part_pos
// Everything is integer work with a deterministic result (ascending rows, stable partitions, stable sort): the plan --
// and with it the summation order of every row -- is a function of the CSR alone.
//
// Three calls (the sizes of the outputs of a call are known to the host from the call before; each call synchronises
// its stream once to hand a few counters to the host -- plan construction is a one-off setup step):
//   gae_spmm_plan_sizes -> gae_spmm_plan_build_rows -> (pinned rows only) gae_spmm_plan_build_pinned
#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"

namespace {

typedef unsigned long long u64;
constexpr int kRowsPerBlock = 1024;            // 256 threads x 4 consecutive rows

__device__ __forceinline__ int column_home(int32_t col)       // the hash ops.column_home uses: (col * 2654435761) >> 13 & 7
{
    return int((u64(uint32_t(col)) * 2654435761ull) >> 13) & 7;
}

// class of a row: 0 empty, 1 light, 2 mid, 3 pinned
__device__ __forceinline__ int row_class(int d, int T, int T2) { return d <= 0 ? 0 : d <= T ? 1 : d <= T2 ? 2 : 3; }

struct RowSums { u64 nl, nm, sm, em, np, ep; };   // light rows | mid rows, their segments, their edges | pinned rows, edges

__device__ __forceinline__ void add_row(RowSums &s, int d, int T, int T2, int seg)
{
    const int c = row_class(d, T, T2);
    if (c == 1) s.nl += 1;
    else if (c == 2) { s.nm += 1; s.sm += u64((d + seg - 1) / seg); s.em += u64(d); }
    else if (c == 3) { s.np += 1; s.ep += u64(d); }
}

__device__ __forceinline__ u64 shfl_up64(u64 v, int d)
{
    const unsigned lo = __shfl_up(unsigned(v), d, 64), hi = __shfl_up(unsigned(v >> 32), d, 64);
    return (u64(hi) << 32) | lo;
}

// exclusive prefix of v over the 256 threads of a block (thread order); total = block sum.  scratch: [4] u64 in LDS
__device__ __forceinline__ u64 block_excl_scan(u64 v, u64 *scratch, u64 &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u64 inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u64 t = shfl_up64(inc, d);
        if (lane >= d) inc += t;
    }
    __syncthreads();
    if (lane == 63) scratch[wave] = inc;
    __syncthreads();
    u64 base = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w < wave) base += scratch[w];
    }
    total = scratch[0] + scratch[1] + scratch[2] + scratch[3];
    return base + inc - v;
}

// per-block sums of the six quantities -> blk[q * n_blocks + b]; totals (atomics: integers) -> totals[0..5], max degree [6]
__global__ __launch_bounds__(256) void plan_count_kernel(const int32_t *__restrict__ indptr, int64_t n_rows, int T, int T2,
                                                         int seg, u64 *__restrict__ blk, int64_t n_blocks,
                                                         u64 *__restrict__ totals)
{
    __shared__ u64 sc[4];
    const int64_t r0 = int64_t(blockIdx.x) * kRowsPerBlock + threadIdx.x * 4;
    RowSums s{0, 0, 0, 0, 0, 0};
    int mx = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t r = r0 + q;
        if (r < n_rows) {
            const int d = indptr[r + 1] - indptr[r];
            add_row(s, d, T, T2, seg);
            mx = d > mx ? d : mx;
        }
    }
    u64 v[6] = {s.nl, s.nm, s.sm, s.em, s.np, s.ep};
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        u64 tot;
        block_excl_scan(v[q], sc, tot);
        if (threadIdx.x == 0) {
            if (blk) blk[q * n_blocks + blockIdx.x] = tot;
            if (totals && tot) atomicAdd(&totals[q], tot);
        }
    }
    if (totals) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const int t = __shfl_down(mx, off, 64); mx = t > mx ? t : mx; }
        if ((threadIdx.x & 63) == 0 && mx > 0) atomicMax(&totals[6], u64(mx));
    }
}

// exclusive prefix sums in place, one block per array: a[b * n .. (b + 1) * n)
__global__ __launch_bounds__(1024) void scan_arrays_kernel(u64 *__restrict__ a, int64_t n)
{
    __shared__ u64 part[1024];
    u64 *arr = a + int64_t(blockIdx.x) * n;
    const int64_t per = (n + 1023) / 1024;
    const int64_t b0 = int64_t(threadIdx.x) * per, b1 = b0 + per < n ? b0 + per : n;
    u64 sum = 0;
    for (int64_t b = b0; b < b1; ++b) sum += arr[b];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 run = 0;
        for (int t = 0; t < 1024; ++t) { const u64 v = part[t]; part[t] = run; run += v; }
    }
    __syncthreads();
    u64 run = part[threadIdx.x];
    for (int64_t b = b0; b < b1; ++b) { const u64 v = arr[b]; arr[b] = run; run += v; }
}

struct FillOut {
    int32_t *light_desc;                                   // [nl][4]
    int32_t *heavy_rows, *heavy_seg_base, *seg_heavy, *seg_desc;   // mid rows
    int32_t *mid_edge_base;                                // [nm + 1] compact edge offset of a mid row (NULL: not compacted)
    int32_t *vh_rows, *pin_edge_base;                      // [np], [np + 1]
};

// every row finds its slot (ascending rows per class) and writes its descriptors
__global__ __launch_bounds__(256) void plan_fill_kernel(const int32_t *__restrict__ indptr, int64_t n_rows, int T, int T2,
                                                        int seg, const u64 *__restrict__ blk, int64_t n_blocks, FillOut o,
                                                        u64 nm_total, u64 np_total, u64 em_total, u64 ep_total)
{
    __shared__ u64 sc[4];
    const int64_t r0 = int64_t(blockIdx.x) * kRowsPerBlock + threadIdx.x * 4;
    int d[4], c[4];
    RowSums s{0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t r = r0 + q;
        d[q] = r < n_rows ? indptr[r + 1] - indptr[r] : 0;
        c[q] = row_class(d[q], T, T2);
        add_row(s, d[q], T, T2, seg);
    }
    u64 v[6] = {s.nl, s.nm, s.sm, s.em, s.np, s.ep}, at[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        u64 tot;
        at[q] = blk[q * n_blocks + blockIdx.x] + block_excl_scan(v[q], sc, tot);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t r = r0 + q;
        if (c[q] == 1) {
            const int32_t e0 = indptr[r];
            if (o.light_desc) *reinterpret_cast<int4 *>(o.light_desc + at[0] * 4) = make_int4(int32_t(r), e0, e0 + d[q], 0);
            at[0] += 1;
        } else if (c[q] == 2) {
            const int64_t m = int64_t(at[1]), sb = int64_t(at[2]);
            const int ns = (d[q] + seg - 1) / seg;
            const int32_t eb = o.mid_edge_base ? int32_t(at[3]) : indptr[r];
            o.heavy_rows[m] = int32_t(r);
            o.heavy_seg_base[m] = int32_t(sb);
            if (o.mid_edge_base) o.mid_edge_base[m] = int32_t(at[3]);
            for (int k = 0; k < ns; ++k) {
                const int32_t a = eb + k * seg, b = (k + 1) * seg < d[q] ? a + seg : eb + d[q];
                *reinterpret_cast<int4 *>(o.seg_desc + (sb + k) * 4) = make_int4(int32_t(r), a, b, ns == 1 ? 1 : 0);
                o.seg_heavy[sb + k] = int32_t(m);
            }
            at[1] += 1; at[2] += u64(ns); at[3] += u64(d[q]);
        } else if (c[q] == 3) {
            o.vh_rows[at[4]] = int32_t(r);
            o.pin_edge_base[at[4]] = int32_t(at[5]);
            at[4] += 1; at[5] += u64(d[q]);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (o.mid_edge_base) o.mid_edge_base[nm_total] = int32_t(em_total);
        if (o.pin_edge_base) o.pin_edge_base[np_total] = int32_t(ep_total);
    }
}

// compact copy of the mid rows' column ids: one wave per row
__global__ __launch_bounds__(256) void mid_copy_kernel(const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                       const int32_t *__restrict__ heavy_rows,
                                                       const int32_t *__restrict__ mid_edge_base, int64_t nm,
                                                       int32_t *__restrict__ mid_ids)
{
    const int64_t m = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (m >= nm) return;
    const int lane = threadIdx.x & 63;
    const int32_t r = heavy_rows[m], e0 = indptr[r], d = indptr[r + 1] - e0, b = mid_edge_base[m];
    for (int t = lane; t < d; t += 64) mid_ids[b + t] = indices[e0 + t];
}

__global__ __launch_bounds__(256) void freq_kernel(const int32_t *__restrict__ ids, int64_t n, int32_t *__restrict__ freq)
{
    for (int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x; e < n; e += int64_t(gridDim.x) * 256)
        atomicAdd(&freq[ids[e] & 0x7fffffff], 1);
}
constexpr int kFreqBins = 4096;
__global__ __launch_bounds__(256) void freq_hist_kernel(const int32_t *__restrict__ freq, int64_t n_cols,
                                                        unsigned *__restrict__ hist)
{
    __shared__ unsigned h[kFreqBins];
    for (int t = threadIdx.x; t < kFreqBins; t += 256) h[t] = 0;
    __syncthreads();
    for (int64_t c = int64_t(blockIdx.x) * 256 + threadIdx.x; c < n_cols; c += int64_t(gridDim.x) * 256) {
        const int f = freq[c];
        if (f > 0) atomicAdd(&h[f < kFreqBins ? f : kFreqBins - 1], 1u);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < kFreqBins; t += 256)
        if (h[t]) atomicAdd(&hist[t], h[t]);
}
__global__ __launch_bounds__(256) void tag_kernel(int32_t *__restrict__ ids, int64_t n, const int32_t *__restrict__ freq,
                                                  int32_t min_freq)
{
    for (int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x; e < n; e += int64_t(gridDim.x) * 256) {
        const int32_t j = ids[e];
        if (freq[j] >= min_freq) ids[e] = int32_t(unsigned(j) | 0x80000000u);
    }
}

// ---- pinned rows ------------------------------------------------------------------------------------------------
// edges per (pinned row, home): one block per row
__global__ __launch_bounds__(256) void home_count_kernel(const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                         const int32_t *__restrict__ vh_rows, int32_t *__restrict__ cnt)
{
    __shared__ int red[4][8];
    const int32_t r = vh_rows[blockIdx.x], e0 = indptr[r], d = indptr[r + 1] - e0;
    int c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = threadIdx.x; t < d; t += 256) {
        const int h = column_home(indices[e0 + t]);
#pragma unroll
        for (int q = 0; q < 8; ++q) c[q] += h == q ? 1 : 0;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c[q] += __shfl_down(c[q], off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][q] = c[q];
    }
    __syncthreads();
    if (threadIdx.x < 8) cnt[int64_t(blockIdx.x) * 8 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// vfirst[g] = number of chunks (virtual rows) of the groups in front of group g = (pinned row, home); vfirst[G] = total.
// One block (G is 8 x the number of pinned rows).
__global__ __launch_bounds__(1024) void chunk_scan_kernel(const int32_t *__restrict__ cnt, int seg, int64_t G,
                                                          int32_t *__restrict__ vfirst)
{
    __shared__ u64 part[1024];
    const int64_t per = (G + 1023) / 1024;
    const int64_t g0 = int64_t(threadIdx.x) * per, g1 = g0 + per < G ? g0 + per : G;
    u64 sum = 0;
    for (int64_t g = g0; g < g1; ++g) sum += u64((cnt[g] + seg - 1) / seg);
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 run = 0;
        for (int t = 0; t < 1024; ++t) { const u64 v = part[t]; part[t] = run; run += v; }
        vfirst[G] = int32_t(run);
    }
    __syncthreads();
    u64 run = part[threadIdx.x];
    for (int64_t g = g0; g < g1; ++g) { vfirst[g] = int32_t(run); run += u64((cnt[g] + seg - 1) / seg); }
}

// keys / values of the virtual rows in (row, home, chunk) order: key = home << 32 | first column of the chunk.  The
// first column is read from the row's CSR list: the chunk's first edge is the (k * seg)-th edge of home h in the row's
// (ascending) list -- found by the block that partitions the row (home_scatter_kernel), which also writes the ids.
// One block per pinned row: stable partition of its ids by home into vh_cols[pin_edge_base[p] + ...], (row, home,
// column) order, and the sort keys / edge ranges of its chunks.
__global__ __launch_bounds__(256) void home_scatter_kernel(const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                           const int32_t *__restrict__ vh_rows,
                                                           const int32_t *__restrict__ pin_edge_base,
                                                           const int32_t *__restrict__ cnt, const int32_t *__restrict__ vfirst,
                                                           int seg, int32_t *__restrict__ vh_cols, u64 *__restrict__ keys,
                                                           int32_t *__restrict__ vals, int32_t *__restrict__ v_e0,
                                                           int32_t *__restrict__ v_len)
{
    __shared__ int wcnt[4][8], running[8], gstart[8], gcnt[8], gvf[8];
    const int64_t p = blockIdx.x;
    const int32_t r = vh_rows[p], e0 = indptr[r], d = indptr[r + 1] - e0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 8) {
        int s = pin_edge_base[p];
        for (int q = 0; q < int(threadIdx.x); ++q) s += cnt[p * 8 + q];
        gstart[threadIdx.x] = s;
        gcnt[threadIdx.x] = cnt[p * 8 + threadIdx.x];
        gvf[threadIdx.x] = vfirst[p * 8 + threadIdx.x];
        running[threadIdx.x] = 0;
    }
    __syncthreads();
    for (int base = 0; base < d; base += 256) {
        const int t = base + threadIdx.x;
        const bool valid = t < d;
        const int32_t col = valid ? indices[e0 + t] : 0;
        const int h = valid ? column_home(col) : -1;
        int rank = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const u64 m = __builtin_amdgcn_ballot_w64(h == q);
            if (h == q) rank = __builtin_popcountll(m & ((1ull << lane) - 1ull));
            if (lane == 0) wcnt[wave][q] = __builtin_popcountll(m);
        }
        __syncthreads();
        if (valid) {
            int off = running[h] + rank;
            for (int w = 0; w < wave; ++w) off += wcnt[w][h];
            vh_cols[gstart[h] + off] = col;
            if (off % seg == 0) {                       // first edge of a chunk: its sort key and edge range
                const int k = off / seg, vid = gvf[h] + k;
                keys[vid] = (u64(h) << 32) | u64(uint32_t(col));
                vals[vid] = vid;
                v_e0[vid] = gstart[h] + off;
                v_len[vid] = gcnt[h] - off < seg ? gcnt[h] - off : seg;
            }
        }
        __syncthreads();
        if (threadIdx.x < 8) running[threadIdx.x] += wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x] + wcnt[2][threadIdx.x] + wcnt[3][threadIdx.x];
        __syncthreads();
    }
}

// first position of every home in the sorted keys: hstart[h] = lower bound of h << 32, hstart[8] = nv
__global__ void home_bounds_kernel(const u64 *__restrict__ keys, int64_t nv, int64_t *__restrict__ hstart)
{
    const int h = threadIdx.x;
    if (h > 8) return;
    if (h == 8) { hstart[8] = nv; return; }
    const u64 key = u64(h) << 32;
    int64_t lo = 0, hi = nv;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (keys[mid] < key) lo = mid + 1; else hi = mid; }
    hstart[h] = lo;
}

struct HStart { int64_t v[9]; };
// sorted index s -> position p of the virtual row: rank inside its home -> 4 consecutive positions per thread block,
// blocks interleaved over the homes ((p / 4) % 8 == home)
__global__ __launch_bounds__(256) void place_kernel(const u64 *__restrict__ keys, const int32_t *__restrict__ vals,
                                                    const int32_t *__restrict__ v_e0, const int32_t *__restrict__ v_len,
                                                    int64_t nv, HStart hs, int32_t *__restrict__ vh_desc,
                                                    int32_t *__restrict__ part_pos)
{
    const int64_t s = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (s >= nv) return;
    const int h = int(keys[s] >> 32);
    const int64_t rank = s - hs.v[h];
    const int64_t pos = (rank / 4) * 32 + h * 4 + (rank % 4);
    const int32_t vid = vals[s], a = v_e0[vid];
    *reinterpret_cast<int4 *>(vh_desc + pos * 4) = make_int4(int32_t(pos), a, a + v_len[vid], 0);
    part_pos[vid] = int32_t(pos);
}

__global__ __launch_bounds__(256) void part_ptr_kernel(const int32_t *__restrict__ vfirst, int64_t np, int32_t *__restrict__ part_ptr)
{
    const int64_t p = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (p <= np) part_ptr[p] = vfirst[p * 8];
}

inline int64_t al(int64_t x) { return (x + 255) / 256 * 256; }

} // namespace

extern "C" int gae_spmm_plan_sizes(const int32_t *indptr, int64_t n_rows, int32_t threshold, int32_t pin_degree,
                                   int32_t segment_edges, int64_t *sizes_host, void *scratch, int64_t scratch_bytes,
                                   void *stream)
{
    GAE_REQUIRE(n_rows >= 0, GAE_E_SIZE, "gae_spmm_plan_sizes: negative n_rows");
    GAE_REQUIRE(threshold >= 1 && pin_degree >= threshold && segment_edges >= 64 && segment_edges % 64 == 0, GAE_E_RANGE,
                "gae_spmm_plan_sizes: threshold >= 1, pin_degree >= threshold, segment_edges a positive multiple of 64");
    GAE_REQUIRE(sizes_host && scratch && scratch_bytes >= 64 && gae::aligned16(scratch), GAE_E_NULL,
                "gae_spmm_plan_sizes: NULL / small scratch (64 bytes)");
    for (int k = 0; k < 8; ++k) sizes_host[k] = 0;
    if (n_rows == 0) return GAE_OK;
    GAE_REQUIRE(indptr != nullptr, GAE_E_NULL, "gae_spmm_plan_sizes: indptr is NULL");
    hipStream_t s = gae::as_stream(stream);
    u64 *tot = static_cast<u64 *>(scratch);
    GAE_HIP(hipMemsetAsync(tot, 0, 64, s));
    const int64_t nb = (n_rows + kRowsPerBlock - 1) / kRowsPerBlock;
    hipLaunchKernelGGL(plan_count_kernel, dim3(unsigned(nb)), dim3(256), 0, s, indptr, n_rows, int(threshold), int(pin_degree),
                       int(segment_edges), static_cast<u64 *>(nullptr), nb, tot);
    GAE_CHECK_LAUNCH("plan_count_kernel");
    u64 host[8];
    GAE_HIP(hipMemcpyAsync(host, tot, 64, hipMemcpyDeviceToHost, s));
    GAE_HIP(hipStreamSynchronize(s));
    for (int k = 0; k < 7; ++k) sizes_host[k] = int64_t(host[k]);
    return GAE_OK;
}

extern "C" int64_t gae_spmm_plan_scratch_bytes(int64_t n_rows, int64_t n_cols, int64_t n_pinned, int64_t pinned_edges,
                                               int32_t segment_edges)
{
    if (n_rows < 0 || n_cols < 0 || n_pinned < 0 || pinned_edges < 0 || segment_edges < 64) return GAE_E_SIZE;
    const int64_t nb = (n_rows + kRowsPerBlock - 1) / kRowsPerBlock;
    const int64_t nv = pinned_edges / segment_edges + 8 * n_pinned + 8;          // bound on the virtual rows
    size_t sort_bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, sort_bytes, static_cast<u64 *>(nullptr), static_cast<u64 *>(nullptr),
                                    static_cast<int32_t *>(nullptr), static_cast<int32_t *>(nullptr), size_t(nv), 0u, 36u,
                                    hipStream_t(nullptr));
    return 256 + al(6 * nb * 8) + al((n_rows + 1) * 4) + al(n_cols * 4) + al(kFreqBins * 4) + al((n_pinned + 1) * 4) + al(n_pinned * 8 * 4) +
           al((n_pinned * 8 + 1) * 4) + 2 * al(nv * 8) + 4 * al(nv * 4) + al(int64_t(sort_bytes)) + al(16 * 8) + 4096;
}

namespace {
struct Scratch {
    u64 *blk; int32_t *mid_base, *freq; unsigned *hist; int32_t *pin_edge_base, *cnt, *vfirst; u64 *keys_a, *keys_b;
    int32_t *vals_a, *vals_b, *v_e0, *v_len; void *sort_tmp; size_t sort_bytes; int64_t *hstart; int64_t nv_bound;
};
Scratch carve(void *scratch, int64_t n_rows, int64_t n_cols, int64_t n_pinned, int64_t pinned_edges, int seg)
{
    Scratch c{};
    char *p = static_cast<char *>(scratch) + 256;
    const int64_t nb = (n_rows + kRowsPerBlock - 1) / kRowsPerBlock;
    c.nv_bound = pinned_edges / seg + 8 * n_pinned + 8;
    auto take = [&](int64_t bytes) { char *q = p; p += al(bytes); return q; };
    c.blk = reinterpret_cast<u64 *>(take(6 * nb * 8));
    c.mid_base = reinterpret_cast<int32_t *>(take((n_rows + 1) * 4));
    c.freq = reinterpret_cast<int32_t *>(take(n_cols * 4));
    c.hist = reinterpret_cast<unsigned *>(take(kFreqBins * 4));
    c.pin_edge_base = reinterpret_cast<int32_t *>(take((n_pinned + 1) * 4));
    c.cnt = reinterpret_cast<int32_t *>(take(n_pinned * 8 * 4));
    c.vfirst = reinterpret_cast<int32_t *>(take((n_pinned * 8 + 1) * 4));
    c.keys_a = reinterpret_cast<u64 *>(take(c.nv_bound * 8));
    c.keys_b = reinterpret_cast<u64 *>(take(c.nv_bound * 8));
    c.vals_a = reinterpret_cast<int32_t *>(take(c.nv_bound * 4));
    c.vals_b = reinterpret_cast<int32_t *>(take(c.nv_bound * 4));
    c.v_e0 = reinterpret_cast<int32_t *>(take(c.nv_bound * 4));
    c.v_len = reinterpret_cast<int32_t *>(take(c.nv_bound * 4));
    c.sort_bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, c.sort_bytes, static_cast<u64 *>(nullptr), static_cast<u64 *>(nullptr),
                                    static_cast<int32_t *>(nullptr), static_cast<int32_t *>(nullptr), size_t(c.nv_bound), 0u,
                                    36u, hipStream_t(nullptr));
    c.sort_tmp = take(int64_t(c.sort_bytes));
    c.hstart = reinterpret_cast<int64_t *>(take(16 * 8));
    return c;
}
} // namespace

// Outputs (device arrays sized from gae_spmm_plan_sizes: sizes = {n_light, n_mid, mid_segments, mid_edges, n_pinned,
// pinned_edges, max_degree}): light_desc [n_light][4]; heavy_rows, heavy_seg_base [n_mid]; seg_heavy [mid_segments];
// seg_desc [mid_segments][4]; mid_ids [mid_edges] or NULL (hot_columns > 0 needs it); vh_rows [n_pinned]; vh_part_ptr
// [n_pinned + 1].  pinned_host_out = {virtual rows NV, positions V (multiple of 32), min_freq of the hot tags}.
extern "C" int gae_spmm_plan_build_rows(const int32_t *indptr, const int32_t *indices, int64_t n_rows, int64_t n_cols,
                                        int32_t threshold, int32_t pin_degree, int32_t segment_edges, const int64_t *sizes,
                                        int64_t hot_columns, int32_t *light_desc, int32_t *heavy_rows,
                                        int32_t *heavy_seg_base, int32_t *seg_heavy, int32_t *seg_desc, int32_t *mid_ids,
                                        int32_t *vh_rows, int32_t *vh_part_ptr, void *scratch, int64_t scratch_bytes,
                                        int64_t *pinned_host_out, void *stream)
{
    GAE_REQUIRE(n_rows >= 1 && sizes && scratch && pinned_host_out, GAE_E_NULL, "gae_spmm_plan_build_rows: NULL / empty");
    GAE_REQUIRE(threshold >= 1 && pin_degree >= threshold && segment_edges >= 64 && segment_edges % 64 == 0, GAE_E_RANGE,
                "gae_spmm_plan_build_rows: bad thresholds");
    const int64_t nl = sizes[0], nm = sizes[1], sm = sizes[2], em = sizes[3], np = sizes[4], ep = sizes[5];
    GAE_REQUIRE(scratch_bytes >= gae_spmm_plan_scratch_bytes(n_rows, n_cols, np, ep, segment_edges) && gae::aligned16(scratch),
                GAE_E_WORKSPACE, "gae_spmm_plan_build_rows: scratch too small (gae_spmm_plan_scratch_bytes)");
    (void)nl;                                        // (light_desc may be NULL: the list is optional)
    GAE_REQUIRE((nm == 0 || (heavy_rows && heavy_seg_base && seg_heavy && seg_desc)) &&
                    (np == 0 || (vh_rows && vh_part_ptr && indices)), GAE_E_NULL, "gae_spmm_plan_build_rows: NULL output");
    GAE_REQUIRE(!mid_ids || indices, GAE_E_NULL, "gae_spmm_plan_build_rows: the compact id copy needs `indices`");
    GAE_REQUIRE(hot_columns <= 0 || mid_ids || em == 0, GAE_E_NULL, "gae_spmm_plan_build_rows: hot tags need mid_ids");
    GAE_REQUIRE(em < (int64_t(1) << 31) && ep < (int64_t(1) << 31), GAE_E_SIZE, "gae_spmm_plan_build_rows: too many edges");
    hipStream_t s = gae::as_stream(stream);
    Scratch c = carve(scratch, n_rows, n_cols, np, ep, segment_edges);
    const int64_t nb = (n_rows + kRowsPerBlock - 1) / kRowsPerBlock;
    hipLaunchKernelGGL(plan_count_kernel, dim3(unsigned(nb)), dim3(256), 0, s, indptr, n_rows, int(threshold), int(pin_degree),
                       int(segment_edges), c.blk, nb, static_cast<u64 *>(nullptr));
    hipLaunchKernelGGL(scan_arrays_kernel, dim3(6), dim3(1024), 0, s, c.blk, nb);
    int32_t *mid_edge_base = (mid_ids && nm > 0) ? c.mid_base : nullptr;
    FillOut o{light_desc, heavy_rows, heavy_seg_base, seg_heavy, seg_desc, mid_edge_base, vh_rows, c.pin_edge_base};
    hipLaunchKernelGGL(plan_fill_kernel, dim3(unsigned(nb)), dim3(256), 0, s, indptr, n_rows, int(threshold), int(pin_degree),
                       int(segment_edges), c.blk, nb, o, u64(nm), u64(np), u64(em), u64(ep));
    GAE_CHECK_LAUNCH("plan_fill_kernel");
    int64_t min_freq = 0;
    if (mid_ids && nm > 0) {
        hipLaunchKernelGGL(mid_copy_kernel, dim3(unsigned((nm + 3) / 4)), dim3(256), 0, s, indptr, indices, heavy_rows,
                           mid_edge_base, nm, mid_ids);
        GAE_CHECK_LAUNCH("mid_copy_kernel");
    }
    bool need_sync = false;
    if (mid_ids && hot_columns > 0 && em > 0) {
        GAE_HIP(hipMemsetAsync(c.freq, 0, size_t(n_cols) * 4, s));
        GAE_HIP(hipMemsetAsync(c.hist, 0, kFreqBins * 4, s));
        const int64_t g = (em + 255) / 256;
        hipLaunchKernelGGL(freq_kernel, dim3(unsigned(g < 65536 ? g : 65536)), dim3(256), 0, s, mid_ids, em, c.freq);
        const int64_t gc = (n_cols + 255) / 256;
        hipLaunchKernelGGL(freq_hist_kernel, dim3(unsigned(gc < 2048 ? gc : 2048)), dim3(256), 0, s, c.freq, n_cols, c.hist);
        GAE_CHECK_LAUNCH("freq_hist_kernel");
        need_sync = true;
    }
    int64_t nv = 0, V = 0;
    if (np > 0) {
        hipLaunchKernelGGL(home_count_kernel, dim3(unsigned(np)), dim3(256), 0, s, indptr, indices, vh_rows, c.cnt);
        hipLaunchKernelGGL(chunk_scan_kernel, dim3(1), dim3(1024), 0, s, c.cnt, int(segment_edges), np * 8, c.vfirst);
        hipLaunchKernelGGL(part_ptr_kernel, dim3(unsigned((np + 256) / 256)), dim3(256), 0, s, c.vfirst, np, vh_part_ptr);
        GAE_CHECK_LAUNCH("part_ptr_kernel");
        need_sync = true;
    }
    unsigned hist_host[kFreqBins];
    int32_t nv32 = 0;
    if (need_sync) {
        if (mid_ids && hot_columns > 0 && em > 0) GAE_HIP(hipMemcpyAsync(hist_host, c.hist, kFreqBins * 4, hipMemcpyDeviceToHost, s));
        if (np > 0) GAE_HIP(hipMemcpyAsync(&nv32, c.vfirst + np * 8, 4, hipMemcpyDeviceToHost, s));
        GAE_HIP(hipStreamSynchronize(s));
    }
    if (mid_ids && hot_columns > 0 && em > 0) {
        // the frequency of the hot_columns-th most gathered column (a column gathered once is never hot)
        int64_t acc = 0;
        int f = kFreqBins - 1;
        for (; f >= 2; --f) { acc += hist_host[f]; if (acc >= hot_columns) break; }
        min_freq = f < 2 ? 2 : f;
        const int64_t g = (em + 255) / 256;
        hipLaunchKernelGGL(tag_kernel, dim3(unsigned(g < 65536 ? g : 65536)), dim3(256), 0, s, mid_ids, em, c.freq, int32_t(min_freq));
        GAE_CHECK_LAUNCH("tag_kernel");
    }
    if (np > 0) {
        nv = nv32;
        GAE_REQUIRE(nv <= c.nv_bound, GAE_E_WORKSPACE, "gae_spmm_plan_build_rows: internal bound on the virtual rows exceeded");
    }
    pinned_host_out[0] = nv;
    pinned_host_out[1] = V;            // filled by gae_spmm_plan_build_pinned
    pinned_host_out[2] = min_freq;
    return GAE_OK;
}

// Second half for plans with pinned rows.  Call with vh_desc == NULL first: the ids are partitioned (vh_cols [pinned_edges]),
// the chunks sorted, and pinned_host_out[1] = V is reported; then again with vh_desc [V][4] (zeroed by the call) and
// vh_part_pos [NV] to place the chunks.  (The scratch buffer carries the state between the two calls: do not touch it.)
extern "C" int gae_spmm_plan_build_pinned(const int32_t *indptr, const int32_t *indices, int64_t n_rows, int64_t n_cols,
                                          int32_t segment_edges, const int64_t *sizes, const int32_t *vh_rows,
                                          int32_t *vh_cols, int32_t *vh_desc, int32_t *vh_part_pos, void *scratch,
                                          int64_t scratch_bytes, int64_t *pinned_host_out, void *stream)
{
    GAE_REQUIRE(sizes && scratch && pinned_host_out && indptr && indices && vh_rows && vh_cols, GAE_E_NULL,
                "gae_spmm_plan_build_pinned: NULL pointer");
    const int64_t np = sizes[4], ep = sizes[5], nv = pinned_host_out[0];
    GAE_REQUIRE(np > 0 && nv > 0, GAE_E_SIZE, "gae_spmm_plan_build_pinned: the plan has no pinned rows");
    GAE_REQUIRE(scratch_bytes >= gae_spmm_plan_scratch_bytes(n_rows, n_cols, np, ep, segment_edges), GAE_E_WORKSPACE,
                "gae_spmm_plan_build_pinned: scratch too small");
    hipStream_t s = gae::as_stream(stream);
    Scratch c = carve(scratch, n_rows, n_cols, np, ep, segment_edges);
    if (vh_desc == nullptr) {
        hipLaunchKernelGGL(home_scatter_kernel, dim3(unsigned(np)), dim3(256), 0, s, indptr, indices, vh_rows, c.pin_edge_base,
                           c.cnt, c.vfirst, int(segment_edges), vh_cols, c.keys_a, c.vals_a, c.v_e0, c.v_len);
        GAE_CHECK_LAUNCH("home_scatter_kernel");
        size_t bytes = c.sort_bytes;
        GAE_HIP(rocprim::radix_sort_pairs(c.sort_tmp, bytes, c.keys_a, c.keys_b, c.vals_a, c.vals_b, size_t(nv), 0u, 36u, s));
        hipLaunchKernelGGL(home_bounds_kernel, dim3(1), dim3(64), 0, s, c.keys_b, nv, c.hstart);
        GAE_CHECK_LAUNCH("home_bounds_kernel");
        int64_t hs[9];
        GAE_HIP(hipMemcpyAsync(hs, c.hstart, 9 * 8, hipMemcpyDeviceToHost, s));
        GAE_HIP(hipStreamSynchronize(s));
        int64_t L = 0;
        for (int h = 0; h < 8; ++h) L = hs[h + 1] - hs[h] > L ? hs[h + 1] - hs[h] : L;
        L = (L + 3) / 4 * 4;
        pinned_host_out[1] = 8 * L;
        return GAE_OK;
    }
    GAE_REQUIRE(vh_part_pos != nullptr && gae::aligned16(vh_desc), GAE_E_NULL, "gae_spmm_plan_build_pinned: NULL / misaligned output");
    const int64_t V = pinned_host_out[1];
    GAE_REQUIRE(V > 0 && V % 32 == 0, GAE_E_SIZE, "gae_spmm_plan_build_pinned: V was not computed (first call)");
    HStart hs;
    GAE_HIP(hipMemcpyAsync(hs.v, c.hstart, 9 * 8, hipMemcpyDeviceToHost, s));
    GAE_HIP(hipMemsetAsync(vh_desc, 0, size_t(V) * 16, s));
    GAE_HIP(hipStreamSynchronize(s));
    hipLaunchKernelGGL(place_kernel, dim3(unsigned((nv + 255) / 256)), dim3(256), 0, s, c.keys_b, c.vals_b, c.v_e0, c.v_len, nv,
                       hs, vh_desc, vh_part_pos);
    GAE_CHECK_LAUNCH("place_kernel");
    return GAE_OK;
}
