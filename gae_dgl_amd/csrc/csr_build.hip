// Graph-structure kernels: COO -> CSR, degrees / norm, dense adjacency (debug),
// block-diagonal batch gather from a device-resident dataset CSR.
//
// Replaces the DGL graph index the reference builds with DGLGraph.add_edges /
// dgl.batch and traverses in g.update_all (gae_dgl/gae.py:28,
// gae_dgl/train_inductive.py:31-35,44; gae_dgl/train_transductive.py:55-59).
// Integer work only: results are exact and order-deterministic.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"

namespace {

using gae::kWave;

__host__ __device__ inline int bits_for(int64_t n)
{
    int b = 1;
    while ((int64_t(1) << b) < n) ++b;
    return b;
}

inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// key = row << col_bits | col ; out-of-range ids are clamped and flagged
__global__ __launch_bounds__(256) void pack_keys_kernel(const int64_t *__restrict__ row,
                                                        const int64_t *__restrict__ col, int64_t n_edges,
                                                        int64_t n_rows, int64_t n_cols, int col_bits,
                                                        uint64_t *__restrict__ keys, int32_t *status)
{
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    bool bad = false;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < n_edges; e += stride) {
        int64_t r = row[e], c = col[e];
        if (r < 0 || r >= n_rows) { bad = true; r = r < 0 ? 0 : n_rows - 1; }
        if (c < 0 || c >= n_cols) { bad = true; c = c < 0 ? 0 : n_cols - 1; }
        keys[e] = (uint64_t(r) << col_bits) | uint64_t(c);
    }
    if (bad && status) atomicExch(status, 1);
}

// sorted keys -> indices (low bits) and indptr (first position of every row;
// empty rows inherit the next row's start).  One thread per edge + one extra
// sweep for the tail rows.
__global__ __launch_bounds__(256) void unpack_fill_kernel(const uint64_t *__restrict__ keys, int64_t n_edges,
                                                          int64_t n_rows, int col_bits,
                                                          int32_t *__restrict__ indptr,
                                                          int32_t *__restrict__ indices)
{
    const uint64_t cmask = (uint64_t(1) << col_bits) - 1;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    const int64_t t0 = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    for (int64_t e = t0; e < n_edges; e += stride) {
        const uint64_t k = keys[e];
        indices[e] = int32_t(k & cmask);
        const int64_t r = int64_t(k >> col_bits);
        const int64_t rp = e == 0 ? -1 : int64_t(keys[e - 1] >> col_bits);
        for (int64_t q = rp + 1; q <= r; ++q) indptr[q] = int32_t(e);
    }
    // rows after the last non-empty row (and indptr[n_rows]) = n_edges
    const int64_t last = n_edges == 0 ? -1 : int64_t(keys[n_edges - 1] >> col_bits);
    for (int64_t q = last + 1 + t0; q <= n_rows; q += stride) indptr[q] = int32_t(n_edges);
}

__global__ __launch_bounds__(256) void degree_norm_kernel(const int32_t *__restrict__ indptr, int64_t n,
                                                          int32_t *__restrict__ deg, float *__restrict__ norm)
{
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t v = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; v < n; v += stride) {
        const int32_t d = indptr[v + 1] - indptr[v];
        if (deg) deg[v] = d;
        // train_transductive.py:56-57: pow(deg, -0.5), inf -> 0
        if (norm) norm[v] = d > 0 ? 1.0f / sqrtf(float(d)) : 0.0f;
    }
}

__global__ __launch_bounds__(256) void zero2d_kernel(float *__restrict__ out, int64_t n_rows, int64_t n_cols,
                                                     int64_t ld)
{
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    const int64_t total = n_rows * n_cols;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += stride)
        out[(i / n_cols) * ld + (i % n_cols)] = 0.f;
}

// one wave per row; counts are small integers so the float atomics are exact
__global__ __launch_bounds__(256) void csr_to_dense_kernel(const int32_t *__restrict__ indptr,
                                                           const int32_t *__restrict__ indices, int64_t n_rows,
                                                           float *__restrict__ out, int64_t ld)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wave = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) / kWave;
    const int64_t nwaves = int64_t(gridDim.x) * blockDim.x / kWave;
    for (int64_t r = wave; r < n_rows; r += nwaves)
        for (int32_t e = indptr[r] + lane; e < indptr[r + 1]; e += kWave)
            atomicAdd(&out[r * ld + indices[e]], 1.0f);
}

// Exclusive prefix sums of the selected graphs' node / edge counts ON THE DEVICE (the batch plan of dgl.batch,
// train_inductive.py:34: "node ids offset by the prefix sum of node counts").  One block; the scan runs in chunks of
// 1024 graphs with a carried total.  t_indptr may be NULL (symmetric dataset: the transposed structure is the same).
// graph ids of the next batch of an epoch order that already lives on the device: out_ids[b] = order[cursor * B + b];
// the cursor (device int64) advances by one batch, so a replayed HIP graph walks the epoch
__global__ __launch_bounds__(256) void batch_select_kernel(const int64_t *__restrict__ order, int64_t n_order,
                                                           int64_t *__restrict__ cursor, int64_t B,
                                                           int64_t *__restrict__ out_ids)
{
    const int64_t c = *cursor;
    for (int64_t b = threadIdx.x; b < B; b += 256) {
        const int64_t k = c * B + b;
        out_ids[b] = order[k < n_order ? k : n_order - 1];
    }
    __syncthreads();
    if (threadIdx.x == 0) *cursor = c + 1;
}

__global__ __launch_bounds__(1024) void batch_plan_kernel(const int64_t *__restrict__ graph_ptr,
                                                          const int32_t *__restrict__ indptr,
                                                          const int32_t *__restrict__ t_indptr,
                                                          const int64_t *__restrict__ graph_ids, int64_t n_graphs,
                                                          int64_t *__restrict__ node_ptr, int64_t *__restrict__ edge_ptr,
                                                          int64_t *__restrict__ t_edge_ptr,
                                                          // gae_x_batch_plan_next: the ids come from an epoch order
                                                          const int64_t *__restrict__ order, int64_t n_order,
                                                          int64_t *__restrict__ cursor, int64_t *__restrict__ ids_out)
{
    // A thread owns PLAN_ITEMS consecutive graphs of a 1024 x PLAN_ITEMS chunk: the dependent chain order ->
    // graph_ptr -> indptr of all of them is in flight at once and a 4096-graph batch is ONE pass (it was four passes
    // of three round trips and three barriers each: 39.6 -> 33 us inside the inductive step).
    constexpr int PLAN_ITEMS = 4;
    __shared__ long long wsum[3][16];
    __shared__ long long carry[3];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < 3) carry[tid] = 0;
    const int64_t cur = order ? *cursor : 0;
    __syncthreads();
    for (int64_t base = 0; base < n_graphs; base += 1024 * PLAN_ITEMS) {
        const int64_t b0 = base + int64_t(tid) * PLAN_ITEMS;
        int64_t g[PLAN_ITEMS];
#pragma unroll
        for (int i = 0; i < PLAN_ITEMS; ++i) {
            const int64_t b = b0 + i;
            const int64_t bc = b < n_graphs ? b : n_graphs - 1;           // clamped: loads stay branch-free
            if (order) {
                const int64_t k = cur * n_graphs + bc;
                g[i] = order[k < n_order ? k : n_order - 1];
            } else {
                g[i] = graph_ids[bc];
            }
        }
        int64_t n0[PLAN_ITEMS], n1[PLAN_ITEMS];
#pragma unroll
        for (int i = 0; i < PLAN_ITEMS; ++i) { n0[i] = graph_ptr[g[i]]; n1[i] = graph_ptr[g[i] + 1]; }
        long long v[PLAN_ITEMS][3];
#pragma unroll
        for (int i = 0; i < PLAN_ITEMS; ++i) {
            const bool live = b0 + i < n_graphs;
            const long long e = (long long)indptr[n1[i]] - indptr[n0[i]];
            const long long te = t_indptr ? (long long)t_indptr[n1[i]] - t_indptr[n0[i]] : e;
            v[i][0] = live ? n1[i] - n0[i] : 0;
            v[i][1] = live ? e : 0;
            v[i][2] = live ? te : 0;
            if (order && live) ids_out[b0 + i] = g[i];
        }
        long long tot[3], inc[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            tot[q] = 0;
#pragma unroll
            for (int i = 0; i < PLAN_ITEMS; ++i) tot[q] += v[i][q];
            long long x = tot[q];
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const long long y = __shfl_up(x, off, 64);
                if (lane >= off) x += y;
            }
            inc[q] = x;                                      // inclusive over the threads of this wave
            if (lane == 63) wsum[q][wv] = x;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            long long pre = carry[q];
            for (int w2 = 0; w2 < wv; ++w2) pre += wsum[q][w2];
            inc[q] += pre;                                   // inclusive sum up to this thread's last graph
        }
        long long run[3] = {inc[0] - tot[0], inc[1] - tot[1], inc[2] - tot[2]};   // exclusive start of this thread
#pragma unroll
        for (int i = 0; i < PLAN_ITEMS; ++i) {
            const int64_t b = b0 + i;
            if (b < n_graphs) {
                node_ptr[b] = run[0];
                edge_ptr[b] = run[1];
                if (t_edge_ptr) t_edge_ptr[b] = run[2];
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) run[q] += v[i][q];
            if (b == n_graphs - 1) {
                node_ptr[n_graphs] = run[0];
                edge_ptr[n_graphs] = run[1];
                if (t_edge_ptr) t_edge_ptr[n_graphs] = run[2];
            }
        }
        __syncthreads();
        if (tid == 1023) {
#pragma unroll
            for (int q = 0; q < 3; ++q) carry[q] = inc[q];
        }
        __syncthreads();
    }
    if (order && tid == 0) *cursor = cur + 1;      // every thread read the cursor before the first barrier
}

template <typename TI, typename TO>
__device__ __forceinline__ TO feat_convert(TI v);
template <> __device__ __forceinline__ float feat_convert<float, float>(float v) { return v; }
template <> __device__ __forceinline__ unsigned short feat_convert<unsigned short, unsigned short>(unsigned short v) { return v; }
template <> __device__ __forceinline__ float feat_convert<unsigned char, float>(unsigned char v) { return float(v); }

// One wave per selected graph: rebase its indptr slice, its column ids and copy (uint8 -> fp32: expand) its feature
// rows into the batch (block-diagonal) arrays; optionally write the packed neighbour table of the batch CSR
// (gae_spmm_ell_build's format) in the same pass -- the batch's SpMM launches then start with one table load.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void batch_gather_kernel(
    const int64_t *__restrict__ graph_ptr, const int32_t *__restrict__ ds_indptr,
    const int32_t *__restrict__ ds_indices, const TI *__restrict__ ds_feat, int64_t ld_feat, int64_t F,
    const int64_t *__restrict__ graph_ids, int64_t n_graphs, const int64_t *__restrict__ out_node_ptr,
    const int64_t *__restrict__ out_edge_ptr, int32_t *__restrict__ out_indptr,
    int32_t *__restrict__ out_indices, TO *__restrict__ out_feat, int64_t ld_out, int32_t *__restrict__ out_ell,
    int ell_width, int64_t cap_nodes, int64_t cap_edges, int64_t *__restrict__ out_counts,
    // gae_x_batch_gather_next (order != NULL): select + plan + gather in ONE launch for batches of a few hundred graphs --
    // every wave adds up the sizes of the graphs in front of its own itself (two coalesced passes over <= 1024 ids)
    const int64_t *__restrict__ order, int64_t n_order, int64_t *__restrict__ cursor, int64_t *__restrict__ ids_out,
    int64_t *__restrict__ node_ptr_out, int64_t *__restrict__ edge_ptr_out)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t b = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) / kWave;
    int64_t in_on = 0, in_oe = 0, in_tn = 0, in_te = 0, in_g = 0, in_cur = 0;
    // the wave's own graph as the prefix pass saw it (the lane that visited q == b broadcasts: two dependent round
    // trips -- graph_ptr[g], ds_indptr[n0] -- less in front of the copies)
    long long own_g = 0, own_n0 = 0, own_n1 = 0;
    int own_e0 = 0, own_e1 = 0;
    if (order) {
        in_cur = *cursor;
        long long pn = 0, pe = 0, tn = 0, te = 0;
        for (int64_t q = lane; q < n_graphs; q += kWave) {
            const int64_t k = in_cur * n_graphs + q;
            const int64_t gq = order[k < n_order ? k : n_order - 1];
            const int64_t a0 = graph_ptr[gq], a1 = graph_ptr[gq + 1];
            const int ei0 = ds_indptr[a0], ei1 = ds_indptr[a1];
            const long long nn_ = a1 - a0, ee_ = (long long)ei1 - ei0;
            tn += nn_; te += ee_;
            if (q < b) { pn += nn_; pe += ee_; }
            if (q == b) { own_g = gq; own_n0 = a0; own_n1 = a1; own_e0 = ei0; own_e1 = ei1; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            pn += __shfl_xor(pn, off, 64); pe += __shfl_xor(pe, off, 64);
            tn += __shfl_xor(tn, off, 64); te += __shfl_xor(te, off, 64);
        }
        in_on = pn; in_oe = pe; in_tn = tn; in_te = te;
        if (b < n_graphs) {
            const int src = int(b % kWave);
            own_g = __shfl(own_g, src, 64); own_n0 = __shfl(own_n0, src, 64); own_n1 = __shfl(own_n1, src, 64);
            own_e0 = __shfl(own_e0, src, 64); own_e1 = __shfl(own_e1, src, 64);
            in_g = own_g;
            if (lane == 0) { ids_out[b] = in_g; node_ptr_out[b] = in_on; edge_ptr_out[b] = in_oe; }
        }
        if (b == 0 && lane == 0) { node_ptr_out[n_graphs] = in_tn; edge_ptr_out[n_graphs] = in_te; }
        // the cursor advances once every block has read it: the last block to pass here (ticket in out_counts[3])
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long t = atomicAdd(reinterpret_cast<unsigned long long *>(&out_counts[3]), 1ull);
            if (t == gridDim.x - 1ull) { out_counts[3] = 0; *cursor = in_cur + 1; }
        }
    }
    if (b >= n_graphs) {
        // ---- fixed-capacity batch (cap_nodes > 0): the waves behind the last graph turn the rows [N_b, cap_nodes)
        //      into isolated zero-feature nodes -- empty CSR rows, zero features, an empty table row -- so that a
        //      captured HIP graph can run every batch on the same shapes; the true sizes go to out_counts
        if (cap_nodes <= 0) return;
        // Device-side guard: the host sized the buffers from the batches it knew about; a batch that does not fit
        // (a replay past the validated batches, a stale order) must never write behind them.  Only the longest
        // prefix of graphs that fits is gathered (the graph waves below test the same condition), the rest of the
        // buffers becomes padding and the number of dropped graphs is ADDED to out_counts[2] for the host to see.
        int64_t keep = n_graphs, nb, eb;
        if (order) {
            nb = in_tn; eb = in_te;
            if (nb > cap_nodes || eb > cap_edges) {          // (never on a validated order) serial walk to the last fit
                nb = eb = 0; keep = 0;
                for (int64_t q = 0; q < n_graphs; ++q) {
                    const int64_t k = in_cur * n_graphs + q;
                    const int64_t gq = order[k < n_order ? k : n_order - 1];
                    const int64_t a0 = graph_ptr[gq], a1 = graph_ptr[gq + 1];
                    const int64_t e_ = int64_t(ds_indptr[a1]) - ds_indptr[a0];
                    if (nb + (a1 - a0) > cap_nodes || eb + e_ > cap_edges) break;
                    nb += a1 - a0; eb += e_; keep = q + 1;
                }
            }
        } else {
            if (out_node_ptr[n_graphs] > cap_nodes || out_edge_ptr[n_graphs] > cap_edges) {
                int64_t lo = 0, hi = n_graphs;               // largest k with node_ptr[k] / edge_ptr[k] inside
                while (lo < hi) {
                    const int64_t mid = (lo + hi + 1) >> 1;
                    if (out_node_ptr[mid] <= cap_nodes && out_edge_ptr[mid] <= cap_edges) lo = mid; else hi = mid - 1;
                }
                keep = lo;
            }
            nb = out_node_ptr[keep]; eb = out_edge_ptr[keep];
        }
        const int64_t n_pad_waves = (int64_t(gridDim.x) * blockDim.x) / kWave - n_graphs;
        const int64_t w = b - n_graphs;
        if (w == 0 && lane == 0) {
            out_indptr[nb] = int32_t(eb);                    // closes the last kept graph (or opens an empty batch)
            if (out_counts) { out_counts[0] = nb; out_counts[1] = eb; out_counts[2] += n_graphs - keep; }
        }
        for (int64_t i = nb + w; i < cap_nodes; i += n_pad_waves) {
            if (lane == 0) out_indptr[i + 1] = int32_t(eb);
            if (out_ell)
                for (int k = lane; k < ell_width; k += kWave) out_ell[i * ell_width + k] = -1;
            if (out_feat)
                for (int64_t c = lane; c < ld_out; c += kWave) out_feat[i * ld_out + c] = TO(0);
        }
        return;
    }
    const int64_t g = order ? in_g : graph_ids[b];
    const int64_t n0 = order ? int64_t(own_n0) : graph_ptr[g], n1 = order ? int64_t(own_n1) : graph_ptr[g + 1];
    const int64_t on = order ? in_on : out_node_ptr[b], oe = order ? in_oe : out_edge_ptr[b];
    const int32_t e0 = order ? own_e0 : ds_indptr[n0], e1 = order ? own_e1 : ds_indptr[n1];
    const int64_t nn = n1 - n0;
    if (cap_nodes > 0 && (on + nn > cap_nodes || oe + (e1 - e0) > cap_edges)) return;   // see the guard above
    for (int64_t i = lane; i < nn; i += kWave)
        out_indptr[on + i] = int32_t(ds_indptr[n0 + i] - e0 + oe);
    if (b == n_graphs - 1 && lane == 0) out_indptr[on + nn] = int32_t(oe + (e1 - e0));
    const int64_t shift = on - n0;
    for (int32_t e = e0 + lane; e < e1; e += kWave)
        out_indices[oe + (e - e0)] = int32_t(ds_indices[e] + shift);
    // 32-bit index arithmetic below: a member graph has far fewer than 2^31 / ld_out rows, and a 64-bit division per
    // copied element was most of this kernel's time on small batches
    const unsigned nn32 = unsigned(nn);
    if (out_ell && nn32 > 0) {
        // same treatment as the feature copy below: UNR independent (row pointer -> column id) chains per trip,
        // branch-free (clamped addresses, selection after the loads); an edge-less molecule loads no column ids
        constexpr int UNR = 4;
        const unsigned W = unsigned(ell_width), total = nn32 * W;
        const bool has_edges = e1 > e0;                              // wave-uniform
        for (unsigned base = lane; base < total; base += kWave * UNR) {
            int32_t p[UNR], deg[UNR], col[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                unsigned idx = base + u * kWave;
                idx = idx < total ? idx : total - 1;
                const unsigned i = idx / W;
                p[u] = ds_indptr[n0 + i];
                deg[u] = ds_indptr[n0 + i + 1] - p[u];
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                unsigned idx = base + u * kWave;
                idx = idx < total ? idx : total - 1;
                const unsigned i = idx / W, k = idx - i * W;
                int32_t at = p[u] + int32_t(k);
                at = at < e1 ? at : e1 - 1;
                col[u] = has_edges ? ds_indices[at] : 0;
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const unsigned idx = base + u * kWave;
                const unsigned i = idx / W, k = idx - i * W;
                int32_t v;
                if (deg[u] > int32_t(W) && k == W - 1) v = -2;       // row continues in the CSR arrays
                else v = int32_t(k) < deg[u] ? int32_t(col[u] + shift) : -1;
                if (idx < total) out_ell[(on + i) * W + k] = v;
            }
        }
    }
    if (out_feat && nn32 > 0) {            // whole output rows: the pad columns [F, ld_out) are written as zeros
        // a molecule's copy is a chain of dependent round trips (cold rows of the dataset), not a byte stream: every
        // trip issues UNR independent loads -- branch-free: indices clamped into the molecule, values selected after
        // the loads -- before its first store (38 atoms x 40 columns: 3 trips instead of 24)
        constexpr int UNR = 8;
        const unsigned ldo = unsigned(ld_out), total = nn32 * ldo, F32 = unsigned(F);
        for (unsigned base = lane; base < total; base += kWave * UNR) {
            TI raw[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                unsigned i = base + u * kWave;
                i = i < total ? i : total - 1;
                const unsigned r = i / ldo, c = i - r * ldo;
                raw[u] = ds_feat[(n0 + r) * ld_feat + (c < F32 ? c : 0)];
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const unsigned i = base + u * kWave;
                const unsigned r = i / ldo, c = i - r * ldo;
                if (i < total) out_feat[(on + r) * ld_out + c] = c < F32 ? feat_convert<TI, TO>(raw[u]) : TO(0);
            }
        }
    }
}

inline int grid_for(int64_t n, int block = 256, int cap = 256 * 8)
{
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return int(g);
}

size_t sort_temp_bytes(int64_t n_edges, int end_bit)
{
    size_t bytes = 0;
    uint64_t *nul = nullptr;
    (void)rocprim::radix_sort_keys(nullptr, bytes, nul, nul, size_t(n_edges), 0u, unsigned(end_bit),
                                   hipStream_t(0));
    return bytes;
}

} // namespace

extern "C" int64_t gae_csr_from_coo_workspace_bytes(int64_t n_edges, int64_t n_rows)
{
    if (n_edges < 0 || n_rows < 0) return GAE_E_SIZE;
    (void)n_rows;
    const int64_t keys = align_up(n_edges * 8, 256);
    // temp size depends (weakly) on the bit range; query with the widest one
    return 2 * keys + align_up(int64_t(sort_temp_bytes(n_edges > 0 ? n_edges : 1, 64)), 256) + 256;
}

extern "C" int gae_csr_from_coo(const int64_t *row, const int64_t *col, int64_t n_edges, int64_t n_rows,
                                int64_t n_cols, int32_t *indptr, int32_t *indices, void *workspace,
                                int64_t workspace_bytes, int32_t *status_dev, void *stream)
{
    GAE_REQUIRE(n_edges >= 0 && n_rows >= 0 && n_cols >= 0, GAE_E_SIZE, "gae_csr_from_coo: negative size");
    GAE_REQUIRE(n_edges < (int64_t(1) << 31) && n_rows < (int64_t(1) << 31) - 1 && n_cols < (int64_t(1) << 31),
                GAE_E_SIZE, "gae_csr_from_coo: int32 CSR supports < 2^31 edges/rows/cols");
    GAE_REQUIRE(indptr != nullptr, GAE_E_NULL, "gae_csr_from_coo: indptr is NULL");
    GAE_REQUIRE(n_edges == 0 || (row && col && indices && workspace), GAE_E_NULL,
                "gae_csr_from_coo: NULL pointer with n_edges > 0");
    GAE_REQUIRE(n_edges == 0 || (n_rows > 0 && n_cols > 0), GAE_E_SIZE,
                "gae_csr_from_coo: edges given for an empty graph");
    hipStream_t s = gae::as_stream(stream);
    if (n_edges == 0) {
        hipLaunchKernelGGL(unpack_fill_kernel, dim3(grid_for(n_rows + 1)), dim3(256), 0, s, nullptr, int64_t(0),
                           n_rows, 1, indptr, indices);
        GAE_CHECK_LAUNCH("unpack_fill_kernel");
        return GAE_OK;
    }
    const int col_bits = bits_for(n_cols);
    const int end_bit = col_bits + bits_for(n_rows);
    const int64_t keys_bytes = align_up(n_edges * 8, 256);
    size_t temp_bytes = sort_temp_bytes(n_edges, end_bit);
    GAE_REQUIRE(workspace_bytes >= 2 * keys_bytes + int64_t(temp_bytes), GAE_E_WORKSPACE,
                "gae_csr_from_coo: workspace %lld < %lld bytes", (long long)workspace_bytes,
                (long long)(2 * keys_bytes + int64_t(temp_bytes)));
    GAE_REQUIRE(gae::aligned16(workspace), GAE_E_ALIGN, "gae_csr_from_coo: workspace not 16-byte aligned");
    char *ws = static_cast<char *>(workspace);
    uint64_t *keys_a = reinterpret_cast<uint64_t *>(ws);
    uint64_t *keys_b = reinterpret_cast<uint64_t *>(ws + keys_bytes);
    void *temp = ws + 2 * keys_bytes;
    hipLaunchKernelGGL(pack_keys_kernel, dim3(grid_for(n_edges)), dim3(256), 0, s, row, col, n_edges, n_rows,
                       n_cols, col_bits, keys_a, status_dev);
    GAE_CHECK_LAUNCH("pack_keys_kernel");
    GAE_HIP(rocprim::radix_sort_keys(temp, temp_bytes, keys_a, keys_b, size_t(n_edges), 0u, unsigned(end_bit), s));
    hipLaunchKernelGGL(unpack_fill_kernel, dim3(grid_for(n_edges)), dim3(256), 0, s, keys_b, n_edges, n_rows,
                       col_bits, indptr, indices);
    GAE_CHECK_LAUNCH("unpack_fill_kernel");
    return GAE_OK;
}

// ---------------------------------------------------------------------------
// Row pack of the multi-GPU exchange (gae_dgl_amd/parallel.py): out[i] = H[idx[i]] (idx == NULL: H[i]) for
// i < n_rows, zero rows behind up to n_out_rows.  The rows a rank sends to its peers go STRAIGHT into the send
// buffer of the all-to-all, its own rows straight into their slot of the buffer the local CSR indexes -- no ATen
// index_select / cat on the path between two products.  One thread per 16-byte vector (rows of whole vectors),
// one per element otherwise.
// ---------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void rows_pack_kernel(const float *__restrict__ H, int64_t ldh,
                                                        const int64_t *__restrict__ idx, int64_t n_rows,
                                                        int64_t n_out_rows, int64_t F, float *__restrict__ out,
                                                        int64_t ldo)
{
    const int64_t vpr = (F + VEC - 1) / VEC;
    const int64_t total = n_out_rows * vpr;
    for (int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x; e < total; e += int64_t(gridDim.x) * 256) {
        const int64_t i = e / vpr, c = (e - i * vpr) * VEC;
        if (VEC == 4) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < n_rows) v = *reinterpret_cast<const float4 *>(H + (idx ? idx[i] : i) * ldh + c);
            *reinterpret_cast<float4 *>(out + i * ldo + c) = v;
        } else {
            out[i * ldo + c] = i < n_rows ? H[(idx ? idx[i] : i) * ldh + c] : 0.f;
        }
    }
}

extern "C" int gae_rows_pack(const float *H, int64_t ldh, int64_t n_src_rows, const int64_t *idx, int64_t n_rows,
                             int64_t n_out_rows, int64_t F, float *out, int64_t ldo, void *stream)
{
    GAE_REQUIRE(n_rows >= 0 && n_out_rows >= n_rows && F >= 0 && n_src_rows >= 0, GAE_E_SIZE,
                "gae_rows_pack: bad sizes");
    GAE_REQUIRE(ldh >= F && ldo >= F, GAE_E_SIZE, "gae_rows_pack: leading dimension < F");
    GAE_REQUIRE(idx != nullptr || n_rows <= n_src_rows, GAE_E_SIZE, "gae_rows_pack: more rows than the source has");
    if (n_out_rows == 0 || F == 0) return GAE_OK;
    GAE_REQUIRE(out && (n_rows == 0 || H), GAE_E_NULL, "gae_rows_pack: NULL pointer");
    const bool vec = F % 4 == 0 && ldh % 4 == 0 && ldo % 4 == 0 && gae::aligned16(H) && gae::aligned16(out);
    const int64_t total = n_out_rows * (vec ? F / 4 : F);
    const dim3 grid(unsigned(grid_for(total, 256, 256 * 16)));
    if (vec) hipLaunchKernelGGL(rows_pack_kernel<4>, grid, dim3(256), 0, gae::as_stream(stream), H, ldh, idx, n_rows,
                                n_out_rows, F, out, ldo);
    else hipLaunchKernelGGL(rows_pack_kernel<1>, grid, dim3(256), 0, gae::as_stream(stream), H, ldh, idx, n_rows,
                            n_out_rows, F, out, ldo);
    GAE_CHECK_LAUNCH("rows_pack_kernel");
    return GAE_OK;
}

extern "C" int gae_degree_norm(const int32_t *indptr, int64_t n_rows, int32_t *deg_out, float *norm_out,
                               void *stream)
{
    GAE_REQUIRE(n_rows >= 0, GAE_E_SIZE, "gae_degree_norm: negative n_rows");
    GAE_REQUIRE(indptr != nullptr, GAE_E_NULL, "gae_degree_norm: indptr is NULL");
    if (n_rows == 0 || (!deg_out && !norm_out)) return GAE_OK;
    hipLaunchKernelGGL(degree_norm_kernel, dim3(grid_for(n_rows)), dim3(256), 0, gae::as_stream(stream), indptr,
                       n_rows, deg_out, norm_out);
    GAE_CHECK_LAUNCH("degree_norm_kernel");
    return GAE_OK;
}

extern "C" int gae_csr_to_dense(const int32_t *indptr, const int32_t *indices, int64_t n_rows, int64_t n_cols,
                                float *out, int64_t ld, void *stream)
{
    GAE_REQUIRE(n_rows >= 0 && n_cols >= 0 && ld >= n_cols, GAE_E_SIZE, "gae_csr_to_dense: bad sizes");
    if (n_rows == 0 || n_cols == 0) return GAE_OK;
    GAE_REQUIRE(indptr && out, GAE_E_NULL, "gae_csr_to_dense: NULL pointer");
    hipStream_t s = gae::as_stream(stream);
    hipLaunchKernelGGL(zero2d_kernel, dim3(grid_for(n_rows * n_cols)), dim3(256), 0, s, out, n_rows, n_cols, ld);
    GAE_CHECK_LAUNCH("zero2d_kernel");
    hipLaunchKernelGGL(csr_to_dense_kernel, dim3(grid_for(n_rows * kWave)), dim3(256), 0, s, indptr, indices,
                       n_rows, out, ld);
    GAE_CHECK_LAUNCH("csr_to_dense_kernel");
    return GAE_OK;
}

extern "C" int gae_batch_select(const int64_t *order, int64_t n_order, int64_t *cursor_dev, int64_t batch_graphs,
                                int64_t *out_ids, void *stream)
{
    GAE_REQUIRE(n_order > 0 && batch_graphs > 0, GAE_E_SIZE, "gae_batch_select: sizes must be positive");
    GAE_REQUIRE(order && cursor_dev && out_ids, GAE_E_NULL, "gae_batch_select: NULL pointer");
    hipLaunchKernelGGL(batch_select_kernel, dim3(1), dim3(256), 0, gae::as_stream(stream), order, n_order, cursor_dev,
                       batch_graphs, out_ids);
    GAE_CHECK_LAUNCH("batch_select_kernel");
    return GAE_OK;
}

extern "C" int gae_batch_plan(const int64_t *graph_ptr, const int32_t *ds_indptr, const int32_t *ds_t_indptr,
                              const int64_t *graph_ids, int64_t n_graphs, int64_t *out_node_ptr,
                              int64_t *out_edge_ptr, int64_t *out_t_edge_ptr, void *stream)
{
    GAE_REQUIRE(n_graphs >= 0, GAE_E_SIZE, "gae_batch_plan: negative n_graphs");
    GAE_REQUIRE(out_node_ptr && out_edge_ptr, GAE_E_NULL, "gae_batch_plan: NULL output");
    hipStream_t s = gae::as_stream(stream);
    if (n_graphs == 0) {
        GAE_HIP(hipMemsetAsync(out_node_ptr, 0, sizeof(int64_t), s));
        GAE_HIP(hipMemsetAsync(out_edge_ptr, 0, sizeof(int64_t), s));
        if (out_t_edge_ptr) GAE_HIP(hipMemsetAsync(out_t_edge_ptr, 0, sizeof(int64_t), s));
        return GAE_OK;
    }
    GAE_REQUIRE(graph_ptr && ds_indptr && graph_ids, GAE_E_NULL, "gae_batch_plan: NULL pointer");
    hipLaunchKernelGGL(batch_plan_kernel, dim3(1), dim3(1024), 0, s, graph_ptr, ds_indptr, ds_t_indptr, graph_ids,
                       n_graphs, out_node_ptr, out_edge_ptr, out_t_edge_ptr, nullptr, 0, nullptr, nullptr);
    GAE_CHECK_LAUNCH("batch_plan_kernel");
    return GAE_OK;
}

extern "C" int gae_x_batch_plan_next(const int64_t *graph_ptr, const int32_t *ds_indptr, const int32_t *ds_t_indptr,
                                   const int64_t *order, int64_t n_order, int64_t *cursor_dev, int64_t n_graphs,
                                   int64_t *out_ids, int64_t *out_node_ptr, int64_t *out_edge_ptr,
                                   int64_t *out_t_edge_ptr, void *stream)
{
    GAE_REQUIRE(n_graphs > 0 && n_order > 0, GAE_E_SIZE, "gae_x_batch_plan_next: sizes must be positive");
    GAE_REQUIRE(graph_ptr && ds_indptr && order && cursor_dev && out_ids && out_node_ptr && out_edge_ptr, GAE_E_NULL,
                "gae_x_batch_plan_next: NULL pointer");
    hipLaunchKernelGGL(batch_plan_kernel, dim3(1), dim3(1024), 0, gae::as_stream(stream), graph_ptr, ds_indptr,
                       ds_t_indptr, nullptr, n_graphs, out_node_ptr, out_edge_ptr, out_t_edge_ptr, order, n_order,
                       cursor_dev, out_ids);
    GAE_CHECK_LAUNCH("batch_plan_kernel");
    return GAE_OK;
}

extern "C" int gae_batch_gather(const int64_t *graph_ptr, const int32_t *ds_indptr, const int32_t *ds_indices,
                                const void *ds_feat, int64_t ld_feat, int64_t F, int dtype,
                                const int64_t *graph_ids, int64_t n_graphs, const int64_t *out_node_ptr,
                                const int64_t *out_edge_ptr, int64_t n_batch_nodes, int64_t n_batch_edges,
                                int32_t *out_indptr, int32_t *out_indices, void *out_feat, int64_t ld_out,
                                int32_t *out_ell, int32_t ell_width, int64_t cap_nodes, int64_t *out_counts,
                                void *stream)
{
    GAE_REQUIRE(cap_nodes == 0 || cap_nodes >= n_batch_nodes, GAE_E_SIZE,
                "gae_batch_gather: cap_nodes smaller than the batch");
    GAE_REQUIRE(n_graphs >= 0 && F >= 0 && n_batch_nodes >= 0 && n_batch_edges >= 0, GAE_E_SIZE,
                "gae_batch_gather: negative size");
    GAE_REQUIRE(ld_feat >= F && ld_out >= F, GAE_E_SIZE, "gae_batch_gather: leading dimension < F");
    GAE_REQUIRE(dtype == GAE_F32 || dtype == GAE_BF16 || dtype == GAE_U8, GAE_E_DTYPE, "gae_batch_gather: dtype %d",
                dtype);
    GAE_REQUIRE(out_indptr != nullptr, GAE_E_NULL, "gae_batch_gather: out_indptr is NULL");
    GAE_REQUIRE(!out_ell || ell_width == 4 || ell_width == 8 || ell_width == GAE_SPMM_ELL_WIDTH, GAE_E_RANGE,
                "gae_batch_gather: ell_width must be 4, 8 or %d", GAE_SPMM_ELL_WIDTH);
    hipStream_t s = gae::as_stream(stream);
    if (n_graphs == 0) {
        GAE_HIP(hipMemsetAsync(out_indptr, 0, sizeof(int32_t), s));
        return GAE_OK;
    }
    GAE_REQUIRE(graph_ptr && ds_indptr && graph_ids && out_node_ptr && out_edge_ptr, GAE_E_NULL,
                "gae_batch_gather: NULL pointer");
    GAE_REQUIRE(n_batch_edges == 0 || (ds_indices && out_indices), GAE_E_NULL,
                "gae_batch_gather: NULL index pointer");
    GAE_REQUIRE(F == 0 || n_batch_nodes == 0 || !out_feat || ds_feat, GAE_E_NULL,
                "gae_batch_gather: NULL feature pointer");
    // one wave per graph (+ 256 waves that write the padding of a fixed-capacity batch)
    const int64_t blocks = ((n_graphs + (cap_nodes > 0 ? 256 : 0)) * kWave + 255) / 256;
#define GAE_BG(TI, TO)                                                                                              \
    hipLaunchKernelGGL((batch_gather_kernel<TI, TO>), dim3(unsigned(blocks)), dim3(256), 0, s, graph_ptr, ds_indptr,  \
                       ds_indices, static_cast<const TI *>(ds_feat), ld_feat, F, graph_ids, n_graphs, out_node_ptr,  \
                       out_edge_ptr, out_indptr, out_indices, static_cast<TO *>(out_feat), ld_out, out_ell,          \
                       int(ell_width), cap_nodes, n_batch_edges, out_counts, nullptr, int64_t(0), nullptr, nullptr,     \
                       nullptr, nullptr)
    if (dtype == GAE_F32) GAE_BG(float, float);
    else if (dtype == GAE_BF16) GAE_BG(unsigned short, unsigned short);
    else GAE_BG(unsigned char, float);
#undef GAE_BG
    GAE_CHECK_LAUNCH("batch_gather_kernel");
    return GAE_OK;
}

// gae_x_batch_plan_next + gae_batch_gather in one launch (fixed-capacity batches of <= 1024 graphs): the captured
// inductive step of the reference's default batch (128 molecules, gae_dgl/train_inductive.py:24) is bound by its
// number of kernel nodes, and the plan of so few graphs is cheaper to recompute in every wave than to launch.
extern "C" int gae_x_batch_gather_next(const int64_t *graph_ptr, const int32_t *ds_indptr, const int32_t *ds_indices,
                                     const void *ds_feat, int64_t ld_feat, int64_t F, int dtype, const int64_t *order,
                                     int64_t n_order, int64_t *cursor_dev, int64_t n_graphs, int64_t *out_ids,
                                     int64_t *out_node_ptr, int64_t *out_edge_ptr, int64_t cap_nodes, int64_t cap_edges,
                                     int32_t *out_indptr, int32_t *out_indices, void *out_feat, int64_t ld_out,
                                     int32_t *out_ell, int32_t ell_width, int64_t *out_counts, void *stream)
{
    GAE_REQUIRE(n_graphs >= 1 && n_graphs <= 1024 && n_order >= 1, GAE_E_RANGE,
                "gae_x_batch_gather_next: 1 .. 1024 graphs per batch (larger batches: gae_x_batch_plan_next + gae_batch_gather)");
    GAE_REQUIRE(cap_nodes >= 1 && cap_edges >= 0 && F >= 0, GAE_E_SIZE, "gae_x_batch_gather_next: bad capacities");
    GAE_REQUIRE(ld_feat >= F && ld_out >= F, GAE_E_SIZE, "gae_x_batch_gather_next: leading dimension < F");
    GAE_REQUIRE(dtype == GAE_F32 || dtype == GAE_BF16 || dtype == GAE_U8, GAE_E_DTYPE, "gae_x_batch_gather_next: dtype %d", dtype);
    GAE_REQUIRE(graph_ptr && ds_indptr && order && cursor_dev && out_ids && out_node_ptr && out_edge_ptr && out_indptr &&
                    out_counts, GAE_E_NULL, "gae_x_batch_gather_next: NULL pointer");
    GAE_REQUIRE(cap_edges == 0 || (ds_indices && out_indices), GAE_E_NULL, "gae_x_batch_gather_next: NULL index pointer");
    GAE_REQUIRE(F == 0 || !out_feat || ds_feat, GAE_E_NULL, "gae_x_batch_gather_next: NULL feature pointer");
    GAE_REQUIRE(!out_ell || ell_width == 4 || ell_width == 8 || ell_width == GAE_SPMM_ELL_WIDTH, GAE_E_RANGE,
                "gae_x_batch_gather_next: ell_width must be 4, 8 or %d", GAE_SPMM_ELL_WIDTH);
    hipStream_t s = gae::as_stream(stream);
    const int64_t blocks = ((n_graphs + 256) * kWave + 255) / 256;       // one wave per graph + 256 padding waves
#define GAE_BGN(TI, TO)                                                                                              \
    hipLaunchKernelGGL((batch_gather_kernel<TI, TO>), dim3(unsigned(blocks)), dim3(256), 0, s, graph_ptr, ds_indptr,   \
                       ds_indices, static_cast<const TI *>(ds_feat), ld_feat, F, nullptr, n_graphs, nullptr, nullptr,  \
                       out_indptr, out_indices, static_cast<TO *>(out_feat), ld_out, out_ell, int(ell_width),          \
                       cap_nodes, cap_edges, out_counts, order, n_order, cursor_dev, out_ids, out_node_ptr,            \
                       out_edge_ptr)
    if (dtype == GAE_F32) GAE_BGN(float, float);
    else if (dtype == GAE_BF16) GAE_BGN(unsigned short, unsigned short);
    else GAE_BGN(unsigned char, float);
#undef GAE_BGN
    GAE_CHECK_LAUNCH("batch_gather_kernel (select + plan + gather)");
    return GAE_OK;
}
