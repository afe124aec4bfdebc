// The last reduction of the fused decoder + BCE loss (dense partials, edge partials, analytic terms -> the scalar),
// as a device function of ONE block: bce_finalize_kernel (decoder_bce.hip) runs it as its own launch, adam_step_kernel
// (optim.hip) as one extra block of the optimiser launch when the caller deferred it (gae_x_decoder_bce_defer_finalize):
// the scalar is not an input of the backward pass, so inside a captured training step the dependent ~5 us launch
// disappears.  The order of the sums is that of a 1024-thread block whatever the real block size NT (a thread plays
// 1024 / NT virtual threads, a wave 1024 / NT virtual waves): both forms give the same bits.
#pragma once
#include "common.h"

namespace gae {

template <int NT>
__device__ __forceinline__ void bce_finalize_block(const gae_bce_tail &t, double (*red)[16] /* LDS [3][16] */)
{
    static_assert(1024 % NT == 0 && NT % 64 == 0, "virtual 1024-thread block");
    if (t.loss_out == nullptr) return;
    double inv_n2 = t.inv_n2, pad_terms = t.pad_terms;
    if (t.scal) { inv_n2 = t.scal[1]; pad_terms = t.scal[2]; }
    const int lane = threadIdx.x & 63;
    const double2 *dp2 = reinterpret_cast<const double2 *>(t.dense_partial);   // {sum |x|, sum log2 t} pairs
    // every virtual thread keeps the order of the 1024-thread kernel (quads (k, k + 1024, k + 2048, k + 3072) added
    // pairwise, then single elements), but the V virtual threads of a real thread advance side by side: their loads
    // are independent and go out together (one after the other they made the tail block the longest of the launch)
    constexpr int V = 1024 / NT;
    double a[V], l[V], e[V];
    int64_t k[V];
#pragma unroll
    for (int v = 0; v < V; ++v) { a[v] = l[v] = e[v] = 0.0; k[v] = int64_t(threadIdx.x) + v * NT; }
    for (bool any = true; any;) {
        any = false;
#pragma unroll
        for (int v = 0; v < V; ++v)
            if (k[v] + 3 * 1024 < t.n_dense) {
                const double2 v0 = dp2[k[v]], v1 = dp2[k[v] + 1024], v2 = dp2[k[v] + 2048], v3 = dp2[k[v] + 3072];
                a[v] += (v0.x + v1.x) + (v2.x + v3.x);
                l[v] += (v0.y + v1.y) + (v2.y + v3.y);
                k[v] += 4 * 1024;
                any = true;
            }
    }
    for (bool any = true; any;) {
        any = false;
#pragma unroll
        for (int v = 0; v < V; ++v)
            if (k[v] < t.n_dense) {
                const double2 w = dp2[k[v]];
                a[v] += w.x; l[v] += w.y;
                k[v] += 1024;
                any = true;
            }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) k[v] = int64_t(threadIdx.x) + v * NT;
    for (bool any = true; any;) {
        any = false;
#pragma unroll
        for (int v = 0; v < V; ++v)
            if (k[v] < t.n_edge) {
                e[v] += t.edge_partial[k[v]];
                k[v] += 1024;
                any = true;
            }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) {
        double av = a[v], lv = l[v], ev = e[v];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            av += __shfl_down(av, off, 64); lv += __shfl_down(lv, off, 64); ev += __shfl_down(ev, off, 64);
        }
        const int tid = int(threadIdx.x) + v * NT;          // the virtual thread
        if (lane == 0) { red[0][tid >> 6] = av; red[1][tid >> 6] = lv; red[2][tid >> 6] = ev; }
    }
    __syncthreads();
    // an additive term on top (VGAE: the KL partials of gae_x_vgae_head_prep), summed by the first wave: lane l takes
    // partials l, l + 64, ..., then a fixed shuffle tree
    double extra = 0.0;
    if (t.kl_partial != nullptr && threadIdx.x < 64) {
        for (int64_t q = threadIdx.x; q < t.n_kl; q += 64) extra += t.kl_partial[q];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) extra += __shfl_down(extra, off, 64);
    }
    if (threadIdx.x == 0) {
        double a = 0.0, l = 0.0, e = 0.0;
        for (int w = 0; w < 16; ++w) { a += red[0][w]; l += red[1][w]; e += red[2][w]; }
        double sx = 0.0;
        for (int q = 0; q < t.DP; ++q) sx += t.S[q] * t.S[t.DP + q];          // sum_{i in window} sum_j x_ij
        const double dense = 0.5 * sx + 0.5 * a + 0.69314718055994531 * (l - pad_terms);
        const float rec = float((dense + e) * inv_n2);
        float total = rec;
        if (t.kl_partial != nullptr) {
            const float klf = float(extra * t.kl_scale);
            if (t.kl_out) *t.kl_out = klf;
            total = rec + klf;                  // (fp32, as the sum of the two fp32 scalars would be)
        }
        if (t.rec_out) *t.rec_out = rec;
        *t.loss_out = total;
        if (t.bump_draw) *t.bump_draw += 1;     // every read of the counter (prepare) is stream-ordered before this block
    }
}

} // namespace gae
