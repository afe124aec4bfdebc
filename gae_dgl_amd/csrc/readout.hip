// Graph-level readout: mean | sum | max of the node embeddings of every member graph of a batch.
//
// The reference describes it (README.md:54: "GAE feature is a concatenation of mean, sum, and max aggregation of the
// hidden vector H in R^{N x 16}, so its dimension is 48" -- the feature behind the ESOL table, README.md:44-51) but
// does not ship the code; with DGL it is dgl.mean_nodes / sum_nodes / max_nodes over the batched graph of
// train_inductive.py:34.  Segment reduction over graph_ptr: HBM-bound, 4 d bytes read per node, 12 d written per
// graph.  One wave per graph: 64 / DP rows in flight per step (DP = d rounded up to a power of two), a fixed
// butterfly across the row slots -> deterministic.  An empty graph gives zeros in all three blocks.
#include "common.h"

namespace {

template <int DP>
__global__ __launch_bounds__(256) void segment_readout_kernel(const float *__restrict__ Z, int64_t ldz, int d,
                                                              const int64_t *__restrict__ graph_ptr, int64_t n_graphs,
                                                              float *__restrict__ out, int64_t ldo)
{
    constexpr int RS = 64 / DP;                  // row slots of a wave
    const int lane = threadIdx.x & 63, c = lane % DP, slot = lane / DP;
    const int64_t g = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (g >= n_graphs) return;
    const int64_t r0 = graph_ptr[g], r1 = graph_ptr[g + 1];
    float s = 0.f, m = -INFINITY;
    if (c < d)
        for (int64_t r = r0 + slot; r < r1; r += RS) {
            const float v = Z[r * ldz + c];
            s += v;
            m = fmaxf(m, v);
        }
#pragma unroll
    for (int off = DP; off < 64; off <<= 1) {    // slot 0 ends up with (slot 0 + slot 1) + (slot 2 + slot 3) ...
        s += __shfl_xor(s, off, 64);
        m = fmaxf(m, __shfl_xor(m, off, 64));
    }
    if (slot == 0 && c < d) {
        const int64_t cnt = r1 - r0;
        float *o = out + g * ldo;
        o[c] = cnt > 0 ? s / float(cnt) : 0.f;   // mean
        o[d + c] = s;                            // sum
        o[2 * d + c] = cnt > 0 ? m : 0.f;        // max
    }
}

// d > 64: one block column of 64 features per blockIdx.y, rows walked by the whole wave
__global__ __launch_bounds__(256) void segment_readout_wide_kernel(const float *__restrict__ Z, int64_t ldz, int d,
                                                                   const int64_t *__restrict__ graph_ptr,
                                                                   int64_t n_graphs, float *__restrict__ out,
                                                                   int64_t ldo)
{
    const int c = blockIdx.y * 64 + (threadIdx.x & 63);
    const int64_t g = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (g >= n_graphs || c >= d) return;
    const int64_t r0 = graph_ptr[g], r1 = graph_ptr[g + 1];
    float s = 0.f, m = -INFINITY;
    for (int64_t r = r0; r < r1; ++r) {
        const float v = Z[r * ldz + c];
        s += v;
        m = fmaxf(m, v);
    }
    const int64_t cnt = r1 - r0;
    float *o = out + g * ldo;
    o[c] = cnt > 0 ? s / float(cnt) : 0.f;
    o[d + c] = s;
    o[2 * d + c] = cnt > 0 ? m : 0.f;
}

} // namespace

extern "C" int gae_segment_readout(const float *Z, int64_t ldz, int64_t n_nodes, int64_t d, const int64_t *graph_ptr,
                                   int64_t n_graphs, float *out, int64_t ldo, void *stream)
{
    GAE_REQUIRE(n_nodes >= 0 && d >= 0 && n_graphs >= 0, GAE_E_SIZE, "gae_segment_readout: negative size");
    GAE_REQUIRE(ldz >= d && ldo >= 3 * d, GAE_E_SIZE, "gae_segment_readout: leading dimension too small");
    GAE_REQUIRE(d < (int64_t(1) << 20), GAE_E_SIZE, "gae_segment_readout: d too large");
    if (n_graphs == 0 || d == 0) return GAE_OK;
    GAE_REQUIRE(graph_ptr && out && (Z || n_nodes == 0), GAE_E_NULL, "gae_segment_readout: NULL pointer");
    hipStream_t s = gae::as_stream(stream);
    const unsigned gx = unsigned((n_graphs + 3) / 4);
#define GAE_RO(DP) hipLaunchKernelGGL(segment_readout_kernel<DP>, dim3(gx), dim3(256), 0, s, Z, ldz, int(d), graph_ptr, n_graphs, out, ldo)
    if (d <= 1) GAE_RO(1);
    else if (d <= 2) GAE_RO(2);
    else if (d <= 4) GAE_RO(4);
    else if (d <= 8) GAE_RO(8);
    else if (d <= 16) GAE_RO(16);
    else if (d <= 32) GAE_RO(32);
    else if (d <= 64) GAE_RO(64);
    else
        hipLaunchKernelGGL(segment_readout_wide_kernel, dim3(gx, unsigned((d + 63) / 64)), dim3(256), 0, s, Z, ldz, int(d),
                           graph_ptr, n_graphs, out, ldo);
#undef GAE_RO
    GAE_CHECK_LAUNCH("segment_readout_kernel");
    return GAE_OK;
}
