// Reference-shaped loss on MATERIALISED logits: F.binary_cross_entropy_with_logits(adj_logits, adj,
// pos_weight=pos_weight) of gae_dgl/train_inductive.py:48 (mean reduction) and its autograd, for the cases the
// fused never-materialised kernel (decoder_bce.hip) does not take: embedding widths above 64 (the reference's
// optuna_gae.py:29-34 samples hidden dims up to 256).  Element-wise + an ordered two-stage fp64 reduction
// (deterministic); HBM-bound: 8 bytes read + 4 written per logit.
//     l = (1 - y) x + (1 + (pw - 1) y) softplus(-x),   dl/dx = (1 - y) - (1 + (pw - 1) y) sigmoid(-x)
#include "common.h"

namespace {

constexpr int kBlocks = 2048;

__device__ __forceinline__ void bce_elem(float x, float y, float pw, float scale, double &acc, float &g)
{
    const float lw = 1.f + (pw - 1.f) * y;
    const float ax = fabsf(x);
    const float e = __expf(-ax);                              // exp(-|x|) in (0, 1]
    const float sp = fmaxf(-x, 0.f) + log1pf(e);              // softplus(-x)
    const float sn = x >= 0.f ? e / (1.f + e) : 1.f / (1.f + e);   // sigmoid(-x)
    acc += double((1.f - y) * x + lw * sp);
    g = ((1.f - y) - lw * sn) * scale;
}

__global__ __launch_bounds__(256) void bce_logits_kernel(const float *__restrict__ X, int64_t ldx,
                                                         const float *__restrict__ Y, int64_t ldy, int64_t n_rows,
                                                         int64_t n_cols, float pw, float scale, float *G, int64_t ldg,
                                                         double *__restrict__ partial)
{
    __shared__ double red[4];
    double acc = 0.0;
    const int64_t total = n_rows * n_cols;
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < total; i += stride) {
        const int64_t r = i / n_cols, c = i - r * n_cols;
        float g;
        bce_elem(X[r * ldx + c], Y[r * ldy + c], pw, scale, acc, g);
        if (G) G[r * ldg + c] = g;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void bce_finalize_kernel(const double *__restrict__ partial, int n, double scale,
                                                           float *__restrict__ loss)
{
    __shared__ double red[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (int(threadIdx.x) < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = float(red[0] * scale);
}

} // namespace

extern "C" int64_t gae_bce_logits_workspace_bytes(void) { return kBlocks * int64_t(sizeof(double)); }

extern "C" int gae_bce_logits(const float *logits, int64_t ldx, const float *labels, int64_t ldy, int64_t n_rows,
                              int64_t n_cols, float pos_weight, float *loss_out, float *grad, int64_t ldg,
                              void *workspace, int64_t workspace_bytes, void *stream)
{
    GAE_REQUIRE(n_rows >= 0 && n_cols >= 0, GAE_E_SIZE, "gae_bce_logits: negative size");
    GAE_REQUIRE(ldx >= n_cols && ldy >= n_cols && (!grad || ldg >= n_cols), GAE_E_SIZE,
                "gae_bce_logits: leading dimension smaller than n_cols");
    GAE_REQUIRE(loss_out != nullptr, GAE_E_NULL, "gae_bce_logits: loss_out is NULL");
    GAE_REQUIRE(workspace && workspace_bytes >= gae_bce_logits_workspace_bytes(), GAE_E_WORKSPACE,
                "gae_bce_logits: workspace too small");
    hipStream_t s = gae::as_stream(stream);
    const int64_t total = n_rows * n_cols;
    if (total == 0) {
        GAE_HIP(hipMemsetAsync(loss_out, 0, sizeof(float), s));    // mean over an empty set: 0 (torch gives nan)
        return GAE_OK;
    }
    GAE_REQUIRE(logits && labels, GAE_E_NULL, "gae_bce_logits: NULL pointer");
    double *partial = static_cast<double *>(workspace);
    int blocks = int((total + 255) / 256 < kBlocks ? (total + 255) / 256 : kBlocks);
    const double inv = 1.0 / double(total);
    hipLaunchKernelGGL(bce_logits_kernel, dim3(blocks), dim3(256), 0, s, logits, ldx, labels, ldy, n_rows, n_cols,
                       pos_weight, float(inv), grad, ldg, partial);
    GAE_CHECK_LAUNCH("bce_logits_kernel");
    hipLaunchKernelGGL(bce_finalize_kernel, dim3(1), dim3(256), 0, s, partial, blocks, inv, loss_out);
    GAE_CHECK_LAUNCH("bce_finalize_kernel");
    return GAE_OK;
}
