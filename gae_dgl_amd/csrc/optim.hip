// K12: Adam (torch.optim.Adam defaults: no amsgrad, no maximize, L2 weight decay) for the few small parameter
// tensors of the GAE (layers.{i}.apply_mod.linear.{weight,bias}; 1 808 .. 119 056 values), ONE launch per step.
//
// Replaces the optimiser step of gae_dgl/train_inductive.py:40,50-52 / train_transductive.py:43,66-68.  PyTorch's
// own fused Adam needs two multi-tensor launches per step (13 + 4 us on gfx950) for these 16 k values; inside a
// replayed HIP graph of ~25 launches that is 5 % of a Pubmed step and 10 % of a Cora step.
//
// The step counter lives in device memory (state[0]) so that a captured graph advances it on every replay: every
// block reads it when it starts, the LAST block to finish (ticket in state[1]) increments it and clears the ticket.
#include <string.h>

#include "common.h"
#include "bce_finalize.h"

namespace {

constexpr int kAdamMaxTensors = GAE_ADAM_MAX_TENSORS;
constexpr int kAdamChunk = 1024;      // elements per block: 256 threads x float4
}
namespace gae {
// "adam_chunk": elements per block of a tensor whose gradient is a short partial-sum list (<= 32 partials): 256 (one
// per thread), 1024 (four per thread, one after another) or 0 = auto: 256 from 8 partials on (the sum is a chain of
// dependent round trips: Pubmed, 32 partials, 0.232 -> 0.228 ms per step; Cora, 11: 0.083 -> 0.082), 1024 below
// (Citeseer, 4 partials of 118 k elements: 0.105 -> 0.103 with the fewer, longer blocks)
Knob g_adam_chunk{0};
Knob *optim_knob(const char *name) { return strcmp(name, "adam_chunk") == 0 ? &g_adam_chunk : nullptr; }
}
namespace {

struct AdamArgs {
    gae_adam_tensor t[kAdamMaxTensors];
    int32_t chunk[kAdamMaxTensors];             // elements per block of tensor k
    int32_t first_block[kAdamMaxTensors + 1];   // block range of tensor k: [first_block[k], first_block[k + 1])
    int32_t n_tensors;
    int32_t n_blocks;                           // optimiser blocks of the launch (a loss tail may follow them)
};

template <bool TAIL>
__global__ __launch_bounds__(256) void adam_step_kernel(AdamArgs a, float lr, float beta1, float beta2, float eps,
                                                        float weight_decay, unsigned long long *__restrict__ state,
                                                        const gae_bce_tail tail)
{
    if constexpr (TAIL) {
        // one block behind the optimiser's: the deferred final reduction of the loss (gae_x_adam_step_tail).  It takes
        // no ticket: the step counter waits for the a.n_blocks optimiser blocks only.
        if (blockIdx.x == unsigned(a.n_blocks)) {
            __shared__ double red[3][16];
            gae::bce_finalize_block<256>(tail, red);
            return;
        }
    }
    int k = 0;
    while (k + 1 < a.n_tensors && int(blockIdx.x) >= a.first_block[k + 1]) ++k;
    const gae_adam_tensor t = a.t[k];
    // beta^t is carried in the state (doubles at [2..5]: beta1, beta1^steps, beta2, beta2^steps) and advanced by one
    // multiplication per step: two double-precision pow() per thread were a third of this kernel's 6 us.  A state
    // that does not hold this call's betas (fresh / resumed counter, changed hyper-parameters) takes the pow path.
    const double steps_done = double(state[0]);
    const double *sd = reinterpret_cast<const double *>(state);
    const bool cached = sd[2] == double(beta1) && sd[4] == double(beta2);
    const double b1t = (cached ? sd[3] : pow(double(beta1), steps_done)) * double(beta1);    // beta1^step, 1-based
    const double b2t = (cached ? sd[5] : pow(double(beta2), steps_done)) * double(beta2);
    const float bc1 = float(1.0 - b1t);
    const float bc2_sqrt = float(sqrt(1.0 - b2t));
    const float step_size = lr / bc1;
    // (p, m, v are passed in: the deferred branches request them BEFORE they add up the partial sums -- the two round
    //  trips overlap instead of following each other)
    auto update = [&](int64_t e, float g, float p, float m, float v) {
        if (weight_decay != 0.f) g = fmaf(weight_decay, p, g);
        m = fmaf(g - m, 1.0f - beta1, m);                          // lerp(m, g, 1 - beta1)
        v = fmaf(beta2, v, (1.0f - beta2) * g * g);
        const float denom = sqrtf(v) / bc2_sqrt + eps;
        t.param[e] = p - step_size * (m / denom);
        t.exp_avg[e] = m;
        t.exp_avg_sq[e] = v;
    };
    const int lb = int(blockIdx.x) - a.first_block[k];
    if (t.n_partials == 0) {
        const int64_t base = int64_t(lb) * kAdamChunk + threadIdx.x * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t e = base + q;
            if (e < t.n) update(e, t.grad[e], t.param[e], t.exp_avg[e], t.exp_avg_sq[e]);
        }
    } else {
        // ---- deferred reduction: the gradient is still a list of partial sums (what gae_x_xw_wgrad_partials /
        //      gae_x_linear_bwd_partials left in their workspace), added in the library's one order for such lists
        //      (gae::sum_partials: the stand-alone reduction launches give the same bits); the sum is written to
        //      `grad` and used at once.
        const int L = gae::partial_lanes(t.n_partials);
        const unsigned row_len = unsigned(t.row_len), n32 = unsigned(t.n);          // (sizes checked on the host: < 2^31)
        if (L == 1) {
            // ONE element per thread (adjacent lanes read adjacent elements of every partial): the sum is a chain of
            // ceil(n_partials / 16) dependent round trips per element, so elements go side by side, not one after
            // another in a thread (4 per thread: 12.2 us for Pubmed's 16 k weights, 8 round trips)
            const unsigned chunk = unsigned(a.chunk[k]);
            for (unsigned q = 0; q < chunk / 256u; ++q) {
                const unsigned e = unsigned(lb) * chunk + q * 256u + threadIdx.x;
                const unsigned ec = e < n32 ? e : 0u;
                const unsigned r = ec / row_len, c = ec - r * row_len;
                const float p0 = t.param[ec], m0 = t.exp_avg[ec], v0 = t.exp_avg_sq[ec];
                const float g = gae::sum_partials(t.partials + int64_t(r) * t.row_pitch + c, t.n_partials,
                                                  t.partial_stride, 0, 1);
                if (e < n32) {
                    t.grad[e] = g;
                    update(e, g, p0, m0, v0);
                }
            }
        } else {
            const unsigned e = unsigned(lb) * 4u + threadIdx.x / 64u;
            const int lane = threadIdx.x % 64;
            const unsigned ec = e < n32 ? e : 0u;
            const unsigned r = ec / row_len, c = ec - r * row_len;
            const float p0 = t.param[ec], m0 = t.exp_avg[ec], v0 = t.exp_avg_sq[ec];
            const float g = gae::sum_partials(t.partials + int64_t(r) * t.row_pitch + c, t.n_partials, t.partial_stride,
                                              lane, 64);
            if (e < n32 && lane == 0) {
                t.grad[e] = g;
                update(e, g, p0, m0, v0);
            }
        }
    }
    // ---- the last block to finish advances the step counter
    // (no fence: every thread of this block has CONSUMED its reads of the state above -- they are complete -- before
    //  the barrier, and the last block writes the state only after every other block's ticket, i.e. after their reads)
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long ticket = atomicAdd(&state[1], 1ull);
        if (ticket == (unsigned long long)(a.n_blocks) - 1ull) {      // every block has read the state by now (its ticket came after its loads)
            double *sw = reinterpret_cast<double *>(state);
            sw[2] = double(beta1); sw[3] = b1t; sw[4] = double(beta2); sw[5] = b2t;
            state[1] = 0ull;
            state[0] += 1ull;
        }
    }
}

} // namespace

extern "C" int gae_x_adam_step_tail(const gae_adam_tensor *tensors, int32_t n_tensors, float lr, float beta1, float beta2,
                                  float eps, float weight_decay, uint64_t *state_dev, const gae_bce_tail *tail,
                                  void *stream)
{
    const bool with_tail = tail != nullptr && tail->loss_out != nullptr;
    GAE_REQUIRE(!with_tail || (tail->S && (tail->n_dense == 0 || tail->dense_partial) &&
                               (tail->n_edge == 0 || tail->edge_partial) && tail->DP > 0),
                GAE_E_NULL, "gae_x_adam_step_tail: malformed tail (not one written by gae_decoder_bce*)");
    GAE_REQUIRE(n_tensors >= 0 && n_tensors <= kAdamMaxTensors, GAE_E_RANGE,
                "gae_adam_step: %d tensors per call (at most %d)", n_tensors, kAdamMaxTensors);
    GAE_REQUIRE(lr >= 0.f && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f &&
                    weight_decay >= 0.f,
                GAE_E_RANGE, "gae_adam_step: hyper-parameter out of range");
    if (n_tensors == 0) return with_tail ? gae_x_decoder_bce_finalize(tail, stream) : GAE_OK;
    GAE_REQUIRE(tensors && state_dev, GAE_E_NULL, "gae_adam_step: NULL pointer");
    AdamArgs a;
    memset(&a, 0, sizeof(a));
    a.n_tensors = n_tensors;
    int64_t blocks = 0;
    for (int k = 0; k < n_tensors; ++k) {
        const gae_adam_tensor &t = tensors[k];
        GAE_REQUIRE(t.n >= 0, GAE_E_SIZE, "gae_adam_step: tensor %d has a negative size", k);
        GAE_REQUIRE(t.n == 0 || (t.param && t.grad && t.exp_avg && t.exp_avg_sq), GAE_E_NULL,
                    "gae_adam_step: tensor %d has a NULL pointer", k);
        GAE_REQUIRE(t.n_partials >= 0 && (t.n_partials == 0 || (t.partials && t.row_len > 0 && t.partial_stride >= 0)),
                    GAE_E_SIZE, "gae_adam_step: tensor %d has a malformed partial-sum list", k);
        a.t[k] = t;
        a.first_block[k] = int32_t(blocks);
        GAE_REQUIRE(t.n < (int64_t(1) << 31) && t.row_len < (int64_t(1) << 31), GAE_E_SIZE,
                    "gae_adam_step: tensor %d too large", k);
        const int knob = int(gae::g_adam_chunk);
        const int64_t per_block = t.n_partials > 32 ? 256 / 64
                                  : t.n_partials > 0 ? ((knob == 256 || knob == 1024) ? knob : (t.n_partials >= 8 ? 256 : 1024))
                                                     : kAdamChunk;
        a.chunk[k] = int32_t(per_block);
        blocks += (t.n + per_block - 1) / per_block;
        GAE_REQUIRE(blocks < (int64_t(1) << 30), GAE_E_SIZE, "gae_adam_step: too many elements for one launch");
    }
    a.first_block[n_tensors] = int32_t(blocks);
    if (blocks == 0) blocks = 1;        // still advances the step counter (a.t[0].n == 0: no element passes e < n)
    a.n_blocks = int32_t(blocks);
    if (with_tail)
        hipLaunchKernelGGL(adam_step_kernel<true>, dim3(unsigned(blocks) + 1), dim3(256), 0, gae::as_stream(stream), a, lr,
                           beta1, beta2, eps, weight_decay, reinterpret_cast<unsigned long long *>(state_dev), *tail);
    else
        hipLaunchKernelGGL(adam_step_kernel<false>, dim3(unsigned(blocks)), dim3(256), 0, gae::as_stream(stream), a, lr,
                           beta1, beta2, eps, weight_decay, reinterpret_cast<unsigned long long *>(state_dev), gae_bce_tail{});
    GAE_CHECK_LAUNCH("adam_step_kernel");
    return GAE_OK;
}

extern "C" int gae_adam_step(const gae_adam_tensor *tensors, int32_t n_tensors, float lr, float beta1, float beta2,
                             float eps, float weight_decay, uint64_t *state_dev, void *stream)
{
    return gae_x_adam_step_tail(tensors, n_tensors, lr, beta1, beta2, eps, weight_decay, state_dev, nullptr, stream);
}
