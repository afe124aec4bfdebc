// Shared helpers for the libgae_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>

#include "gae_hip.h"
#include "gae_hip_experimental.h"

namespace gae {

// Tuning knob (gae_tuning_set / gae_tuning_get): ONE value per process, read with relaxed atomics at launch time.
// Process-wide on purpose: PyTorch runs the backward of an autograd Function on its engine's worker thread, so a
// per-thread value set from Python would not reach the launches of backward() (the dW and A^T products).
struct Knob {
    std::atomic<int> v;
    explicit constexpr Knob(int x) : v(x) {}
    operator int() const { return v.load(std::memory_order_relaxed); }
    Knob &operator=(int x) { v.store(x, std::memory_order_relaxed); return *this; }
};

// thread-local last-error message (defined in api.hip)
void set_error(const char *fmt, ...);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

#define GAE_REQUIRE(cond, code, ...)      \
    do {                                  \
        if (!(cond)) {                    \
            gae::set_error(__VA_ARGS__);  \
            return (code);                \
        }                                 \
    } while (0)

// after a launch: report launch-configuration errors as positive hipError_t
#define GAE_CHECK_LAUNCH(what)                                                       \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess) {                                                     \
            gae::set_error("%s: %s", (what), hipGetErrorString(e__));                \
            return static_cast<int>(e__);                                            \
        }                                                                            \
    } while (0)

#define GAE_HIP(call)                                                                \
    do {                                                                             \
        hipError_t e__ = (call);                                                     \
        if (e__ != hipSuccess) {                                                     \
            gae::set_error("%s: %s", #call, hipGetErrorString(e__));                 \
            return static_cast<int>(e__);                                            \
        }                                                                            \
    } while (0)

constexpr int kWave = 64;  // gfx950 wavefront
constexpr int kNumXcd = 8; // MI355X: 8 XCDs, private L2 each; block b runs on XCD b % 8

// Bijective XCD-aware remap: consecutive logical ids land on the same XCD so a
// contiguous row range shares one L2 (cdna guide T1, bijective form).
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk)
{
    const unsigned q = nblk / kNumXcd, r = nblk % kNumXcd;
    const unsigned xcd = bid % kNumXcd, k = bid / kNumXcd;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

__device__ __forceinline__ float bf16_to_f32(unsigned short v)
{
    return __uint_as_float(static_cast<unsigned>(v) << 16);
}
// round-to-nearest-even, NaN preserved
__device__ __forceinline__ unsigned short f32_to_bf16(float f)
{
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return static_cast<unsigned short>((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return static_cast<unsigned short>(u >> 16);
}

// ---------------------------------------------------------------------------
// bf16 x 3 products on the matrix cores: v = hi + lo with hi = bf16(v), lo = bf16(v - hi); hi.hi + hi.lo + lo.hi in
// fp32 accumulators drops only the lo.lo term (2^-16 relative).  fp32 MFMA shares the fp32 FMA lanes with the
// VALU on gfx950 (tools/probes/mfma_valu_overlap.hip), bf16 MFMA runs on the matrix pipe proper.
// ---------------------------------------------------------------------------
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef short v4s __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef __bf16 v2bf __attribute__((ext_vector_type(2)));

// split 4 fp32 values into bf16 hi and bf16 lo = bf16(v - hi): v_cvt_pk_bf16_f32 x 4, ~5 VALU ops per pair
__device__ __forceinline__ void split_bf16x4(const v4f &v, v4s &hi, v4s &lo)
{
    unsigned h[2], l[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const v2f a = {v[2 * q], v[2 * q + 1]};
        const unsigned hu = __builtin_bit_cast(unsigned, __builtin_convertvector(a, v2bf));
        const v2f back = {__uint_as_float(hu << 16), __uint_as_float(hu & 0xffff0000u)};
        h[q] = hu;
        l[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(a - back, v2bf));
    }
    struct U { unsigned a, b; } uh{h[0], h[1]}, ul{l[0], l[1]};
    hi = __builtin_bit_cast(v4s, uh);
    lo = __builtin_bit_cast(v4s, ul);
}

// Three bf16 pieces of 4 fp32 values: v = hi + mid + lo EXACTLY (hi = bf16(v), mid = bf16(v - hi), lo = bf16(v - hi - mid):
// 8 + 8 + 8 mantissa bits).  A product of two such operands over the six piece pairs down to 2^-24 relative
// (hi.hi, hi.mid, mid.hi, mid.mid, hi.lo, lo.hi) is an fp32-grade product on the bf16 matrix pipe: every pair is exact in
// the fp32 accumulator, the dropped pairs are below one ulp of the term.
// NON-FINITE / OVERFLOW (documented behaviour, include/gae_hip.h "numerical contract"): for v = +-Inf, and for finite
// |v| > 3.3895e38 (the largest bf16; rounding to bf16 overflows to Inf -- the top 0.4 % of the fp32 exponent range), the
// residual v - hi is Inf - Inf = NaN, so a product that an fp32 MFMA would return as +-Inf (or as a huge finite number)
// comes back as NaN.  NaN inputs stay NaN.  Either way the result is non-finite and torch's anomaly checks fire; a
// select per element that would keep Inf as Inf costs two VALU ops per value in loops that are issue-bound.
__device__ __forceinline__ void split_bf16x4_3(const v4f &v, v4s &hi, v4s &mid, v4s &lo)
{
    unsigned h[2], m[2], l[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const v2f a = {v[2 * q], v[2 * q + 1]};
        const unsigned hu = __builtin_bit_cast(unsigned, __builtin_convertvector(a, v2bf));
        const v2f r1 = a - v2f{__uint_as_float(hu << 16), __uint_as_float(hu & 0xffff0000u)};
        const unsigned mu = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, v2bf));
        const v2f r2 = r1 - v2f{__uint_as_float(mu << 16), __uint_as_float(mu & 0xffff0000u)};
        h[q] = hu; m[q] = mu;
        l[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, v2bf));
    }
    struct U { unsigned a, b; } uh{h[0], h[1]}, um{m[0], m[1]}, ul{l[0], l[1]};
    hi = __builtin_bit_cast(v4s, uh);
    mid = __builtin_bit_cast(v4s, um);
    lo = __builtin_bit_cast(v4s, ul);
}

// ---------------------------------------------------------------------------
// THE order in which a list of partial sums (per-block pieces of a weight gradient) is added, wherever it is added --
// the stand-alone reduction launches of gae_xw_wgrad / gae_linear_bwd and gae_adam_step's deferred reduction -- so
// that a training step gives the same bits whichever of them runs:
//   lists of <= 32 partials: one lane per element (L = 1), partials 0, 1, 2, ... in order (16 loads in flight);
//   longer lists: 64 lanes per element (L = 64), lane l adds partials l, l + 64, ... in order, then the lanes meet
//   in the fixed shuffle-down tree 32, 16, ..., 1 (lane 0 holds the sum; all 64 lanes must call).
// ---------------------------------------------------------------------------
__device__ __forceinline__ int partial_lanes(int64_t n_partials) { return n_partials > 32 ? 64 : 1; }

__device__ __forceinline__ float sum_partials(const float *__restrict__ p0, int64_t n_partials, int64_t stride, int lane,
                                              int L)
{
    float g = 0.f;
    for (int64_t q0 = lane; q0 < n_partials; q0 += int64_t(16) * L) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int64_t q = q0 + int64_t(u) * L;
            v[u] = p0[(q < n_partials ? q : n_partials - 1) * stride];        // clamped: loads stay branch-free
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) g += q0 + int64_t(u) * L < n_partials ? v[u] : 0.f;
    }
    if (L == 64) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) g += __shfl_down(g, off, 64);
    }
    return g;
}

// one list of partial sums and where its sums go: element e of `n` reads partial q at
// base[q * stride + (e / row_len) * row_pitch + e % row_len] and is written to out[(e / row_len) * out_pitch + e % row_len]
struct PartialList {
    const float *base;
    float *out;
    int64_t n, n_partials, stride, row_len, row_pitch, out_pitch;
};
// reduction launch over up to two lists (a weight gradient and its bias gradient), defined in xw.hip
int launch_partials_reduce(const PartialList &a, const PartialList &b, hipStream_t s);

// ---------------------------------------------------------------------------
// Philox4x32-10 counter RNG (dropout masks, VGAE noise): 4 x 32 random bits per (counter, seed)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1)
{
    const uint64_t p0 = uint64_t(0xD2511F53u) * c[0];
    const uint64_t p1 = uint64_t(0xCD9E8D57u) * c[2];
    const uint32_t n0 = uint32_t(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n2 = uint32_t(p0 >> 32) ^ c[3] ^ k1;
    c[1] = uint32_t(p1); c[3] = uint32_t(p0); c[0] = n0; c[2] = n2;
}

// `draw` selects one of 2^64 disjoint streams of the same seed through the high counter words: consecutive draws
// over tensors of different sizes (inductive batches) can never reuse a 128-bit block
__device__ __forceinline__ void philox4x32_10(uint64_t ctr, uint64_t draw, uint64_t seed, uint32_t (&c)[4])
{
    c[0] = uint32_t(ctr); c[1] = uint32_t(ctr >> 32); c[2] = uint32_t(draw); c[3] = uint32_t(draw >> 32);
    uint32_t k0 = uint32_t(seed), k1 = uint32_t(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// inverted-dropout multiplier of one 32-bit draw: 0 with probability p, else scale = 1 / (1 - p)
__device__ __forceinline__ float dropout_multiplier(uint32_t bits, float p, float scale)
{
    const float u = float(bits >> 8) * (1.0f / 16777216.0f);  // [0, 1)
    return u >= p ? scale : 0.f;
}

} // namespace gae
