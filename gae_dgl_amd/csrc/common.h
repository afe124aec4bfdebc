// Shared helpers for the libgae_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "gae_hip.h"

namespace gae {

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

} // namespace gae
