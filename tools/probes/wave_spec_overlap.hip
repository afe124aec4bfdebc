// Probe (round 6): do the matrix pipe and the VALU of ONE SIMD overlap when DIFFERENT waves feed them?
//
// tools/probes/mfma_valu_overlap.hip / inst_cost.hip let every wave issue both kinds in program order and found the times
// ADD (mfma + 4 v_fma: 12.8 ns against 7.5 + 6.0).  MI355X_MICROARCH.md says an MFMA-only wave and a VALU-only wave run
// concurrently.  Here a block has 8 or 16 waves (2 or 4 per SIMD: wave w sits on SIMD w % 4) and the same total work per
// SIMD is issued three ways:
//     mixed        every wave: [1 MFMA, V VALU] x iters, in program order (4 independent accumulator chains, 16
//                  independent VALU chains)
//     specialised  the first half of the waves (one or two per SIMD) issue ONLY the MFMAs of two mixed waves, the second
//                  half ONLY the VALU instructions of two mixed waves
//     mfma / valu  one kind alone, in every wave (the two floors)
// If the pipes overlap across waves, specialised ~ max(mfma, valu); if a SIMD issues one instruction at a time whatever
// its kind, specialised ~ mixed ~ mfma + valu.  Also with s_setprio 3 on the MFMA waves.
//   hipcc --offload-arch=gfx950 -O3 wave_spec_overlap.hip -o bin/wave_spec_overlap && bin/wave_spec_overlap
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

#define MFMA(k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, sa), __builtin_bit_cast(bf16x8, sb), acc[k], 0, 0, 0)
#define FMA4(b) asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5" \
                             : "+v"(r[b]), "+v"(r[b + 1]), "+v"(r[b + 2]), "+v"(r[b + 3]) : "v"(c), "v"(d))
#define EXP1(b) asm volatile("v_exp_f32 %0, %0" : "+v"(r[b]))

// MODE 0 mixed, 1 specialised, 2 mfma only, 3 valu only.  VPM: VALU instructions per MFMA (4 or 8).  TR: one of every four
// VALU instructions is a transcendental.  PRIO: s_setprio on the MFMA waves of the specialised form.
template <int MODE, int VPM, bool TR, int PRIO, int THREADS>
__global__ __launch_bounds__(THREADS) void k(float *out, int iters)
{
    f32x4 acc[4];
    float r[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = 1.0f + 0.001f * float((threadIdx.x + i) & 63);
    const s16x8 sa = {short(threadIdx.x), 2, 3, 4, 5, 6, 7, 8}, sb = {5, 6, short(threadIdx.x), 8, 1, 2, 3, 4};
    const float c = 0.9999f, d = 0.0001f;
    const int wave = threadIdx.x >> 6, half = THREADS / 128;          // waves per half
    const bool mfma_wave = MODE == 2 || (MODE == 1 && wave < half);
    const bool valu_wave = MODE == 3 || (MODE == 1 && wave >= half);
    if (MODE == 1 && mfma_wave && PRIO) __builtin_amdgcn_s_setprio(PRIO);
    if (MODE == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                MFMA(g);
                FMA4(4 * g);
                if (VPM == 8) { if (TR) { EXP1(4 * g); FMA4(4 * g); } else FMA4(4 * g); }
            }
        }
    } else if (mfma_wave) {
        // MODE 1: this wave issues the MFMAs of TWO mixed waves (its own and its VALU partner's); MODE 2: its own only
        const int reps = MODE == 1 ? 2 : 1;
        for (int it = 0; it < iters * reps; ++it) {
#pragma unroll
            for (int g = 0; g < 4; ++g) MFMA(g);
        }
    } else if (valu_wave) {
        const int reps = MODE == 1 ? 2 : 1;
        for (int it = 0; it < iters * reps; ++it) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                FMA4(4 * g);
                if (VPM == 8) { if (TR) { EXP1(4 * g); FMA4(4 * g); } else FMA4(4 * g); }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
}

template <int MODE, int VPM, bool TR, int PRIO, int THREADS>
float run(float *out, int iters, int blocks)
{
    hipLaunchKernelGGL((k<MODE, VPM, TR, PRIO, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, out, 16);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, VPM, TR, PRIO, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best * 1e3f;   // us
}

template <int VPM, bool TR, int THREADS>
void table(float *out, const char *what)
{
    const int iters = 20000, blocks = 256;               // one block per CU: THREADS / 256 waves per SIMD
    const float mixed = run<0, VPM, TR, 0, THREADS>(out, iters, blocks);
    const float spec = run<1, VPM, TR, 0, THREADS>(out, iters, blocks);
    const float spec3 = run<1, VPM, TR, 3, THREADS>(out, iters, blocks);
    const float m = run<2, VPM, TR, 0, THREADS>(out, iters, blocks);
    const float v = run<3, VPM, TR, 0, THREADS>(out, iters, blocks);
    // per SIMD and group (1 MFMA + VPM VALU): waves/SIMD x 4 groups x iters groups were issued on every SIMD
    const double groups = double(THREADS / 256) * 4.0 * iters;
    auto ns = [&](float us) { return us * 1e3 / groups; };
    printf("%-44s waves/SIMD %d | per group and SIMD: mfma-only %5.2f ns, valu-only %5.2f ns (sum %5.2f, max %5.2f) | mixed %5.2f | "
           "specialised %5.2f | specialised, MFMA waves at s_setprio 3 %5.2f\n",
           what, THREADS / 256, ns(m), ns(v), ns(m) + ns(v), ns(m) > ns(v) ? ns(m) : ns(v), ns(mixed), ns(spec), ns(spec3));
}

int main()
{
    float *out; hipMalloc(&out, 64);
    table<4, false, 512>(out, "1 mfma_16x16x32_bf16 + 4 v_fma_f32");
    table<4, false, 1024>(out, "1 mfma_16x16x32_bf16 + 4 v_fma_f32");
    table<8, false, 512>(out, "1 mfma_16x16x32_bf16 + 8 v_fma_f32");
    table<8, false, 1024>(out, "1 mfma_16x16x32_bf16 + 8 v_fma_f32");
    table<8, true, 512>(out, "1 mfma_16x16x32_bf16 + 1 v_exp + 8 v_fma");
    table<8, true, 1024>(out, "1 mfma_16x16x32_bf16 + 1 v_exp + 8 v_fma");
    return 0;
}
