// Probe: issue cost (cycles per wave64 instruction on one SIMD) of the instruction kinds in the fused decoder + BCE
// kernel, measured issue-bound: 8 waves per SIMD, 16 independent destination registers per lane, no memory.
//   hipcc --offload-arch=gfx950 -O3 inst_cost.hip -o bin/inst_cost
// cycles = (wave-cycles the SIMD spent) / instructions issued on it, with the shader clock read through
// s_memtime around the loop (clock64()).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, long long *cyc, int iters)
{
    float r[16];
    f32x2 q[16];
    f32x4 acc[4];
    unsigned u[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { r[i] = 1.0f + 0.01f * float(threadIdx.x + i); q[i] = f32x2{r[i], r[i] + 1.f}; u[i] = threadIdx.x * 7u + i; }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const s16x4 sa = {short(threadIdx.x), 2, 3, 4}, sb = {5, 6, short(threadIdx.x), 8};
    const s16x8 sa8 = {short(threadIdx.x), 2, 3, 4, 5, 6, 7, 8}, sb8 = {5, 6, short(threadIdx.x), 8, 1, 2, 3, 4};
    const float c = 0.999f, d = 0.001f;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#define X_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c), "v"(d));
#define X_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(d));
#define X_PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 15]));
#define X_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 15]));
#define X_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 15]));
#define X_EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
#define X_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
#define X_LOG(i) asm volatile("v_log_f32 %0, %0" : "+v"(r[i]));
#define X_CVTPK(i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(r[i]), "v"(r[(i + 1) & 15]));
#define X_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 15]), "v"(0x07060302u));
#define X_BFI(i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(u[i]) : "v"(0x7fffffffu), "v"(u[(i + 1) & 15]));
#define X_AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(0xffff0000u));
#define X_LSHL(i) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(u[i]));
#define X_ABSADD(i) asm volatile("v_add_f32 %0, %0, |%1|" : "+v"(r[i]) : "v"(r[(i + 1) & 15]));
#define X_MFMA(i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(sa, sb, acc[i & 3], 0, 0, 0);
#define X_MIX(i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(sa, sb, acc[i & 3], 0, 0, 0); \
                 asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %3, %3, %1, %2\n\tv_fma_f32 %4, %4, %1, %2\n\tv_fma_f32 %5, %5, %1, %2" \
                              : "+v"(r[i]), "+v"(r[(i + 4) & 15]), "+v"(r[(i + 8) & 15]), "+v"(r[(i + 12) & 15]) : "v"(c), "v"(d) : );
#define X_MIXT(i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(sa, sb, acc[i & 3], 0, 0, 0); \
                 asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %4, %4, %2, %3" \
                              : "+v"(r[i]), "+v"(r[(i + 4) & 15]) : "v"(c), "v"(d), "v"(r[(i + 8) & 15]) : );
#define X_TV(i) asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %4, %4, %2, %3\n\tv_fma_f32 %5, %5, %2, %3" \
                              : "+v"(r[i]), "+v"(r[(i + 4) & 15]) : "v"(c), "v"(d), "v"(r[(i + 8) & 15]), "v"(r[(i + 12) & 15]) : );
#define X_MFMA32(i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, sa8), __builtin_bit_cast(bf16x8, sb8), acc[i & 3], 0, 0, 0);
#define X_MIX32(i) X_MFMA32(i) \
                 asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %3, %3, %1, %2\n\tv_fma_f32 %4, %4, %1, %2\n\tv_fma_f32 %5, %5, %1, %2" \
                              : "+v"(r[i]), "+v"(r[(i + 4) & 15]), "+v"(r[(i + 8) & 15]), "+v"(r[(i + 12) & 15]) : "v"(c), "v"(d) : );
        if (OP == 0) { REP16(X_FMA) }
        else if (OP == 1) { REP16(X_ADD) }
        else if (OP == 2) { REP16(X_PKADD) }
        else if (OP == 3) { REP16(X_PKMUL) }
        else if (OP == 4) { REP16(X_PKFMA) }
        else if (OP == 5) { REP16(X_EXP) }
        else if (OP == 6) { REP16(X_RCP) }
        else if (OP == 7) { REP16(X_LOG) }
        else if (OP == 8) { REP16(X_CVTPK) }
        else if (OP == 9) { REP16(X_PERM) }
        else if (OP == 10) { REP16(X_BFI) }
        else if (OP == 11) { REP16(X_AND) }
        else if (OP == 12) { REP16(X_LSHL) }
        else if (OP == 13) { REP16(X_ABSADD) }
        else if (OP == 14) { REP16(X_MFMA) }
        else if (OP == 15) { REP16(X_MIX) }      // 1 MFMA + 4 v_fma per group
        else if (OP == 16) { REP16(X_MIXT) }     // 1 MFMA + 1 v_exp + 2 v_fma per group
        else if (OP == 17) { REP16(X_TV) }       // 1 v_exp + 3 v_fma per group
        else if (OP == 18) { REP16(X_MFMA32) }
        else if (OP == 19) { REP16(X_MIX32) }
    }
    const long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i] + q[i][0] + q[i][1] + __uint_as_float(u[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0];
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int OP>
void run(const char *name, int per_group, float *out, long long *cyc)
{
    const int iters = 2000;
    for (int wps : {1, 2, 4}) {       // waves per SIMD (the kernel uses ~90 VGPRs: 4 waves per SIMD are co-resident)
        const int blocks = 256 * wps;  // 256 CUs x 4 SIMDs, one wave of each block per SIMD
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double groups = double(iters) * 16;                  // per wave
        // s_memtime ticks at 100 MHz on this part?  report both the tick count and the event time
        printf("%-34s waves/SIMD %d: %8.2f ns per group per wave-slot  (%.1f ticks/group)  -> %6.2f ns per SIMD per group of %d\n",
               name, wps, ms * 1e6 / groups, double(c) / groups, ms * 1e6 / groups / wps, per_group);
    }
}

int main()
{
    float *out; long long *cyc;
    hipMalloc(&out, 64); hipMalloc(&cyc, 64);
    run<0>("v_fma_f32", 1, out, cyc);
    run<1>("v_add_f32", 1, out, cyc);
    run<2>("v_pk_add_f32", 1, out, cyc);
    run<3>("v_pk_mul_f32", 1, out, cyc);
    run<4>("v_pk_fma_f32", 1, out, cyc);
    run<5>("v_exp_f32", 1, out, cyc);
    run<6>("v_rcp_f32", 1, out, cyc);
    run<7>("v_log_f32", 1, out, cyc);
    run<8>("v_cvt_pk_bf16_f32", 1, out, cyc);
    run<9>("v_perm_b32", 1, out, cyc);
    run<10>("v_bfi_b32", 1, out, cyc);
    run<11>("v_and_b32", 1, out, cyc);
    run<12>("v_lshlrev_b32", 1, out, cyc);
    run<13>("v_add_f32 |abs|", 1, out, cyc);
    run<14>("v_mfma_f32_16x16x16_bf16", 1, out, cyc);
    run<15>("mfma + 4 v_fma", 5, out, cyc);
    run<16>("mfma + v_exp + 2 v_fma", 4, out, cyc);
    run<17>("v_exp + 3 v_fma", 4, out, cyc);
    run<18>("v_mfma_f32_16x16x32_bf16", 1, out, cyc);
    run<19>("mfma x32 + 4 v_fma", 5, out, cyc);
    return 0;
}
