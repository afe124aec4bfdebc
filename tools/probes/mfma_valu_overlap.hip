// Probe: do MFMA and VALU work overlap on one SIMD?  f32-input MFMA vs bf16 MFMA.
// hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o probe && ./probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // bit0: f32 MFMA, bit1: VALU fma chain, bit2: bf16 MFMA, bit3: transcendentals
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    float v0 = a, v1 = a + 1, v2 = a + 2, v3 = a + 3, v4 = a + 4, v5 = a + 5, v6 = a + 6, v7 = a + 7;
    s16x4 sa = {1, 2, 3, 4}, sb = {5, 6, 7, 8};
    for (int i = 0; i < iters; ++i) {
        if (MODE & 1) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc1, 0, 0, 0);
        }
        if (MODE & 4) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(sa, sb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(sb, sa, acc1, 0, 0, 0);
        }
        if (MODE & 2) {   // 16 independent-ish VALU fmas
            v0 = fmaf(v0, b, a); v1 = fmaf(v1, b, a); v2 = fmaf(v2, b, a); v3 = fmaf(v3, b, a);
            v4 = fmaf(v4, b, a); v5 = fmaf(v5, b, a); v6 = fmaf(v6, b, a); v7 = fmaf(v7, b, a);
            v0 = fmaf(v0, b, a); v1 = fmaf(v1, b, a); v2 = fmaf(v2, b, a); v3 = fmaf(v3, b, a);
            v4 = fmaf(v4, b, a); v5 = fmaf(v5, b, a); v6 = fmaf(v6, b, a); v7 = fmaf(v7, b, a);
        }
        if (MODE & 8) {   // 4 transcendentals
            v0 = __builtin_amdgcn_exp2f(v0); v1 = __builtin_amdgcn_logf(v1); v2 = __builtin_amdgcn_rcpf(v2);
            v3 = __builtin_amdgcn_exp2f(v3);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc0[0] + acc1[1] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}

template <int MODE>
float run(float *d, int iters, int blocks)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
}
int main()
{
    float *d; hipMalloc(&d, 256 * 4096 * 4);
    const int iters = 20000;
    for (int blocks : {256 * 2, 256 * 4}) {   // 2 / 4 waves per SIMD
        printf("blocks/CU=%d (waves/SIMD=%d), %d iters: per-iteration ns per wave-slot\n", blocks / 256, blocks / 256, iters);
        float t1 = run<1>(d, iters, blocks), t2 = run<2>(d, iters, blocks), t3 = run<3>(d, iters, blocks);
        float t4 = run<4>(d, iters, blocks), t6 = run<6>(d, iters, blocks), t8 = run<8>(d, iters, blocks);
        float t9 = run<9>(d, iters, blocks), t12 = run<12>(d, iters, blocks), t10 = run<10>(d, iters, blocks);
        printf("  f32 MFMA x2 only      %8.1f us\n  VALU fma x16 only     %8.1f us\n  f32 MFMA + VALU       %8.1f us (sum %.1f, max %.1f)\n", t1, t2, t3, t1 + t2, t1 > t2 ? t1 : t2);
        printf("  bf16 MFMA x2 only     %8.1f us\n  bf16 MFMA + VALU      %8.1f us (sum %.1f, max %.1f)\n", t4, t6, t4 + t2, t4 > t2 ? t4 : t2);
        printf("  trans x4 only         %8.1f us\n  f32 MFMA + trans      %8.1f us (sum %.1f)\n  bf16 MFMA + trans     %8.1f us (sum %.1f)\n  VALU + trans          %8.1f us (sum %.1f)\n",
               t8, t9, t1 + t8, t12, t4 + t8, t10, t2 + t8);
    }
    return 0;
}
