// Probe: sustained VALU rates on gfx950 for the op mix of the fused decoder + BCE kernel: plain fp32 (v_fma_f32,
// v_add_f32, v_mul_f32), packed fp32, and the transcendentals it uses (v_exp_f32, v_log_f32, v_rcp_f32).
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o bin/valu_rate
// 8 independent chains per lane, 8 waves per SIMD: issue-bound, no memory traffic.  Prints lane-ops per second and
// the ratio to the v_fma_f32 rate (= the weights of the loss kernel's VALU roofline in bench.py).
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = seed + 0.001f * float(threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) x[i] = __builtin_fmaf(x[i], 0.999f, 0.001f);
            else if (OP == 1) x[i] = __builtin_amdgcn_exp2f(x[i] * 0.5f) ;            // v_exp_f32 (+1 mul)
            else if (OP == 2) x[i] = __builtin_amdgcn_logf(x[i] + 2.0f);              // v_log_f32 (+1 add)
            else if (OP == 3) x[i] = __builtin_amdgcn_rcpf(x[i] + 1.5f);              // v_rcp_f32 (+1 add)
            else if (OP == 4) x[i] = x[i] * 0.999f;                                   // v_mul_f32
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 12345.678f) out[0] = s;
}

template <int OP>
double run(float *out, int extra_ops)
{
    const int iters = 4096, blocks = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5f);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double inst = double(blocks) * 256 * iters * 8;      // lane-level op groups
    (void)extra_ops;
    return inst / (best * 1e-3);
}

int main()
{
    float *out; hipMalloc(&out, 64);
    const double fma = run<0>(out, 0), mul = run<4>(out, 0);
    const double ex = run<1>(out, 1), lg = run<2>(out, 1), rc = run<3>(out, 1);
    printf("v_fma_f32            %.3e lane-ops/s\n", fma);
    printf("v_mul_f32            %.3e lane-ops/s  (x%.2f)\n", mul, fma / mul);
    // each transcendental iteration also carries one plain op: t_pair = t_plain + t_trans
    printf("v_exp_f32 (+1 mul)   %.3e pairs/s -> exp alone costs %.2f plain ops\n", ex, fma / ex - 1.0);
    printf("v_log_f32 (+1 add)   %.3e pairs/s -> log alone costs %.2f plain ops\n", lg, fma / lg - 1.0);
    printf("v_rcp_f32 (+1 add)   %.3e pairs/s -> rcp alone costs %.2f plain ops\n", rc, fma / rc - 1.0);
    return 0;
}
