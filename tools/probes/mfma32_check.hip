// Check: v_mfma_f32_16x16x32_bf16 on concatenated K = 16 fragments [a1 | a2] x [b1 | b2] == a1 b1 + a2 b2
//   hipcc --offload-arch=gfx950 -O3 mfma32_check.hip -o bin/mfma32_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ void k(const s16x4 *a, const s16x4 *b, f32x4 *o)
{
    const int t = threadIdx.x;
    const s16x4 a1 = a[t], a2 = a[t + 64], b1 = b[t], b2 = b[t + 64];
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a1, b1, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a2, b2, c, 0, 0, 0);
    const s16x8 A = {a1[0], a1[1], a1[2], a1[3], a2[0], a2[1], a2[2], a2[3]};
    const s16x8 B = {b1[0], b1[1], b1[2], b1[3], b2[0], b2[1], b2[2], b2[3]};
    f32x4 d = {0, 0, 0, 0};
    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), d, 0, 0, 0);
    o[t] = c; o[t + 64] = d;
}
int main()
{
    short ha[128 * 4], hb[128 * 4];
    for (int i = 0; i < 512; ++i) {
        float x = (rand() % 2001 - 1000) / 500.0f, y = (rand() % 2001 - 1000) / 500.0f;
        unsigned ux, uy; memcpy(&ux, &x, 4); memcpy(&uy, &y, 4);
        ha[i] = short(ux >> 16); hb[i] = short(uy >> 16);
    }
    short *da, *db; float *dout; float ho[128 * 4];
    hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dout, sizeof ho);
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, (const s16x4 *)da, (const s16x4 *)db, (f32x4 *)dout);
    hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
    double md = 0, mx = 0;
    for (int i = 0; i < 256; ++i) { double e = fabs(double(ho[i]) - ho[256 + i]); if (e > md) md = e; if (fabs(ho[i]) > mx) mx = fabs(ho[i]); }
    printf("max |mfma16 chain - mfma32 concat| = %.3e (scale %.3e)\n", md, mx);
    return 0;
}
