// What ds_read_b64_tr_b16 (gfx950) returns: every lane reads 8 bytes at its own LDS address, the 16 lanes of a group
// exchange 16-bit elements.  Prints, for lanes 0..63, which (source lane, element) each of the 4 result elements is.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/ds_tr_probe.hip -o tools/probes/ds_tr_probe && tools/probes/ds_tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short *out)
{
    __shared__ __attribute__((aligned(16))) short buf[256];
    const int lane = threadIdx.x;
    for (int e = 0; e < 4; ++e) buf[lane * 4 + e] = short(lane * 4 + e);       // element id = 4 lane + e
    __syncthreads();
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(buf + lane * 4));
    *(s16x4 *)(out + lane * 4) = v;
}
int main()
{
    short *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int e = 0; e < 4; ++e) printf("  (L%2d,e%d)", h[l * 4 + e] / 4, h[l * 4 + e] % 4);
        printf("\n");
    }
    return 0;
}
