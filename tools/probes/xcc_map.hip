// Probe: which XCD runs workgroup b?  (HW_REG_XCC_ID via s_getreg)  hipcc --offload-arch=gfx950 xcc_map.hip -o xcc_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int *out, int nx) {
    if (threadIdx.x == 0) {
        unsigned v;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
        out[blockIdx.y * nx + blockIdx.x] = int(v & 0xf);
    }
}
int main() {
    for (int cfg = 0; cfg < 4; ++cfg) {
        dim3 grid = cfg == 0 ? dim3(64) : cfg == 1 ? dim3(4936) : cfg == 2 ? dim3(617, 16) : dim3(100003);
        int n = grid.x * grid.y;
        int *d; hipMalloc(&d, n * 4); hipMemset(d, 0xff, n * 4);
        hipLaunchKernelGGL(k, grid, dim3(256), 0, 0, d, int(grid.x));
        std::vector<int> h(n); hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
        int match = 0; for (int b = 0; b < n; ++b) match += (h[b] == b % 8);
        printf("grid (%u,%u): %d blocks, xcc == linear_id %% 8 for %d (%.1f%%); first 24:", grid.x, grid.y, n, match, 100.0 * match / n);
        for (int b = 0; b < 24 && b < n; ++b) printf(" %d", h[b]);
        printf("\n");
        hipFree(d);
    }
    return 0;
}
