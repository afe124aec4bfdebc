// Probe: how fast can a CU gather random row PIECES that are resident in its XCD's L2?
//   hipcc --offload-arch=gfx950 -O3 gather_l2.hip -o bin/gather_l2
// A table of R rows x STRIDE bytes; every group of LPR lanes reads a piece of LPR*16 bytes of a pseudo-random row at
// the column offset of its XCD (blockIdx % 8) -- the access pattern of the XCD-tiled SpMM gather without any
// index loads, adds kept to a minimum, no stores.  Sweeps piece width, loads in flight per wave, waves per CU,
// slice size (L2-resident or not).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int LPR, int NL>
__global__ __launch_bounds__(256) void gather(const void *tab, unsigned bytes, unsigned rows, unsigned stride,
                                              unsigned piece_off_per_xcd, int iters, unsigned *sink)
{
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(tab), 0, int(bytes), 0x00020000);
    const unsigned lane = threadIdx.x & 63, lig = lane % LPR;
    unsigned seed = (blockIdx.x * 256u + threadIdx.x / LPR) * 2654435761u + 12345u;
    const unsigned col = (blockIdx.x % 8) * piece_off_per_xcd + lig * 16;
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        u32x4 v[NL];
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            seed = seed * 1664525u + 1013904223u;
            const unsigned r = (seed >> 8) % rows;
            v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, r * stride + col, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < NL; ++k) acc ^= v[k];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc.x;
}

template <int LPR, int NL>
float run(const void *tab, unsigned bytes, unsigned rows, unsigned stride, unsigned poff, int blocks, int iters, unsigned *sink)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((gather<LPR, NL>), dim3(blocks), dim3(256), 0, 0, tab, bytes, rows, stride, poff, iters, sink);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((gather<LPR, NL>), dim3(blocks), dim3(256), 0, 0, tab, bytes, rows, stride, poff, iters, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    const unsigned stride = 2048;
    unsigned *sink; hipMalloc(&sink, 64);
    void *tab; const size_t cap = size_t(1) << 28; hipMalloc(&tab, cap); hipMemset(tab, 1, cap);
    printf("piece_B loads_in_flight blocks/CU rows slice_MB/XCD  TB/s   B/clk/CU(2.1GHz)\n");
    for (unsigned rows : {1000u, 10000u, 19717u, 100000u}) {
        for (int bpc : {2, 4, 8}) {
            const int blocks = 256 * bpc;
#define RUN(LPR, NL)                                                                                              \
    {                                                                                                             \
        const int iters = 4096 / NL / bpc;                                                                        \
        const float ms = run<LPR, NL>(tab, rows * stride, rows, stride, LPR * 16 >= 2048 ? 0 : LPR * 16, blocks, iters, sink); \
        const double bytes = double(blocks) * 4 * iters * NL * 1024.0;                                            \
        printf("%6d %8d %9d %7u %8.2f   %6.2f   %6.1f\n", LPR * 16, NL, bpc, rows, rows * LPR * 16 / 1e6,          \
               bytes / ms / 1e9, bytes / ms * 1e3 / 256 / 2.1e9);                                                  \
    }
            RUN(8, 4) RUN(8, 8) RUN(16, 2) RUN(16, 4) RUN(16, 8) RUN(16, 16) RUN(32, 4) RUN(32, 8) RUN(64, 4) RUN(64, 8)
        }
    }
    return 0;
}
