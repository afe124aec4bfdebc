// Probe (round 4): a gather stream that mixes HOT rows (a per-XCD set that fits the XCD's L2) with COLD rows (a 2 GiB
// table, no reuse) -- the access pattern of the heavy rows of an R-MAT graph.  Which cache-policy bits on the COLD
// loads keep the hot set resident?  aux of raw_buffer_load: 1 = sc0, 2 = nt, 16 = sc1 (gfx950).  Also: the rate of
// random 128-byte row fetches over footprints between the aggregate L2 (32 MiB) and the Infinity Cache (256 MiB).
//   hipcc --offload-arch=gfx950 -O3 -o gather_mix gather_mix.hip && ./gather_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int AUX>
__global__ __launch_bounds__(256) void mix(const void *tab, unsigned bytes, unsigned hot_rows, unsigned cold_base,
                                           unsigned cold_rows, unsigned hot_thresh, int iters, unsigned *sink)
{
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(tab), 0, int(bytes), 0x00020000);
    const unsigned lane = threadIdx.x & 63, lig = lane & 7;
    const unsigned gid = blockIdx.x * 32 + threadIdx.x / 8;
    const unsigned xcd = blockIdx.x % 8;
    unsigned seed = gid * 2654435761u + 12345u;
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        u32x4 vh[8], vc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            seed = seed * 1664525u + 1013904223u;
            const bool hot = (seed >> 16) < hot_thresh;            // 16-bit threshold
            seed = seed * 1664525u + 1013904223u;
            const unsigned r = hot ? xcd * hot_rows + (seed >> 8) % hot_rows : cold_base + (seed >> 4) % cold_rows;
            const unsigned off = r * 128u + lig * 16u;
            vh[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, hot ? off : 0xfffffff0u, 0, 0);
            vc[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, hot ? 0xfffffff0u : off, 0, AUX);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc ^= vh[k] ^ vc[k];
    }
    if (acc[0] == 0x12345678u && acc[1] == 1u) sink[0] = acc[2] ^ acc[3];
}

template <int AUX>
float run(const void *tab, unsigned bytes, unsigned hot_rows, unsigned cold_base, unsigned cold_rows, float p_hot, int iters,
          unsigned *sink)
{
    const unsigned thresh = unsigned(p_hot * 65536.f);
    const int grid = 256 * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((mix<AUX>), dim3(grid), dim3(256), 0, 0, tab, bytes, hot_rows, cold_base, cold_rows, thresh, iters, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((mix<AUX>), dim3(grid), dim3(256), 0, 0, tab, bytes, hot_rows, cold_base, cold_rows, thresh, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double rows = double(grid) * 32 * 8 * iters;
    return float(rows * 128 / (ms * 1e-3) / 1e12);     // TB/s of gathered rows
}

int main()
{
    const size_t bytes = (size_t(1) << 31) + (size_t(1) << 30);      // 3 GiB table
    void *tab; unsigned *sink;
    hipMalloc(&tab, bytes); hipMalloc(&sink, 64);
    hipMemset(tab, 1, bytes);
    const unsigned total_rows = unsigned(bytes / 128);
    printf("== mixed stream: hot set per XCD + cold rows over 2 GiB; TB/s of gathered rows\n");
    printf("%8s %6s | %7s %7s %7s %7s %7s %7s %7s\n", "hot_rows", "p_hot", "plain", "sc0", "nt", "sc0nt", "sc1", "sc0sc1", "sc1nt");
    const unsigned cold_base = 8u << 20, cold_rows = 1u << 24;       // cold rows start 1 GiB in
    for (unsigned hr : {4096u, 8192u, 16384u, 24576u, 32768u})
        for (float p : {0.5f, 0.7f}) {
            printf("%8u %6.2f |", hr, p);
            printf(" %7.2f", run<0>(tab, unsigned(bytes - 1), hr, cold_base, cold_rows, p, 64, sink));
            printf(" %7.2f", run<1>(tab, unsigned(bytes - 1), hr, cold_base, cold_rows, p, 64, sink));
            printf(" %7.2f", run<2>(tab, unsigned(bytes - 1), hr, cold_base, cold_rows, p, 64, sink));
            printf(" %7.2f", run<3>(tab, unsigned(bytes - 1), hr, cold_base, cold_rows, p, 64, sink));
            printf(" %7.2f", run<16>(tab, unsigned(bytes - 1), hr, cold_base, cold_rows, p, 64, sink));
            printf(" %7.2f", run<17>(tab, unsigned(bytes - 1), hr, cold_base, cold_rows, p, 64, sink));
            printf(" %7.2f\n", run<18>(tab, unsigned(bytes - 1), hr, cold_base, cold_rows, p, 64, sink));
            fflush(stdout);
        }
    printf("== all-hot stream over a footprint (8 x hot_rows x 128 B), plain loads: L2 -> Infinity Cache -> HBM\n");
    for (unsigned hr : {16384u, 32768u, 65536u, 131072u, 262144u, 524288u, 1048576u, 2097152u}) {
        const float t = run<0>(tab, unsigned(bytes - 1), hr, cold_base, cold_rows, 1.0f, 64, sink);
        printf("footprint %7.1f MiB: %6.2f TB/s\n", 8.0 * hr * 128 / 1048576.0, t);
        fflush(stdout);
    }
    (void)total_rows;
    return 0;
}
