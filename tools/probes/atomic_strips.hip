// Probe: the mirror tiles of the symmetric loss kernel (decoder_bce.hip) as fp32 ATOMIC adds into one [N][16] buffer
// against the strips it writes today (one 4 KB tile per (row panel, column tile), folded later by the edge kernel).
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomic_strips.hip -o bin/atomic_strips
// Shapes: Pubmed (N = 19717, 128-row panels, 11 tiles per block) and a 4096-molecule batch (N = 95000, 256-row panels).
// Every block owns (panel I, chunk c) and emits one [64 columns][16] tile per column tile right of its panel, 256
// threads x 4 floats; nothing else (no MFMA work in between: the raw rate of the memory side).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: strip stores (dwordx4), 1: atomic adds (4 x f32), 2: atomic adds, tile order rotated per block
__global__ __launch_bounds__(256) void emit(float *out, long n, int panel_rows, int tiles_per_block, long n_tiles, int chunks)
{
    const long I = blockIdx.x / chunks, c = blockIdx.x % chunks;
    const long first_tile = (I + 1) * panel_rows / 64;              // tiles right of the panel
    const long t0 = first_tile + c * tiles_per_block;
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
    for (int k = 0; k < tiles_per_block; ++k) {
        int kk = k;
        if (MODE == 2) kk = (k + blockIdx.x) % tiles_per_block;
        const long t = t0 + kk;
        if (t >= n_tiles) { if (MODE == 2) continue; else break; }
        if (MODE == 0) {
            float *p = out + (I * n_tiles + t) * 1024 + threadIdx.x * 4;      // a strip of its own per panel
            *reinterpret_cast<f32x4 *>(p) = v;
        } else {
            float *p = out + t * 1024 + threadIdx.x * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) atomicAdd(p + q, v[q]);
        }
    }
}

template <int MODE>
float run(float *buf, long n, int panel_rows, int tiles_per_block)
{
    const long n_tiles = (n + 63) / 64, panels = (n + panel_rows - 1) / panel_rows;
    const int chunks = int((n_tiles + tiles_per_block - 1) / tiles_per_block);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(emit<MODE>, dim3(unsigned(panels * chunks)), dim3(256), 0, 0, buf, n, panel_rows, tiles_per_block, n_tiles, chunks);
    hipEventRecord(a);
    const int it = 20;
    for (int w = 0; w < it; ++w) hipLaunchKernelGGL(emit<MODE>, dim3(unsigned(panels * chunks)), dim3(256), 0, 0, buf, n, panel_rows, tiles_per_block, n_tiles, chunks);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / it * 1e3f;
}

int main()
{
    struct { const char *name; long n; int pr, tpb; } cases[] = {{"pubmed", 19717, 128, 11}, {"zinc-4096", 95000, 256, 53}};
    for (auto &c : cases) {
        const long n_tiles = (c.n + 63) / 64, panels = (c.n + c.pr - 1) / c.pr;
        float *buf; hipMalloc(&buf, size_t(panels) * n_tiles * 1024 * 4);
        hipMemset(buf, 0, size_t(panels) * n_tiles * 1024 * 4);
        double tiles = 0; for (long I = 0; I < panels; ++I) tiles += double(n_tiles - (I + 1) * c.pr / 64 > 0 ? n_tiles - (I + 1) * c.pr / 64 : 0);
        const float s = run<0>(buf, c.n, c.pr, c.tpb), a1 = run<1>(buf, c.n, c.pr, c.tpb), a2 = run<2>(buf, c.n, c.pr, c.tpb);
        printf("%-10s N %6ld: %.0f tiles = %.1f MB of strips | strip stores %.1f us | atomic adds %.1f us (%.1f G atomics/s) | rotated order %.1f us\n",
               c.name, c.n, tiles, tiles * 4096 / 1e6, s, a1, tiles * 1024 / a1 / 1e3, a2);
        hipFree(buf);
    }
    return 0;
}
