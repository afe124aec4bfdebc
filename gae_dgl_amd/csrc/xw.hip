// K3/K4, "stream" family: the two dense products of a GCN layer whose input is WIDE and whose output is narrow
// (layer 1 of the reference model: 500 / 1433 / 3703 input features -> 32, gae_dgl/gae.py:10,13-16,26-31)
//
//     forward   P [n, J]  = act(X [n, K] W [J, K]^T + b)        J <= 32
//     backward  dW [J, K] = G [n, J]^T X [n, K],   db [J] = colsum(D (.) [Ym > 0])
//
// Both are passes over X that move 4 K bytes per row and do 2 J flops per byte: HBM-bound (intensity J / 2 = 16
// flop/B against 157 TFLOP/s of fp32 MFMA / 8 TB/s = 20), provided X is read exactly ONCE, every CU pulls on HBM
// for the whole launch, and nothing else of comparable size moves.  The kernels this file replaces for these
// shapes (dense.hip: linear_fwd_pieces_kernel, atb_bf16_kernel) re-read the weights once per 32-row block (as many
// L2 bytes as X has HBM bytes), lived for ~10 dependent round trips per wave and left a partial round of CUs idle:
// 2.7 TB/s forward, 2.9 TB/s backward on Pubmed (profiles/r02_linear_bench.txt).  Here:
//   * forward: W is STATIONARY in registers -- the 8 waves of a block split K into 256-byte (512-byte) column
//     slices, each wave holds its 32 x 64 (32 x 128) slice of W as MFMA B-fragments for the whole launch and streams
//     the block's row tiles (16 rows) through them: 4 (8) raw-buffer dwordx4 loads per wave and tile, three tiles
//     in flight, no LDS and no barrier in front of the MFMAs; the 8 partial tiles meet in a double-buffered LDS
//     area once per tile, in wave order (deterministic).  One block per CU, ceil(tiles / CUs) tiles each.
//   * backward: a block owns (row partition, 64-column slice); its 8 waves interleave over the partition's 4-row
//     groups, lane (n, g) loads X[row + g][slice + 4 n ..] -- one dwordx4 = 4 rows x 256 contiguous bytes per
//     instruction -- whose 4 values are the B operands of 4 MFMAs (output columns 4 n + q); A = G^T comes as two
//     dword loads.  The per-(partition, slice) partials (2 MB for Pubmed instead of one 64 KB tile per block) are
//     added in partition order by a second small launch, which also finishes db.
//   * rows / tiles / slices outside the operands are addressed BEHIND the raw buffer: the bounds check returns
//     zeros without touching memory, so the pipelines are branch-free; columns >= K are zeroed by selects (the pad
//     columns of a row may hold anything).
// Products: exact fp32 (v_mfma_f32_16x16x4_f32 == an fmaf chain) for fp32 storage; bf16-stored X (BASELINE config 5)
// feeds v_mfma_f32_16x16x32_bf16 directly in the forward, with W = hi + lo split into two bf16 fragments (fp32
// accumulation), and is widened in registers for the fp32 MFMAs of the backward.
#include <string.h>

#include "common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
using gae::v4f;

constexpr int kXwMaxWaves = 8;
constexpr unsigned kBehind = 0xF0000000u;      // byte offset behind every operand (all below 0xE0000000 bytes, xw_usable):
                                               // the bounds check returns zeros, also with a small offset added

struct XwFwdArgs {
    const void *X;
    const float *W, *bias;
    float *out;
    int64_t n, ldo, split_stride;      // out + blockIdx.y * split_stride receives this split's rows (ldo floats apart)
    unsigned x_bytes, ldx_bytes, w_bytes;
    int K, J, ldw, act;
    int tiles_per_block, k_per_block;  // columns per (block, split)
    unsigned long long *stamps;        // (experiments, DBG = 3) [block][wave][16] s_memtime stamps
};

// workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() also drains the vector-memory
// counter (s_waitcnt vmcnt(0)): every row tile the wave has in flight would have to land before each barrier and
// the prefetch ring would be worth nothing.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// zero the elements of one 16-byte load that lie at or behind column K (`left` = K - first column of the load)
template <typename TX>
__device__ __forceinline__ u32x4 mask_tail(u32x4 v, int left)
{
    if constexpr (sizeof(TX) == 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = r < left ? v[r] : 0u;
    } else {
#pragma unroll
        for (int d = 0; d < 4; ++d) v[d] = 2 * d + 1 < left ? v[d] : (2 * d < left ? (v[d] & 0xffffu) : 0u);
    }
    return v;
}

// ---------------------------------------------------------------------------------------------------- forward
template <typename TX, int NH, int SLICE_BYTES, int DBG = 0, int DEPTH = 2, bool P3 = false, int TCV = 6>
__global__ __launch_bounds__(64 * kXwMaxWaves) void xw_fwd_kernel(const XwFwdArgs a)
{
    // P3 (fp32 storage): the products on the bf16 matrix pipe from three bf16 pieces per operand, six piece pairs
    // (gae::split_bf16x4_3: fp32-grade, every term to 2^-24) -- 12 v_mfma_f32_16x16x32_bf16 per 64 columns and NH
    // instead of 16 v_mfma_f32_16x16x4_f32 that take four times as long each on the fp32 lanes.
    static_assert(!P3 || sizeof(TX) == 4, "P3 splits fp32 storage");
    constexpr int ES = int(sizeof(TX));
    constexpr int NL = SLICE_BYTES / 64;            // 16-byte loads per lane and row tile
    constexpr int KW = SLICE_BYTES / ES;            // columns of a wave's slice
    constexpr int CPL = 16 / ES;                    // columns per 16-byte load of one lane (4 fp32 / 8 bf16)
    // DEPTH: row tiles in flight (this one + DEPTH - 1 ahead)
    constexpr int OUTW = 16 * NH;
    // partial tiles of up to TC row tiles x 8 waves: the waves run their tiles WITHOUT meeting (a barrier per tile kept
    // the two waves of a SIMD in lockstep and the matrix pipe idle while they reduced: 11 us for the MFMAs alone on
    // Pubmed, 4.3 us of which is issue time); one barrier and one reduction per chunk of TC tiles
    constexpr int TC = TCV;
    __shared__ __attribute__((aligned(16))) float red[TC][kXwMaxWaves][NH * 256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = int(blockDim.x >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    auto stamp = [&](int k) {
        if constexpr (DBG == 3) {
            if (lane == 0 && k < 16)
                a.stamps[(int64_t(blockIdx.x) * kXwMaxWaves + wave) * 16 + k] = __builtin_amdgcn_s_memtime();
        }
    };
    stamp(0);
    const int kb = int(blockIdx.y) * a.k_per_block + wave * KW;     // first column of this wave's slice
    const bool has_k = wave * KW < a.k_per_block && kb < a.K;

    // ---- the wave's slice of W as B fragments, held for the whole launch
    //      fp32: wf[i][r][nh] = W[16 nh + l15][kb + 16 i + 4 g + r]                      (MFMA (i, r) contracts those k)
    //      bf16: whi / wlo[i][nh] = 8 consecutive k of row 16 nh + l15 from kb + 32 i + 8 g, W = hi + lo
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.W), 0, int(a.w_bytes), 0x00020000);
    float wf[(sizeof(TX) == 4 && !P3) ? NL : 1][4][NH];
    bf16x8 whi[sizeof(TX) == 2 ? NL : 1][NH], wlo[sizeof(TX) == 2 ? NL : 1][NH];
    // P3: step s = loads (2 s, 2 s + 1); lane (l15, g) contracts k in {16 (2 s) + 4 g + r} U {16 (2 s + 1) + 4 g + r}
    gae::v4s w3[P3 ? NL : 1][3][NH];                 // [load i][piece][nh]: 4 of the 8 k of step i / 2 (half i % 2)
    u32x4 wraw[P3 ? NL : 1][NH];                     // P3: the raw vectors; shifted and split AFTER the first row tiles
    int wsh[P3 ? NL : 1];                            //     are requested (the split needs the data, the requests do not)
    if constexpr (P3) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int k = kb + i * 16 + 4 * g;
            const int ku = k <= a.K - 4 ? k : a.K - 4;
            wsh[i] = k - ku;
#pragma unroll
            for (int nh = 0; nh < NH; ++nh) {
                const int j = 16 * nh + l15;
                const bool in = has_k && j < a.J && k < a.K;
                wraw[i][nh] = __builtin_amdgcn_raw_buffer_load_b128(rw, in ? unsigned(j * a.ldw + ku) * 4u : kBehind, 0, 0);
            }
        }
    } else
#pragma unroll
    for (int i = 0; i < NL; ++i)
#pragma unroll
        for (int nh = 0; nh < NH; ++nh) {
            const int j = 16 * nh + l15;
            float v[CPL];
#pragma unroll
            for (int c4 = 0; c4 < CPL; c4 += 4) {
                // one 16-byte raw-buffer load per 4 columns (16 rows x 64 bytes per instruction), branch-free so that
                // the first row tiles are requested before the weights have arrived: the row's last, partial vector
                // starts at K - 4 and is shifted into place by selects; rows >= J and slices >= K read behind the
                // buffer (zeros)
                const int k = kb + i * (64 / ES) + CPL * g + c4;
                const int ku = k <= a.K - 4 ? k : a.K - 4;
                const bool in = has_k && j < a.J && k < a.K;
                const u32x4 l = __builtin_amdgcn_raw_buffer_load_b128(rw, in ? unsigned(j * a.ldw + ku) * 4u : kBehind, 0, 0);
                const int sh = k - ku;                        // 0 except in the tail vector (then 1..3)
                v[c4 + 0] = __uint_as_float(sh == 0 ? l[0] : sh == 1 ? l[1] : sh == 2 ? l[2] : l[3]);
                v[c4 + 1] = __uint_as_float(sh == 0 ? l[1] : sh == 1 ? l[2] : sh == 2 ? l[3] : 0u);
                v[c4 + 2] = __uint_as_float(sh == 0 ? l[2] : sh == 1 ? l[3] : 0u);
                v[c4 + 3] = __uint_as_float(sh == 0 ? l[3] : 0u);
            }
            if constexpr (sizeof(TX) == 4) {
#pragma unroll
                for (int c = 0; c < 4; ++c) wf[i][c][nh] = v[c];
            } else {
                gae::v4s h0, l0, h1, l1;
                gae::split_bf16x4(v4f{v[0], v[1], v[2], v[3]}, h0, l0);
                gae::split_bf16x4(v4f{v[4], v[5], v[6], v[7]}, h1, l1);
                struct P { gae::v4s a, b; };
                whi[i][nh] = __builtin_bit_cast(bf16x8, (P{h0, h1}));
                wlo[i][nh] = __builtin_bit_cast(bf16x8, (P{l0, l1}));
            }
        }

    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.X), 0, int(a.x_bytes), 0x00020000);
    const int64_t tile0 = int64_t(blockIdx.x) * a.tiles_per_block;
    const int64_t n_tiles = (a.n + 15) / 16;
    const int nt = int(min<int64_t>(a.tiles_per_block, n_tiles - tile0));      // tiles of this block (>= 1)
    const unsigned col_off = unsigned(kb + CPL * g) * ES;
    int left[NL];                                   // columns in front of K, counted from each load's first column
#pragma unroll
    for (int i = 0; i < NL; ++i) left[i] = a.K - (kb + i * (64 / ES) + CPL * g);

    u32x4 st[DEPTH][NL];
    auto issue = [&](u32x4 (&s)[NL], int t) {
        const int64_t row = (tile0 + t) * 16 + l15;
        const bool ok = has_k && t < nt && row < a.n;
        const unsigned base = (ok && DBG != 2) ? unsigned(row) * a.ldx_bytes + col_off : kBehind;
#pragma unroll
        for (int i = 0; i < NL; ++i) s[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, base + unsigned(i * 64), 0, 0);
    };
    float *out = a.out + int64_t(blockIdx.y) * a.split_stride;
    const bool final_pass = a.bias != nullptr || a.act != GAE_ACT_IDENTITY;     // (split launches pass neither)
    const bool out_vec = (a.ldo & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15u) == 0;

    auto tile = [&](const u32x4 (&s)[NL], int t) {
        v4f acc[NH];
#pragma unroll
        for (int nh = 0; nh < NH; ++nh) acc[nh] = v4f{0.f, 0.f, 0.f, 0.f};
        if constexpr (P3) {
            struct P8 { gae::v4s a, b; };
#pragma unroll
            for (int i = 0; i < NL; i += 2) {
                gae::v4s xp[2][3];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const u32x4 x = mask_tail<TX>(s[i + h], left[i + h]);
                    gae::split_bf16x4_3(v4f{__uint_as_float(x[0]), __uint_as_float(x[1]), __uint_as_float(x[2]),
                                            __uint_as_float(x[3])}, xp[h][0], xp[h][1], xp[h][2]);
                }
                const bf16x8 xh = __builtin_bit_cast(bf16x8, (P8{xp[0][0], xp[1][0]}));
                const bf16x8 xm = __builtin_bit_cast(bf16x8, (P8{xp[0][1], xp[1][1]}));
                const bf16x8 xl = __builtin_bit_cast(bf16x8, (P8{xp[0][2], xp[1][2]}));
                if constexpr (DBG == 1) {            // (experiment: the loads and the split, no MFMAs)
#pragma unroll
                    for (int nh = 0; nh < NH; ++nh)
                        acc[nh][0] += __builtin_bit_cast(float, int(xh[0]) ^ int(xm[1]) ^ int(xl[2]));
                } else
#pragma unroll
                for (int nh = 0; nh < NH; ++nh) {
                    const bf16x8 wh = __builtin_bit_cast(bf16x8, (P8{w3[i][0][nh], w3[i + 1][0][nh]}));
                    const bf16x8 wm = __builtin_bit_cast(bf16x8, (P8{w3[i][1][nh], w3[i + 1][1][nh]}));
                    const bf16x8 wl = __builtin_bit_cast(bf16x8, (P8{w3[i][2][nh], w3[i + 1][2][nh]}));
                    v4f c = acc[nh];                 // smallest terms first
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl, wh, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, wl, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xm, wm, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xm, wh, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, wm, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, wh, c, 0, 0, 0);
                    acc[nh] = c;
                }
            }
        } else
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const u32x4 x = mask_tail<TX>(s[i], left[i]);
            if constexpr (DBG == 1) {
#pragma unroll
                for (int nh = 0; nh < NH; ++nh) acc[nh][0] += __uint_as_float(x[0] ^ x[1] ^ x[2] ^ x[3]);
            } else if constexpr (sizeof(TX) == 4) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int nh = 0; nh < NH; ++nh)
                        acc[nh] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(x[r]), wf[i][r][nh], acc[nh], 0, 0, 0);
            } else {
                const bf16x8 xa = __builtin_bit_cast(bf16x8, x);
#pragma unroll
                for (int nh = 0; nh < NH; ++nh) {
                    acc[nh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa, wlo[i][nh], acc[nh], 0, 0, 0);
                    acc[nh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa, whi[i][nh], acc[nh], 0, 0, 0);
                }
            }
        }
        // ---- the partial tile goes to this wave's LDS slot as a [16 rows][OUTW] image (4-byte stores: the two row groups
        //      a half-wave holds meet on the same banks, a 2-way conflict that costs a ds_write_b32 nothing); chunks of TC
        //      tiles are then added in wave order by 16-byte reads of consecutive lanes and leave as ONE 16-byte store per
        //      thread (four dword stores per thread from one register were serialised by a vmcnt(0) each: 3000 cycles)
        float *mine = &red[t % TC][wave][0];
#pragma unroll
        for (int nh = 0; nh < NH; ++nh)
#pragma unroll
            for (int q = 0; q < 4; ++q) mine[(4 * g + q) * OUTW + 16 * nh + l15] = acc[nh][q];
        stamp(2 + t);
        if ((t + 1) % TC == 0 || t + 1 == nt) {
            const int c0 = t / TC * TC, cnt = t + 1 - c0;
            lds_barrier();
            constexpr int QPR = OUTW / 4;               // column quads per row
            for (int e = tid; e < cnt * 16 * QPR; e += int(blockDim.x)) {
                const int tt = e / (16 * QPR), sl = e % (16 * QPR);       // sl = row * QPR + quad: consecutive 16 bytes
                const int orow = sl / QPR, oc0 = (sl % QPR) * 4;
                v4f v[kXwMaxWaves];
#pragma unroll
                for (int w = 0; w < kXwMaxWaves; ++w)                                          // all requested together
                    v[w] = *reinterpret_cast<const v4f *>(&red[tt][w < nw ? w : 0][sl * 4]);
                v4f y = v[0];
#pragma unroll
                for (int w = 1; w < kXwMaxWaves; ++w) y += w < nw ? v[w] : v4f{0.f, 0.f, 0.f, 0.f};     // wave order
                const int64_t r = (tile0 + c0 + tt) * 16 + orow;
                if (r < a.n && oc0 < a.J) {
                    if (final_pass) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            if (a.bias && oc0 + q < a.J) y[q] += a.bias[oc0 + q];
                            if (a.act == GAE_ACT_RELU) y[q] = fmaxf(y[q], 0.f);
                        }
                    }
                    float *dst = out + r * a.ldo + oc0;
                    if (out_vec && oc0 + 4 <= a.J) *reinterpret_cast<v4f *>(dst) = y;
                    else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (oc0 + q < a.J) dst[q] = y[q];
                    }
                }
            }
            if (t + 1 < nt) lds_barrier();          // the next chunk overwrites the slots
            stamp(12);
        }
    };

#define GAE_PIN() __builtin_amdgcn_sched_barrier(0)
    if constexpr (P3) GAE_PIN();
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) issue(st[d], d);
    GAE_PIN();
    stamp(1);
    if constexpr (P3) {
#pragma unroll
        for (int i = 0; i < NL; ++i)
#pragma unroll
            for (int nh = 0; nh < NH; ++nh) {
                const u32x4 l = wraw[i][nh];
                const int sh = wsh[i];                        // 0 except in the row's last, partial vector (then 1..3)
                const v4f v = {__uint_as_float(sh == 0 ? l[0] : sh == 1 ? l[1] : sh == 2 ? l[2] : l[3]),
                               __uint_as_float(sh == 0 ? l[1] : sh == 1 ? l[2] : sh == 2 ? l[3] : 0u),
                               __uint_as_float(sh == 0 ? l[2] : sh == 1 ? l[3] : 0u), __uint_as_float(sh == 0 ? l[3] : 0u)};
                gae::split_bf16x4_3(v, w3[i][0][nh], w3[i][1][nh], w3[i][2][nh]);
            }
        GAE_PIN();
    }
    for (int t = 0; t < nt; t += DEPTH) {
        bool more = true;
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (more) {
                issue(st[(d + DEPTH - 1) % DEPTH], t + d + DEPTH - 1); GAE_PIN(); tile(st[d], t + d); GAE_PIN();
                more = t + d + 1 < nt;
            }
        }
    }
#undef GAE_PIN
}

// out[r][c] = act(bias[c] + sum_s partial[s][r][c])  -- second pass of a forward that was split along K
__global__ __launch_bounds__(256) void xw_split_reduce_kernel(const float *__restrict__ partial, int splits, int64_t n,
                                                              int J, const float *__restrict__ bias, int act,
                                                              float *__restrict__ out, int64_t ldo)
{
    const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (e >= n * J) return;
    float y = 0.f;
    for (int s = 0; s < splits; ++s) y += partial[int64_t(s) * n * J + e];     // split order
    const int64_t r = e / J;
    const int c = int(e - r * J);
    if (bias) y += bias[c];
    if (act == GAE_ACT_RELU) y = fmaxf(y, 0.f);
    out[r * ldo + c] = y;
}

// ---------------------------------------------------------------------------------------------------- backward
struct XtgArgs {
    const void *X;
    const float *G, *Gmask;            // dW = (G (.) [Gmask > 0])^T X        (Gmask may be NULL)
    const float *D, *Dmask;            // db = colsum(D (.) [Dmask > 0])      (D may be NULL: no db)
    float *part, *dbpart;              // part[p][32][kp], dbpart[p][32]
    int64_t n;
    unsigned x_bytes, ldx_bytes, g_bytes, ldg_bytes, gm_bytes, ldgm_bytes;
    int64_t ldd, lddm;
    int K, J, kp, n_slices, xcd_map;
    int64_t rows_per_part;             // multiple of 32
};

// rows of G a block of the GLDS form can stage (16 NH floats each): 128 KB of the CU's 160 KB
constexpr int kXtgGldsBytes = 128 * 1024;

// GLDS: the partition's rows of G (gated by Gmask) are staged in LDS once per block and the A operands are read from
// there -- the main loop requests nothing but X.  (Round 4 took the kernel apart: X loads + MFMAs without the G loads
// 9.8 us, everything 12.9 us on Pubmed -- the 8-byte G requests, one more per row group in the same in-order queue as
// the X stream, cost the last 3 us although they all hit L2.)
template <typename TX, int NH, bool MASKED, int DBG = 0, int NS = 4, int U = 2, bool GLDS = false>
__global__ __launch_bounds__(512) void xtg_kernel(const XtgArgs a)
{
    constexpr int ES = int(sizeof(TX));
    constexpr int NQ = 16 / ES;                     // columns per lane (4 fp32 / 8 bf16): output columns NQ n + q
    constexpr int SW = 16 * NQ;                     // columns of a block's slice (64 / 128)
    // U: 4-row groups per pipeline stage, NS: stages (NS - 1 of them in flight while one is multiplied)
    constexpr int DBU = 8;                          // db: elements per thread in flight
    constexpr int GW = 16 * NH;                     // floats per staged row of G
    constexpr int RED_FLOATS = 8 * 16 * NH * (SW + 4);
    constexpr int SMEM_FLOATS = GLDS ? (RED_FLOATS > kXtgGldsBytes / 4 ? RED_FLOATS : kXtgGldsBytes / 4) : RED_FLOATS;
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];       // Gs during the main loop, then the waves' tiles
    float (*red)[16 * NH][SW + 4] = reinterpret_cast<float (*)[16 * NH][SW + 4]>(smem);
    float *Gs = smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    // block -> (partition, slice): consecutive (partition, slice) pairs run on ONE XCD next to each other (bijective
    // XCD remap of the block id), so the 256-byte pieces of a row are requested from one L2 at about the same time
    // (with slice = id % 8 every XCD pulled its own column stripe out of every DRAM page: 18.3 -> 17.0 us on Pubmed)
    const unsigned lid = a.xcd_map ? gae::xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int part = int(lid / unsigned(a.n_slices)), slice = int(lid % unsigned(a.n_slices));
    const int64_t rbeg = int64_t(part) * a.rows_per_part;
    const int64_t rend = min(a.n, rbeg + a.rows_per_part);

    // ---- db: the partition's rows are dealt to its column-slice blocks; this block adds D (.) [Dmask > 0] over its
    //      share -- element e = tid + 512 i of the share's [rows][32] array, all requested before the main loop --
    //      and leaves 32 column sums in dbpart[partition][slice]
    float dsum = 0.f;
    const bool want_db = a.D != nullptr;
    float dbv[DBU], dbm[DBU];                       // the first 16 DBU rows of the share: requested HERE, added after the
    int64_t db_d0 = 0, db_d1 = 0;                   // main loop (an add in front of it would wait for them and hold up
    if (want_db) {                                  // the first X loads: + 3 us on Pubmed)
        const int64_t share = (a.rows_per_part / 32 + a.n_slices - 1) / a.n_slices * 32;    // rows per slice block
        db_d0 = rbeg + share * slice; db_d1 = min(rend, db_d0 + share);
        const int j = tid & 31;
#pragma unroll
        for (int u = 0; u < DBU; ++u) {
            const int64_t r = db_d0 + (tid >> 5) + 16 * u;
            const int64_t rc = r < db_d1 ? r : (db_d1 > db_d0 ? db_d1 - 1 : 0);                // clamped: branch-free loads
            dbv[u] = j < a.J ? a.D[rc * a.ldd + j] : 0.f;
            dbm[u] = (a.Dmask != nullptr && j < a.J) ? a.Dmask[rc * a.lddm + j] : 1.f;
        }
    }

    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.X), 0, int(a.x_bytes), 0x00020000);
    __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.G), 0, int(a.g_bytes), 0x00020000);
    __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.Gmask), 0,
                                                                  a.Gmask ? int(a.gm_bytes) : 0, 0x00020000);
    constexpr bool masked = MASKED;
    const unsigned xcol = unsigned(slice * SW + NQ * l15) * ES;
    v4f acc[NH][NQ];
#pragma unroll
    for (int mh = 0; mh < NH; ++mh)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[mh][q] = v4f{0.f, 0.f, 0.f, 0.f};

    // A = G^T: MFMA mh of a step takes output row j = NH l15 + mh from lane (l15, g) -- the NH values of a lane are
    // adjacent in G: ONE 4- / 8-byte load per step (rows j >= J: behind the buffer)
    struct Stage { u32x4 x[U]; unsigned gv[U][NH], gm[MASKED ? U : 1][NH]; };
    const unsigned gcol = unsigned(NH * l15) * 4u;
    const bool j_ok = NH * l15 < a.J;           // (J odd: the pair's second value is masked below)
    // group `it` of this wave: rows rbeg + 4 (wave + 8 (U it + u)) + g -- the 8 waves sweep 32 consecutive rows
    auto issue = [&](Stage &s, int64_t it) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t row = rbeg + 4 * (wave + 8 * (U * it + u)) + g;
            const bool ok = row < rend;
            s.x[u] = __builtin_amdgcn_raw_buffer_load_b128(rx, (ok && DBG != 2) ? unsigned(row) * a.ldx_bytes + xcol : kBehind, 0, 0);
            if constexpr (GLDS) continue;           // the A operands come from LDS (compute)
            const unsigned go = (ok && j_ok && DBG != 3) ? unsigned(row) * a.ldg_bytes + gcol : kBehind;
            const unsigned mo = (ok && j_ok) ? unsigned(row) * a.ldgm_bytes + gcol : kBehind;
            if constexpr (NH == 2) {
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rg, go, 0, 0));
                s.gv[u][0] = v[0]; s.gv[u][1] = v[1];
                if (masked) {
                    const u32x2 m = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rm, mo, 0, 0));
                    s.gm[u][0] = m[0]; s.gm[u][1] = m[1];
                }
            } else {
                s.gv[u][0] = __builtin_amdgcn_raw_buffer_load_b32(rg, go, 0, 0);
                if (masked) s.gm[u][0] = __builtin_amdgcn_raw_buffer_load_b32(rm, mo, 0, 0);
            }
        }
    };
    auto compute = [&](const Stage &s, int64_t it) {
        float gl[U][NH];
        if constexpr (GLDS) {                       // all of the stage's A operands requested before the first MFMA
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float *gp = Gs + (4 * (wave + 8 * (U * int(it) + u)) + g) * GW + NH * l15;
                if constexpr (NH == 2) {
                    const gae::v2f t = *reinterpret_cast<const gae::v2f *>(gp);
                    gl[u][0] = t[0]; gl[u][1] = t[1];
                } else {
                    gl[u][0] = *gp;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float xv[NQ];
            if constexpr (sizeof(TX) == 4) {
#pragma unroll
                for (int q = 0; q < 4; ++q) xv[q] = __uint_as_float(s.x[u][q]);
            } else {
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    xv[2 * d] = __uint_as_float(s.x[u][d] << 16);
                    xv[2 * d + 1] = __uint_as_float(s.x[u][d] & 0xffff0000u);
                }
            }
#pragma unroll
            for (int mh = 0; mh < NH; ++mh) {
                float gval;
                if constexpr (GLDS) {
                    gval = gl[u][mh];               // (gated and zero-padded when it was staged)
                } else {
                    const bool on = NH * l15 + mh < a.J && (!masked || __uint_as_float(s.gm[u][mh]) > 0.f);
                    gval = on ? __uint_as_float(s.gv[u][mh]) : 0.f;
                }
                if constexpr (DBG == 1) {
                    acc[mh][0][0] += gval * (xv[0] + xv[NQ - 1]);
                } else {
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
                        acc[mh][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(gval, xv[q], acc[mh][q], 0, 0, 0);
                }
            }
        }
    };
    const int64_t groups = (rend - rbeg + 3) / 4;                               // 4-row groups of the partition
    const int64_t my_groups = groups > wave ? (groups - wave + 7) / 8 : 0;       // ... of this wave
    const int64_t iters = (my_groups + U - 1) / U;
#define GAE_PIN() __builtin_amdgcn_sched_barrier(0)
    // GLDS: rows the waves read from LDS = those of wave 0's stages (the longest list), staged with zeros behind the
    // partition: 16-byte pieces, requested BEFORE the first X stages and written to LDS behind them (the vector-memory
    // counter is in order: waiting for the G pieces leaves the X stages in flight)
    constexpr int GPT = 8;                          // 16-byte pieces per thread and trip
    const int cap_rows = int((((groups + 7) / 8 + U - 1) / U) * U * 32);
    const int n_pieces = GLDS ? cap_rows * (GW / 4) : 0;
    u32x4 gq[GLDS ? GPT : 1], gmq[(GLDS && MASKED) ? GPT : 1];
    auto g_request = [&](int e0) {
        if constexpr (!GLDS) return;
#pragma unroll
        for (int t = 0; t < (GLDS ? GPT : 1); ++t) {
            const int e = e0 + 512 * t + tid;
            const int64_t row = rbeg + e / (GW / 4);
            const unsigned c4 = unsigned(e % (GW / 4)) * 16u;
            const bool ok = e < n_pieces && row < rend && DBG != 3;
            gq[t] = __builtin_amdgcn_raw_buffer_load_b128(rg, ok ? unsigned(row) * a.ldg_bytes + c4 : kBehind, 0, 0);
            if constexpr (MASKED)
                gmq[t] = __builtin_amdgcn_raw_buffer_load_b128(rm, ok ? unsigned(row) * a.ldgm_bytes + c4 : kBehind, 0, 0);
        }
    };
    auto g_store = [&](int e0) {
        if constexpr (!GLDS) return;
#pragma unroll
        for (int t = 0; t < (GLDS ? GPT : 1); ++t) {
            const int e = e0 + 512 * t + tid;
            if (e >= n_pieces) continue;
            const int j0 = (e % (GW / 4)) * 4;
            v4f v;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool on = j0 + q < a.J && (!masked || __uint_as_float(gmq[MASKED ? t : 0][q]) > 0.f);
                v[q] = on ? __uint_as_float(gq[t][q]) : 0.f;
            }
            *reinterpret_cast<v4f *>(Gs + int64_t(e) * 4) = v;
        }
    };
    Stage st[NS];
    const bool run = iters > 0 && DBG != 4;
    if constexpr (GLDS) g_request(0);               // (every wave stages, also one without row groups of its own)
    if (run) {
#pragma unroll
        for (int d = 0; d < NS - 1; ++d) issue(st[d], d);
    }
    GAE_PIN();
    if constexpr (GLDS) {
        g_store(0);
        for (int e0 = 512 * GPT; e0 < n_pieces; e0 += 512 * GPT) { g_request(e0); g_store(e0); }
        lds_barrier();
    }
    if (run) {
        for (int64_t it = 0; it < iters; it += NS) {
            bool more = true;
#pragma unroll
            for (int d = 0; d < NS; ++d) {
                if (more) {
                    issue(st[(d + NS - 1) % NS], it + d + NS - 1); GAE_PIN(); compute(st[d], it + d); GAE_PIN();
                    more = it + d + 1 < iters;
                }
            }
        }
    }
#undef GAE_PIN
    if constexpr (GLDS) lds_barrier();              // the waves' tiles below overwrite the staged rows
    // ---- the 8 waves' tiles meet in LDS: acc[mh][q][r] = dW[NH (4 g + r) + mh][slice SW + NQ l15 + q]
#pragma unroll
    for (int mh = 0; mh < NH; ++mh)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q4 = 0; q4 < NQ; q4 += 4)
                *reinterpret_cast<v4f *>(&red[wave][NH * (4 * g + r) + mh][NQ * l15 + q4]) =
                    v4f{acc[mh][q4][r], acc[mh][q4 + 1][r], acc[mh][q4 + 2][r], acc[mh][q4 + 3][r]};
    __syncthreads();
    constexpr int VPR = SW / 4;                                  // float4 per output row of the slice
    for (int e = tid; e < 16 * NH * VPR; e += 512) {
        const int j = e / VPR, c4 = e % VPR;
        v4f t = *reinterpret_cast<const v4f *>(&red[0][j][4 * c4]);
#pragma unroll
        for (int w = 1; w < 8; ++w) t += *reinterpret_cast<const v4f *>(&red[w][j][4 * c4]);
        *reinterpret_cast<v4f *>(a.part + (int64_t(part) * 32 + j) * a.kp + slice * SW + 4 * c4) = t;
    }
    if (want_db) {
        const int j = tid & 31;
#pragma unroll
        for (int u = 0; u < DBU; ++u) dsum += (db_d0 + (tid >> 5) + 16 * u < db_d1 && dbm[u] > 0.f) ? dbv[u] : 0.f;
        for (int64_t r0 = db_d0 + (tid >> 5) + 16 * DBU; r0 < db_d1; r0 += 16 * DBU) {      // shares of > 128 rows
            float dv[DBU], mv[DBU];
#pragma unroll
            for (int u = 0; u < DBU; ++u) {
                const int64_t r = r0 + 16 * u;
                const int64_t rc = r < db_d1 ? r : db_d1 - 1;
                dv[u] = j < a.J ? a.D[rc * a.ldd + j] : 0.f;
                mv[u] = (a.Dmask != nullptr && j < a.J) ? a.Dmask[rc * a.lddm + j] : 1.f;
            }
#pragma unroll
            for (int u = 0; u < DBU; ++u) dsum += (r0 + 16 * u < db_d1 && mv[u] > 0.f) ? dv[u] : 0.f;
        }
        __syncthreads();
        float *dred = &red[0][0][0];
        dred[tid] = dsum;                              // [sub = tid / 32][j = tid % 32]
        __syncthreads();
        if (tid < 32) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += dred[q * 32 + tid];
            a.dbpart[(int64_t(part) * a.n_slices + slice) * 32 + tid] = t;
        }
    }
}

// the stand-alone reduction of one or two partial lists (gae::sum_partials order): blocks [0, nb_a) take list a
__global__ __launch_bounds__(256) void partials_reduce_kernel(const gae::PartialList a, const gae::PartialList b, unsigned nb_a)
{
    const bool second = blockIdx.x >= nb_a;
    const gae::PartialList &t = second ? b : a;
    const unsigned lb = second ? blockIdx.x - nb_a : blockIdx.x;
    const int L = gae::partial_lanes(t.n_partials);
    const int64_t e = int64_t(lb) * (256 / L) + threadIdx.x / L;
    const int lane = threadIdx.x % L;
    const bool live = e < t.n;
    const int64_t ec = live ? e : 0;
    const int64_t r = ec / t.row_len, c = ec % t.row_len;
    const float g = gae::sum_partials(t.base + r * t.row_pitch + c, t.n_partials, t.stride, lane, L);
    if (live && lane == 0) t.out[r * t.out_pitch + c] = g;
}

// ---------------------------------------------------------------------------------------------------- plans
struct FwdPlan { int slice_bytes, nw, splits, k_per_block, tiles_per_block; int64_t row_blocks; };

gae::Knob g_xw_rows{0};       // "xw_rows": rows per block of the forward (0 = auto: one block per CU)
gae::Knob g_xw_parts{0};      // "xw_parts": row partitions of the backward (0 = auto)
gae::Knob g_xw{1};            // "xw": 0 = never use this family (dense.hip kernels instead)
gae::Knob g_xw_depth{0};      // "xw_depth" (experiments): other ring depths of the fp32 kernels (forward 3 / 4 / 5 tiles, default 2;
                              // backward (stages, groups) (2, 4) / (3, 4) / (4, 4) / (6, 2), default (4, 2))
gae::Knob g_xw_glds{1};       // "xw_glds": 1 = the backward stages its partition's rows of G in LDS (J > 16), 0 = loads them per row group
gae::Knob g_xw_xcd{1};        // "xw_xcd": XCD-aware block order of the backward (1) or slice-major ids (0); same sums
gae::Knob g_xw_stamps{0}, g_xw_stamps_hi{0};     // "xw_stamps" / "xw_stamps_hi" (experiments, with xw_dbg = 3): low / high half of the device address of [blocks][8][16] uint64 time stamps
gae::Knob g_xw_tc{0};         // "xw_tc" (experiments): row tiles per reduction chunk of the forward (LDS: 16 KB each; 0 = 6)
gae::Knob g_xw_bpc{1};        // "xw_bpc": blocks per CU the forward's grid is sized for
gae::Knob g_xw_p3{1};         // "xw_p3": fp32-stored X, forward: 1 = bf16 x 3-piece products on the matrix pipe (six pairs, fp32-grade), 0 = exact fp32 MFMAs
gae::Knob g_xw_dbg{0};        // "xw_dbg" (experiments, wrong results): 1 = without MFMAs, 2 = without X loads; backward also 3 = without G loads, 4 = without its main loop

FwdPlan fwd_plan(int64_t n, int K, int elem)
{
    // A block streams `tiles_per_block` row tiles through the W slices its waves hold, so it fetches its share of W
    // once per launch: W bytes per block / X bytes per block = 4 J / (16 elem tiles_per_block).  Tall operands
    // (Pubmed: 1233 tiles) give every CU ceil(tiles / 256) tiles and the whole of K.  Operands with few rows (Cora 170
    // tiles, Citeseer 208) would run one tile per block and read W -- as large as X -- from L2 once per tile: they get 4
    // tiles per block and are split along K over blocks instead (partials added in split order by a second launch).
    FwdPlan p{};
    const int64_t tiles = (n + 15) / 16;
    const int64_t slots = 256 * (g_xw_bpc > 0 ? int64_t(g_xw_bpc) : 1);
    int64_t t = g_xw_rows > 0 ? (g_xw_rows + 15) / 16 : (tiles + slots - 1) / slots;
    if (g_xw_rows == 0 && t < 4) t = tiles < 4 ? tiles : 4;
    if (t < 1) t = 1;
    if (t > (1 << 20)) t = 1 << 20;
    p.tiles_per_block = int(t);
    p.row_blocks = (tiles + t - 1) / t;
    const int64_t k_bytes = int64_t(K) * elem;
    int splits = int((k_bytes + 4095) / 4096);                      // at most 8 waves x 512 bytes per block
    const int want = int(256 / p.row_blocks);                        // ... and enough blocks for every CU
    if (want > splits) splits = want;
    const int max_splits = int((k_bytes + 1023) / 1024);            // a block keeps at least 4 waves x 256 bytes busy
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    const int per = (K + splits - 1) / splits;
    p.slice_bytes = int64_t(per) * elem <= 2048 ? 256 : 512;
    const int kw = p.slice_bytes / elem;
    p.nw = (per + kw - 1) / kw;
    if (p.nw < 1) p.nw = 1;
    p.k_per_block = p.nw * kw;
    p.splits = (K + p.k_per_block - 1) / p.k_per_block;              // no empty split
    return p;
}

struct BwdPlan { int sw, n_slices, kp, parts; int64_t rows_per_part; };

BwdPlan bwd_plan(int64_t n, int K, int elem)
{
    BwdPlan p{};
    p.sw = 16 * (16 / elem);
    p.n_slices = (K + p.sw - 1) / p.sw;
    p.kp = p.n_slices * p.sw;
    // ONE round of blocks: slices x partitions <= CUs (a block is 8 waves of ~160 VGPRs: one per CU; 288 blocks on
    // 256 CUs ran as two rounds, 17 -> 34 us on Pubmed)
    int parts = g_xw_parts > 0 ? int(g_xw_parts) : 256 / p.n_slices;
    if (parts < 1) parts = 1;
    int64_t rpp = (n + parts - 1) / parts;
    rpp = (rpp + 31) / 32 * 32;                    // whole sweeps of the 8 waves
    if (rpp < 32) rpp = 32;
    p.rows_per_part = rpp;
    p.parts = int((n + rpp - 1) / rpp);
    if (p.parts < 1) p.parts = 1;
    return p;
}

} // namespace

namespace gae {

Knob *xw_knob(const char *name)
{
    if (strcmp(name, "xw_rows") == 0) return &g_xw_rows;
    if (strcmp(name, "xw_parts") == 0) return &g_xw_parts;
    if (strcmp(name, "xw") == 0) return &g_xw;
    if (strcmp(name, "xw_dbg") == 0) return &g_xw_dbg;
    if (strcmp(name, "xw_xcd") == 0) return &g_xw_xcd;
    if (strcmp(name, "xw_glds") == 0) return &g_xw_glds;
    if (strcmp(name, "xw_depth") == 0) return &g_xw_depth;
    if (strcmp(name, "xw_p3") == 0) return &g_xw_p3;
    if (strcmp(name, "xw_tc") == 0) return &g_xw_tc;
    if (strcmp(name, "xw_stamps") == 0) return &g_xw_stamps;
    if (strcmp(name, "xw_stamps_hi") == 0) return &g_xw_stamps_hi;
    if (strcmp(name, "xw_bpc") == 0) return &g_xw_bpc;
    return nullptr;
}

// can the stream family run these operands?  (rows of whole 16-byte vectors, X addressable through one raw buffer)
bool xw_usable(const void *X, int64_t ldx, int64_t n, int64_t K, int64_t J, int elem)
{
    return g_xw != 0 && n > 0 && K >= 193 && K < (1 << 24) && J >= 1 && J <= 32 && (ldx * elem) % 16 == 0 &&
           ldx >= K && aligned16(X) && n * ldx * elem < int64_t(0xE0000000u);
}

int64_t xw_fwd_workspace_bytes(int64_t n, int64_t K, int64_t J, int elem)
{
    const FwdPlan p = fwd_plan(n, int(K), elem);
    return p.splits > 1 ? (int64_t(p.splits) * n * J * 4 + 255) / 256 * 256 : 0;
}

int xw_fwd_splits(int64_t n, int64_t K, int elem) { return fwd_plan(n, int(K), elem).splits; }

int xw_fwd_launch(const void *X, int64_t ldx, int64_t n, int K, int elem, const float *W, int64_t ldw, const float *bias,
                  int J, int act, float *out, int64_t ldo, void *ws, int64_t ws_bytes, hipStream_t s, bool keep_splits)
{
    FwdPlan p = fwd_plan(n, K, elem);
    if (p.splits > 1 && (ws == nullptr || ws_bytes < int64_t(p.splits) * n * J * 4)) {
        set_error("xw_fwd: operands split along K need %lld bytes of workspace", (long long)(int64_t(p.splits) * n * J * 4));
        return GAE_E_SIZE;
    }
    XwFwdArgs a{};
    a.X = X; a.W = W; a.n = n; a.K = K; a.J = J; a.ldw = int(ldw);
    a.x_bytes = unsigned(n * ldx * elem); a.ldx_bytes = unsigned(ldx * elem);
    a.w_bytes = unsigned(((int64_t(J) - 1) * ldw + K) * 4);
    a.tiles_per_block = p.tiles_per_block; a.k_per_block = p.k_per_block;
    a.stamps = reinterpret_cast<unsigned long long *>((uint64_t(uint32_t(int(g_xw_stamps_hi))) << 32) | uint32_t(int(g_xw_stamps)));
    if (g_xw_dbg == 3 && a.stamps == nullptr) {      // the stamping variant writes through this pointer
        set_error("xw_fwd: xw_dbg=3 (s_memtime stamps) needs the stamp buffer's address in xw_stamps / xw_stamps_hi");
        return GAE_E_NULL;
    }
    if (p.splits > 1) { a.bias = nullptr; a.act = GAE_ACT_IDENTITY; a.out = static_cast<float *>(ws); a.ldo = J; a.split_stride = n * J; }
    else { a.bias = bias; a.act = act; a.out = out; a.ldo = ldo; a.split_stride = 0; }
    const dim3 grid(unsigned(p.row_blocks), unsigned(p.splits)), block(unsigned(64 * p.nw));
#define GAE_XW(TX, NH, SB) hipLaunchKernelGGL((xw_fwd_kernel<TX, NH, SB>), grid, block, 0, s, a)
    const bool wide = J > 16;
    if (elem == 4) {
        if (g_xw_p3 != 0 && p.slice_bytes == 256 && wide && (g_xw_dbg || g_xw_depth || g_xw_tc)) {     // experiments
            const int dp = g_xw_depth ? int(g_xw_depth) : 2, tc = g_xw_tc ? int(g_xw_tc) : 6, dbg = int(g_xw_dbg);
            bool launched = false;
#define GAE_XWE(DB, DP, TC) if (dbg == DB && dp == DP && tc == TC) { hipLaunchKernelGGL((xw_fwd_kernel<float, 2, 256, DB, DP, true, TC>), grid, block, 0, s, a); launched = true; }
            GAE_XWE(0, 2, 6); GAE_XWE(0, 3, 6); GAE_XWE(0, 4, 6); GAE_XWE(0, 2, 3); GAE_XWE(0, 3, 3); GAE_XWE(0, 4, 3);
            GAE_XWE(1, 2, 6); GAE_XWE(2, 2, 6); GAE_XWE(1, 4, 3); GAE_XWE(2, 4, 3); GAE_XWE(3, 2, 6);
#undef GAE_XWE
            if (!launched) {      // only the instantiated (xw_dbg, xw_depth, xw_tc) combinations exist: never return uninitialised output
                set_error("xw_fwd: no kernel for the knob combination xw_dbg=%d xw_depth=%d xw_tc=%d (with xw_p3=1: depth 2|3|4 x "
                          "tc 3|6 at dbg 0; dbg 1|2 at (2,6) and (4,3); dbg 3 at (2,6))", dbg, dp, tc);
                return GAE_E_RANGE;
            }
        }
        else if (g_xw_p3 != 0) {
#define GAE_XW3(NH, SB) hipLaunchKernelGGL((xw_fwd_kernel<float, NH, SB, 0, 2, true>), grid, block, 0, s, a)
            if (p.slice_bytes == 256) { if (wide) GAE_XW3(2, 256); else GAE_XW3(1, 256); }
            else { if (wide) GAE_XW3(2, 512); else GAE_XW3(1, 512); }
#undef GAE_XW3
        }
        else if (p.slice_bytes == 256 && wide && g_xw_depth == 4) hipLaunchKernelGGL((xw_fwd_kernel<float, 2, 256, 0, 4>), grid, block, 0, s, a);
        else if (p.slice_bytes == 256 && wide && g_xw_depth == 5) hipLaunchKernelGGL((xw_fwd_kernel<float, 2, 256, 0, 5>), grid, block, 0, s, a);
        else if (p.slice_bytes == 256 && wide && g_xw_depth == 3) hipLaunchKernelGGL((xw_fwd_kernel<float, 2, 256, 0, 3>), grid, block, 0, s, a);
        else if (p.slice_bytes == 256 && wide && g_xw_dbg == 1) hipLaunchKernelGGL((xw_fwd_kernel<float, 2, 256, 1>), grid, block, 0, s, a);
        else if (p.slice_bytes == 256 && wide && g_xw_dbg == 2) hipLaunchKernelGGL((xw_fwd_kernel<float, 2, 256, 2>), grid, block, 0, s, a);
        else if (p.slice_bytes == 256) { if (wide) GAE_XW(float, 2, 256); else GAE_XW(float, 1, 256); }
        else { if (wide) GAE_XW(float, 2, 512); else GAE_XW(float, 1, 512); }
    } else {
        if (p.slice_bytes == 256) { if (wide) GAE_XW(unsigned short, 2, 256); else GAE_XW(unsigned short, 1, 256); }
        else { if (wide) GAE_XW(unsigned short, 2, 512); else GAE_XW(unsigned short, 1, 512); }
    }
#undef GAE_XW
    GAE_CHECK_LAUNCH("xw_fwd_kernel");
    if (p.splits > 1 && !keep_splits) {
        const int64_t ne = n * J;
        hipLaunchKernelGGL(xw_split_reduce_kernel, dim3(unsigned((ne + 255) / 256)), dim3(256), 0, s,
                           static_cast<const float *>(ws), p.splits, n, J, bias, act, out, ldo);
        GAE_CHECK_LAUNCH("xw_split_reduce_kernel");
    }
    return GAE_OK;
}

int launch_partials_reduce(const PartialList &a, const PartialList &b, hipStream_t s)
{
    auto blocks = [](const PartialList &t) -> int64_t {
        if (t.n <= 0 || t.out == nullptr) return 0;
        const int per = t.n_partials > 32 ? 256 / 64 : 256;
        return (t.n + per - 1) / per;
    };
    const int64_t na = blocks(a), nb = blocks(b);
    if (na + nb == 0) return GAE_OK;
    hipLaunchKernelGGL(partials_reduce_kernel, dim3(unsigned(na + nb)), dim3(256), 0, s, a, b, unsigned(na));
    GAE_CHECK_LAUNCH("partials_reduce_kernel");
    return GAE_OK;
}

int64_t xtg_workspace_bytes(int64_t n, int64_t K, int elem)
{
    const BwdPlan p = bwd_plan(n, int(K), elem);
    return (int64_t(p.parts) * 32 * p.kp + int64_t(p.parts) * (p.n_slices + 1) * 32) * 4 + 256;
}

int xtg_launch(const void *X, int64_t ldx, int64_t n, int K, int elem, const float *G, int64_t ldg, const float *Gmask,
               int64_t ldgm, const float *D, int64_t ldd, const float *Dmask, int64_t lddm, int J, float *dW,
               int64_t lddw, float *db, void *ws, int64_t ws_bytes, hipStream_t s, int64_t *layout_only = nullptr)
{
    const BwdPlan p = bwd_plan(n, K, elem);
    if (ws == nullptr || ws_bytes < xtg_workspace_bytes(n, K, elem)) {
        set_error("xtg: workspace of %lld bytes needed", (long long)xtg_workspace_bytes(n, K, elem));
        return GAE_E_SIZE;
    }
    XtgArgs a{};
    a.X = X; a.G = G; a.Gmask = Gmask; a.D = db ? D : nullptr; a.Dmask = Dmask;
    a.part = static_cast<float *>(ws);
    a.dbpart = a.part + int64_t(p.parts) * 32 * p.kp;
    a.n = n; a.K = K; a.J = J; a.kp = p.kp; a.n_slices = p.n_slices; a.rows_per_part = p.rows_per_part;
    a.x_bytes = unsigned(n * ldx * elem); a.ldx_bytes = unsigned(ldx * elem);
    a.g_bytes = unsigned(n * ldg * 4); a.ldg_bytes = unsigned(ldg * 4);
    a.gm_bytes = unsigned(n * ldgm * 4); a.ldgm_bytes = unsigned(ldgm * 4);
    a.ldd = ldd; a.lddm = lddm;
    bool want_dw = dW != nullptr, want_db = db != nullptr && D != nullptr;
    if (layout_only) {                 // partials only: dW / db are flags here, nothing is written to them
        layout_only[0] = p.parts; layout_only[1] = int64_t(32) * p.kp; layout_only[2] = p.kp;
        layout_only[3] = int64_t(p.parts) * 32 * p.kp; layout_only[4] = int64_t(p.parts) * p.n_slices; layout_only[5] = 32;
    }
    if (!want_db) a.D = nullptr;
    if (!want_dw) { a.x_bytes = 0; a.g_bytes = 0; a.gm_bytes = 0; a.Gmask = nullptr; }     // every load behind its buffer: zeros
    // (db alone still sweeps the slices: its rows are dealt to the slice blocks; the products of an unwanted dW go to
    // the workspace and are not reduced)
    a.xcd_map = g_xw_xcd != 0;
    const dim3 grid(unsigned(p.n_slices) * unsigned(p.parts));
    const bool wide = J > 16;
    // staged G ("xw_glds", default on): the rows the waves read -- wave 0's stages of 2 x 32 rows -- must fit 128 KB
    const int64_t groups_max = (p.rows_per_part + 3) / 4, cap_rows = (((groups_max + 7) / 8 + 1) / 2) * 2 * 32;
    const bool glds = g_xw_glds != 0 && want_dw && cap_rows * 32 * 4 <= kXtgGldsBytes;
#define GAE_XTG(TX, NH, ...) do { if (a.Gmask) hipLaunchKernelGGL((xtg_kernel<TX, NH, true, __VA_ARGS__>), grid, dim3(512), 0, s, a); \
                                   else hipLaunchKernelGGL((xtg_kernel<TX, NH, false, __VA_ARGS__>), grid, dim3(512), 0, s, a); } while (0)
    if (elem == 4) {
        if (wide && g_xw_dbg == 1) GAE_XTG(float, 2, 1);
        else if (wide && g_xw_dbg == 2) GAE_XTG(float, 2, 2);
        else if (wide && g_xw_dbg == 3) GAE_XTG(float, 2, 3);
        else if (wide && g_xw_dbg == 4) GAE_XTG(float, 2, 4);
        else if (wide && g_xw_depth == 2) GAE_XTG(float, 2, 0, 2, 4);
        else if (wide && g_xw_depth == 4) GAE_XTG(float, 2, 0, 4, 4);
        else if (wide && g_xw_depth == 3) GAE_XTG(float, 2, 0, 3, 4);
        else if (wide && g_xw_depth == 6) GAE_XTG(float, 2, 0, 6, 2);
        else if (wide && glds) GAE_XTG(float, 2, 0, 4, 2, true);
        else if (wide) GAE_XTG(float, 2, 0);
        else GAE_XTG(float, 1, 0);
    } else {
        if (wide && glds) GAE_XTG(unsigned short, 2, 0, 4, 2, true);
        else if (wide) GAE_XTG(unsigned short, 2, 0);
        else GAE_XTG(unsigned short, 1, 0);
    }
#undef GAE_XTG
    GAE_CHECK_LAUNCH("xtg_kernel");
    if (layout_only) return GAE_OK;
    PartialList la{}, lb{};
    if (want_dw) la = PartialList{a.part, dW, int64_t(J) * K, p.parts, int64_t(32) * p.kp, K, p.kp, lddw};
    if (want_db) lb = PartialList{a.dbpart, db, J, int64_t(p.parts) * p.n_slices, 32, J, J, J};
    return launch_partials_reduce(la, lb, s);
}

} // namespace gae

// ---------------------------------------------------------------------------------------------------- C ABI
extern "C" int gae_xw_usable(const void *X, int64_t ldx, int dtype, int64_t n, int64_t f_in, int64_t f_out)
{
    if (dtype != GAE_F32 && dtype != GAE_BF16) return 0;
    return gae::xw_usable(X, ldx, n, f_in, f_out, dtype == GAE_F32 ? 4 : 2) ? 1 : 0;
}

extern "C" int64_t gae_xw_fwd_workspace_bytes(int64_t n, int64_t f_in, int64_t f_out, int dtype)
{
    if (n < 0 || f_in < 1 || f_out < 1 || f_out > 32 || f_in >= (1 << 24) || (dtype != GAE_F32 && dtype != GAE_BF16))
        return GAE_E_SIZE;
    return gae::xw_fwd_workspace_bytes(n, f_in, f_out, dtype == GAE_F32 ? 4 : 2);
}

extern "C" int64_t gae_xw_fwd_splits(int64_t n, int64_t f_in, int64_t f_out, int dtype)
{
    if (n < 1 || f_in < 1 || f_out < 1 || f_out > 32 || f_in >= (1 << 24) || (dtype != GAE_F32 && dtype != GAE_BF16))
        return GAE_E_SIZE;
    return gae::xw_fwd_splits(n, f_in, dtype == GAE_F32 ? 4 : 2);
}

extern "C" int gae_xw_fwd(const void *X, int64_t ldx, int dtype, int64_t n, int64_t f_in, const float *W, int64_t ldw,
                          const float *b, int64_t f_out, int act, float *P, int64_t ldp, void *workspace,
                          int64_t workspace_bytes, int keep_splits, void *stream)
{
    GAE_REQUIRE(!keep_splits || (b == nullptr && act == GAE_ACT_IDENTITY), GAE_E_RANGE,
                "gae_xw_fwd: keep_splits leaves bias and activation to the consumer");
    GAE_REQUIRE(dtype == GAE_F32 || dtype == GAE_BF16, GAE_E_DTYPE, "gae_xw_fwd: dtype %d", dtype);
    GAE_REQUIRE(n >= 0 && f_in >= 0 && f_out >= 0, GAE_E_SIZE, "gae_xw_fwd: negative size");
    GAE_REQUIRE(act == GAE_ACT_IDENTITY || act == GAE_ACT_RELU, GAE_E_DTYPE, "gae_xw_fwd: activation %d", act);
    if (n == 0 || f_out == 0) return GAE_OK;
    GAE_REQUIRE(X && W && (P || keep_splits), GAE_E_NULL, "gae_xw_fwd: NULL pointer");
    GAE_REQUIRE(ldw >= f_in && ldp >= f_out, GAE_E_SIZE, "gae_xw_fwd: leading dimension too small");
    const int elem = dtype == GAE_F32 ? 4 : 2;
    GAE_REQUIRE(gae::xw_usable(X, ldx, n, f_in, f_out, elem), GAE_E_RANGE,
                "gae_xw_fwd: needs f_in >= 193, f_out <= 32, rows of whole 16-byte vectors and X below 4 GiB "
                "(gae_xw_usable); use gae_linear_fwd");
    GAE_REQUIRE(!workspace || gae::aligned16(workspace), GAE_E_ALIGN, "gae_xw_fwd: workspace not 16-byte aligned");
    return gae::xw_fwd_launch(X, ldx, n, int(f_in), elem, W, ldw, b, int(f_out), act, P, ldp, workspace, workspace_bytes,
                              gae::as_stream(stream), keep_splits != 0);
}

extern "C" int64_t gae_xw_wgrad_workspace_bytes(int64_t n, int64_t f_in, int dtype)
{
    if (n < 0 || f_in < 1 || f_in >= (1 << 24) || (dtype != GAE_F32 && dtype != GAE_BF16)) return GAE_E_SIZE;
    return gae::xtg_workspace_bytes(n, f_in, dtype == GAE_F32 ? 4 : 2);
}

extern "C" int gae_xw_wgrad(const void *X, int64_t ldx, int dtype, int64_t n, int64_t f_in, const float *G, int64_t ldg,
                            const float *Gmask, int64_t ldgm, const float *D, int64_t ldd, const float *Dmask,
                            int64_t lddm, int64_t f_out, float *dW, int64_t lddw, float *db, void *workspace,
                            int64_t workspace_bytes, void *stream)
{
    GAE_REQUIRE(dtype == GAE_F32 || dtype == GAE_BF16, GAE_E_DTYPE, "gae_xw_wgrad: dtype %d", dtype);
    GAE_REQUIRE(n >= 0 && f_in >= 0 && f_out >= 0, GAE_E_SIZE, "gae_xw_wgrad: negative size");
    if (f_out == 0 || (!dW && !db)) return GAE_OK;
    GAE_REQUIRE(n > 0, GAE_E_SIZE, "gae_xw_wgrad: needs at least one row");
    GAE_REQUIRE(!dW || (X && G && ldg >= f_out && lddw >= f_in), GAE_E_NULL, "gae_xw_wgrad: dW needs X, G and lddw >= f_in");
    GAE_REQUIRE(!Gmask || ldgm >= f_out, GAE_E_SIZE, "gae_xw_wgrad: ldgm < f_out");
    GAE_REQUIRE(!db || (D && ldd >= f_out && (!Dmask || lddm >= f_out)), GAE_E_NULL, "gae_xw_wgrad: db needs D");
    const int elem = dtype == GAE_F32 ? 4 : 2;
    GAE_REQUIRE(!dW || gae::xw_usable(X, ldx, n, f_in, f_out, elem), GAE_E_RANGE,
                "gae_xw_wgrad: needs f_in >= 193, f_out <= 32, rows of whole 16-byte vectors and X below 4 GiB "
                "(gae_xw_usable); use gae_linear_bwd");
    GAE_REQUIRE(n * ldg * 4 < int64_t(0xE0000000u) && (!Gmask || n * ldgm * 4 < int64_t(0xE0000000u)), GAE_E_SIZE,
                "gae_xw_wgrad: G larger than a raw buffer resource addresses");
    GAE_REQUIRE(workspace && gae::aligned16(workspace), GAE_E_ALIGN, "gae_xw_wgrad: workspace missing or not 16-byte aligned");
    return gae::xtg_launch(X, ldx, n, int(f_in), elem, G, ldg, Gmask, ldgm, D, ldd, Dmask, lddm, int(f_out), dW, lddw, db,
                           workspace, workspace_bytes, gae::as_stream(stream));
}

extern "C" int gae_x_xw_wgrad_partials(const void *X, int64_t ldx, int dtype, int64_t n, int64_t f_in, const float *G,
                                     int64_t ldg, const float *Gmask, int64_t ldgm, const float *D, int64_t ldd,
                                     const float *Dmask, int64_t lddm, int64_t f_out, int want_dW, int want_db,
                                     void *workspace, int64_t workspace_bytes, int64_t *layout_out, void *stream)
{
    GAE_REQUIRE(dtype == GAE_F32 || dtype == GAE_BF16, GAE_E_DTYPE, "gae_x_xw_wgrad_partials: dtype %d", dtype);
    GAE_REQUIRE(n > 0 && f_in > 0 && f_out > 0 && layout_out, GAE_E_SIZE, "gae_x_xw_wgrad_partials: bad sizes");
    GAE_REQUIRE(want_dW || want_db, GAE_E_RANGE, "gae_x_xw_wgrad_partials: nothing to compute");
    GAE_REQUIRE(!want_dW || (X && G && ldg >= f_out), GAE_E_NULL, "gae_x_xw_wgrad_partials: dW needs X and G");
    GAE_REQUIRE(!Gmask || ldgm >= f_out, GAE_E_SIZE, "gae_x_xw_wgrad_partials: ldgm < f_out");
    GAE_REQUIRE(!want_db || (D && ldd >= f_out && (!Dmask || lddm >= f_out)), GAE_E_NULL, "gae_x_xw_wgrad_partials: db needs D");
    const int elem = dtype == GAE_F32 ? 4 : 2;
    GAE_REQUIRE(!want_dW || gae::xw_usable(X, ldx, n, f_in, f_out, elem), GAE_E_RANGE,
                "gae_x_xw_wgrad_partials: operand not accepted (gae_xw_usable)");
    GAE_REQUIRE(n * ldg * 4 < int64_t(0xE0000000u) && (!Gmask || n * ldgm * 4 < int64_t(0xE0000000u)), GAE_E_SIZE,
                "gae_x_xw_wgrad_partials: G larger than a raw buffer resource addresses");
    GAE_REQUIRE(workspace && gae::aligned16(workspace), GAE_E_ALIGN, "gae_x_xw_wgrad_partials: workspace missing or unaligned");
    float flag = 0.f;           // any non-NULL pointer: xtg_launch only tests dW / db for NULL in this mode
    return gae::xtg_launch(X, ldx, n, int(f_in), elem, G, ldg, Gmask, ldgm, D, ldd, Dmask, lddm, int(f_out),
                           want_dW ? &flag : nullptr, f_in, want_db ? &flag : nullptr, workspace, workspace_bytes,
                           gae::as_stream(stream), layout_out);
}
