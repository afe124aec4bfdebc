// Dense halves of a two-layer GCN encoder on a graph with MILLIONS of rows (BASELINE config 4: R-MAT, 2^24 nodes,
// widths 32 -> 32 -> 16), written for one pass over every operand.
//
// The reference evaluates  H1 = relu((A X) W1^T + b1),  Z = (A H1) W2^T + b2  (gae_dgl/gae.py:26-31,36-45) and its
// autograd (train_inductive.py:51).  With M1 = A X stored, the same values (up to fp32 rounding of the re-associated
// products, the tolerance contract of DESIGN.md section 6) come from
//   forward   T  = H1 W2^T                  (gae_linear2_fwd: M1 -> H1 and T in ONE pass, H1 never re-read)
//             Z  = A T + b2                 (gae_spmm_csr_ep: the second aggregation runs at width 16, not 32)
//   backward  G  = A^T dZ                   (gae_spmm_csr at width 16)
//             dW2 = G^T H1, db2 = colsum(dZ), dY1 = (G W2) (.) [H1 > 0], dW1 = dY1^T M1, db1 = colsum(dY1)
//                                           (gae_gcn2_bwd_dense: ONE pass over G, dZ, H1, M1; dY1 is never stored)
// instead of four Linear / weight-gradient launches that each stream a [N, 32] operand again, and the aggregate of
// layer 2 (A H1, [N, 32]) is neither written nor read.
//
// Both kernels are HBM-bound streams (roofline: bytes of the operands they read and write once); the products run on
// v_mfma_f32_32x32x2_f32 (exact fp32) and hide behind the loads.  A WAVE owns whole 32-row tiles; the weights stay in
// its registers for its whole row range; no LDS in the row loop.
//
// MFMA bookkeeping (32x32x2, D[i][n] += sum_k A[i][k] B[k][n]): lane l = (i = l & 31, h = l >> 5) supplies
// A[i][k = h] and B[k = h][n = i]; accumulator register r of lane (n, h) is D[rho(r, h)][n] with
// rho(r, h) = (r & 3) + 8 (r >> 2) + 4 h.  Which k a (step, h) pair stands for is free as long as both operands agree.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int rho(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// 8 fp32 values -> bf16 hi and bf16 lo = bf16(v - hi) fragments (K = 8 consecutive steps of a 32x32x16 MFMA)
__device__ __forceinline__ void split8(const float *v, s16x8 &hi, s16x8 &lo)
{
    gae::v4s h0, l0, h1, l1;
    gae::split_bf16x4(gae::v4f{v[0], v[1], v[2], v[3]}, h0, l0);
    gae::split_bf16x4(gae::v4f{v[4], v[5], v[6], v[7]}, h1, l1);
    hi = s16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
    lo = s16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
}
// three pieces: v = hi + mid + lo exactly (gae::split_bf16x4_3)
__device__ __forceinline__ void split8_3(const float *v, s16x8 &hi, s16x8 &mid, s16x8 &lo)
{
    gae::v4s h0, m0, l0, h1, m1, l1;
    gae::split_bf16x4_3(gae::v4f{v[0], v[1], v[2], v[3]}, h0, m0, l0);
    gae::split_bf16x4_3(gae::v4f{v[4], v[5], v[6], v[7]}, h1, m1, l1);
    hi = s16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
    mid = s16x8{m0[0], m0[1], m0[2], m0[3], m1[0], m1[1], m1[2], m1[3]};
    lo = s16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
}
__device__ __forceinline__ f32x16 mfma16(const s16x8 &a, const s16x8 &b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// D += A B with A = ah + am + al, B = bh + bm + bl: the six piece pairs down to 2^-24 relative, smallest first -- an
// fp32-grade product on the bf16 matrix pipe (knob atb_bf16 = 1, the default of the weight-gradient products)
__device__ __forceinline__ f32x16 mfma_split6(const s16x8 &ah, const s16x8 &am, const s16x8 &al, const s16x8 &bh,
                                              const s16x8 &bm, const s16x8 &bl, f32x16 c)
{
    c = mfma16(al, bh, c); c = mfma16(ah, bl, c); c = mfma16(am, bm, c);
    c = mfma16(am, bh, c); c = mfma16(ah, bm, c); c = mfma16(ah, bh, c);
    return c;
}
// D += A B with A = ah + al, B = bh + bl on the bf16 matrix pipe, the three leading terms (the library's rule for
// weight-gradient products, knob atb_bf16: 16 mantissa bits per operand, fp32 accumulation), smallest terms first
__device__ __forceinline__ f32x16 mfma_split3(const s16x8 &ah, const s16x8 &al, const s16x8 &bh, const s16x8 &bl, f32x16 c)
{
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al), __builtin_bit_cast(bf16x8, bh), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bl), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bh), c, 0, 0, 0);
    return c;
}

// ---------------------------------------------------------------------------------------------------------------
// gae_linear2_fwd:  Y1 = act(A W1^T + b1) [n, J1],  T = Y1 W2^T [n, J2];  K1 <= 64, J1 <= 32, J2 <= 32.
// Both products are evaluated TRANSPOSED (D1[j][row], D2[j2][row]): the rows of the tile sit in the lanes, so
// the accumulator registers of the first product ARE the B operands of the second (step r, half h <-> j = rho(r, h)).
// ---------------------------------------------------------------------------------------------------------------
template <int KB, int NS = 2>
__global__ __launch_bounds__(256, 3) void linear2_rows_kernel(const float *__restrict__ A, int64_t lda,
                                                           const float *__restrict__ W1, int64_t ldw1,
                                                           const float *__restrict__ b1, int act1,
                                                           const float *__restrict__ W2, int64_t ldw2,
                                                           float *__restrict__ Y1, int64_t ldy1, float *__restrict__ T,
                                                           int64_t ldt, int64_t n, int K1, int J1, int J2,
                                                           int tiles_per_wave, const uint8_t *__restrict__ a_dead,
                                                           const int32_t *__restrict__ rows)
{
    // rows != NULL (list mode): n counts LIST ENTRIES; entry p stands for row rows[p] of A / Y1 / T (ascending).  On a
    // power-law graph most rows of an aggregate are zero rows (no in-edges): the pass then runs its two products on the
    // rows that have edges only -- it is bound by the fp32 matrix pipe, not by bytes -- and gae_linear2_fill_dead writes
    // the one row all the others share.
    constexpr int LDY = 36, LDT = 36;           // floats per LDS row: 32 + 4 (bank spread, 16-byte aligned)
    __shared__ __attribute__((aligned(16))) float lds[4 * (32 * LDY + 32 * LDT)];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5;
    // weights of this wave: w1[kb][q] = W1[j = i][k = 8 kb + 4 h + q], w2[r] = W2[j2 = i][j = rho(r, h)], bias of the
    // 16 outputs this lane holds of its row
    float w1[KB][4], w2[16], bv[16];
    {
        const int jc = i < J1 ? i : J1 - 1, j2c = i < J2 ? i : J2 - 1;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = kb * 8 + 4 * h + q;
                const float v = W1[int64_t(jc) * ldw1 + (k < K1 ? k : K1 - 1)];
                w1[kb][q] = (k < K1 && i < J1) ? v : 0.f;
            }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = rho(r, h);
            const int jj = j < J1 ? j : J1 - 1;
            const float v = W2[int64_t(j2c) * ldw2 + jj];
            w2[r] = (j < J1 && i < J2) ? v : 0.f;
            const float b = b1 ? b1[jj] : 0.f;
            bv[r] = (b1 && j < J1) ? b : 0.f;
        }
    }
    const int K4 = (K1 + 3) & ~3;
    struct Stage { float a[KB][4]; unsigned rid; };
    // a_dead: rows of A that ARE zero and were never written (an aggregate's rows without edges, GAE_SPMM_SKIP_ROWS):
    // not read -- on a power-law graph most rows, i.e. most of this pass's input bytes.  The mask byte of a tile is
    // requested one tile AHEAD of the tile's row loads (in front of them it would add a round trip to every tile).
    // what a tile needs to know about its row BEFORE its loads: bit 31 = dead (a_dead), bits 0..30 = the row id
    auto dead_of = [&](int64_t row0) -> unsigned {
        const int64_t row = row0 + i;
        const int64_t rc = row < n ? row : n - 1;
        const unsigned rid = rows != nullptr ? unsigned(rows[rc]) : unsigned(rc);
        return rid | ((a_dead != nullptr && rows == nullptr && a_dead[rc]) ? 0x80000000u : 0u);
    };
    auto load = [&](Stage &st, int64_t row0, unsigned dinfo) {
        const bool dead = (dinfo >> 31) != 0;
        const int64_t rc = int64_t(dinfo & 0x7fffffffu);
        st.rid = unsigned(rc);
        const float *ap = A + rc * lda;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const int k = kb * 8 + 4 * h;
            float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!dead) t4 = *reinterpret_cast<const float4 *>(ap + (k <= K4 - 4 ? k : K4 - 4));
            st.a[kb][0] = t4.x; st.a[kb][1] = t4.y; st.a[kb][2] = t4.z; st.a[kb][3] = t4.w;
        }
    };
    auto tile = [&](const Stage &st, int64_t row0) {
        const int64_t row = row0 + i;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float av = (kb * 8 + 4 * h + q < K1) ? st.a[kb][q] : 0.f;      // pad columns may hold anything
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[kb][q], av, acc, 0, 0, 0);
            }
        // lane (row i, h) now holds D1[j = rho(r, h)][row]: bias, activation, store, and feed the second product
        f32x16 acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float y = acc[r] + bv[r];
            if (act1 == GAE_ACT_RELU) y = fmaxf(y, 0.f);
            acc[r] = y;
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w2[r], y, acc2, 0, 0, 0);
        }
        // The rows sit in the lanes: stored from here, every instruction would write 16-byte pieces of 32 different lines
        // (measured: the pass ran at 3.2 TB/s).  The tile goes through this wave's LDS slab instead and leaves as whole
        // rows: 8 lanes x 16 bytes per row of Y1, 8 rows per instruction.  (One wave writes and reads its own slab:
        // LDS operations of a wave execute in order, no block barrier.)
        float *ly = lds + wave * (32 * LDY + 32 * LDT), *lt = ly + 32 * LDY;
        (void)row;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int j0 = 8 * q4 + 4 * h;
            *reinterpret_cast<float4 *>(ly + i * LDY + j0) = make_float4(acc[4 * q4], acc[4 * q4 + 1], acc[4 * q4 + 2], acc[4 * q4 + 3]);
            *reinterpret_cast<float4 *>(lt + i * LDT + j0) = make_float4(acc2[4 * q4], acc2[4 * q4 + 1], acc2[4 * q4 + 2], acc2[4 * q4 + 3]);
        }
        __builtin_amdgcn_wave_barrier();
        const int c4 = (lane & 7) * 4, rsub = lane >> 3;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int r = p * 8 + rsub;
            const int64_t orow = int64_t(__shfl(st.rid, r, 64));     // lane r (h = 0) holds the id of the tile's row r
            const float4 y4 = *reinterpret_cast<const float4 *>(ly + r * LDY + c4);
            const float4 t4 = *reinterpret_cast<const float4 *>(lt + r * LDT + c4);
            if (row0 + r < n) {
                if (Y1 != nullptr) {
                    float *yp = Y1 + orow * ldy1 + c4;
                    if (c4 + 4 <= J1) *reinterpret_cast<float4 *>(yp) = y4;
                    else { if (c4 < J1) yp[0] = y4.x; if (c4 + 1 < J1) yp[1] = y4.y; if (c4 + 2 < J1) yp[2] = y4.z; }
                }
                float *tp = T + orow * ldt + c4;
                if (c4 + 4 <= J2) *reinterpret_cast<float4 *>(tp) = t4;
                else { if (c4 < J2) tp[0] = t4.x; if (c4 + 1 < J2) tp[1] = t4.y; if (c4 + 2 < J2) tp[2] = t4.z; }
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
    const int64_t t0 = (int64_t(blockIdx.x) * 4 + wave) * tiles_per_wave;
    const int64_t n_tiles = (n + 31) / 32;
    if (t0 >= n_tiles) return;
    const int64_t t1 = t0 + tiles_per_wave < n_tiles ? t0 + tiles_per_wave : n_tiles;
    // Ring of NS stages: NS - 1 tiles of row loads are in flight while one is multiplied; the row id / dead bit of a tile is
    // requested one tile ahead of its loads.  Round 6 (tools/r06/tall_sq.sh: 38 - 54 % of the fp32 matrix pipe, the waves parked
    // at s_waitcnt for 35 - 50 % of their time): scheduling barriers keep the next tile's loads IN FRONT of this tile's MFMAs --
    // same-box A/B, same bits: list mode 0.638 -> 0.630 ms, a contiguous 5 M rows 0.324 -> 0.300 - 0.319, all 2^24 rows 1.00 ->
    // 0.82 or 1.00 (bimodal from run to run).  NS = 3 / 4 (152 -> 168 VGPRs with 4 / 14 spilled) are slower: 1.04 ms / 0.34 -
    // 0.39 / 0.66 - 0.71 ms.  The pass is neither a clean stream nor pipe-bound; see MEASUREMENTS.md, round 6.
    Stage st[NS];
    unsigned dnext = dead_of((t0 + NS - 1) * 32);
#pragma unroll
    for (int d = 0; d < NS - 1; ++d)
        if (t0 + d < t1) load(st[d], (t0 + d) * 32, dead_of((t0 + d) * 32));
    for (int64_t tt = t0; tt < t1; tt += NS) {
#pragma unroll
        for (int d = 0; d < NS; ++d) {
            if (tt + d < t1) {
                const unsigned dcur = dnext;
                dnext = dead_of((tt + d + NS) * 32);
                if (tt + d + NS - 1 < t1) load(st[(d + NS - 1) % NS], (tt + d + NS - 1) * 32, dcur);
                __builtin_amdgcn_sched_barrier(0);
                tile(st[d], (tt + d) * 32);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// gae_gcn2_bwd_dense: per-block partial sums of
//     dW2 = G^T Y1 [J2, J1],  db2 = colsum(dZ) [J2],  dY1 = (G W2) (.) act1'(Y1),  dW1 = dY1^T M1 [J1, K1],
//     db1 = colsum(dY1) [J1]                                            (J2, J1, K1 <= 32)
// in ONE pass over G, dZ [n, J2], Y1 [n, J1] and M1 [n, K1].  Per 32-row tile:
//   dH = G W2  : A = G rows (lane = row; k = j2), B = W2 -> accumulator lane (j, h), register r = dH[rho(r, h)][j];
//   that register file, gated by Y1 > 0, is directly the A operand of dW1 += dY1^T M1 (k = the tile's rows, step r
//   <-> row rho(r, h)); its B operand is M1 read column-per-lane in the same row order; dW2 += G^T Y1 likewise with
//   G and Y1 read column-per-lane.  No LDS in the row loop; the 4 waves of a block meet in LDS once, in fixed order.
// partial layout per block: [dW1 J1 x K1 | db1 J1 | dW2 J2 x J1 | db2 J2]  (gae_gcn2_bwd_dense_layout)
// ---------------------------------------------------------------------------------------------------------------
// RECOMP: Y1 is not read but RECOMPUTED from the tile of M1 that the pass reads anyway, Y1 = act1(M1 W1^T + b1), with
// the forward's own products in the forward's own order (gae_linear2_fwd: same k sequence, same fma chain -> the same
// bits): the forward then never stores Y1 (2 GiB per step on R-MAT s24) and this pass never reads it.
template <int KB2, bool RELU, bool RECOMP, int BF = 0>     // BF: 0 exact fp32 MFMAs, 1 three bf16 pieces (six pairs), 2 two pieces (three pairs)
__global__ __launch_bounds__(256, (RECOMP && BF) ? 3 : 2) void gcn2_bwd_rows_kernel(const float *__restrict__ G, int64_t ldg,
                                                            const float *__restrict__ dZ, int64_t lddz,
                                                            const float *__restrict__ Y1, int64_t ldy1,
                                                            const float *__restrict__ M1, int64_t ldm1,
                                                            const float *__restrict__ W2, int64_t ldw2, int64_t n,
                                                            int K1, int J1, int J2, int64_t tiles_per_wave,
                                                            float *__restrict__ partial, int64_t stride,
                                                            const float *__restrict__ W1, int64_t ldw1,
                                                            const float *__restrict__ b1,
                                                            const uint8_t *__restrict__ m1_dead,
                                                            const uint8_t *__restrict__ g_dead,
                                                            const int32_t *__restrict__ rows)
{
    // rows != NULL (list mode, RECOMP only): n counts LIST ENTRIES, entry p stands for row rows[p] of G / dZ / M1, and
    // g_dead is indexed by ENTRY (m1_dead is not read: listed rows have an M1 row).  The rows left out are zero rows of
    // M1; their share of the gradients comes from gcn2_dead_sums_kernel / gcn2_dead_terms_kernel.
    // RECOMP: the tile's rows of M1 and G arrive once, rows in the lanes (16-byte loads); the column layout the two
    // weight-gradient products need (lane = column, registers = rows) comes out of a per-wave LDS slab instead of 32 more
    // 4-byte loads per tile -- those kept the CU's address unit busier than its matrix pipes.  The slabs share their
    // memory with the buffers of the final reduction.
    constexpr int LDS_LD = 36;                          // floats per slab row: 32 + 4 (bank spread, 16-byte aligned)
    constexpr int SLAB = 2 * 32 * LDS_LD;               // M1 tile + G tile of one wave
    constexpr int RED = 2 * 34 * 64;
    __shared__ __attribute__((aligned(16))) float lds_all[(RECOMP && 4 * SLAB > RED) ? 4 * SLAB : RED];
    float (*red)[34 * 64] = reinterpret_cast<float (*)[34 * 64]>(lds_all);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5;
    // W2 as the B operand of dH = G W2: w2[kb][q] = W2[j2 = 8 kb + 4 h + q][j = i]
    float w2[KB2][4];
    {
        const int jc = i < J1 ? i : J1 - 1;
#pragma unroll
        for (int kb = 0; kb < KB2; ++kb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j2 = kb * 8 + 4 * h + q;
                const float v = W2[int64_t(j2 < J2 ? j2 : J2 - 1) * ldw2 + jc];
                w2[kb][q] = (j2 < J2 && i < J1) ? v : 0.f;
            }
    }
    // (RECOMP) W1 as the B operand of Y1 = M1 W1^T: w1[kb][q] = W1[j = i][k = 8 kb + 4 h + q]; bias of column j = i
    float w1[RECOMP ? 4 : 1][4], b1v = 0.f;
    if (RECOMP) {
        const int jc = i < J1 ? i : J1 - 1;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = kb * 8 + 4 * h + q;
                const float v = W1[int64_t(jc) * ldw1 + (k < K1 ? k : K1 - 1)];
                w1[kb][q] = (k < K1 && i < J1) ? v : 0.f;
            }
        const float b = b1 ? b1[jc] : 0.f;
        b1v = (b1 && i < J1) ? b : 0.f;
    }
    const int K4 = (K1 + 3) & ~3;
    f32x16 accW1, accW2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { accW1[r] = 0.f; accW2[r] = 0.f; }
    float sdb1 = 0.f;                  // column j = i of dY1, rows rho(., h) of every tile
    float sz[KB2][4];                  // columns 8 kb + 4 h + q of dZ, row i of every tile
#pragma unroll
    for (int kb = 0; kb < KB2; ++kb)
#pragma unroll
        for (int q = 0; q < 4; ++q) sz[kb][q] = 0.f;

    const int J24 = (J2 + 3) & ~3;
    const int jy = i < J1 ? i : J1 - 1, km = i < K1 ? i : K1 - 1, jg = i < J2 ? i : J2 - 1;
    struct Stage { float g[KB2][4], z[KB2][4], y[16], m[RECOMP ? 1 : 16], gc[RECOMP ? 1 : 16]; };
    // (RECOMP) m1_dead / g_dead: rows of M1 / G that ARE zero and were never written (GAE_SPMM_SKIP_ROWS): not read.
    // dead_of: bit 0 = M1 row dead, bit 1 = G row dead; requested one tile ahead of the tile's loads.
    // bits 0..29: the row id, bit 30 = M1 row dead, bit 31 = G row dead
    auto dead_of = [&](int64_t row0) -> unsigned {
        const int64_t pos = row0 + i < n ? row0 + i : n - 1;
        unsigned d = (RECOMP && rows != nullptr) ? unsigned(rows[pos]) : unsigned(pos);
        if (RECOMP && rows == nullptr && m1_dead != nullptr) d |= m1_dead[pos] ? 0x40000000u : 0u;
        if (RECOMP && g_dead != nullptr) d |= g_dead[pos] ? 0x80000000u : 0u;
        return d;
    };
    auto load = [&](Stage &st, int64_t row0, unsigned dead = 0xffffffffu) {
        if (dead == 0xffffffffu) dead = unsigned(row0 + i < n ? row0 + i : n - 1);      // (callers without masks / lists)
        const int64_t row = int64_t(dead & 0x3fffffffu);
        const bool mdead = (dead & 0x40000000u) != 0, gdead = (dead & 0x80000000u) != 0;
#pragma unroll
        for (int kb = 0; kb < KB2; ++kb) {
            const int k = kb * 8 + 4 * h, kc = k <= J24 - 4 ? k : J24 - 4;
            float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!gdead) g4 = *reinterpret_cast<const float4 *>(G + row * ldg + kc);
            const float4 z4 = *reinterpret_cast<const float4 *>(dZ + row * lddz + kc);
            st.g[kb][0] = g4.x; st.g[kb][1] = g4.y; st.g[kb][2] = g4.z; st.g[kb][3] = g4.w;
            st.z[kb][0] = z4.x; st.z[kb][1] = z4.y; st.z[kb][2] = z4.z; st.z[kb][3] = z4.w;
        }
        if (RECOMP) {                     // y[4 kb + q] = M1[row i][8 kb + 4 h + q]: the rows of the tile in the lanes
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const int k = kb * 8 + 4 * h;
                float4 m4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!mdead) m4 = *reinterpret_cast<const float4 *>(M1 + row * ldm1 + (k <= K4 - 4 ? k : K4 - 4));
                st.y[4 * kb] = m4.x; st.y[4 * kb + 1] = m4.y; st.y[4 * kb + 2] = m4.z; st.y[4 * kb + 3] = m4.w;
            }
        }
        if (!RECOMP) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t rr = row0 + rho(r, h) < n ? row0 + rho(r, h) : n - 1;
                st.y[r] = Y1[rr * ldy1 + jy];
                st.m[r] = M1[rr * ldm1 + km];
                st.gc[r] = G[rr * ldg + jg];
            }
        }
    };
    auto tile = [&](const Stage &st, int64_t row0) {
        const bool rv = row0 + i < n;
        float mcol[16], gcol[16];                        // column layout: M1[rho(r, h)][k = i], G[rho(r, h)][j2 = i]
        if (!RECOMP) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { mcol[r] = st.m[r]; gcol[r] = st.gc[r]; }
        }
        if (RECOMP) {
            float *lm = lds_all + wave * SLAB, *lg = lm + 32 * LDS_LD;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
                *reinterpret_cast<float4 *>(lm + i * LDS_LD + 8 * kb + 4 * h) =
                    make_float4(st.y[4 * kb], st.y[4 * kb + 1], st.y[4 * kb + 2], st.y[4 * kb + 3]);
#pragma unroll
            for (int kb = 0; kb < KB2; ++kb)
                *reinterpret_cast<float4 *>(lg + i * LDS_LD + 8 * kb + 4 * h) =
                    make_float4(st.g[kb][0], st.g[kb][1], st.g[kb][2], st.g[kb][3]);
            __builtin_amdgcn_wave_barrier();             // (one wave writes and reads its own slab: LDS operations of a
#pragma unroll                                           //  wave execute in order, no block barrier)
            for (int r = 0; r < 16; ++r) {
                mcol[r] = lm[rho(r, h) * LDS_LD + km];
                gcol[r] = lg[rho(r, h) * LDS_LD + jg];
            }
            __builtin_amdgcn_wave_barrier();
        }
        f32x16 dh;
#pragma unroll
        for (int r = 0; r < 16; ++r) dh[r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < KB2; ++kb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool kv = kb * 8 + 4 * h + q < J2;
                dh = __builtin_amdgcn_mfma_f32_32x32x2f32((rv && kv) ? st.g[kb][q] : 0.f, w2[kb][q], dh, 0, 0, 0);
                sz[kb][q] += (rv && kv) ? st.z[kb][q] : 0.f;
            }
        f32x16 yv;
        if (RECOMP) {
#pragma unroll
            for (int r = 0; r < 16; ++r) yv[r] = 0.f;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool kv = kb * 8 + 4 * h + q < K1;
                    yv = __builtin_amdgcn_mfma_f32_32x32x2f32((rv && kv) ? st.y[4 * kb + q] : 0.f, w1[kb][q], yv, 0, 0, 0);
                }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float y = yv[r] + b1v;
                if (RELU) y = fmaxf(y, 0.f);
                yv[r] = y;
            }
        }
        // BF: the two weight-gradient products of the tile (K = its 32 rows) as 2 x 3 bf16 MFMAs of K = 16 each instead of
        // 2 x 16 fp32 MFMAs of K = 2 (1024 -> 192 matrix-pipe cycles per product and tile): the pass is then bound by
        // its operand stream, not by the matrix pipe.  A lane's 8 K-steps of MFMA t are its rows rho(8 t + 0..7, h) in
        // BOTH operands, so no data moves between lanes.
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float dyv[8], yy[8], mv[8], gv[8];
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) {
                const int r = 8 * t + r8;
                const bool rr = row0 + rho(r, h) < n;
                const float y = (rr && i < J1) ? (RECOMP ? yv[r] : st.y[r]) : 0.f;
                float dy = rr ? dh[r] : 0.f;                       // dH[rho(r, h)][j = i]
                if (RELU) dy = y > 0.f ? dy : 0.f;
                sdb1 += dy;
                const float m = (rr && i < K1) ? mcol[r] : 0.f, g = (rr && i < J2) ? gcol[r] : 0.f;
                if (BF) {
                    dyv[r8] = dy; yy[r8] = y; mv[r8] = m; gv[r8] = g;
                } else {
                    accW1 = __builtin_amdgcn_mfma_f32_32x32x2f32(dy, m, accW1, 0, 0, 0);
                    accW2 = __builtin_amdgcn_mfma_f32_32x32x2f32(g, y, accW2, 0, 0, 0);
                }
            }
            if (BF == 2) {
                s16x8 ah, al, bh, bl;
                split8(dyv, ah, al); split8(mv, bh, bl);
                accW1 = mfma_split3(ah, al, bh, bl, accW1);
                split8(gv, ah, al); split8(yy, bh, bl);
                accW2 = mfma_split3(ah, al, bh, bl, accW2);
            } else if (BF) {
                s16x8 ah, am, al, bh, bm, bl;
                split8_3(dyv, ah, am, al); split8_3(mv, bh, bm, bl);
                accW1 = mfma_split6(ah, am, al, bh, bm, bl, accW1);
                split8_3(gv, ah, am, al); split8_3(yy, bh, bm, bl);
                accW2 = mfma_split6(ah, am, al, bh, bm, bl, accW2);
            }
        }
    };
    const int64_t n_tiles = (n + 31) / 32;
    const int64_t t0 = (int64_t(blockIdx.x) * 4 + wave) * tiles_per_wave;
    const int64_t t1 = t0 + tiles_per_wave < n_tiles ? t0 + tiles_per_wave : n_tiles;
    if (t0 < t1) {
        if (RECOMP) {
            // one stage: with the recomputation the kernel holds 4 accumulator tiles; a second stage of operands would
            // leave one wave per SIMD (252 VGPRs) and nothing to overlap the loads with (measured: 2.28 ms vs 1.36 ms
            // for the form that reads Y1).  Two waves per SIMD cover each other's loads instead.
            Stage s0;
            unsigned dn = dead_of(t0 * 32);
            for (int64_t tt = t0; tt < t1; ++tt) {
                const unsigned dc = dn;
                dn = dead_of((tt + 1) * 32);             // (next tile's mask: in flight under this tile)
                load(s0, tt * 32, dc);
                __builtin_amdgcn_sched_barrier(0);
                tile(s0, tt * 32);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            Stage s0, s1;
            load(s0, t0 * 32);
            for (int64_t tt = t0; tt < t1; tt += 2) {
                if (tt + 1 < t1) load(s1, (tt + 1) * 32);
                __builtin_amdgcn_sched_barrier(0);
                tile(s0, tt * 32);
                __builtin_amdgcn_sched_barrier(0);
                if (tt + 1 >= t1) break;
                if (tt + 2 < t1) load(s0, (tt + 2) * 32);
                __builtin_amdgcn_sched_barrier(0);
                tile(s1, (tt + 1) * 32);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (RECOMP) __syncthreads();       // the slabs become the reduction buffers
    // ---- column sums: db1 over the two halves; db2 over the 32 rows-in-lanes (fixed shuffle tree)
    sdb1 += __shfl_xor(sdb1, 32, 64);
#pragma unroll
    for (int kb = 0; kb < KB2; ++kb)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) sz[kb][q] += __shfl_xor(sz[kb][q], off, 64);
    // ---- the 4 waves meet in LDS, fixed tree order: (w) += (w + 2), then (0) += (1)
    float extra[2] = {sdb1, 0.f};     // per-lane scalars parked behind the accumulators: db1, db2 (below)
    // db2 value of column j2 = 8 kb + 4 h + q sits (equal) in all lanes of half h: lane (i, h) keeps column i if it owns it
#pragma unroll
    for (int kb = 0; kb < KB2; ++kb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j2 = kb * 8 + 4 * h + q;
            if (j2 == i) extra[1] = sz[kb][q];            // lane (i, h) with i in the half's column set
        }
    // (a column j2 with ((j2 >> 2) & 1) == h is held by lane (j2, h); the other half's lane (j2, 1 - h) holds 0)
#pragma unroll
    for (int half = 2; half >= 1; half >>= 1) {
        if (wave >= half && wave < 2 * half) {
            float *rp = red[wave - half];
#pragma unroll
            for (int r = 0; r < 16; ++r) { rp[r * 64 + lane] = accW1[r]; rp[(16 + r) * 64 + lane] = accW2[r]; }
            rp[32 * 64 + lane] = extra[0]; rp[33 * 64 + lane] = extra[1];
        }
        __syncthreads();
        if (wave < half) {
            const float *rp = red[wave];
#pragma unroll
            for (int r = 0; r < 16; ++r) { accW1[r] += rp[r * 64 + lane]; accW2[r] += rp[(16 + r) * 64 + lane]; }
            extra[0] += rp[32 * 64 + lane]; extra[1] += rp[33 * 64 + lane];
        }
        if (half > 1) __syncthreads();
    }
    if (wave != 0) return;
    float *pp = partial + int64_t(blockIdx.x) * stride;
    float *pW1 = pp, *pb1 = pp + J1 * K1, *pW2 = pb1 + J1, *pb2 = pW2 + J2 * J1;
    // accW1 register r of lane (n = k, h) = dW1[j = rho(r, h)][k]; accW2 register r of lane (n = j, h) = dW2[j2 = rho(r, h)][j]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int o = rho(r, h);
        if (o < J1 && i < K1) pW1[o * K1 + i] = accW1[r];
        if (o < J2 && i < J1) pW2[o * J1 + i] = accW2[r];
    }
    if (h == 0 && i < J1) pb1[i] = extra[0];
    const float b2 = extra[1] + __shfl_xor(extra[1], 32, 64);      // the owning half + 0
    if (h == 0 && i < J2) pb2[i] = b2;
}

// T[r][:] = act1(b1) W2^T for the rows r with dead[r] != 0: the image of a zero row of A under gae_linear2_fwd -- what
// every row without in-edges shares (list mode of the forward pass).  The vector is computed once per block.
__global__ __launch_bounds__(256) void linear2_fill_dead_kernel(const float *__restrict__ b1, int act1,
                                                                const float *__restrict__ W2, int64_t ldw2, int J1, int J2,
                                                                const uint8_t *__restrict__ dead, int64_t n,
                                                                float *__restrict__ T, int64_t ldt)
{
    __shared__ float t0[32];
    if (threadIdx.x < 32) {
        float acc = 0.f;
        if (int(threadIdx.x) < J2) {
            for (int j = 0; j < J1; ++j) {               // j ascending: one fma chain per output, as everywhere
                float y = b1 ? b1[j] : 0.f;
                if (act1 == GAE_ACT_RELU) y = fmaxf(y, 0.f);
                acc = fmaf(y, W2[int64_t(threadIdx.x) * ldw2 + j], acc);
            }
        }
        t0[threadIdx.x] = acc;
    }
    __syncthreads();
    const int nvec = (J2 + 3) / 4;
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t t = int64_t(blockIdx.x) * 256 + threadIdx.x; t < n * nvec; t += stride) {
        const int64_t row = t / nvec;
        const int c4 = int(t - row * nvec) * 4;
        if (!dead[row]) continue;
        float *tp = T + row * ldt + c4;
        if (c4 + 4 <= J2) *reinterpret_cast<float4 *>(tp) = make_float4(t0[c4], t0[c4 + 1], t0[c4 + 2], t0[c4 + 3]);
        else for (int q = 0; c4 + q < J2; ++q) tp[q] = t0[c4 + q];
    }
}

// Backward of the rows gae_gcn2_bwd_dense does not visit in list mode (M1 row = 0, so H1 row = act1(b1) =: y0):
//   dW1 += 0,  db1 += (s_G W2) (.) act1'(y0),  dW2 += s_G (x) y0,  db2 += s_Z,   s_G = sum of their G rows (rows marked
//   in g_dead are zero rows that were never written: not read), s_Z = sum of their dZ rows.
// Stage 1: fixed grid, each block owns a contiguous range of rows, thread (row slot, 4 columns), tree in LDS -> [2][32].
constexpr int kDeadBlocks = 2048;
__global__ __launch_bounds__(256) void gcn2_dead_sums_kernel(const float *__restrict__ G, int64_t ldg,
                                                             const float *__restrict__ dZ, int64_t lddz, int J2,
                                                             const uint8_t *__restrict__ m1_dead,
                                                             const uint8_t *__restrict__ g_dead, int64_t n,
                                                             float *__restrict__ part /*[blocks][64]*/)
{
    // a byte stream (the dZ rows of most rows of the graph): 4 rows per thread and trip, all requested before the first add
    __shared__ float red[2][32][33];
    const int J24 = (J2 + 3) & ~3, nv = J24 / 4;                 // threads per row
    const int slots = 256 / nv;                                  // rows per sweep of the block
    const int c4 = (int(threadIdx.x) % nv) * 4, slot = int(threadIdx.x) / nv;
    const int64_t per = ((n + gridDim.x - 1) / gridDim.x + 3) / 4 * 4;
    const int64_t r0 = int64_t(blockIdx.x) * per, r1 = r0 + per < n ? r0 + per : n;
    float sg[4] = {0.f, 0.f, 0.f, 0.f}, sz[4] = {0.f, 0.f, 0.f, 0.f};
    constexpr int U = 4;
    for (int64_t rb = r0 + slot; rb < r1; rb += int64_t(U) * slots) {
        float4 z[U], g[U];
        bool zl[U], gl[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = rb + int64_t(u) * slots;
            const bool in = r < r1 && slot < slots;
            const unsigned md = in ? m1_dead[r] : 0u;
            zl[u] = in && md != 0;
            gl[u] = zl[u] && (g_dead == nullptr || g_dead[r] == 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = rb + int64_t(u) * slots;
            z[u] = make_float4(0.f, 0.f, 0.f, 0.f); g[u] = z[u];
            if (zl[u]) z[u] = *reinterpret_cast<const float4 *>(dZ + r * lddz + c4);
            if (gl[u]) g[u] = *reinterpret_cast<const float4 *>(G + r * ldg + c4);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {                             // row order per thread: deterministic
            sz[0] += z[u].x; sz[1] += z[u].y; sz[2] += z[u].z; sz[3] += z[u].w;
            sg[0] += g[u].x; sg[1] += g[u].y; sg[2] += g[u].z; sg[3] += g[u].w;
        }
    }
    // columns c4 .. c4 + 3 of row slot `slot`: the slots meet in LDS, added in slot order
    for (int k = threadIdx.x; k < 2 * 32 * 33; k += 256) (&red[0][0][0])[k] = 0.f;
    __syncthreads();
    // (slots may exceed 32: fold slot s into LDS column s % 32 in two rounds, low slots first)
    for (int round = 0; round * 32 < slots; ++round) {
        if (slot / 32 == round) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { red[0][c4 + q][slot % 32] += sg[q]; red[1][c4 + q][slot % 32] += sz[q]; }
        }
        __syncthreads();
    }
    if (threadIdx.x < 64) {
        const int w = threadIdx.x >> 5, c = threadIdx.x & 31;
        float t = 0.f;
        for (int k = 0; k < 32; ++k) t += red[w][c][k];
        part[int64_t(blockIdx.x) * 64 + threadIdx.x] = (c < J2) ? t : 0.f;
    }
}

// Stage 2 (one block): the block sums in block order, then the four gradient terms as ONE more partial block of the
// pass's partial list ([dW1 J1 x K1 | db1 J1 | dW2 J2 x J1 | db2 J2], gae_gcn2_bwd_dense's layout)
__global__ __launch_bounds__(256) void gcn2_dead_terms_kernel(const float *__restrict__ part, int n_blocks,
                                                              const float *__restrict__ b1, int act1,
                                                              const float *__restrict__ W2, int64_t ldw2, int K1, int J1,
                                                              int J2, float *__restrict__ out)
{
    __shared__ float s4[4][64], s[64], y0[32];
    {   // 4 threads per output, each adds every 4th block's partial (8 loads in flight), then the four meet in order
        const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
        float t = 0.f;
        int b = q;
        for (; b + 28 < n_blocks; b += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[int64_t(b + 4 * u) * 64 + o];
#pragma unroll
            for (int u = 0; u < 8; ++u) t += v[u];
        }
        for (; b < n_blocks; b += 4) t += part[int64_t(b) * 64 + o];
        s4[q][o] = t;
    }
    __syncthreads();
    if (threadIdx.x < 64) s[threadIdx.x] = (s4[0][threadIdx.x] + s4[1][threadIdx.x]) + (s4[2][threadIdx.x] + s4[3][threadIdx.x]);
    if (threadIdx.x >= 64 && threadIdx.x < 96) {
        const int j = threadIdx.x - 64;
        float y = (b1 && j < J1) ? b1[j] : 0.f;
        if (act1 == GAE_ACT_RELU) y = fmaxf(y, 0.f);
        y0[j] = j < J1 ? y : 0.f;
    }
    __syncthreads();
    float *pW1 = out, *pb1 = out + J1 * K1, *pW2 = pb1 + J1, *pb2 = pW2 + J2 * J1;
    for (int e = threadIdx.x; e < J1 * K1; e += 256) pW1[e] = 0.f;
    if (int(threadIdx.x) < J1) {
        const int j = threadIdx.x;
        float v = 0.f;
        for (int j2 = 0; j2 < J2; ++j2) v = fmaf(s[j2], W2[int64_t(j2) * ldw2 + j], v);
        if (act1 == GAE_ACT_RELU && !(y0[j] > 0.f)) v = 0.f;
        pb1[j] = v;
    }
    for (int e = threadIdx.x; e < J2 * J1; e += 256) pW2[e] = s[e / J1] * y0[e % J1];
    if (int(threadIdx.x) < J2) pb2[threadIdx.x] = s[32 + threadIdx.x];
}

inline int64_t tall_tiles_per_wave(int64_t n, int64_t max_blocks)
{
    const int64_t n_tiles = (n + 31) / 32;
    int64_t tpw = (n_tiles + max_blocks * 4 - 1) / (max_blocks * 4);
    return tpw < 1 ? 1 : tpw;
}

} // namespace

extern "C" int gae_linear2_fwd(const float *A, int64_t lda, int64_t n, int64_t f_in, const float *W1, int64_t ldw1,
                               const float *b1, int64_t f_mid, int act1, const float *W2, int64_t ldw2, int64_t f_out,
                               float *Y1, int64_t ldy1, float *T, int64_t ldt, const uint8_t *a_dead,
                               const int32_t *rows, int64_t n_listed, void *stream)
{
    GAE_REQUIRE(rows == nullptr || (n_listed >= 0 && n_listed <= n && a_dead == nullptr), GAE_E_SIZE,
                "gae_linear2_fwd: a row list has 0 .. n entries and replaces a_dead");
    GAE_REQUIRE(n < (int64_t(1) << 31), GAE_E_SIZE, "gae_linear2_fwd: n too large");
    const int64_t n_all = n;
    if (rows != nullptr) n = n_listed;            // the kernel counts list entries
    (void)n_all;
    GAE_REQUIRE(n >= 0 && f_in >= 1 && f_mid >= 1 && f_out >= 1, GAE_E_SIZE, "gae_linear2_fwd: bad size");
    GAE_REQUIRE(f_in <= 64 && f_mid <= 32 && f_out <= 32, GAE_E_RANGE,
                "gae_linear2_fwd: needs f_in <= 64, f_mid <= 32, f_out <= 32 (use two gae_linear_fwd calls)");
    GAE_REQUIRE(act1 == GAE_ACT_IDENTITY || act1 == GAE_ACT_RELU, GAE_E_RANGE, "gae_linear2_fwd: act %d", act1);
    if (n == 0) return GAE_OK;
    GAE_REQUIRE(A && W1 && W2 && T, GAE_E_NULL, "gae_linear2_fwd: NULL pointer");
    GAE_REQUIRE(lda >= (f_in + 3) / 4 * 4 && lda % 4 == 0 && gae::aligned16(A), GAE_E_ALIGN,
                "gae_linear2_fwd: rows of A must be whole 16-byte vectors");
    GAE_REQUIRE(ldw1 >= f_in && ldw2 >= f_mid, GAE_E_SIZE, "gae_linear2_fwd: weight leading dimension too small");
    GAE_REQUIRE((!Y1 || (ldy1 >= f_mid && ldy1 % 4 == 0 && gae::aligned16(Y1))) && ldt >= f_out && ldt % 4 == 0 &&
                    gae::aligned16(T), GAE_E_ALIGN, "gae_linear2_fwd: rows of Y1 / T must be whole 16-byte vectors");
    hipStream_t s = gae::as_stream(stream);
    const int64_t tpw = tall_tiles_per_wave(n, 4096);
    const int64_t n_tiles = (n + 31) / 32;
    const dim3 grid(unsigned((n_tiles + 4 * tpw - 1) / (4 * tpw)));
    const int kb = int((f_in + 7) / 8);
#define GAE_L2F(KBV)                                                                                                     \
    hipLaunchKernelGGL((linear2_rows_kernel<KBV>), grid, dim3(256), 0, s, A, lda, W1, ldw1, b1, act1, W2, ldw2, Y1, ldy1, \
                       T, ldt, n, int(f_in), int(f_mid), int(f_out), int(tpw), a_dead, rows)
    switch (kb) {
    case 1: GAE_L2F(1); break; case 2: GAE_L2F(2); break; case 3: GAE_L2F(3); break; case 4: GAE_L2F(4); break;
    case 5: GAE_L2F(5); break; case 6: GAE_L2F(6); break; case 7: GAE_L2F(7); break; default: GAE_L2F(8); break;
    }
#undef GAE_L2F
    GAE_CHECK_LAUNCH("linear2_rows_kernel");
    return GAE_OK;
}

extern "C" int gae_linear2_fill_dead(const float *b1, int64_t f_mid, int act1, const float *W2, int64_t ldw2, int64_t f_out,
                                     const uint8_t *dead, int64_t n, float *T, int64_t ldt, void *stream)
{
    GAE_REQUIRE(n >= 0 && f_mid >= 1 && f_mid <= 32 && f_out >= 1 && f_out <= 32, GAE_E_SIZE, "gae_linear2_fill_dead: bad size");
    GAE_REQUIRE(act1 == GAE_ACT_IDENTITY || act1 == GAE_ACT_RELU, GAE_E_RANGE, "gae_linear2_fill_dead: act %d", act1);
    if (n == 0) return GAE_OK;
    GAE_REQUIRE(W2 && dead && T, GAE_E_NULL, "gae_linear2_fill_dead: NULL pointer");
    GAE_REQUIRE(ldw2 >= f_mid && ldt >= f_out && ldt % 4 == 0 && gae::aligned16(T), GAE_E_ALIGN,
                "gae_linear2_fill_dead: rows of T must be whole 16-byte vectors");
    const int64_t want = (n * ((f_out + 3) / 4) + 255) / 256;
    hipLaunchKernelGGL(linear2_fill_dead_kernel, dim3(unsigned(want < 8192 ? want : 8192)), dim3(256), 0,
                       gae::as_stream(stream), b1, act1, W2, ldw2, int(f_mid), int(f_out), dead, n, T, ldt);
    GAE_CHECK_LAUNCH("linear2_fill_dead_kernel");
    return GAE_OK;
}

// layout_out = {n_partials (blocks), floats per partial, offset of db1, offset of dW2, offset of db2} (dW1 at 0)
static void gcn2_layout(int64_t n, int64_t f_in, int64_t f_mid, int64_t f_out, int64_t *lay, int64_t *tpw_out)
{
    const int64_t tpw = tall_tiles_per_wave(n, 1024);
    const int64_t n_tiles = (n + 31) / 32;
    lay[0] = (n_tiles + 4 * tpw - 1) / (4 * tpw);
    lay[2] = f_mid * f_in;
    lay[3] = lay[2] + f_mid;
    lay[4] = lay[3] + f_out * f_mid;
    lay[1] = (lay[4] + f_out + 3) / 4 * 4;
    if (tpw_out) *tpw_out = tpw;
}

extern "C" int64_t gae_gcn2_bwd_dense_workspace_bytes(int64_t n, int64_t f_in, int64_t f_mid, int64_t f_out)
{
    if (n < 0 || f_in < 1 || f_mid < 1 || f_out < 1 || f_in > 32 || f_mid > 32 || f_out > 32) return GAE_E_SIZE;
    int64_t lay[5];
    gcn2_layout(n > 0 ? n : 1, f_in, f_mid, f_out, lay, nullptr);
    // list mode lays the partials out for n_listed <= n rows, and the block count is NOT monotone in the row count
    // once tiles-per-wave exceeds 1 (n = 300000 -> 782 blocks, 131072 -> 1024): size for the most any m <= n can take
    const int64_t n_tiles = ((n > 0 ? n : 1) + 31) / 32;
    const int64_t lay0_max = (n_tiles + 3) / 4 < 1024 ? (n_tiles + 3) / 4 : 1024;
    if (lay[0] < lay0_max) lay[0] = lay0_max;
    return (lay[0] + 1) * lay[1] * 4 + kDeadBlocks * 64 * 4 + 256;       // (+ list mode: one more partial, the dead-row sums)
}

extern "C" int gae_gcn2_bwd_dense(const float *G, int64_t ldg, const float *dZ, int64_t lddz, const float *Y1, int64_t ldy1,
                                  int act1, const float *M1, int64_t ldm1, const float *W2, int64_t ldw2, int64_t n,
                                  int64_t f_in, int64_t f_mid, int64_t f_out, float *dW1, float *db1, float *dW2,
                                  float *db2, void *workspace, int64_t workspace_bytes, int64_t *layout_out,
                                  const float *W1, int64_t ldw1, const float *b1, const uint8_t *m1_dead,
                                  const uint8_t *g_dead, const int32_t *rows, int64_t n_listed,
                                  const uint8_t *g_dead_listed, void *stream)
{
    GAE_REQUIRE(n >= 1 && f_in >= 1 && f_mid >= 1 && f_out >= 1, GAE_E_SIZE, "gae_gcn2_bwd_dense: bad size");
    GAE_REQUIRE(Y1 == nullptr || (m1_dead == nullptr && g_dead == nullptr && rows == nullptr), GAE_E_RANGE,
                "gae_gcn2_bwd_dense: dead-row masks / row lists belong to the recomputing form (Y1 == NULL)");
    GAE_REQUIRE(rows == nullptr || (m1_dead != nullptr && n_listed >= 0 && n_listed <= n), GAE_E_RANGE,
                "gae_gcn2_bwd_dense: a row list (the rows with an M1 row) comes with m1_dead (all the others)");
    GAE_REQUIRE(n < (int64_t(1) << 30), GAE_E_SIZE, "gae_gcn2_bwd_dense: n too large");
    GAE_REQUIRE(f_in <= 32 && f_mid <= 32 && f_out <= 32, GAE_E_RANGE, "gae_gcn2_bwd_dense: widths above 32");
    GAE_REQUIRE(act1 == GAE_ACT_IDENTITY || act1 == GAE_ACT_RELU, GAE_E_RANGE, "gae_gcn2_bwd_dense: act %d", act1);
    GAE_REQUIRE(G && dZ && M1 && W2 && workspace, GAE_E_NULL, "gae_gcn2_bwd_dense: NULL pointer");
    const bool recomp = Y1 == nullptr;
    GAE_REQUIRE(!recomp || (W1 && ldw1 >= f_in && ldm1 >= (f_in + 3) / 4 * 4 && ldm1 % 4 == 0 && gae::aligned16(M1)),
                GAE_E_NULL, "gae_gcn2_bwd_dense: without Y1 the pass recomputes it: W1 and rows of M1 of whole 16-byte "
                            "vectors are needed");
    const int64_t j24 = (f_out + 3) / 4 * 4;
    GAE_REQUIRE(ldg >= j24 && ldg % 4 == 0 && gae::aligned16(G) && lddz >= j24 && lddz % 4 == 0 && gae::aligned16(dZ),
                GAE_E_ALIGN, "gae_gcn2_bwd_dense: rows of G / dZ must be whole 16-byte vectors");
    GAE_REQUIRE((recomp || ldy1 >= f_mid) && ldm1 >= f_in && ldw2 >= f_mid, GAE_E_SIZE, "gae_gcn2_bwd_dense: leading dimension too small");
    GAE_REQUIRE(workspace_bytes >= gae_gcn2_bwd_dense_workspace_bytes(n, f_in, f_mid, f_out) && gae::aligned16(workspace),
                GAE_E_WORKSPACE, "gae_gcn2_bwd_dense: workspace too small or misaligned");
    int64_t lay[5], tpw = 1;
    const bool listed = rows != nullptr;
    const int64_t n_all = n;
    const uint8_t *g_dead_rows = g_dead;  // (row-indexed: the dead-row sums read it)
    if (listed) {                         // the main kernel walks the list; its masks are indexed by list entry
        n = n_listed > 0 ? n_listed : 1;
        g_dead = g_dead_listed;
    }
    gcn2_layout(n, f_in, f_mid, f_out, lay, &tpw);
    GAE_REQUIRE(workspace_bytes >= (lay[0] + 1) * lay[1] * 4 + kDeadBlocks * 64 * 4 + 256, GAE_E_WORKSPACE,
                "gae_gcn2_bwd_dense: workspace too small for the listed rows' partials");
    hipStream_t s = gae::as_stream(stream);
    float *partial = static_cast<float *>(workspace);
    if (listed && n_listed == 0) { lay[0] = 0; }
    const dim3 grid{unsigned(lay[0] > 0 ? lay[0] : 1)};
    const int kb2 = int((f_out + 7) / 8);
    int64_t bfk = 1;
    gae_tuning_get("atb_bf16", &bfk);       // the library's switch for weight-gradient products on the bf16 matrix pipe
#define GAE_G2L(KBV, RL, RC, BFV)                                                                                        \
    hipLaunchKernelGGL((gcn2_bwd_rows_kernel<KBV, RL, RC, BFV>), grid, dim3(256), 0, s, G, ldg, dZ, lddz, Y1, ldy1, M1,   \
                       ldm1, W2, ldw2, n, int(f_in), int(f_mid), int(f_out), tpw, partial, lay[1], W1, ldw1, b1, m1_dead, g_dead, rows)
#define GAE_G2B(KBV, RL)                                                                                                 \
    do {                                                                                                                 \
        if (recomp) { if (bfk == 2) GAE_G2L(KBV, RL, true, 2); else if (bfk) GAE_G2L(KBV, RL, true, 1); else GAE_G2L(KBV, RL, true, 0); } \
        else { if (bfk == 2) GAE_G2L(KBV, RL, false, 2); else if (bfk) GAE_G2L(KBV, RL, false, 1); else GAE_G2L(KBV, RL, false, 0); }    \
    } while (0)
#define GAE_G2K(RL)                                                                                                      \
    do { if (kb2 == 1) GAE_G2B(1, RL); else if (kb2 == 2) GAE_G2B(2, RL); else if (kb2 == 3) GAE_G2B(3, RL); else GAE_G2B(4, RL); } while (0)
    if (lay[0] > 0) { if (act1 == GAE_ACT_RELU) GAE_G2K(true); else GAE_G2K(false); }
#undef GAE_G2K
#undef GAE_G2B
#undef GAE_G2L
    GAE_CHECK_LAUNCH("gcn2_bwd_rows_kernel");
    if (listed) {
        // the rows that are not listed: column sums of their G / dZ rows, turned into one more partial of the list
        float *sums = partial + (lay[0] + 1) * lay[1];
        hipLaunchKernelGGL(gcn2_dead_sums_kernel, dim3(kDeadBlocks), dim3(256), 0, s, G, ldg, dZ, lddz, int(f_out), m1_dead,
                           g_dead_rows, n_all, sums);
        GAE_CHECK_LAUNCH("gcn2_dead_sums_kernel");
        hipLaunchKernelGGL(gcn2_dead_terms_kernel, dim3(1), dim3(256), 0, s, sums, kDeadBlocks, b1, act1, W2, ldw2,
                           int(f_in), int(f_mid), int(f_out), partial + lay[0] * lay[1]);
        GAE_CHECK_LAUNCH("gcn2_dead_terms_kernel");
        lay[0] += 1;
    }
    if (layout_out) {
        for (int k = 0; k < 5; ++k) layout_out[k] = lay[k];
        return GAE_OK;                   // partials only: the caller (gae_adam_step's deferred reduction) adds them
    }
    using gae::PartialList;
    PartialList a{partial, dW1, f_mid * f_in, lay[0], lay[1], f_mid * f_in, f_mid * f_in, f_mid * f_in};
    PartialList b{partial + lay[2], db1, f_mid, lay[0], lay[1], f_mid, f_mid, f_mid};
    int rc = gae::launch_partials_reduce(a, b, s);
    if (rc) return rc;
    PartialList c{partial + lay[3], dW2, f_out * f_mid, lay[0], lay[1], f_out * f_mid, f_out * f_mid, f_out * f_mid};
    PartialList d{partial + lay[4], db2, f_out, lay[0], lay[1], f_out, f_out, f_out};
    return gae::launch_partials_reduce(c, d, s);
}
