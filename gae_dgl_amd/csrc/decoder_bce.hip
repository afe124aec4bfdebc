// K7+K8+K9 fused: inner-product decoder + weighted BCE-with-logits (mean) and
// its gradient, without ever materialising the N x N logits or labels.
//
// Replaces, for training, the chain of gae_dgl/train_inductive.py:44-51
//     adj = g.adjacency_matrix().to_dense(); pos_weight = (N^2 - sum adj)/sum adj
//     adj_logits = model.forward(g)        (gae_dgl/gae.py:70-71: Zt Zt^T)
//     loss = binary_cross_entropy_with_logits(adj_logits, adj, pos_weight=...)
//     loss.backward()
//
// Math.  x_ij = zt_i . zt_j,  y_ij = #edges j->i  (CSR rows = destination):
//   N^2 loss = sum_ij [(1 - y) x + (1 + (pw - 1) y) softplus(-x)]
//            = sum_ij softplus(x_ij)                                   <- dense, label-free
//            + sum_{edges (i,j)} [-x_ij + (pw - 1) softplus(-x_ij)]    <- sparse
//   dL/dx_ij = [sigmoid(x_ij) + y_ij ((pw - 1) sigmoid(x_ij) - pw)] / N^2
//   dZt = (G + G^T) Zt  ->  dense part 2 sigmoid(X) Zt / N^2 (X symmetric), sparse part over
//                           in-edges (CSR) and out-edges (CSR of A^T).
//
// Dense part, trimmed for the VALU (which co-limits with the fp32 matrix rate):
//   with e = exp(-|x|), t = 1 + e, r = 1/t:
//     softplus(x) = (x + |x|)/2 + ln2 * log2(t)      sum_ij x_ij = (sum_i zt_i).(sum_j zt_j)  analytic
//     sigmoid(x)  = 1/2 + copysign(r - 1/2, x)       sum_j (1/2) zt_j                          analytic
//   so the kernel accumulates only sum|x|, sum log2(t) and O' = sum_j copysign(r - 1/2, x_ij) zt_j:
//   ~8 VALU ops (2 transcendental: v_exp, v_rcp; the logs are taken of products of 16 t's) per logit.  Zero-padded columns contribute exactly log2(2) = 1 to
//   sum log2(t) and 0 to everything else: corrected analytically, no masking in the loop.
//
// Dense kernel: flash-style.  A wave owns RI 16-row subtiles and streams 64-column tiles of Zt through
// double-buffered LDS.  S^T = Zj Zi^T on the matrix cores, either
//   * bf16 x 3 (default): Zt = hi + lo (two bf16), S = hi.hi + hi.lo + lo.hi on v_mfma_f32_16x16x16_bf16
//     (fp32 accumulate; the dropped lo.lo term is 2^-16 relative), which runs on the matrix pipe proper and
//     leaves the fp32 FMA lanes to the VALU work; or
//   * exact fp32 v_mfma_f32_16x16x4_f32 (knob bce_s_bf16 = 0), which shares the fp32 lanes with the VALU.
// O' += P Zj on v_mfma_f32_16x16x4_f32: the S^T accumulator registers are directly the A fragments (lane =
// row i, reg r <-> j = 4 (lane >> 4) + r), so P never leaves registers.
// Reductions are two-stage, ordered, fp64 for the loss: bit-stable run to run.
#include <string.h>

#include "common.h"
#include "bce_finalize.h"

namespace {

using gae::kWave;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int TJ = 64;   // columns staged per iteration
constexpr int PREP_ROWS = 64;    // rows per block of the prepare kernel
// tuning knobs (gae_tuning_set): "bce_ri" 16-row subtiles per wave (rows / block = 64 RI),
// "bce_s_bf16" 1 = bf16x3 S product, 0 = exact fp32 S product
gae::Knob g_bce_ri{2};
gae::Knob g_bce_s_bf16{3};        // S = Zt Zt^T (and P V of the symmetric kernel): 3 = two fp16 pieces per operand where the symmetric kernel runs
                                  // (22 mantissa bits, range-guarded; default), three bf16 pieces elsewhere; 2 = three bf16 pieces
                                  // (24 bits); 1 = two bf16 pieces (16 bits: rounds 1-3); 0 = exact fp32 MFMA
gae::Knob g_bce_grid{2048};       // "bce_grid": target size of the (row block, column split) grid of the full-square kernel
constexpr int kChipCus = 256;             // MI355X: the launch-shape heuristics below are written for this part
gae::Knob g_bce_strip_store{-1};  // "bce_strip_store": -1 auto (non-temporal from 32 k rows on: GBs of strips, 2.93 -> 2.88 ms on a ZINC
                                          // batch; plain below: Pubmed 170 vs 174 us), 0 plain, 1 non-temporal, 2 write-through
gae::Knob g_bce_fold_mirror{1};   // "bce_fold_mirror": 1 = the edge kernel folds the mirror strips (no separate reduction launch)
gae::Knob g_bce_sym_tiles{0};     // "bce_sym_tiles": 64-column tiles per block of the symmetric kernel (0 = auto)
gae::Knob g_bce_sym_grid{16384};  // "bce_sym_grid": target size of the (panel, chunk) grid of the symmetric kernel
                             // (many short blocks even out the triangular work: ZINC batch 3.64 -> 3.35 ms)
gae::Knob g_bce_last_kind{0}; // "bce_last_kind" (telemetry, read with gae_tuning_get): dense kernel of the last loss call on this
                               // process -- 0 none yet, 1 full square, 2 symmetric 128-row panels, 3 symmetric 256-row panels
gae::Knob g_bce_sym_bal{1};   // "bce_sym_bal": balanced schedule of the symmetric kernel (every block the same number of column
                              // tiles, one resident round): 1 = below 65 536 rows, 2 = always, 0 = never (the 2-D (panel, column
                              // chunk) grid of rounds 2-5)
gae::Knob g_bce_sym{1};       // "bce_sym": 1 = symmetric dense kernel for full-square launches with d <= 16
gae::Knob g_bce_pv_bf16{1};   // "bce_pv_bf16": 1 = bf16x3 for O' += P V as well (P split on the fly), 0 = exact fp32

__device__ __forceinline__ void softplus_sigmoid(float x, float &sp, float &sg)
{
    // e = exp(-|x|) in (0, 1]; raw v_exp/v_log/v_rcp: an underflowing e flushes to 0, which is the
    // correctly rounded answer for both outputs (no denormal fix-up code needed)
    const float e = __builtin_amdgcn_exp2f(-1.44269504088896341f * fabsf(x));
    const float t = 1.0f + e;
    const float r = __builtin_amdgcn_rcpf(t);
    sp = fmaf(__builtin_amdgcn_logf(t), 0.69314718055994531f, fmaxf(x, 0.f));
    sg = (x >= 0.f ? 1.0f : e) * r;
}


// The per-logit terms of four logits y_r = x_r log2(e) from ONE reciprocal.  With t_r = 1 + 2^-|y_r| in [1, 2]:
//   sum log2(t) rides in the running product (tP *= t0 t1 t2 t3), sum |y| in tA, and
//   sigmoid(x_r) - 1/2 = copysign(1 / t_r - 1/2, x_r),  1 / t_0 = (R t2 t3) t1 with R = 1 / (t0 t1 t2 t3), ...
// i.e. 1 v_rcp + 2 products + 4 FMAs (the -1/2 rides in them) where the straight form issues 4 v_rcp + 4 adds -- a
// transcendental costs 2.5 issue slots of a v_fma (tools/probes/inst_cost.hip), and the products t0 t1, t2 t3 are
// the first level of the product tree anyway.  Error of 1 / t_r: the reciprocal's 1 ulp + three roundings, ~3e-7.
// SCALAR_TREE: the three products of the tree as plain v_mul_f32 as well (same values).  Measured on one box, whole
// training steps: 128-row panels (RI = 2, Pubmed) 0.2233 -> 0.2218 ms with the compiler's pairing, -> 0.2172 ms with
// scalar products.  256-row panels (RI = 4, ZINC batch of 4096): with the ROLLED column loop 2.952 -> 2.880 ms with the
// compiler's pairing and 2.957 ms with scalar products; the four-logit form freed enough registers to UNROLL that loop
// (250 VGPRs, no spill: 2.853 -> 2.794 ms), and unrolled the scalar products win there as well (234 VGPRs, 2.81 -> 2.72 ms).
// The symmetric kernels therefore all take the scalar form; the full-square kernel (small graphs) follows its RI.
template <bool SCALAR_TREE>
__device__ __forceinline__ f32x4 quad_terms(const f32x4 &y, float &tP, float &tA)
{
    float t[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        t[r] = 1.0f + __builtin_amdgcn_exp2f(-fabsf(y[r]));
        tA += fabsf(y[r]);
    }
    float t01, t23, q;
    if constexpr (SCALAR_TREE) {    // (paired, the compiler spends two v_mov and two v_pk_mul on these three products)
        asm("v_mul_f32 %0, %1, %2" : "=v"(t01) : "v"(t[0]), "v"(t[1]));
        asm("v_mul_f32 %0, %1, %2" : "=v"(t23) : "v"(t[2]), "v"(t[3]));
        asm("v_mul_f32 %0, %1, %2" : "=v"(q) : "v"(t01), "v"(t23));
    } else {
        t01 = t[0] * t[1]; t23 = t[2] * t[3]; q = t01 * t23;
    }
    tP *= q;
    const float R = __builtin_amdgcn_rcpf(q);
    // (scalar VALU instructions on purpose: left to itself the compiler pairs these into v_pk_mul / v_pk_fma, which
    //  cost what two scalar ones cost AND need a v_mov per operand to line the register pairs up -- the saving was gone)
    // (the two products that READ the reciprocal stay with the compiler: a VALU instruction that uses a transcendental's
    //  result needs wait states, which the hazard recogniser does not insert in front of inline assembly -- as inline
    //  assembly they read a stale R in the rolled 256-row-panel loop)
    const float r01 = R * t23, r23 = R * t01;
    float s0, s1, s2, s3;
    asm("v_fma_f32 %0, %1, %2, -0.5" : "=v"(s0) : "v"(r01), "v"(t[1]));
    asm("v_fma_f32 %0, %1, %2, -0.5" : "=v"(s1) : "v"(r01), "v"(t[0]));
    asm("v_fma_f32 %0, %1, %2, -0.5" : "=v"(s2) : "v"(r23), "v"(t[3]));
    asm("v_fma_f32 %0, %1, %2, -0.5" : "=v"(s3) : "v"(r23), "v"(t[2]));
    return f32x4{copysignf(s0, y[0]), copysignf(s1, y[1]), copysignf(s2, y[2]), copysignf(s3, y[3])};
}

// ---------------------------------------------------------------------------
// prepare: Zt[n][DP] = Z (.) mask zero padded to DP = 16 KS columns (clean 16-byte rows, no mask loads or
// feature bounds checks downstream), its bf16 hi / lo split, and per-block fp64 column sums of Zt over all
// rows and over the row window (for the analytic terms).
// ---------------------------------------------------------------------------
// drop_p > 0: the inverted-dropout multipliers of this draw are generated here (the Philox stream of
// gae_dropout_mask: element e = i d + k of the [n, d] mask), applied, and stored to `mask` for the backward pass.
__global__ __launch_bounds__(256) void bce_prepare_kernel(const float *__restrict__ Z, float *__restrict__ mask,
                                                          int64_t ldz, int64_t n, int d, int DP, int64_t row_begin,
                                                          int64_t row_end, float *__restrict__ Zt,
                                                          unsigned short *__restrict__ Zhi,
                                                          unsigned short *__restrict__ Zlo,
                                                          double *__restrict__ colsum_partial /*[grid][2][DP]*/,
                                                          float drop_p, float drop_scale, uint64_t seed,
                                                          uint64_t offset, const uint64_t *__restrict__ draw_dev,
                                                          const int64_t *__restrict__ counts /*{nodes, edges} or NULL*/,
                                                          double *__restrict__ scal /*[3] or NULL*/, double all_pairs)
{
    __shared__ double red[2][256];
    const int tid = threadIdx.x;
    // fixed-capacity batch (gae_decoder_bce_padded): rows >= counts[0] are padding -- their Zt rows are zero (logit 0
    // with everything, no gradient to anybody) and pos_weight / 1 / N^2 / the count of zero-logit pairs come from the
    // device-side counts
    const int64_t n_valid = counts ? counts[0] : n;
    if (counts && blockIdx.x == 0 && tid == 0) {
        const double nv = double(counts[0]), ev = double(counts[1]);
        scal[0] = ev > 0 ? (nv * nv - ev) / ev : 0.0;      // pos_weight (train_inductive.py:46)
        scal[1] = nv > 0 ? 1.0 / (nv * nv) : 0.0;          // mean over the N^2 real pairs
        scal[2] = all_pairs - nv * nv;                     // evaluated pairs with a zero operand: log2(1 + e^0) = 1 each
    }
    const int k = tid % DP, rl = tid / DP, rpp = 256 / DP;   // rows per pass
    const int64_t r0 = int64_t(blockIdx.x) * PREP_ROWS;
    double s_all = 0.0, s_win = 0.0;
    const bool draw = drop_p > 0.f;
    const uint64_t draw_idx = (draw && draw_dev) ? *draw_dev : 0;   // a stream of its own per draw (as gae_dropout_mask)
    for (int rr = rl; rr < PREP_ROWS; rr += rpp) {
        const int64_t i = r0 + rr;
        if (i >= n) break;
        float v = 0.f;
        if (k < d && i >= n_valid) {
            if (draw) mask[i * ldz + k] = 0.f;
        } else if (k < d) {
            v = Z[i * ldz + k];
            if (draw) {
                const int64_t e = i * d + k;
                uint32_t c[4];
                gae::philox4x32_10(offset + uint64_t(e >> 2), draw_idx, seed, c);
                const uint32_t bits = (e & 2) ? ((e & 1) ? c[3] : c[2]) : ((e & 1) ? c[1] : c[0]);
                const float m = gae::dropout_multiplier(bits, drop_p, drop_scale);
                mask[i * ldz + k] = m;
                v *= m;
            } else if (mask) {
                v *= mask[i * ldz + k];
            }
        }
        Zt[i * DP + k] = v;
        const unsigned short hi = gae::f32_to_bf16(v);
        Zhi[i * DP + k] = hi;
        Zlo[i * DP + k] = gae::f32_to_bf16(v - gae::bf16_to_f32(hi));
        s_all += double(v);
        if (i >= row_begin && i < row_end) s_win += double(v);
    }
    red[0][tid] = s_all; red[1][tid] = s_win;
    __syncthreads();
    if (tid < DP) {
        double a = 0.0, w = 0.0;
        for (int q = 0; q < rpp; ++q) { a += red[0][q * DP + tid]; w += red[1][q * DP + tid]; }
        colsum_partial[(int64_t(blockIdx.x) * 2 + 0) * DP + tid] = a;
        colsum_partial[(int64_t(blockIdx.x) * 2 + 1) * DP + tid] = w;
    }
}

// one block: ordered sum of the per-block column sums -> S_all / S_win (double) and S_all as float.
// 256 threads = DP columns x (256 / DP) interleaved block partitions, combined in LDS (`red`, 2 x 256 doubles)
// in fixed order.  Runs as one extra block of the dense launch (nothing in that kernel reads its result).
__device__ __forceinline__ void bce_colsum_block(const double *__restrict__ colsum_partial, int64_t n_blocks, int DP,
                                                 double *__restrict__ S /*[2][DP]*/,
                                                 float *__restrict__ S_all_f /*[DP]*/, double *red /*LDS*/)
{
    const int k = threadIdx.x % DP, q = threadIdx.x / DP, nq = 256 / DP;
    double a = 0.0, w = 0.0;
    for (int64_t b = q; b < n_blocks; b += nq) {
        a += colsum_partial[(b * 2 + 0) * DP + k];
        w += colsum_partial[(b * 2 + 1) * DP + k];
    }
    red[threadIdx.x] = a; red[256 + threadIdx.x] = w;
    __syncthreads();
    if (threadIdx.x < DP) {
        a = 0.0; w = 0.0;
        for (int j = 0; j < nq; ++j) { a += red[j * DP + k]; w += red[256 + j * DP + k]; }
        S[k] = a; S[DP + k] = w;
        S_all_f[k] = float(a);
    }
}

// ---------------------------------------------------------------------------
// dense part.  loss_partial[blk] = {sum |x|, sum log2(1 + exp(-|x|))} over the block's (row, column) window.
// ---------------------------------------------------------------------------
// K = 32 fragments: two K = 16 fragments side by side (lane group g holds k = 4 g + r of either half)
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ s16x8 cat(const s16x4 &a, const s16x4 &b)
{
    return s16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
__device__ __forceinline__ void put_half(s16x8 &v, int h, const s16x4 &x)
{
#pragma unroll
    for (int e = 0; e < 4; ++e) v[4 * h + e] = x[e];
}
__device__ __forceinline__ s16x4 get_half(const s16x8 &v, int h)
{
    return s16x4{v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]};
}
__device__ __forceinline__ f32x4 mfma32(const s16x8 &a, const s16x8 &b, const f32x4 &c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// 8 bytes per lane from LDS, the 16-bit elements exchanged inside every group of 16 lanes (gfx950 ds_read_b64_tr_b16,
// tools/probes/ds_tr_probe.hip): with lane L of a group pointing at chunk L % 4 (4 elements) of row L / 4 of a
// 4 x 16 block, lane l receives column l of the 4 rows -- a K-major LDS tile read straight into MFMA A / B fragments
// (lane = m or n, registers = 4 consecutive k).
__device__ __forceinline__ s16x4 lds_read_tr(const unsigned short *p)
{
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p));
}

// split 4 fp32 values into bf16 hi and bf16 lo = bf16(v - hi): v_cvt_pk_bf16_f32 x4, 5 VALU ops per pair
__device__ __forceinline__ void split_bf16x4(const f32x4 &v, s16x4 &hi, s16x4 &lo)
{
    unsigned h[2], l[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const f32x2 a = {v[2 * q], v[2 * q + 1]};
        const unsigned hu = __builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16x2));
        const f32x2 back = {__uint_as_float(hu << 16), __uint_as_float(hu & 0xffff0000u)};
        h[q] = hu;
        l[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(a - back, bf16x2));
    }
    struct U { unsigned a, b; } uh{h[0], h[1]}, ul{l[0], l[1]};
    hi = __builtin_bit_cast(s16x4, uh);
    lo = __builtin_bit_cast(s16x4, ul);
}

// Three bf16 pieces of fp32 values: v = hi + lo + lo2 with hi = bf16(v), lo = bf16(v - hi), lo2 = bf16(v - hi - lo):
// 24 mantissa bits, i.e. v itself up to its last bit.  The S product of the fused loss uses all pairs of pieces down to
// 2^-24 relative (hi.hi, hi.lo, lo.hi, lo.lo, hi.lo2, lo2.hi): an fp32-grade logit on the bf16 matrix pipe (round 4;
// the two-piece product, 16 mantissa bits per operand, left 8e-5 of the gradient's scale on embeddings whose large
// components cancel -- tools/r04/loss_condition.py).
__device__ __forceinline__ void split_bf16x4_3(const f32x4 &v, s16x4 &hi, s16x4 &lo, s16x4 &lo2)
{
    unsigned h[2], l[2], l2[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const f32x2 a = {v[2 * q], v[2 * q + 1]};
        const unsigned hu = __builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16x2));
        const f32x2 r1 = a - f32x2{__uint_as_float(hu << 16), __uint_as_float(hu & 0xffff0000u)};
        const unsigned lu = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
        const f32x2 r2 = r1 - f32x2{__uint_as_float(lu << 16), __uint_as_float(lu & 0xffff0000u)};
        h[q] = hu; l[q] = lu;
        l2[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
    }
    struct U { unsigned a, b; } uh{h[0], h[1]}, ul{l[0], l[1]}, ul2{l2[0], l2[1]};
    hi = __builtin_bit_cast(s16x4, uh);
    lo = __builtin_bit_cast(s16x4, ul);
    lo2 = __builtin_bit_cast(s16x4, ul2);
}
// third piece of values whose first two pieces are given (the column tiles: Zhi / Zlo come from the prepare step)
__device__ __forceinline__ s16x4 third_piece(const f32x4 &v, const s16x4 &hi, const s16x4 &lo)
{
    unsigned l2[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const f32x2 a = {v[2 * q], v[2 * q + 1]};
        const f32x2 fh = {gae::bf16_to_f32((unsigned short)hi[2 * q]), gae::bf16_to_f32((unsigned short)hi[2 * q + 1])};
        const f32x2 fl = {gae::bf16_to_f32((unsigned short)lo[2 * q]), gae::bf16_to_f32((unsigned short)lo[2 * q + 1])};
        l2[q] = __builtin_bit_cast(unsigned, __builtin_convertvector((a - fh) - fl, bf16x2));
    }
    struct U { unsigned a, b; } u{l2[0], l2[1]};
    return __builtin_bit_cast(s16x4, u);
}

// ---- fp16 pieces (round 4).  v = hi + lo with hi = fp16(v), lo = fp16(v - hi): 22 mantissa bits in TWO pieces (bf16
// needs three for 24), so all four partial products of S cost two K = 32 MFMAs instead of three, and the split is one
// v_cvt_pkrtz per pair and piece plus a mixed-precision subtract.  The price is fp16's range: the pieces are exact to
// 2^-22 |v| only while 2^-14 <= |v| <= 65504 (below: absolute error <= 2^-25, harmless; above: overflow).  The kernels
// that use them therefore report embeddings beyond kF16Max, and the three-piece bf16 form then recomputes the call's
// outputs (the range guard of bce_dense_sym_kernel).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
constexpr float kF16Max = 32768.f;      // largest |Zt| the fp16 pieces are used for (Zt log2(e) must fit as well)
__device__ __forceinline__ f32x4 mfma32h(const s16x8 &a, const s16x8 &b, const f32x4 &c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ f32x4 mfma_k32(const s16x8 &a, const s16x8 &b, const f32x4 &c)
{
    if constexpr (F16) return mfma32h(a, b, c);
    else return mfma32(a, b, c);
}
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
// v - float(half of hh): one mixed-precision FMA (the fp16 half is read in place, no v_cvt_f32_f16)
__device__ __forceinline__ float minus_f16_lo(float v, unsigned hh)
{
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hh), "v"(v));
    return r;
}
__device__ __forceinline__ float minus_f16_hi(float v, unsigned hh)
{
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hh), "v"(v));
    return r;
}
// hi = fp16(v) and lo = fp16(v - hi), both round-to-nearest-even (v_cvt_pk_f16_f32); v - hi is exact in fp32:
// |v - hi - lo| <= 2^-22 |v| for 2^-14 <= |v| <= 65504.  4 VALU instructions per pair of values.
// (v_fma_mixlo / mixhi_f16 would round v - hi straight into the fp16 halves, one instruction less per pair: measured
//  SLOWER -- ZINC-95 k step 2.92 -> 3.01 ms -- and not the same bits; not used.)
__device__ __forceinline__ void split_f16x4(const f32x4 &v, s16x4 &hi, s16x4 &lo)
{
    unsigned h[2], l[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const f32x2 a = {v[2 * q], v[2 * q + 1]};
        const unsigned hu = __builtin_bit_cast(unsigned, __builtin_convertvector(a, f16x2));
        const f32x2 r = {minus_f16_lo(a[0], hu), minus_f16_hi(a[1], hu)};
        h[q] = hu;
        l[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
    }
    struct U { unsigned a, b; } uh{h[0], h[1]}, ul{l[0], l[1]};
    hi = __builtin_bit_cast(s16x4, uh);
    lo = __builtin_bit_cast(s16x4, ul);
}
// fp32 values that ARE fp16 values (outputs of an MFMA against the identity) back to their 16-bit form
__device__ __forceinline__ s16x4 exact_f16(const f32x4 &d)
{
    struct U { unsigned a, b; } u{__builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(d[0], d[1])),
                                  __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(d[2], d[3]))};
    return __builtin_bit_cast(s16x4, u);
}

template <int KS, bool WITH_GRAD, int RI, int MINW, bool SBF16, bool PBF16, bool TRV = false, bool S3 = false>
__global__ __launch_bounds__(256, MINW) void bce_dense_kernel(
    const float *__restrict__ Zt /*[n][16 KS]*/, const unsigned short *__restrict__ Zhi,
    const unsigned short *__restrict__ Zlo, int64_t n, int64_t row_begin, int64_t n_local, int64_t cols_per_split,
    float *__restrict__ O_partial /*[splits][n_local][KS*16]*/, double *__restrict__ loss_partial /*[blocks][2]*/,
    const double *__restrict__ colsum_partial, int64_t n_prep_blocks, double *__restrict__ S,
    float *__restrict__ S_all_f, unsigned n_row_blocks)
{
    constexpr int ROWS_PER_BLOCK = 4 * RI * 16;
    constexpr int DP = KS * 16;          // padded feature width
    constexpr int LDA = DP + 4;          // fp32 LDS row stride (floats): 16-byte aligned, breaks the power of two
    constexpr int LDH = DP + 4;          // bf16 LDS row stride (elements): 8-byte aligned rows
    constexpr int V4 = TJ * DP / 4 / 256;  // float4 per thread and staged tile (1, 2, 4)
    constexpr int LDT = TJ + 4;          // transposed bf16 tile row stride (elements), 8-byte aligned rows
    static_assert(!S3 || SBF16, "the three-piece S product is a bf16 form");
    constexpr bool NEED_F32 = !SBF16 || (WITH_GRAD && !PBF16);
    constexpr bool LOAD_F32 = NEED_F32 || S3;            // S3: the staged fp32 values give the third piece
    constexpr bool NEED_BF = SBF16 || (WITH_GRAD && PBF16);
    static_assert(!TRV || (SBF16 && PBF16 && WITH_GRAD), "transpose reads take the V fragments from the bf16 [j][k] tiles");
    constexpr bool NEED_T = WITH_GRAD && PBF16 && !TRV;
    __shared__ __attribute__((aligned(16))) float Zs[2][NEED_F32 ? TJ * LDA : 4];            // fp32 column tile [j][k]
    __shared__ __attribute__((aligned(16))) unsigned short Hs[2][SBF16 ? TJ * LDH : 4];      // bf16 hi [j][k]
    __shared__ __attribute__((aligned(16))) unsigned short Ls[2][SBF16 ? TJ * LDH : 4];      // bf16 lo [j][k]
    __shared__ __attribute__((aligned(16))) unsigned short L2s[2][S3 ? TJ * LDH : 4];        // bf16 lo2 [j][k]
    __shared__ __attribute__((aligned(16))) unsigned short HT[2][NEED_T ? DP * LDT : 4];     // bf16 hi [k][j]
    __shared__ __attribute__((aligned(16))) unsigned short LT[2][NEED_T ? DP * LDT : 4];     // bf16 lo [k][j]
    __shared__ double red[4][2];

    if (blockIdx.x >= n_row_blocks) {   // the extra block column: column sums of Zt for the kernels that follow
        static_assert(sizeof(Hs) >= 4096 || sizeof(Zs) >= 4096, "column-sum scratch aliases a staging tile");
        if (blockIdx.y == 0)
            bce_colsum_block(colsum_partial, n_prep_blocks, DP, S, S_all_f,
                             reinterpret_cast<double *>(sizeof(Hs) >= 4096 ? static_cast<void *>(&Hs[0][0])
                                                                           : static_cast<void *>(&Zs[0][0])));
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    // rows are LOCAL ids (0 .. n_local) of the window [row_begin, row_begin + n_local) of Zt
    const int64_t row_base = int64_t(blockIdx.x) * ROWS_PER_BLOCK + wave * (RI * 16);
    const int64_t col_begin = int64_t(blockIdx.y) * cols_per_split;
    int64_t col_end = col_begin + cols_per_split;
    if (col_end > n) col_end = n;

    // B fragments of S^T = Zj Zi^T: lane (i = l15, g) holds LOG2E * Zt[i][16 c + 4 g + r], r = 0..3 (fp32, or
    // split into hi / lo bf16 here, once per wave).  The row operand carries the log2(e) factor, so the
    // accumulator is y = x log2(e) and exp(-|x|) = exp2(-|y|) needs no multiply per logit (sign and |.|
    // sums are rescaled at the end).
    constexpr float LOG2E = 1.44269504088896341f;
    // bf16 x 3 on K = 32 MFMAs (see bce_dense_sym_kernel): K2 fragments per row subtile, each two K = 16 fragments
    // side by side -- chunk pairs [c | c + 1] for d > 16, the same chunk twice for d <= 16.
    constexpr int K2 = KS == 1 ? 1 : KS / 2;
    f32x4 bfrag[RI][KS];
    s16x8 bhh[RI][K2], bll[RI][K2];
    s16x8 b3[RI][S3 ? K2 : 1];        // S3: KS == 1: [bhi | blo2]; chunk pairs: [blo2 | blo2']
#pragma unroll
    for (int ri = 0; ri < RI; ++ri) {
        const int64_t i = row_base + ri * 16 + l15;
        const int64_t gi = row_begin + (i < n_local ? i : 0);
#pragma unroll
        for (int c = 0; c < KS; ++c) {
            f32x4 b = *reinterpret_cast<const f32x4 *>(Zt + gi * DP + 16 * c + 4 * g);
            if (i >= n_local) b = f32x4{0.f, 0.f, 0.f, 0.f};
            b *= LOG2E;
            bfrag[ri][c] = b;
            if (SBF16) {
                s16x4 bh, bl, bl2;
                if (S3) split_bf16x4_3(b, bh, bl, bl2);
                else split_bf16x4(b, bh, bl);
                if (KS == 1) {
                    bhh[ri][0] = cat(bh, bh);
                    bll[ri][0] = cat(bl, bl);
                    if (S3) b3[ri][0] = cat(bh, bl2);
                } else {
                    put_half(bhh[ri][c / 2], c & 1, bh);
                    put_half(bll[ri][c / 2], c & 1, bl);
                    if (S3) put_half(b3[ri][c / 2], c & 1, bl2);
                }
            }
        }
    }
    f32x4 oacc[RI][KS];
#pragma unroll
    for (int ri = 0; ri < RI; ++ri)
#pragma unroll
        for (int c = 0; c < KS; ++c) oacc[ri][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    double sumA[RI], sumL[RI];
#pragma unroll
    for (int ri = 0; ri < RI; ++ri) { sumA[ri] = 0.0; sumL[ri] = 0.0; }

    // staging: thread -> 16-byte piece #(tid + 256 q) of the fp32 tile and 8-byte piece of each bf16 tile;
    // tile rows are contiguous in Zt / Zhi / Zlo.  Rows >= col_end are staged as zeros (branch-free: the
    // row index is clamped and the value selected).
    struct Stage { f32x4 f[V4]; s16x4 h[V4], l[V4]; };
    auto load_tile = [&](int64_t j0, Stage &st) {
#pragma unroll
        for (int q = 0; q < V4; ++q) {
            const int idx = tid + 256 * q;            // 4-element piece index inside the tile
            const int jj = idx / (DP / 4), kk = (idx % (DP / 4)) * 4;
            const bool jv = j0 + jj < col_end;
            const int64_t j = jv ? j0 + jj : col_begin;
            if (LOAD_F32) {
                st.f[q] = *reinterpret_cast<const f32x4 *>(Zt + j * DP + kk);
                if (!jv) st.f[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (NEED_BF) {
                st.h[q] = *reinterpret_cast<const s16x4 *>(Zhi + j * DP + kk);
                st.l[q] = *reinterpret_cast<const s16x4 *>(Zlo + j * DP + kk);
                if (!jv) { st.h[q] = s16x4{0, 0, 0, 0}; st.l[q] = s16x4{0, 0, 0, 0}; }
            }
        }
    };
    auto store_tile = [&](int buf, const Stage &st) {
#pragma unroll
        for (int q = 0; q < V4; ++q) {
            const int idx = tid + 256 * q;
            const int jj = idx / (DP / 4), kk = (idx % (DP / 4)) * 4;
            if (NEED_F32) *reinterpret_cast<f32x4 *>(&Zs[buf][jj * LDA + kk]) = st.f[q];
            if (SBF16) {
                *reinterpret_cast<s16x4 *>(&Hs[buf][jj * LDH + kk]) = st.h[q];
                *reinterpret_cast<s16x4 *>(&Ls[buf][jj * LDH + kk]) = st.l[q];
                if (S3) *reinterpret_cast<s16x4 *>(&L2s[buf][jj * LDH + kk]) = third_piece(st.f[q], st.h[q], st.l[q]);
            }
            if (NEED_T) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    HT[buf][(kk + e) * LDT + jj] = (unsigned short)st.h[q][e];
                    LT[buf][(kk + e) * LDT + jj] = (unsigned short)st.l[q][e];
                }
            }
        }
    };

    auto compute_tile = [&](int buf) {
        const float *zs = Zs[buf];
        // fp32 partials of this 64-column tile (16 logits per lane and subtile): sum |x| and PRODUCT of
        // t = 1 + exp(-|x|) in [1, 2] (<= 2^16): sum log2(t) = log2(prod t) costs one v_log per 16 logits
        float tA[RI], tP[RI];
#pragma unroll
        for (int ri = 0; ri < RI; ++ri) { tA[ri] = 0.f; tP[ri] = 1.f; }
#pragma unroll
        for (int jp = 0; jp < TJ / 32; ++jp) {           // pairs of 16-column subtiles (K = 32 for O' += P V)
            s16x8 ph[RI], pl[RI];                        // bf16 P of the pair: [subtile 2 jp | subtile 2 jp + 1]
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int jt = 2 * jp + h;
                f32x4 sacc[RI];
#pragma unroll
                for (int ri = 0; ri < RI; ++ri) sacc[ri] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (SBF16) {
                    // A fragments: lane (j = l15, g) -> 4 bf16 at [jt*16 + j][16 c + 4 g ..]
                    auto frag = [&](const unsigned short *base, int c) {
                        return *reinterpret_cast<const s16x4 *>(&base[(jt * 16 + l15) * LDH + 16 * c + 4 * g]);
                    };
                    if (KS == 1) {        // S = [ah | al] x [bhi | bhi] + [ah | al] x [blo | blo]  (+ [al2 | ah] x [bhi | blo2])
                        const s16x4 fh = frag(Hs[buf], 0);
                        const s16x8 ahl = cat(fh, frag(Ls[buf], 0));
#pragma unroll
                        for (int ri = 0; ri < RI; ++ri) {
                            if (S3) sacc[ri] = mfma32(cat(frag(L2s[buf], 0), fh), b3[ri][0], sacc[ri]);
                            sacc[ri] = mfma32(ahl, bll[ri][0], sacc[ri]);
                            sacc[ri] = mfma32(ahl, bhh[ri][0], sacc[ri]);
                        }
                    } else {              // chunk pairs: lo.hi + hi.lo + hi.hi, 3 MFMAs per 32 features (S3: + lo.lo, hi.lo2, lo2.hi)
#pragma unroll
                        for (int c2 = 0; c2 < K2; ++c2) {
                            const s16x8 ah = cat(frag(Hs[buf], 2 * c2), frag(Hs[buf], 2 * c2 + 1));
                            const s16x8 al = cat(frag(Ls[buf], 2 * c2), frag(Ls[buf], 2 * c2 + 1));
                            if (S3) {
                                const s16x8 al2 = cat(frag(L2s[buf], 2 * c2), frag(L2s[buf], 2 * c2 + 1));
#pragma unroll
                                for (int ri = 0; ri < RI; ++ri) {
                                    sacc[ri] = mfma32(al2, bhh[ri][c2], sacc[ri]);
                                    sacc[ri] = mfma32(ah, b3[ri][c2], sacc[ri]);
                                    sacc[ri] = mfma32(al, bll[ri][c2], sacc[ri]);
                                }
                            }
#pragma unroll
                            for (int ri = 0; ri < RI; ++ri) {
                                sacc[ri] = mfma32(al, bhh[ri][c2], sacc[ri]);
                                sacc[ri] = mfma32(ah, bll[ri][c2], sacc[ri]);
                                sacc[ri] = mfma32(ah, bhh[ri][c2], sacc[ri]);
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < KS; ++c) {
                        const f32x4 af = *reinterpret_cast<const f32x4 *>(&zs[(jt * 16 + l15) * LDA + 16 * c + 4 * g]);
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int ri = 0; ri < RI; ++ri)
                                sacc[ri] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r], bfrag[ri][c][r], sacc[ri], 0, 0, 0);
                    }
                }
                // sacc[ri][r] = x(i = l15 of subtile ri, j = jt*16 + 4 g + r)
                f32x4 p[RI];
#pragma unroll
                for (int ri = 0; ri < RI; ++ri) p[ri] = quad_terms<RI == 2>(sacc[ri], tP[ri], tA[ri]);   // sigmoid(x) - 1/2 (sacc = x_ij * log2(e))
                if (WITH_GRAD && PBF16) {     // P = hi + lo (bf16) on the fly
#pragma unroll
                    for (int ri = 0; ri < RI; ++ri) {
                        s16x4 h4, l4;
                        split_bf16x4(p[ri], h4, l4);
                        put_half(ph[ri], h, h4);
                        put_half(pl[ri], h, l4);
                    }
                }
                if (WITH_GRAD && !PBF16) {
                    // B fragments of O' += P V: lane (nn = l15, g) -> V[jt*16 + 4 g + r][16 c + nn]
#pragma unroll
                    for (int c = 0; c < KS; ++c) {
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = zs[(jt * 16 + 4 * g + r) * LDA + 16 * c + l15];
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int ri = 0; ri < RI; ++ri)
                                oacc[ri][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(p[ri][r], v[r], oacc[ri][c], 0, 0, 0);
                    }
                }
            }
            if (WITH_GRAD && PBF16) {
                // B fragments: lane (nn = l15, g) -> bf16 V[j][16 c + nn] for the 4 + 4 columns j this lane group
                // holds of the two subtiles, read from the transposed tiles; O' += lo.hi + hi.lo + hi.hi, K = 32
#pragma unroll
                for (int c = 0; c < KS; ++c) {
                    s16x8 vh, vl;
                    if constexpr (TRV) {      // LDS transpose reads of the [j][k] tiles (see lds_read_tr)
                        const int o = (jp * 32 + 4 * g + (l15 >> 2)) * LDH + 16 * c + 4 * (l15 & 3);
                        vh = cat(lds_read_tr(&Hs[buf][o]), lds_read_tr(&Hs[buf][o + 16 * LDH]));
                        vl = cat(lds_read_tr(&Ls[buf][o]), lds_read_tr(&Ls[buf][o + 16 * LDH]));
                    } else {
                        const int at = (16 * c + l15) * LDT + jp * 32 + 4 * g;
                        vh = cat(*reinterpret_cast<const s16x4 *>(&HT[buf][at]),
                                 *reinterpret_cast<const s16x4 *>(&HT[buf][at + 16]));
                        vl = cat(*reinterpret_cast<const s16x4 *>(&LT[buf][at]),
                                 *reinterpret_cast<const s16x4 *>(&LT[buf][at + 16]));
                    }
#pragma unroll
                    for (int ri = 0; ri < RI; ++ri) {
                        oacc[ri][c] = mfma32(pl[ri], vh, oacc[ri][c]);
                        oacc[ri][c] = mfma32(ph[ri], vl, oacc[ri][c]);
                        oacc[ri][c] = mfma32(ph[ri], vh, oacc[ri][c]);
                    }
                }
            }
        }
#pragma unroll
        for (int ri = 0; ri < RI; ++ri) {
            sumA[ri] += double(tA[ri]);
            sumL[ri] += double(__builtin_amdgcn_logf(tP[ri]));
        }
    };

    Stage stage;
    if (col_begin < col_end) {
        load_tile(col_begin, stage);
        store_tile(0, stage);
    }
    __syncthreads();
    int buf = 0;
    for (int64_t j0 = col_begin; j0 < col_end; j0 += TJ, buf ^= 1) {
        const bool more = j0 + TJ < col_end;
        if (more) load_tile(j0 + TJ, stage);        // in flight while this tile is consumed
        compute_tile(buf);
        if (more) store_tile(buf ^ 1, stage);
        __syncthreads();
    }
    // ---- O' partial: oacc[ri][c][r] = O'(i = 4 g + r, nn = l15) of subtile ri, feature 16 c + nn
    if (WITH_GRAD) {
        float *op = O_partial + int64_t(blockIdx.y) * n_local * DP;
#pragma unroll
        for (int ri = 0; ri < RI; ++ri)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t i = row_base + ri * 16 + 4 * g + r;
                if (i < n_local) {
#pragma unroll
                    for (int c = 0; c < KS; ++c) op[i * DP + 16 * c + l15] = oacc[ri][c][r];
                }
            }
    }
    // ---- loss partials: drop rows >= n_local, wave reduce (fixed tree) -> block
    double la = 0.0, ll = 0.0;
#pragma unroll
    for (int ri = 0; ri < RI; ++ri) {
        const bool rv = (row_base + ri * 16 + l15) < n_local;
        la += rv ? sumA[ri] * 0.69314718055994531 : 0.0;    // sum |y| / log2(e) = sum |x|
        ll += rv ? sumL[ri] : 0.0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { la += __shfl_down(la, off, 64); ll += __shfl_down(ll, off, 64); }
    if (lane == 0) { red[wave][0] = la; red[wave][1] = ll; }
    __syncthreads();
    if (tid == 0) {
        const int64_t b = int64_t(blockIdx.y) * n_row_blocks + blockIdx.x;
        loss_partial[2 * b + 0] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        loss_partial[2 * b + 1] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    }
}

// ---------------------------------------------------------------------------
// Symmetric form of the dense part (full square, d <= 16, bf16x3 products): X = Zt Zt^T is symmetric, so only the
// tiles on and above the block diagonal are evaluated.  A block owns a panel of 128 rows (4 waves x 2 x 16) and
// one chunk of the columns at or right of the panel:
//   * tiles inside the panel's own 128 x 128 square are evaluated as in bce_dense_kernel (weight 1);
//   * tiles right of it are evaluated ONCE (the 2 transcendentals + ~6 VALU ops per logit are the cost of this
//     kernel) and serve both halves: their loss terms count twice, O'_I += P Z_J as before, and the mirror
//     O'_J += P^T Z_I is a second small matrix product whose A operand is P transposed through a wave-private
//     LDS tile (P sits in registers as lane = row i, regs = columns j; the mirror needs lane = column j).
//     The 4 waves' mirror tiles meet in LDS (fixed order) and go to a triangular strip buffer
//     Wmir[panel][f][j] that bce_mirror_reduce_kernel folds into O'_mirror[j][f] in panel order.
// Mirror traffic: N^2 / 4 bytes written and read once (Pubmed 97 MB against 1.9 x 10^8 logits saved).
// Zero-padded columns (j >= n) are corrected in the kernel (their log2(1 + e^0) = 1 is subtracted).
// ---------------------------------------------------------------------------
// rows per panel = 64 RI (4 waves x RI subtiles of 16 rows); "bce_sym_ri" = 0 (auto) | 2 | 4.  Taller panels halve
// the mirror strips (N^2 / 8 bytes: 2.2 -> 1.1 GB written and read back on a ZINC batch) at the price of registers
// (2 waves per SIMD): they pay from ~32 k rows on (ZINC batch of 95 k rows: 3.00 -> 2.92 ms; Pubmed, 20 k rows:
// 185 -> 194 us).  With the K = 32 fragments the fully unrolled 256-row body spilled 43 VGPRs (3.48 ms) and its
// column-pair loop was left rolled (238 VGPRs) -- until the one-reciprocal-per-four-logits form (quad_terms) freed
// enough registers: unrolled it now takes 234 VGPRs without a spill (ZINC step 2.853 -> 2.72 ms).
gae::Knob g_bce_sym_ri{0};
gae::Knob g_bce_sym_tr{1};    // "bce_sym_tr": 1 = V fragments by LDS transpose reads (ds_read_b64_tr_b16), 0 = from transposed tile copies

// the upper 16 bits of four fp32 values (exact when they are bf16 values): one v_perm_b32 per pair
__device__ __forceinline__ s16x4 upper_halves(const f32x4 &d)
{
    struct U { unsigned a, b; } u{__builtin_amdgcn_perm(__float_as_uint(d[1]), __float_as_uint(d[0]), 0x07060302u),
                                  __builtin_amdgcn_perm(__float_as_uint(d[3]), __float_as_uint(d[2]), 0x07060302u)};
    return __builtin_bit_cast(s16x4, u);
}

// float offset of panel I's strip in Wmir: strips are [(NP - PR (I + 1)) / 64 column tiles][16][64] with NP = n
// rounded up to 64
__host__ __device__ inline int64_t sym_strip_offset(int64_t I, int64_t NP, int64_t PR)
{
    return 16 * (I * NP - PR * (I * (I + 1) / 2));
}

// TRV (default): the V fragments of O' += P V come from LDS transpose reads of the [j][k] tiles -- no second,
// transposed copy of every tile (16 ds_write_b16 per thread and tile, 8.7 KB of LDS): Pubmed 170 -> 166 us, a ZINC
// batch 2.92 -> 2.88 ms.  TRV = false keeps the round-2 form (knob "bce_sym_tr" = 0).
template <bool WITH_GRAD, int RI, bool TRV, bool S3 = false, bool F16 = false, bool PERSIST = false, bool BAL = false>
__global__ __launch_bounds__(256, (RI == 2 && !PERSIST) ? 3 : 2) void bce_dense_sym_kernel(
    const float *__restrict__ Zt /*[n][16]*/, const unsigned short *__restrict__ Zhi,
    const unsigned short *__restrict__ Zlo, int64_t n, int64_t cols_per_chunk,
    float *__restrict__ O_partial /*[chunks][n][16]*/, float *__restrict__ Wmir,
    double *__restrict__ loss_partial /*[chunks * panels][2]*/, const double *__restrict__ colsum_partial,
    int64_t n_prep_blocks, double *__restrict__ S, float *__restrict__ S_all_f, unsigned n_panels, int exp_strip,
    unsigned *__restrict__ range_flag, int flag_mode, unsigned ticket, unsigned n_chunks, int bal_tpb)
{
    // Range guard of the fp16 pieces.  flag_mode 1 (the F16 launch): a thread that meets |Zt| > kF16Max (or a NaN)
    // writes this call's ticket to *range_flag; the launch's results are then meaningless.  flag_mode 2 (the
    // three-piece bf16 launch that follows it): does nothing unless *range_flag holds the ticket, else recomputes every
    // output of the first launch.  The flag lives in the caller's workspace, which nobody initialises: the ticket (a
    // process-wide counter, scrambled) tells this call's report from whatever the memory held, and the edge kernel
    // (always after both launches) clears it, so a replayed HIP graph -- whose ticket is frozen -- starts clean as
    // well.  A stale match costs one redundant bf16 pass, never a wrong result.  Embeddings beyond 32768 do not
    // occur in a trained GAE: in practice the second launch costs the start of its few blocks (PERSIST: a 1-D grid of
    // a few hundred blocks that would walk the (panel, chunk) units of the first launch's 2-D grid).
    if (flag_mode == 2 && *range_flag != ticket) return;
    static_assert(!(S3 && F16), "three bf16 pieces OR two fp16 pieces");
    bool out_of_range = false;
    constexpr bool STAGE32 = S3 || F16;  // the column tiles are staged as fp32 and split at LDS-store time
    constexpr int DP = 16, SYM_PR = 64 * RI;
    constexpr int LDH = DP + 4;          // bf16 LDS row stride (elements): 8-byte aligned rows
    constexpr int V4 = TJ * DP / 4 / 256;  // = 1
    constexpr int LDT = TJ + 4;          // transposed bf16 tile row stride (elements)
    constexpr int LDM = TJ + 4;          // mirror tile row stride (floats)
    // hi and lo pieces of a column tile INTERLEAVED per row: [hi k 0..3 | lo k 0..3 | hi k 4..7 | lo k 4..7 | ...], rows of
    // 2 DP + 8 elements (80 bytes: the 16-byte pieces of 16 consecutive rows fall on disjoint banks).  A lane's [ah | al]
    // operand of the S product is ONE ds_read_b128 straight into four consecutive registers (two 8-byte reads from two
    // arrays needed four v_mov per subtile to line them up); the LDS transpose reads take the pieces at stride 16 bytes.
    constexpr int LDHL = 2 * DP + 8;
    __shared__ __attribute__((aligned(16))) unsigned short HLs[2][TJ * LDHL];
    __shared__ __attribute__((aligned(16))) unsigned short L2s[2][S3 ? TJ * LDH : 4];       // bf16 lo2 [j][k] (three-piece S)
    __shared__ __attribute__((aligned(16))) unsigned short HT[2][TRV ? 4 : DP * LDT];       // bf16 hi [k][j]
    __shared__ __attribute__((aligned(16))) unsigned short LT[2][TRV ? 4 : DP * LDT];       // bf16 lo [k][j]
    // mirror tiles [buffer][wave][f][j].  Two buffers (round 4): the tile after next rewrites a buffer only behind the NEXT tile's
    // barrier, which every thread passes after it has flushed this one -- the second block barrier per tile is gone.  (Round 2
    // tried this with the transposed tile copies still in LDS: 54 KB per block, 2 instead of 3 blocks per CU, 157 -> 179 us; since
    // the LDS transpose reads the block holds 45 KB and three still fit.)
    __shared__ __attribute__((aligned(16))) float MR[2][4][16 * LDM];
    __shared__ double red[4][2];

  double sumA = 0.0, sumL = 0.0;    // loss sums over this lane's rows of all RI subtiles (rows >= n excluded as they are added): of one
                                    // unit (2-D grid) or of all the units of the block (balanced schedule)
  // one unit of work: panel I (rows [SYM_PR I, SYM_PR (I + 1))) against the columns [col_begin, col_end) at or right of it; its
  // O' partial goes to slot op_slot of the panel's rows; lp_index >= 0: its loss sums to loss_partial[2 lp_index ..], < 0: added
  // carried on in sumA / sumL
  auto unit = [&](const int64_t I, const int64_t col_begin, const int64_t col_end, const int64_t op_slot, const int64_t lp_index) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int64_t NP = (n + 63) / 64 * 64;
    const int64_t row_base = SYM_PR * I + wave * (RI * 16);
    const int64_t diag_end = SYM_PR * (I + 1);      // tiles starting below this column lie in the panel's own square

    constexpr float LOG2E = 1.44269504088896341f;
    static_assert(RI % 2 == 0, "row subtiles are paired into K = 32 mirror products");
    s16x8 bhh[RI];               // B fragments of S^T = Zj Zi^T (row operand, log2(e) folded in): [hi | hi]
    s16x8 bll[RI];               // ... and [lo | lo]: S = (ah + al)(bhi + blo), all four partial products
    s16x8 b3[S3 ? RI : 1];       // S3: [hi | lo2] against [al2 | ah]: + al2.bhi + ah.blo2
    s16x8 zTh[RI / 2], zTl[RI / 2];   // B fragments of the mirror product, row subtiles (2 rp, 2 rp + 1) concatenated:
                                      // lane (f = l15, g) -> Zt[i = 4 g + r][f]
#pragma unroll
    for (int ri = 0; ri < RI; ++ri) {
        const int64_t i = row_base + ri * 16 + l15;
        f32x4 b = *reinterpret_cast<const f32x4 *>(Zt + (i < n ? i : 0) * DP + 4 * g);
        if (i >= n) b = f32x4{0.f, 0.f, 0.f, 0.f};
        b *= LOG2E;
        s16x4 bhi, blo, blo2;
        if (S3) { split_bf16x4_3(b, bhi, blo, blo2); b3[ri] = cat(bhi, blo2); }
        else if (F16) {
            out_of_range |= !(fmaxf(fmaxf(fabsf(b[0]), fabsf(b[1])), fmaxf(fabsf(b[2]), fabsf(b[3]))) <= kF16Max * LOG2E);
            split_f16x4(b, bhi, blo);
        } else split_bf16x4(b, bhi, blo);
        bhh[ri] = cat(bhi, bhi);
        bll[ri] = cat(blo, blo);
        s16x4 th, tl;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t ir = row_base + ri * 16 + 4 * g + r;
            const bool v = ir < n;
            if (F16) continue;
            th[r] = v ? short(Zhi[(v ? ir : 0) * DP + l15]) : short(0);
            tl[r] = v ? short(Zlo[(v ? ir : 0) * DP + l15]) : short(0);
        }
        if (F16) {                   // fp16 pieces of the rows, from the fp32 values
            f32x4 zr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t ir = row_base + ri * 16 + 4 * g + r;
                zr[r] = ir < n ? Zt[ir * DP + l15] : 0.f;
            }
            split_f16x4(zr, th, tl);
        }
        put_half(zTh[ri / 2], ri & 1, th);
        put_half(zTl[ri / 2], ri & 1, tl);
    }
    s16x4 ident;                 // B fragment of the 16 x 16 identity: lane (n = l15, g) -> [k = 4 g + r == n]
#pragma unroll
    for (int r = 0; r < 4; ++r) ident[r] = (4 * g + r == l15) ? short(F16 ? 0x3C00 : 0x3F80) : short(0);
    f32x4 oacc[RI];
#pragma unroll
    for (int ri = 0; ri < RI; ++ri) oacc[ri] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!BAL) { sumA = 0.0; sumL = 0.0; }
    bool rvalid[RI];
#pragma unroll
    for (int ri = 0; ri < RI; ++ri) rvalid[ri] = (row_base + ri * 16 + l15) < n;
    // zero-padded columns (j >= n) exist only in the last tile of the last chunk: their count per lane, once
    const int64_t pad_j0 = (col_end == n && (n % TJ) != 0) ? n / TJ * TJ : -1;
    float pad_lane = 0.f;
    if (pad_j0 >= 0) {
#pragma unroll
        for (int jt = 0; jt < TJ / 16; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) pad_lane += (pad_j0 + jt * 16 + 4 * g + r >= n) ? 1.f : 0.f;
    }

    // S3: only the fp32 values are staged (4 registers instead of 4 + 4 + 4); hi / lo / lo2 are split off at store time
    // with the prepare step's own roundings (v_cvt_pk_bf16_f32 = round to nearest even: the same hi / lo bits)
    struct Stage { s16x4 h[STAGE32 ? 1 : V4], l[STAGE32 ? 1 : V4]; f32x4 f[STAGE32 ? V4 : 1]; };
    auto load_tile = [&](int64_t j0, Stage &st) {
#pragma unroll
        for (int q = 0; q < V4; ++q) {
            const int idx = tid + 256 * q;
            const int jj = idx / (DP / 4), kk = (idx % (DP / 4)) * 4;
            const bool jv = j0 + jj < col_end;
            const int64_t j = jv ? j0 + jj : col_begin;
            if (STAGE32) {
                st.f[q] = *reinterpret_cast<const f32x4 *>(Zt + j * DP + kk);
                if (!jv) st.f[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
                st.h[q] = *reinterpret_cast<const s16x4 *>(Zhi + j * DP + kk);
                st.l[q] = *reinterpret_cast<const s16x4 *>(Zlo + j * DP + kk);
                if (!jv) { st.h[q] = s16x4{0, 0, 0, 0}; st.l[q] = s16x4{0, 0, 0, 0}; }
            }
        }
    };
    auto store_tile = [&](int buf, const Stage &st) {
#pragma unroll
        for (int q = 0; q < V4; ++q) {
            const int idx = tid + 256 * q;
            const int jj = idx / (DP / 4), kk = (idx % (DP / 4)) * 4;
            s16x4 h4, l4;
            if (S3) {
                s16x4 l24;
                split_bf16x4_3(st.f[q], h4, l4, l24);
                *reinterpret_cast<s16x4 *>(&L2s[buf][jj * LDH + kk]) = l24;
            } else if (F16) {
                const f32x4 f = st.f[q];
                out_of_range |= !(fmaxf(fmaxf(fabsf(f[0]), fabsf(f[1])), fmaxf(fabsf(f[2]), fabsf(f[3]))) <= kF16Max);
                split_f16x4(f, h4, l4);
            } else {
                h4 = st.h[q]; l4 = st.l[q];
            }
            *reinterpret_cast<s16x8 *>(&HLs[buf][jj * LDHL + 2 * kk]) = cat(h4, l4);
            if (WITH_GRAD && !TRV) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    HT[buf][(kk + e) * LDT + jj] = (unsigned short)h4[e];
                    LT[buf][(kk + e) * LDT + jj] = (unsigned short)l4[e];
                }
            }
        }
    };
    // sum the 4 waves' mirror tiles of the tile that started at column j0 and store it to this panel's strip
    float *strip = Wmir + sym_strip_offset(I, NP, SYM_PR);
    auto flush_mirror = [&](int64_t j0, int mb) {
        const int f = tid >> 4, jq = (tid & 15) * 4;
        f32x4 v = *reinterpret_cast<const f32x4 *>(&MR[mb][0][f * LDM + jq]);
#pragma unroll
        for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const f32x4 *>(&MR[mb][w][f * LDM + jq]);
        // tile-major strip: the 16 x 64 block of one column tile is 4 KB of contiguous memory (written here by one
        // block, read back by one block of the reduction) instead of 16 pieces of 256 bytes
        f32x4 *sp = reinterpret_cast<f32x4 *>(strip + (j0 - diag_end) * 16 + f * TJ + jq);
        if (exp_strip == 1) __builtin_nontemporal_store(v, sp);
        else if (exp_strip == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(sp), "v"(v) : "memory");
        else *sp = v;
    };

    // ---- one 64-column tile.  gfx950 runs v_mfma_f32_16x16x32_bf16 in the same 4 passes as the 16x16x16 form
    //      (tools/probes/inst_cost.hip: 7.5 ns per SIMD either way), and on this chip a SIMD's MFMA and VALU time
    //      ADD (mfma + 4 v_fma: 11.2 - 12.8 ns against 7.4 + 5.7): every MFMA saved is time saved.  A K = 32 MFMA on
    //      concatenated fragments [a1 | a2] x [b1 | b2] is a1 b1 + a2 b2 (lane group g holds k = 4 g + r of both
    //      halves), so
    //        S     = [ah | al] x [bhi | bhi]  +  [ah | al] x [blo | blo]   2 instead of 3 MFMAs per 16 x 16 logits
    //        O'   += [p(jt) | p(jt+1)] x [v(jt) | v(jt+1)]  (x 3)      3 per TWO column subtiles
    //        mirror = [q(ri) | q(ri+1)] x [z(ri) | z(ri+1)]  (x 3)     3 per TWO row subtiles
    //      -- 56 instead of 88 MFMAs per tile and wave (RI = 2).
    auto compute_tile = [&](int buf, int64_t j0, bool offdiag) {
        float tA[RI], tP[RI];
#pragma unroll
        for (int ri = 0; ri < RI; ++ri) { tA[ri] = 0.f; tP[ri] = 1.f; }
        // (256-row panels: unrolled since the four-logit form -- 234 VGPRs; the three-piece fallback stays rolled: it spills unrolled)
#pragma unroll((RI == 4 && S3) ? 1 : TJ / 32)
        for (int jp = 0; jp < TJ / 32; ++jp) {           // pairs of 16-column subtiles
            s16x8 ph[RI], pl[RI];                        // P of the pair: [subtile 2 jp | subtile 2 jp + 1]
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int jt = 2 * jp + h;
                const s16x8 ahl = *reinterpret_cast<const s16x8 *>(&HLs[buf][(jt * 16 + l15) * LDHL + 8 * g]);
                const s16x4 ah = get_half(ahl, 0);
                f32x4 sacc[RI];
#pragma unroll
                for (int ri = 0; ri < RI; ++ri) {
                    sacc[ri] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (S3)       // the smallest terms first: al2.bhi + ah.blo2
                        sacc[ri] = mfma32(cat(*reinterpret_cast<const s16x4 *>(&L2s[buf][(jt * 16 + l15) * LDH + 4 * g]), ah),
                                          b3[ri], sacc[ri]);
                    sacc[ri] = mfma_k32<F16>(ahl, bll[ri], sacc[ri]);
                    sacc[ri] = mfma_k32<F16>(ahl, bhh[ri], sacc[ri]);
                }
                // sacc[ri][r] = y(i = l15 of subtile ri, j = jt*16 + 4 g + r), y = x log2(e)
#pragma unroll
                for (int ri = 0; ri < RI; ++ri) {
                    const f32x4 p = quad_terms<true>(sacc[ri], tP[ri], tA[ri]);     // sigmoid(x) - 1/2
                    if (WITH_GRAD) {
                        s16x4 h4, l4;
                        if (F16) split_f16x4(p, h4, l4);
                        else split_bf16x4(p, h4, l4);
                        put_half(ph[ri], h, h4);
                        put_half(pl[ri], h, l4);
                    }
                }
            }
            if (WITH_GRAD) {
                const int jc = jp * 32 + 4 * g;
                s16x8 vh, vl;                // lane (f = l15, g): Z[j = jc + r][f] | Z[j = jc + 16 + r][f]
                if constexpr (TRV) {
                    const int o = (jc + (l15 >> 2)) * LDHL + 8 * (l15 & 3);
                    vh = cat(lds_read_tr(&HLs[buf][o]), lds_read_tr(&HLs[buf][o + 16 * LDHL]));
                    vl = cat(lds_read_tr(&HLs[buf][o + 4]), lds_read_tr(&HLs[buf][o + 4 + 16 * LDHL]));
                } else {
                    vh = cat(*reinterpret_cast<const s16x4 *>(&HT[buf][l15 * LDT + jc]),
                             *reinterpret_cast<const s16x4 *>(&HT[buf][l15 * LDT + jc + 16]));
                    vl = cat(*reinterpret_cast<const s16x4 *>(&LT[buf][l15 * LDT + jc]),
                             *reinterpret_cast<const s16x4 *>(&LT[buf][l15 * LDT + jc + 16]));
                }
#pragma unroll
                for (int ri = 0; ri < RI; ++ri) {
                    oacc[ri] = mfma_k32<F16>(pl[ri], vh, oacc[ri]);
                    oacc[ri] = mfma_k32<F16>(ph[ri], vl, oacc[ri]);
                    oacc[ri] = mfma_k32<F16>(ph[ri], vh, oacc[ri]);
                }
                if (offdiag) {
                    // mirror: (P^T Z_I)[j][f] += sum_i P[i][j] Z[i][f] needs P with lane = column j, registers = rows
                    // i; this lane holds P[i = l15][j = 4 g + r].
                    // One MFMA against the identity re-lays P out on the matrix pipe: D = P_hi I has the C layout
                    // lane = j, regs = i, and its fp32 values are exactly the bf16 inputs, so their upper halves
                    // are the A fragments wanted.  (Wave-private LDS planes + transpose reads instead -- 16 MFMAs and
                    // 32 v_perm fewer per tile and wave, 16 KB more LDS traffic -- measured slower: 170 -> 178 us.)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        f32x4 macc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int rp = 0; rp < RI / 2; ++rp) {       // pairs of row subtiles: K = 32 rows
                            s16x8 qh, ql;
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                const int ri = 2 * rp + e;
                                const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
                                f32x4 dh, dl;
                                if constexpr (F16) {
                                    dh = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, get_half(ph[ri], h)), __builtin_bit_cast(f16x4, ident), z4, 0, 0, 0);
                                    dl = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, get_half(pl[ri], h)), __builtin_bit_cast(f16x4, ident), z4, 0, 0, 0);
                                    put_half(qh, e, exact_f16(dh));
                                    put_half(ql, e, exact_f16(dl));
                                } else {
                                    dh = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(get_half(ph[ri], h), ident, z4, 0, 0, 0);
                                    dl = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(get_half(pl[ri], h), ident, z4, 0, 0, 0);
                                    put_half(qh, e, upper_halves(dh));
                                    put_half(ql, e, upper_halves(dl));
                                }
                            }
                            macc = mfma_k32<F16>(ql, zTh[rp], macc);
                            macc = mfma_k32<F16>(qh, zTl[rp], macc);
                            macc = mfma_k32<F16>(qh, zTh[rp], macc);
                        }
                        // macc[r] = mirror(j = jt*16 + 4 g + r, f = l15)  ->  MR[wave][f][j]
                        *reinterpret_cast<f32x4 *>(&MR[buf][wave][l15 * LDM + (2 * jp + h) * 16 + 4 * g]) = macc;
                    }
                }
            }
        }
        // zero-padded columns of the last tile (j >= n): each contributed log2(1 + e^0) = 1 per row
        const float padcnt = j0 == pad_j0 ? pad_lane : 0.f;
        const double w = offdiag ? 2.0 : 1.0;
        float a = 0.f, l = 0.f;
#pragma unroll
        for (int ri = 0; ri < RI; ++ri) {
            a += rvalid[ri] ? tA[ri] : 0.f;
            l += rvalid[ri] ? __builtin_amdgcn_logf(tP[ri]) - padcnt : 0.f;
        }
        sumA += w * double(a);
        sumL += w * double(l);
    };

    Stage stage;
    load_tile(col_begin, stage);
    store_tile(0, stage);
    __syncthreads();
    int buf = 0;
    for (int64_t j0 = col_begin; j0 < col_end; j0 += TJ, buf ^= 1) {
        const bool more = j0 + TJ < col_end;
        if (more) load_tile(j0 + TJ, stage);        // in flight while this tile is consumed
        const bool offdiag = j0 >= diag_end;
        compute_tile(buf, j0, offdiag);
        if (more) store_tile(buf ^ 1, stage);
        __syncthreads();
        if (WITH_GRAD && offdiag) flush_mirror(j0, buf);    // block-uniform; (the next tile writes the other mirror buffer)
    }
    // ---- O' partial: oacc[ri][r] = O'(i = 4 g + r, f = l15) of subtile ri
    if (WITH_GRAD) {
        float *op = O_partial + op_slot * n * DP;
#pragma unroll
        for (int ri = 0; ri < RI; ++ri)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t i = row_base + ri * 16 + 4 * g + r;
                if (i < n) op[i * DP + l15] = oacc[ri][r];
            }
    }
    if (F16 && flag_mode == 1 && out_of_range) atomicExch(range_flag, ticket);
    if (BAL) return;
    double la = sumA * 0.69314718055994531, ll = sumL;     // sum |y| / log2(e) = sum |x|
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { la += __shfl_down(la, off, 64); ll += __shfl_down(ll, off, 64); }
    if (lane == 0) { red[wave][0] = la; red[wave][1] = ll; }
    __syncthreads();
    if (tid == 0) {
        loss_partial[2 * lp_index + 0] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        loss_partial[2 * lp_index + 1] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    }
  };
  // the (panel bx, column chunk by) unit of the 2-D grid
  auto unit_xy = [&](const unsigned bx, const unsigned by) {
    if (bx >= n_panels) {   // the extra block column: column sums of Zt for the kernels that follow
        if (by == 0)
            bce_colsum_block(colsum_partial, n_prep_blocks, DP, S, S_all_f, reinterpret_cast<double *>(&MR[0][0][0]));
        return;
    }
    const int64_t col_begin = SYM_PR * int64_t(bx) + int64_t(by) * cols_per_chunk;
    const int64_t lp_index = int64_t(by) * n_panels + bx;
    if (col_begin >= n) {            // chunk beyond this panel's columns
        if (threadIdx.x == 0) { loss_partial[2 * lp_index] = 0.0; loss_partial[2 * lp_index + 1] = 0.0; }
        return;
    }
    const int64_t ce = col_begin + cols_per_chunk;
    unit(bx, col_begin, ce > n ? n : ce, by, lp_index);
  };
    if constexpr (BAL) {
        // ---- BALANCED schedule (round 6): the column tiles of all panels form ONE sequence (panel-major; panel I has
        //      NT - TPP I tiles, NT = NP / 64); block b takes tiles [b tpb, (b + 1) tpb) of it -- every block the same number
        //      of tiles, all blocks resident at once (grid = 2 or 3 per CU), none waits for a second round.  A block's share
        //      of panel I is one unit; its O' partial goes to slot (b - first block of panel I): a panel has
        //      <= tiles / tpb + 2 partials (Pubmed <= 11, mean 5; a ZINC batch 4) where the 2-D grid wrote up to 29.
        //      One loss partial per block.  The last (virtual) block does the column sums.
        // Virtual blocks: the launch that does the work has one block per tile range (+ 1: the column sums); the fp16 range
        // guard's fallback launch walks the same ranges with a few blocks (it returns at once unless the guard fired).
        const unsigned n_vb = n_chunks;
        auto vblock = [&](const unsigned vb) {
        if (vb + 1 == n_vb) {
            bce_colsum_block(colsum_partial, n_prep_blocks, DP, S, S_all_f, reinterpret_cast<double *>(&MR[0][0][0]));
            return;
        }
        sumA = 0.0; sumL = 0.0;
        // (32-bit tile arithmetic: W < 2^31 for every n whose strips fit the 8 GiB cap)
        constexpr int TPP = SYM_PR / 64;
        const int NT = int((n + 63) / 64), T = int(n_panels);
        auto prefix = [&](int I) { return I * NT - TPP * (I * (I - 1) / 2); };      // tiles of the panels before I
        const int W = prefix(T);
        int t0 = int(vb) * bal_tpb, t1 = t0 + bal_tpb;
        if (t1 > W) t1 = W;
        if (t0 < t1) {
            // panel of tile t0: prefix(I) <= t0 < prefix(I + 1) -- from the quadratic's root, then exact
            const float hb = float(NT) + 0.5f * float(TPP);
            int I = int((hb - sqrtf(fmaxf(hb * hb - 2.0f * float(TPP) * float(t0), 0.0f))) / float(TPP));
            if (I < 0) I = 0;
            if (I > T - 1) I = T - 1;
            I = __builtin_amdgcn_readfirstlane(I);     // (the float root left it in a vector register: everything derived
                                                       //  from it -- column range, strip pointer, slot -- would follow)
            while (I > 0 && prefix(I) > t0) --I;
            while (I + 1 < T && prefix(I + 1) <= t0) ++I;
            while (t0 < t1) {
                const int pI = prefix(I), tiles = NT - TPP * I;
                const int lt = t0 - pI;
                int cnt = tiles - lt;
                if (cnt > t1 - t0) cnt = t1 - t0;
                const int64_t cb = int64_t(SYM_PR) * I + int64_t(lt) * 64;
                int64_t ce = cb + int64_t(cnt) * 64;
                if (ce > n) ce = n;
                if (cb < n) unit(I, cb, ce, int(vb) - int(unsigned(pI) / unsigned(bal_tpb)), -1);
                __syncthreads();                   // the next unit reuses the LDS tiles
                t0 += cnt; ++I;
            }
        }
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        double la = sumA * 0.69314718055994531, ll = sumL;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { la += __shfl_down(la, off, 64); ll += __shfl_down(ll, off, 64); }
        if (lane == 0) { red[wave][0] = la; red[wave][1] = ll; }
        __syncthreads();
        if (tid == 0) {
            loss_partial[2 * vb + 0] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
            loss_partial[2 * vb + 1] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
        }
        };
        if constexpr (PERSIST) {                       // (the guard's fallback launch: a few blocks walk all the ranges)
            for (unsigned vb = blockIdx.x; vb < n_vb; vb += gridDim.x) {
                vblock(vb);
                __syncthreads();                       // red and the LDS tiles are reused by the next virtual block
            }
        } else {
            vblock(blockIdx.x);
        }
    } else if constexpr (PERSIST) {
        const unsigned nx = n_panels + 1, total = nx * n_chunks;
        for (unsigned u = blockIdx.x; u < total; u += gridDim.x) {
            unit_xy(u % nx, u / nx);
            __syncthreads();                       // the next unit reuses the LDS tiles
        }
    } else {
        unit_xy(blockIdx.x, blockIdx.y);
    }
}

// O'_mirror[j][f] = sum over the panels left of j's panel, in panel order, of their strip entries.
// Block = one 64-column tile; thread (f, 4 columns); up to 8 strips in flight.
__global__ __launch_bounds__(256) void bce_mirror_reduce_kernel(const float *__restrict__ Wmir, int64_t n,
                                                               float *__restrict__ Omir /*[n][16]*/, int SYM_PR)
{
    const int64_t NP = (n + 63) / 64 * 64;
    const int64_t j0 = int64_t(blockIdx.x) * TJ;
    const int f = threadIdx.x >> 4, jq = (threadIdx.x & 15) * 4;
    const int64_t n_left = j0 / SYM_PR;       // panels 0 .. n_left - 1 hold a mirror tile for these columns
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    auto at = [&](int64_t I) {
        const int64_t de = SYM_PR * (I + 1);
        return *reinterpret_cast<const f32x4 *>(Wmir + sym_strip_offset(I, NP, SYM_PR) + (j0 - de) * 16 + f * TJ + jq);
    };
    int64_t I = 0;
    for (; I + 8 <= n_left; I += 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = at(I + u);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; I < n_left; ++I) acc += at(I);
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (j0 + jq + e < n) Omir[(j0 + jq + e) * 16 + f] = acc[e];
}

// ---------------------------------------------------------------------------
// Sparse part + assembly.  A group of LPR lanes owns node i (4 features per lane): in-edges from the CSR
// give the loss terms and G_s Zt, out-edges from the CSR of A^T give G_s^T Zt; then
//   dZ[i] = mask[i] * ( 2 (sum_splits O'[s][i] + S_all / 2) + sparse ) / N^2.
// ---------------------------------------------------------------------------
template <int VEC, int LPR, bool WITH_GRAD>
__global__ __launch_bounds__(256) void bce_edges_kernel(
    const float *__restrict__ Zt /*[n][DP] = Z (.) mask, zero padded*/, const float *__restrict__ mask, int64_t ldz,
    int64_t row_begin, int64_t n_local, int d, const int32_t *__restrict__ indptr,
    const int32_t *__restrict__ indices, const int32_t *__restrict__ t_indptr, const int32_t *__restrict__ t_indices,
    float pw, float inv_n2, const float *__restrict__ O_partial, int n_splits, int DP,
    const float *__restrict__ S_all_f, float *__restrict__ dZ, int64_t lddz, double *__restrict__ loss_partial,
    const float *__restrict__ O_mirror /*[n][16] or NULL*/, int64_t sym_cols_per_chunk, int SYM_PR,
    const int64_t *__restrict__ counts, const double *__restrict__ scal,
    const float *__restrict__ Wmir /*mirror strips: fold them here instead of reading O_mirror (LPR == 4), or NULL*/,
    unsigned *__restrict__ range_flag /*the dense launches' range guard, cleared here for the next call (or NULL)*/)
{
    static_assert(VEC == 4, "edge kernel reads the padded Zt rows as float4");
    if (range_flag != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *range_flag = 0u;
    // Block b owns rows [64 b, 64 b + 64).  When it also folds the mirror strips its work grows with b (column tile b
    // has a strip from every panel left of it: up to N / SYM_PR tiles of 4 KB): the heaviest blocks go FIRST, so the
    // light ones fill the tail (dispatch is in block-id order; lightest-first left the longest blocks for last --
    // Pubmed: 308 blocks on 256 CUs).  Same work per row, same sums.
    const unsigned bid = (LPR == 4 && Wmir != nullptr) ? gridDim.x - 1u - blockIdx.x : blockIdx.x;
    // ---- symmetric dense kernel, d <= 16: this block's 64 rows are exactly one 64-column tile of the mirror strips.
    //      The strip reduction of bce_mirror_reduce_kernel (same thread mapping, same panel order, same 8-deep
    //      batches: bit-identical sums) runs here and hands its result over through LDS: one kernel node and the
    //      O'_mirror round trip less, and its byte stream overlaps with the latency-bound edge walks of other blocks.
    __shared__ __attribute__((aligned(16))) float Om[LPR == 4 ? TJ * 20 : 4];
    const bool fold = LPR == 4 && Wmir != nullptr;
    if (LPR == 4 && fold) {
        const int64_t NP = (n_local + 63) / 64 * 64;
        const int64_t j0 = int64_t(bid) * TJ;
        const int f = threadIdx.x >> 4, jq = (threadIdx.x & 15) * 4;
        const int64_t n_left = j0 / SYM_PR;       // panels 0 .. n_left - 1 hold a mirror tile for these columns
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        auto at = [&](int64_t I) {
            const int64_t de = SYM_PR * (I + 1);
            return *reinterpret_cast<const f32x4 *>(Wmir + sym_strip_offset(I, NP, SYM_PR) + (j0 - de) * 16 + f * TJ + jq);
        };
        // (16 strips in flight and the ragged tail requested as one predicated batch were measured: Pubmed 28.6 -> 35.8 us,
        //  a ZINC batch 236 -> 245 us -- eight plain loads per batch it stays.  The fold is a byte stream, not a latency chain:
        //  as a launch of its own, cut into four pieces per column tile (1232 blocks), it takes 18.0 us = 97 MB at 5.4 TB/s
        //  and the edge kernel 13.2 us behind it -- 31.2 us against 28.6 us fused; profiles/r05_loss_fold.txt)
        int64_t I = 0;
        for (; I + 8 <= n_left; I += 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = at(I + u);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; I < n_left; ++I) acc += at(I);
#pragma unroll
        for (int e = 0; e < 4; ++e) Om[(jq + e) * 20 + f] = acc[e];
        __syncthreads();
    }
    if (scal) { pw = float(scal[0]); inv_n2 = float(scal[1]); }      // fixed-capacity batch: true sizes on the device
    const int64_t n_valid = counts ? counts[0] : n_local;
    __shared__ double red[4];
    constexpr int RPB = 256 / LPR;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lig = tid % LPR;
    const int64_t i = int64_t(bid) * RPB + tid / LPR;   // local row
    const int64_t gi = row_begin + i;                          // its row in Zt / mask
    const int64_t n = n_local;
    const int f0 = lig * VEC;
    const bool rowv = i < n;
    const bool fv = f0 < DP;     // lanes beyond the padded width idle (LPR is a power of two >= DP / 4)

    // Everything that does not depend on the neighbour gathers is requested first (row of Zt, both edge ranges,
    // the dense kernel's partial O' rows): the kernel is a chain of dependent round trips, not a byte stream.
    float zi[VEC], acc_in[VEC], acc_out[VEC];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    {
        const f32x4 t = (rowv && fv) ? *reinterpret_cast<const f32x4 *>(Zt + gi * DP + f0) : zero4;
#pragma unroll
        for (int q = 0; q < VEC; ++q) { acc_in[q] = acc_out[q] = 0.f; zi[q] = t[q]; }
    }
    // ... and what the LAST lines of the kernel need (the column sums' share of dZ, the dropout multipliers): asked for
    // there, inside `if (f < d)`, they came out as eight dependent round trips at the end of every wave
    float sall[VEC], mk[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
        const bool in = WITH_GRAD && rowv && f0 + q < d;
        sall[q] = WITH_GRAD ? S_all_f[in ? f0 + q : 0] : 0.f;
        mk[q] = (WITH_GRAD && mask != nullptr) ? mask[in ? gi * ldz + f0 + q : 0] : 1.f;
    }
    int32_t posA = rowv ? indptr[i] : 0;
    int32_t endA = rowv ? indptr[i + 1] : 0;
    int32_t posB = (WITH_GRAD && rowv) ? t_indptr[i] : 0;
    int32_t endB = (WITH_GRAD && rowv) ? t_indptr[i + 1] : 0;
    // rows with more than kLongRow edges in a list (the hubs of the real citation graphs: 100 - 170) are left out of the
    // lane group's own walk below -- 8 edges per pair of round trips: a 156-edge row held its block for 40 us -- and
    // taken by the whole wave afterwards, one row at a time
    constexpr int kLongRow = 16;
    const bool is_long = max(endA - posA, endB - posB) > kLongRow;
    const int32_t longA0 = posA, longA1 = endA, longB0 = posB, longB1 = endB;
    if (is_long) { endA = posA; endB = posB; }
    f32x4 osum = zero4;
    if (WITH_GRAD && rowv && fv) {
        const float *op = O_partial + i * DP + f0;
        const int64_t sstride = n * DP;
        if (O_mirror) {     // symmetric dense kernel: row i's panel wrote ceil((n - panel start) / chunk) partials
            if (sym_cols_per_chunk < 0) {      // balanced schedule, -sym_cols_per_chunk column tiles per block: the blocks
                const int64_t tpb = -sym_cols_per_chunk, I = i / SYM_PR, TPP = SYM_PR / 64, NT = (n + 63) / 64;   // that hold tiles of panel I
                const int64_t pI = I * NT - TPP * (I * (I - 1) / 2), pI1 = pI + NT - TPP * I;
                n_splits = int((pI1 - 1) / tpb - pI / tpb + 1);
            } else {
                const int64_t rem = n - (i / SYM_PR) * SYM_PR;
                n_splits = int((rem + sym_cols_per_chunk - 1) / sym_cols_per_chunk);
            }
            osum = fold ? *reinterpret_cast<const f32x4 *>(&Om[(tid / LPR) * 20 + f0])
                        : *reinterpret_cast<const f32x4 *>(O_mirror + i * DP + f0);
        }
        int sp = 0;
        for (; sp + 8 <= n_splits; sp += 8) {        // 8 independent loads per trip, added in split order
            f32x4 o[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) o[u] = *reinterpret_cast<const f32x4 *>(op + (sp + u) * sstride);
#pragma unroll
            for (int u = 0; u < 8; ++u) osum += o[u];
        }
        if (sp < n_splits) {                         // tail: one more trip, clamped loads, selected to zero
            f32x4 o[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int su = sp + u < n_splits ? sp + u : n_splits - 1;
                o[u] = *reinterpret_cast<const f32x4 *>(op + su * sstride);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) osum += (sp + u < n_splits) ? o[u] : zero4;
        }
    }
    double lsum = 0.0;
    // In-edges (CSR: loss + G_s Zt) and out-edges (CSR of A^T: G_s^T Zt) of node i advance TOGETHER, NE edges of
    // each per trip: the ids come from two coalesced loads per list, the 2 NE neighbour rows are all in flight
    // before the first dot product (one memory round trip per trip for both lists; rows of <= 8 edges: one trip).
    constexpr int NE = LPR >= 4 ? 8 : 4;
    const int glane0 = (lane / LPR) * LPR;
    while (posA < endA || posB < endB) {
        int32_t mineA[NE / 4], mineB[NE / 4];
#pragma unroll
        for (int h = 0; h < NE / 4; ++h) {
            const int32_t eA = posA + 4 * h + (LPR >= 4 ? (lig & 3) : 0), eB = posB + 4 * h + (LPR >= 4 ? (lig & 3) : 0);
            mineA[h] = eA < endA ? indices[eA] : 0;
            mineB[h] = (WITH_GRAD && eB < endB) ? t_indices[eB] : 0;
        }
        float zj[2 * NE][VEC], dot[2 * NE];
#pragma unroll
        for (int u = 0; u < 2 * NE; ++u) {
            const bool second = u >= NE;
            if (second && !WITH_GRAD) { dot[u] = 0.f; continue; }
            const int32_t pos = second ? posB : posA, end = second ? endB : endA;
            const int uu = u % NE;
            const int32_t mine = second ? mineB[uu / 4] : mineA[uu / 4];
            const int32_t j = LPR >= 4 ? __shfl(mine, glane0 + (uu & 3), 64)
                                       : (pos + uu < end ? (second ? t_indices : indices)[pos + uu] : 0);
            const bool ev = pos + uu < end;
            dot[u] = 0.f;
            const f32x4 t = (ev && fv) ? *reinterpret_cast<const f32x4 *>(Zt + int64_t(j) * DP + f0) : zero4;
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                zj[u][q] = t[q];
                dot[u] = fmaf(zi[q], t[q], dot[u]);
            }
        }
#pragma unroll
        for (int off = LPR / 2; off > 0; off >>= 1)
#pragma unroll
            for (int u = 0; u < (WITH_GRAD ? 2 * NE : NE); ++u) dot[u] += __shfl_xor(dot[u], off, 64);
#pragma unroll
        for (int u = 0; u < (WITH_GRAD ? 2 * NE : NE); ++u) {
            const bool second = u >= NE;
            const int32_t pos = second ? posB : posA, end = second ? endB : endA;
            if (pos + (u % NE) < end) {
                const float x = dot[u];
                float spn, sgn;               // softplus(-x), sigmoid(-x)
                softplus_sigmoid(-x, spn, sgn);
                const float sg = 1.0f - sgn;  // sigmoid(x)
                if (!second && lig == 0) lsum += double(-x + (pw - 1.0f) * spn);
                const float c = (pw - 1.0f) * sg - pw;
#pragma unroll
                for (int q = 0; q < VEC; ++q) {
                    if (second) acc_out[q] = fmaf(c, zj[u][q], acc_out[q]);
                    else acc_in[q] = fmaf(c, zj[u][q], acc_in[q]);
                }
            }
        }
        posA = min(posA + NE, endA);
        posB = min(posB + NE, endB);
    }
    // ---- long rows: CH edges of each list per trip (ids: one coalesced load per list), lane group g takes ids g,
    //      g + G, ... -- up to 4 of each list, 8 rows of Zt in flight per lane --, its shares of the two sums meet in a
    //      butterfly over the lane bits above the group; the loss terms stay where they were computed (lsum is summed
    //      over the wave below anyway).  Same terms as the walk above, added in another order.
    {
        // (requesting the first long row's first ids in front of the short rows' walk: measured, no gain)
        constexpr int G = 64 / LPR, CU = LPR < 4 ? LPR : 4, CH = CU * G;      // CH <= 64: one id per lane
        unsigned long long todo = __builtin_amdgcn_ballot_w64(is_long && lig == 0);
        const int grp = lane / LPR;
        while (todo != 0) {                                            // scalar
            const int hl = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int32_t eA = __builtin_amdgcn_readlane(longA1, hl), eB = __builtin_amdgcn_readlane(longB1, hl);
            int32_t a0 = __builtin_amdgcn_readlane(longA0, hl), b0 = __builtin_amdgcn_readlane(longB0, hl);
            float zr[VEC], pin[VEC], pout[VEC];
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                zr[q] = __shfl(zi[q], hl + lig, 64);
                pin[q] = pout[q] = 0.f;
            }
            for (; a0 < eA || b0 < eB; a0 += CH, b0 += CH) {
                const int32_t idA = (lane < CH && a0 + lane < eA) ? indices[a0 + lane] : 0;
                const int32_t idB = (WITH_GRAD && lane < CH && b0 + lane < eB) ? t_indices[b0 + lane] : 0;
                float zj[2 * CU][VEC], dot[2 * CU];
#pragma unroll
                for (int u = 0; u < 2 * CU; ++u) {
                    const bool second = u >= CU;
                    dot[u] = 0.f;
                    if (second && !WITH_GRAD) continue;
                    const int slot = grp + G * (u % CU);
                    const int32_t j = __shfl(second ? idB : idA, slot, 64);
                    const bool ev = (second ? b0 : a0) + slot < (second ? eB : eA);
                    const f32x4 t = (ev && fv) ? *reinterpret_cast<const f32x4 *>(Zt + int64_t(j) * DP + f0) : zero4;
#pragma unroll
                    for (int q = 0; q < VEC; ++q) {
                        zj[u][q] = t[q];
                        dot[u] = fmaf(zr[q], t[q], dot[u]);
                    }
                }
#pragma unroll
                for (int off = LPR / 2; off > 0; off >>= 1)
#pragma unroll
                    for (int u = 0; u < (WITH_GRAD ? 2 * CU : CU); ++u) dot[u] += __shfl_xor(dot[u], off, 64);
#pragma unroll
                for (int u = 0; u < (WITH_GRAD ? 2 * CU : CU); ++u) {
                    const bool second = u >= CU;
                    const int slot = grp + G * (u % CU);
                    if ((second ? b0 : a0) + slot < (second ? eB : eA)) {
                        const float x = dot[u];
                        float spn, sgn;
                        softplus_sigmoid(-x, spn, sgn);
                        const float sg = 1.0f - sgn;
                        if (!second && lig == 0) lsum += double(-x + (pw - 1.0f) * spn);
                        const float c = (pw - 1.0f) * sg - pw;
#pragma unroll
                        for (int q = 0; q < VEC; ++q) {
                            if (second) pout[q] = fmaf(c, zj[u][q], pout[q]);
                            else pin[q] = fmaf(c, zj[u][q], pin[q]);
                        }
                    }
                }
            }
#pragma unroll
            for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
                for (int q = 0; q < VEC; ++q) {
                    pin[q] += __shfl_xor(pin[q], off, 64);
                    if (WITH_GRAD) pout[q] += __shfl_xor(pout[q], off, 64);
                }
            if ((lane & ~(LPR - 1)) == hl) {
#pragma unroll
                for (int q = 0; q < VEC; ++q) { acc_in[q] += pin[q]; acc_out[q] += pout[q]; }
            }
        }
    }
    if (WITH_GRAD && rowv) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            const int f = f0 + q;
            if (f < d) {
                const float o = 0.5f * sall[q] + osum[q];
                float v = (2.0f * o + (acc_in[q] + acc_out[q])) * inv_n2;
                if (mask) v *= mk[q];
                if (i >= n_valid) v = 0.f;                     // padding rows receive no gradient
                dZ[i * lddz + f] = v;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lsum += __shfl_down(lsum, off, 64);
    if (lane == 0) red[wave] = lsum;
    __syncthreads();
    if (tid == 0) loss_partial[bid] = (red[0] + red[1]) + (red[2] + red[3]);
}

// single block: ordered sums of all partials -> mean
//   N^2 loss = 1/2 S_win.S_all + 1/2 sum|x| + ln2 (sum log2(t) - pad_cols * n_local) + edges
__global__ __launch_bounds__(1024) void bce_finalize_kernel(const gae_bce_tail t)
{
    __shared__ double red[3][16];
    gae::bce_finalize_block<1024>(t, red);
}

struct BcePlan {
    int64_t row_blocks, n_splits, cols_per_split, edge_blocks, prep_blocks;
    int KS, DP, LPR, VEC, RI;
    int64_t o_bytes, zt_bytes, zh_bytes, cs_bytes, s_bytes, n_dense, total_bytes;
    double pad_terms;
    bool sym;                     // symmetric dense kernel (full square, d <= 16, bf16x3 products)
    int sym_pr;                   // its panel height (rows)
    int bal_tpb;                  // balanced schedule: column tiles per block (0 = the 2-D grid)
    int64_t bal_blocks;           // ... and the blocks that have tiles
    int64_t wmir_bytes, omir_bytes;
};

inline int64_t align256(int64_t x) { return (x + 255) / 256 * 256; }

bool bce_plan(int64_t n, int64_t n_local, int64_t d, bool vec_ok, BcePlan &p)
{
    const int ri = (g_bce_ri == 1 || g_bce_ri == 4) ? g_bce_ri : 2;
    const int64_t ROWS_PER_BLOCK = 64 * ri;
    p.RI = ri;
    if (d > 64) return false;
    p.KS = int((d + 15) / 16);
    if (p.KS == 3) p.KS = 4;
    if (p.KS < 1) p.KS = 1;
    p.DP = p.KS * 16;
    p.row_blocks = (n_local + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    if (p.row_blocks < 1) p.row_blocks = 1;
    int64_t col_tiles = (n + TJ - 1) / TJ;
    if (col_tiles < 1) col_tiles = 1;
    int64_t want = (g_bce_grid + p.row_blocks - 1) / p.row_blocks;  // ~8 blocks per CU
    if (want > col_tiles) want = col_tiles;
    if (want > 32) want = 32;                                 // each split is one more partial O' per row to add
    if (want < 1) want = 1;
    const int64_t tiles_per_split = (col_tiles + want - 1) / want;
    p.cols_per_split = tiles_per_split * TJ;
    p.n_splits = (col_tiles + tiles_per_split - 1) / tiles_per_split;
    p.pad_terms = double(col_tiles * TJ - n) * double(n_local);   // zero columns: log2(1 + e^0) = 1 each
    (void)vec_ok;
    p.VEC = 4;
    const int nvec = p.DP / 4;   // the edge kernel reads the padded Zt rows
    int lpr = 1;
    while (lpr < nvec) lpr <<= 1;
    p.LPR = lpr;
    p.edge_blocks = (n_local + (256 / lpr) - 1) / (256 / lpr);
    if (p.edge_blocks < 1) p.edge_blocks = 1;
    p.prep_blocks = (n + PREP_ROWS - 1) / PREP_ROWS;
    // Symmetric form: every tile right of the block diagonal is evaluated once for both halves.
    p.sym = false;
    p.sym_pr = 128;
    p.bal_tpb = 0;
    p.bal_blocks = 0;
    p.wmir_bytes = p.omir_bytes = 0;
    // balanced schedule (round 6): every block the same number of column tiles, one resident round.  It removes the
    // quantisation of the 2-D grid into rounds of blocks, which is what kept 256-row panels from paying below ~32 k rows:
    // with it they win from ~5 k rows on (whole loss sequence, 2-D grid + 128-row panels -> balanced + 256-row panels:
    // N = 8000 63.3 -> 53.8 us, Pubmed 161.8 -> 144.4 us, 26 k 247.7 -> 220.8 us, 40 k 521.8 -> 466.4 us; N = 5000 43.7 ->
    // 42.5 us; below that the full-square kernel wins: N = 3327 32.7 vs 37.3 us).  Launches of ten rounds and more keep the
    // 2-D grid (N = 95 k: 2470 vs 2520 us -- the balanced instantiation of the 256-row kernel spills 4 VGPRs).
    const bool bal = g_bce_sym_bal == 2 || (g_bce_sym_bal == 1 && n < 65536);
    if (g_bce_sym && n_local == n && p.KS == 1 && g_bce_s_bf16 && g_bce_pv_bf16 &&
        n >= (g_bce_sym > 1 ? 512 : (bal ? 5120 : 8192))) {   // below that the extra launch costs more than it saves
        const int64_t SYM_PR = (g_bce_sym_ri == 4 || (g_bce_sym_ri == 0 && (n >= 32768 || bal))) ? 256 : 128;
        p.sym_pr = int(SYM_PR);
        const int64_t T = (n + SYM_PR - 1) / SYM_PR, NP = (n + 63) / 64 * 64;
        int64_t chunks = (g_bce_sym_grid + T - 1) / T; // half of the (panel, chunk) grid is live
        if (chunks > 28) chunks = 28;                  // every chunk is one more partial O' per row to write and add
        if (chunks > col_tiles) chunks = col_tiles;
        if (chunks < 1) chunks = 1;
        int64_t cpc = ((n + chunks - 1) / chunks + TJ - 1) / TJ * TJ;
        if (g_bce_sym_tiles > 0) {
            cpc = int64_t(g_bce_sym_tiles) * TJ;
        } else {
            // Launches of a few rounds of blocks (Pubmed: ~2000 live blocks on 768 slots): pick the tiles per block,
            // within a quarter of the value above, that fills the last round best (12 -> 11 tiles: 2060 -> 2250
            // blocks = 2.93 rounds instead of 2.68; 169 -> 166 us).  Long launches keep the value: fewer partials.
            const int64_t slots = int64_t(kChipCus) * (SYM_PR == 128 ? 3 : 2);
            auto live_blocks = [&](int64_t t) {
                int64_t L = 0;
                for (int64_t I = 0; I < T; ++I) L += (n - SYM_PR * I + t * TJ - 1) / (t * TJ);
                return L;
            };
            const int64_t t0 = cpc / TJ;
            if ((live_blocks(t0) + slots - 1) / slots <= 6) {
                double best = -1.0;
                int64_t best_t = t0;
                for (int64_t t = t0 - t0 / 4; t <= t0 + t0 / 4; ++t) {
                    if (t < 1) continue;
                    const int64_t L = live_blocks(t), rounds = (L + slots - 1) / slots;
                    const double fill = double(L) / double(rounds * slots) - 0.002 * double(t > t0 ? t - t0 : t0 - t);
                    if (fill > best) { best = fill; best_t = t; }
                }
                cpc = best_t * TJ;
            }
        }
        const int64_t last_len = NP - SYM_PR * T;      // strip length of the last panel (<= 0: it has no strip)
        const int64_t wfloats = sym_strip_offset(T - 1, NP, SYM_PR) + 16 * (last_len > 0 ? last_len : 0);
        if (wfloats * 4 <= (int64_t(8) << 30)) {
            p.sym = true;
            p.row_blocks = T;
            p.cols_per_split = cpc;
            p.n_splits = (n + cpc - 1) / cpc;
            p.pad_terms = 0.0;                         // padded columns are corrected inside the kernel
            p.wmir_bytes = align256(wfloats * 4);
            p.omir_bytes = align256(n * 16 * 4);
            if (bal) {
                // balanced schedule (see the kernel): one sequence of W column tiles, tpb of them per block, all blocks
                // resident at once; n_splits = the most partials any panel gets (the edge kernel derives a row's count
                // from the same integers)
                const int64_t TPP = SYM_PR / TJ, NT = NP / TJ;
                auto prefix = [&](int64_t I) { return I * NT - TPP * (I * (I - 1) / 2); };
                const int64_t W = prefix(T), slots = int64_t(kChipCus) * (SYM_PR == 128 ? 3 : 2);
                const int64_t tpb = (W + slots - 1) / slots;
                int64_t most = 1;
                for (int64_t I = 0; I < T; ++I) {
                    const int64_t pieces = (prefix(I + 1) - 1) / tpb - prefix(I) / tpb + 1;
                    if (pieces > most) most = pieces;
                }
                p.bal_tpb = int(tpb);
                p.bal_blocks = (W + tpb - 1) / tpb;
                p.n_splits = most;
            }
        }
    }
    p.o_bytes = align256(p.n_splits * n_local * p.DP * 4);
    p.zt_bytes = align256(n * p.DP * 4);
    p.zh_bytes = align256(n * p.DP * 2);
    p.cs_bytes = align256(((n + 15) / 16 + 1) * 2 * p.DP * 8);    // up to one partial per 16 rows (a producer kernel's blocks: gae_x_gcn_layer_fused_prep)
    p.s_bytes = align256(2 * p.DP * 8 + p.DP * 4 + 4 * 8);       // column sums (double x 2, float) + 3 scalars
    p.n_dense = p.bal_tpb ? p.bal_blocks : p.row_blocks * p.n_splits;
    p.total_bytes = p.o_bytes + p.zt_bytes + 2 * p.zh_bytes + p.cs_bytes + p.s_bytes +
                    align256((2 * p.n_dense + p.edge_blocks) * 8) + p.wmir_bytes + p.omir_bytes;
    return true;
}

template <bool WITH_GRAD>
int launch_dense(const BcePlan &p, const float *Zt, const unsigned short *Zhi, const unsigned short *Zlo, int64_t n,
                 int64_t row_begin, int64_t n_local, float *O, double *lp, const double *cs, double *S, float *S_all_f,
                 hipStream_t s)
{
    const dim3 grid(unsigned(p.row_blocks) + 1, unsigned(p.n_splits));   // + 1: the column-sum block
#define GAE_BD(KS, RI, MW, SB)                                                                                     \
    do {                                                                                                           \
        if (SB && g_bce_s_bf16 >= 2 && g_bce_pv_bf16 && WITH_GRAD && g_bce_sym_tr)                                  \
            hipLaunchKernelGGL((bce_dense_kernel<KS, WITH_GRAD, RI, MW, SB, SB, SB && WITH_GRAD, SB>), grid, dim3(256), 0, s, \
                               Zt, Zhi, Zlo, n, row_begin, n_local, p.cols_per_split, O, lp, cs, p.prep_blocks, S, \
                               S_all_f, unsigned(p.row_blocks));                                                   \
        else if (SB && g_bce_s_bf16 >= 2 && g_bce_pv_bf16)                                                          \
            hipLaunchKernelGGL((bce_dense_kernel<KS, WITH_GRAD, RI, MW, SB, SB, false, SB>), grid, dim3(256), 0, s, \
                               Zt, Zhi, Zlo, n, row_begin, n_local, p.cols_per_split, O, lp, cs, p.prep_blocks, S, \
                               S_all_f, unsigned(p.row_blocks));                                                   \
        else if (SB && g_bce_s_bf16 >= 2)                                                                           \
            hipLaunchKernelGGL((bce_dense_kernel<KS, WITH_GRAD, RI, MW, SB, false, false, SB>), grid, dim3(256), 0, s, \
                               Zt, Zhi, Zlo, n, row_begin, n_local, p.cols_per_split, O, lp, cs, p.prep_blocks, S, \
                               S_all_f, unsigned(p.row_blocks));                                                   \
        else if (SB && g_bce_pv_bf16 && WITH_GRAD && g_bce_sym_tr)                                                  \
            hipLaunchKernelGGL((bce_dense_kernel<KS, WITH_GRAD, RI, MW, SB, SB, SB && WITH_GRAD>), grid, dim3(256), 0, s, \
                               Zt, Zhi, Zlo, n, row_begin, n_local, p.cols_per_split, O, lp, cs, p.prep_blocks, S, \
                               S_all_f, unsigned(p.row_blocks));                                                   \
        else if (SB && g_bce_pv_bf16)                                                                              \
            hipLaunchKernelGGL((bce_dense_kernel<KS, WITH_GRAD, RI, MW, SB, SB>), grid, dim3(256), 0, s, Zt, Zhi,  \
                               Zlo, n, row_begin, n_local, p.cols_per_split, O, lp, cs, p.prep_blocks, S, S_all_f, \
                               unsigned(p.row_blocks));                                                            \
        else                                                                                                       \
            hipLaunchKernelGGL((bce_dense_kernel<KS, WITH_GRAD, RI, MW, SB, false>), grid, dim3(256), 0, s, Zt,    \
                               Zhi, Zlo, n, row_begin, n_local, p.cols_per_split, O, lp, cs, p.prep_blocks, S,     \
                               S_all_f, unsigned(p.row_blocks));                                                   \
    } while (0)
    const bool sb = g_bce_s_bf16 != 0;
    if (p.KS == 1) {
        if (p.RI == 1) { if (sb) GAE_BD(1, 1, 1, true); else GAE_BD(1, 1, 1, false); }
        else if (p.RI == 4) { if (sb) GAE_BD(1, 4, 1, true); else GAE_BD(1, 4, 1, false); }
        else { if (sb) GAE_BD(1, 2, 1, true); else GAE_BD(1, 2, 1, false); }
    } else if (p.KS == 2) {
        if (p.RI == 1) { if (sb) GAE_BD(2, 1, 1, true); else GAE_BD(2, 1, 1, false); }
        else { if (sb) GAE_BD(2, 2, 1, true); else GAE_BD(2, 2, 1, false); }
    } else {
        if (p.RI == 1) { if (sb) GAE_BD(4, 1, 1, true); else GAE_BD(4, 1, 1, false); }
        else { if (sb) GAE_BD(4, 2, 1, true); else GAE_BD(4, 2, 1, false); }
    }
#undef GAE_BD
    GAE_CHECK_LAUNCH("bce_dense_kernel");
    return GAE_OK;
}

template <int VEC, bool WITH_GRAD>
int launch_edges(const BcePlan &p, const float *Zt, const float *mask, int64_t ldz, int64_t row_begin, int64_t n, int d,
                 const int32_t *ip, const int32_t *ix, const int32_t *tp, const int32_t *tx, float pw, float inv_n2,
                 const float *O, const float *S_all_f, float *dZ, int64_t lddz, double *lp, const float *Omir,
                 const int64_t *counts, const double *scal, const float *Wmir, unsigned *range_flag, hipStream_t s)
{
#define GAE_EDGE(LPR)                                                                                              \
    hipLaunchKernelGGL((bce_edges_kernel<VEC, LPR, WITH_GRAD>), dim3(unsigned(p.edge_blocks)), dim3(256), 0, s, Zt, \
                       mask, ldz, row_begin, n, d, ip, ix, tp, tx, pw, inv_n2, O, int(p.n_splits), p.DP, S_all_f,  \
                       dZ, lddz, lp, Omir, (p.bal_tpb ? -int64_t(p.bal_tpb) : p.cols_per_split), p.sym_pr, counts, scal, Wmir, range_flag)
    switch (p.LPR) {
    case 1: GAE_EDGE(1); break;
    case 2: GAE_EDGE(2); break;
    case 4: GAE_EDGE(4); break;
    case 8: GAE_EDGE(8); break;
    default: GAE_EDGE(16); break;
    }
#undef GAE_EDGE
    GAE_CHECK_LAUNCH("bce_edges_kernel");
    return GAE_OK;
}

} // namespace

namespace gae {
Knob *bce_knob(const char *name)
{
    if (strcmp(name, "bce_ri") == 0) return &g_bce_ri;
    if (strcmp(name, "bce_s_bf16") == 0) return &g_bce_s_bf16;
    if (strcmp(name, "bce_pv_bf16") == 0) return &g_bce_pv_bf16;
    if (strcmp(name, "bce_sym") == 0) return &g_bce_sym;
    if (strcmp(name, "bce_sym_bal") == 0) return &g_bce_sym_bal;
    if (strcmp(name, "bce_last_kind") == 0) return &g_bce_last_kind;
    if (strcmp(name, "bce_sym_grid") == 0) return &g_bce_sym_grid;
    if (strcmp(name, "bce_grid") == 0) return &g_bce_grid;
    if (strcmp(name, "bce_sym_tiles") == 0) return &g_bce_sym_tiles;
    if (strcmp(name, "bce_fold_mirror") == 0) return &g_bce_fold_mirror;
    if (strcmp(name, "bce_strip_store") == 0) return &g_bce_strip_store;
    if (strcmp(name, "bce_sym_ri") == 0) return &g_bce_sym_ri;
    if (strcmp(name, "bce_sym_tr") == 0) return &g_bce_sym_tr;
    return nullptr;
}
} // namespace gae

extern "C" int64_t gae_decoder_bce_workspace_bytes(int64_t n, int64_t n_local, int64_t d)
{
    if (n < 0 || d < 0 || n_local < 0 || n_local > n) return GAE_E_SIZE;
    BcePlan p;
    if (!bce_plan(n, n_local, d, true, p)) return GAE_E_RANGE;
    return p.total_bytes + 256;
}

namespace {
// pairs the dense kernel leaves in its log2 sum: the symmetric kernel corrects its own pad columns (the n x n square
// remains), the full kernel counts whole column tiles
inline double bce_all_pairs(const BcePlan &p, int64_t n, int64_t n_local)
{
    return p.sym ? double(n) * double(n) : double(n_local) * double((n + TJ - 1) / TJ * TJ);
}

thread_local gae_bce_tail *t_tail_out = nullptr;     // gae_x_decoder_bce_defer_finalize: receives the next call's final reduction

int decoder_bce_impl(const float *Z, float *mask, int64_t ldz, int64_t n, int64_t d,
                     int64_t row_begin, int64_t n_local, const int32_t *indptr,
                     const int32_t *indices, const int32_t *t_indptr, const int32_t *t_indices,
                     float pos_weight, float dropout_p, uint64_t seed, uint64_t offset,
                     uint64_t *draw_dev, float *loss_out, float *dZ, int64_t lddz, void *workspace,
                     int64_t workspace_bytes, const int64_t *counts, void *stream, int64_t prepared_blocks = -1)
{
    // prepared_blocks >= 0: Zt / hi / lo / the column-sum partials (that many) / the padded-batch scalars are already
    // in the workspace (gae_x_gcn_layer_fused_prep wrote them): no prepare launch, Z is not read
    const bool prepared = prepared_blocks >= 0;
    GAE_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, GAE_E_RANGE, "gae_decoder_bce: dropout_p = %g outside [0, 1)",
                double(dropout_p));
    GAE_REQUIRE(dropout_p == 0.f || mask, GAE_E_NULL, "gae_decoder_bce: dropout_p > 0 needs the mask output buffer");
    GAE_REQUIRE(n > 0 && d > 0, GAE_E_SIZE, "gae_decoder_bce: n and d must be positive");
    GAE_REQUIRE(row_begin >= 0 && n_local >= 0 && row_begin + n_local <= n, GAE_E_SIZE,
                "gae_decoder_bce: row window outside [0, n)");
    GAE_REQUIRE(d <= 64, GAE_E_RANGE, "gae_decoder_bce: d = %lld > 64 not supported by the fused kernel",
                (long long)d);
    GAE_REQUIRE(n < (int64_t(1) << 31), GAE_E_SIZE, "gae_decoder_bce: n too large");
    GAE_REQUIRE(ldz >= d && (!dZ || lddz >= d), GAE_E_SIZE, "gae_decoder_bce: leading dimension too small");
    GAE_REQUIRE((Z || prepared) && loss_out && workspace && (n_local == 0 || indptr), GAE_E_NULL, "gae_decoder_bce: NULL pointer");
    GAE_REQUIRE(!dZ || n_local == 0 || t_indptr, GAE_E_NULL, "gae_decoder_bce: the gradient needs the CSR of A^T");
    BcePlan p;
    bce_plan(n, n_local, d, true, p);
    GAE_REQUIRE(workspace_bytes >= p.total_bytes, GAE_E_WORKSPACE, "gae_decoder_bce: workspace %lld < %lld bytes",
                (long long)workspace_bytes, (long long)p.total_bytes);
    if (prepared) {
        GAE_REQUIRE(prepared_blocks >= 1 && prepared_blocks * 2 * p.DP * 8 <= p.cs_bytes, GAE_E_RANGE,
                    "gae_x_decoder_bce_prepared: %lld column-sum partials do not fit the workspace", (long long)prepared_blocks);
        p.prep_blocks = prepared_blocks;
    }
    GAE_REQUIRE(gae::aligned16(workspace), GAE_E_ALIGN, "gae_decoder_bce: workspace not 16-byte aligned");
    hipStream_t s = gae::as_stream(stream);
    char *w = static_cast<char *>(workspace);
    float *O = reinterpret_cast<float *>(w); w += p.o_bytes;
    float *Zt = reinterpret_cast<float *>(w); w += p.zt_bytes;
    unsigned short *Zhi = reinterpret_cast<unsigned short *>(w); w += p.zh_bytes;
    unsigned short *Zlo = reinterpret_cast<unsigned short *>(w); w += p.zh_bytes;
    double *cs = reinterpret_cast<double *>(w); w += p.cs_bytes;
    double *S = reinterpret_cast<double *>(w);
    float *S_all_f = reinterpret_cast<float *>(w + 2 * p.DP * 8);
    double *scal = counts ? reinterpret_cast<double *>(w + 2 * p.DP * 8 + ((p.DP * 4 + 7) & ~7)) : nullptr;
    unsigned *range_flag = reinterpret_cast<unsigned *>(w + 2 * p.DP * 8 + ((p.DP * 4 + 7) & ~7) + 3 * 8);   // the 4th scalar slot
    w += p.s_bytes;
    double *lp = reinterpret_cast<double *>(w); w += align256((2 * p.n_dense + p.edge_blocks) * 8);
    float *Wmir = reinterpret_cast<float *>(w); w += p.wmir_bytes;
    float *Omir = reinterpret_cast<float *>(w);
    const double inv_n2 = 1.0 / (double(n) * double(n));
    if (n_local == 0) {
        GAE_HIP(hipMemsetAsync(loss_out, 0, sizeof(float), s));
        if (t_tail_out) { memset(t_tail_out, 0, sizeof(gae_bce_tail)); t_tail_out = nullptr; }   // nothing left to do
        return GAE_OK;
    }
    if (!prepared) {
        hipLaunchKernelGGL(bce_prepare_kernel, dim3(unsigned(p.prep_blocks)), dim3(256), 0, s, Z, mask, ldz, n, int(d), p.DP,
                           row_begin, row_begin + n_local, Zt, Zhi, Zlo, cs, dropout_p, 1.0f / (1.0f - dropout_p), seed,
                           offset, draw_dev, counts, scal, bce_all_pairs(p, n, n_local));
        GAE_CHECK_LAUNCH("bce_prepare_kernel");
    }
    int rc;
    g_bce_last_kind = p.sym ? (p.sym_pr == 256 ? 3 : 2) : 1;
    if (p.sym) {
        // 2-D grid: + 1 block column for the column sums; balanced schedule: 1-D, + 1 block for them
        const dim3 grid = p.bal_tpb ? dim3(unsigned(p.bal_blocks) + 1) : dim3(unsigned(p.row_blocks) + 1, unsigned(p.n_splits));
        static std::atomic<unsigned> call_counter{0};
        const unsigned ticket = (call_counter.fetch_add(1) * 2654435761u) | 0x80000001u;    // never 0 (= cleared)
        // fp16 pieces (knob bce_s_bf16 = 3, the default): the F16 launch reports out-of-range embeddings through
        // range_flag, the three-piece bf16 launch behind it runs only then (see the kernel)
#define GAE_SYM3(WG, R, T, S3V, F16V, FM, PERS, GRID)                                                                \
    do { if (p.bal_tpb) GAE_SYM4(WG, R, T, S3V, F16V, FM, PERS, true, GRID); else GAE_SYM4(WG, R, T, S3V, F16V, FM, PERS, false, GRID); } while (0)
#define GAE_SYM4(WG, R, T, S3V, F16V, FM, PERS, BALV, GRID)                                                          \
    hipLaunchKernelGGL((bce_dense_sym_kernel<WG, R, T, S3V, F16V, PERS, BALV>), GRID, dim3(256), 0, s, Zt, Zhi, Zlo, n, p.cols_per_split, O, Wmir, \
                       lp, cs, p.prep_blocks, S, S_all_f, unsigned(p.row_blocks),                                    \
                       g_bce_strip_store >= 0 ? g_bce_strip_store : (n >= 32768 ? 1 : 0), range_flag, FM, ticket,  \
                       (p.bal_tpb ? unsigned(p.bal_blocks) + 1u : unsigned(p.n_splits)), p.bal_tpb)
#define GAE_SYM(WG, R, T) do { if (g_bce_s_bf16 == 4) GAE_SYM3(WG, R, T, false, true, 0, false, grid);   /* experiments: fp16 pieces WITHOUT the range guard */ \
    else if (g_bce_s_bf16 >= 3) { GAE_SYM3(WG, R, T, false, true, 1, false, grid); GAE_SYM3(WG, R, T, true, false, 2, true, (p.bal_tpb ? dim3(64) : dim3(2 * kChipCus))); } \
    else if (g_bce_s_bf16 == 2) GAE_SYM3(WG, R, T, true, false, 0, false, grid); else GAE_SYM3(WG, R, T, false, false, 0, false, grid); } while (0)
        if (!dZ) { if (p.sym_pr == 256) GAE_SYM(false, 4, false); else GAE_SYM(false, 2, false); }
        else if (g_bce_sym_tr) { if (p.sym_pr == 256) GAE_SYM(true, 4, true); else GAE_SYM(true, 2, true); }
        else { if (p.sym_pr == 256) GAE_SYM(true, 4, false); else GAE_SYM(true, 2, false); }
#undef GAE_SYM
#undef GAE_SYM3
#undef GAE_SYM4
        GAE_CHECK_LAUNCH("bce_dense_sym_kernel");
        if (dZ && !(g_bce_fold_mirror && p.LPR == 4)) {     // otherwise the edge kernel folds the strips itself
            hipLaunchKernelGGL(bce_mirror_reduce_kernel, dim3(unsigned((n + TJ - 1) / TJ)), dim3(256), 0, s, Wmir, n,
                               Omir, p.sym_pr);
            GAE_CHECK_LAUNCH("bce_mirror_reduce_kernel");
        }
        rc = GAE_OK;
    } else {
        rc = dZ ? launch_dense<true>(p, Zt, Zhi, Zlo, n, row_begin, n_local, O, lp, cs, S, S_all_f, s)
                : launch_dense<false>(p, Zt, Zhi, Zlo, n, row_begin, n_local, O, lp, cs, S, S_all_f, s);
    }
    if (rc) return rc;
    double *lpe = lp + 2 * p.n_dense;
    rc = dZ ? launch_edges<4, true>(p, Zt, mask, ldz, row_begin, n_local, int(d), indptr, indices, t_indptr, t_indices,
                                    pos_weight, float(inv_n2), O, S_all_f, dZ, lddz, lpe, p.sym ? Omir : nullptr,
                                    counts, scal, (p.sym && g_bce_fold_mirror && p.LPR == 4) ? Wmir : nullptr,
                                    p.sym ? range_flag : nullptr, s)
            : launch_edges<4, false>(p, Zt, mask, ldz, row_begin, n_local, int(d), indptr, indices, t_indptr,
                                     t_indices, pos_weight, float(inv_n2), O, S_all_f, dZ, lddz, lpe, nullptr, counts,
                                     scal, nullptr, p.sym ? range_flag : nullptr, s);
    if (rc) return rc;
    gae_bce_tail tail;
    memset(&tail, 0, sizeof(tail));
    tail.dense_partial = lp; tail.n_dense = p.n_dense;
    tail.edge_partial = lpe; tail.n_edge = p.edge_blocks;
    tail.S = S; tail.DP = p.DP;
    tail.pad_terms = p.pad_terms; tail.inv_n2 = inv_n2;
    tail.loss_out = loss_out; tail.bump_draw = dropout_p > 0.f ? draw_dev : nullptr; tail.scal = scal;
    if (t_tail_out) {               // armed by gae_x_decoder_bce_defer_finalize: the caller launches (or hands on) the reduction
        *t_tail_out = tail;
        t_tail_out = nullptr;
        return GAE_OK;
    }
    hipLaunchKernelGGL(bce_finalize_kernel, dim3(1), dim3(1024), 0, s, tail);
    GAE_CHECK_LAUNCH("bce_finalize_kernel");
    return GAE_OK;
}
} // namespace

extern "C" int gae_x_decoder_bce_defer_finalize(gae_bce_tail *tail_out)
{
    t_tail_out = tail_out;          // NULL disarms
    return GAE_OK;
}

extern "C" int gae_x_decoder_bce_finalize(const gae_bce_tail *tail, void *stream)
{
    GAE_REQUIRE(tail != nullptr, GAE_E_NULL, "gae_x_decoder_bce_finalize: tail is NULL");
    if (tail->loss_out == nullptr) return GAE_OK;
    GAE_REQUIRE(tail->S && (tail->n_dense == 0 || tail->dense_partial) && (tail->n_edge == 0 || tail->edge_partial) &&
                    tail->DP > 0 && tail->n_dense >= 0 && tail->n_edge >= 0,
                GAE_E_NULL, "gae_x_decoder_bce_finalize: malformed tail (not one written by gae_decoder_bce*)");
    hipLaunchKernelGGL(bce_finalize_kernel, dim3(1), dim3(1024), 0, gae::as_stream(stream), *tail);
    GAE_CHECK_LAUNCH("bce_finalize_kernel");
    return GAE_OK;
}

extern "C" int gae_decoder_bce_rows(const float *Z, float *mask, int64_t ldz, int64_t n, int64_t d,
                                    int64_t row_begin, int64_t n_local, const int32_t *indptr,
                                    const int32_t *indices, const int32_t *t_indptr, const int32_t *t_indices,
                                    float pos_weight, float dropout_p, uint64_t seed, uint64_t offset,
                                    uint64_t *draw_dev, float *loss_out, float *dZ, int64_t lddz, void *workspace,
                                    int64_t workspace_bytes, void *stream)
{
    return decoder_bce_impl(Z, mask, ldz, n, d, row_begin, n_local, indptr, indices, t_indptr, t_indices, pos_weight,
                            dropout_p, seed, offset, draw_dev, loss_out, dZ, lddz, workspace, workspace_bytes, nullptr,
                            stream);
}

extern "C" int gae_decoder_bce_padded(const float *Z, float *mask, int64_t ldz, int64_t n_cap, int64_t d,
                                      const int32_t *indptr, const int32_t *indices, const int32_t *t_indptr,
                                      const int32_t *t_indices, const int64_t *counts_dev, float dropout_p,
                                      uint64_t seed, uint64_t offset, uint64_t *draw_dev, float *loss_out, float *dZ,
                                      int64_t lddz, void *workspace, int64_t workspace_bytes, void *stream)
{
    GAE_REQUIRE(counts_dev != nullptr, GAE_E_NULL, "gae_decoder_bce_padded: counts_dev is NULL");
    return decoder_bce_impl(Z, mask, ldz, n_cap, d, 0, n_cap, indptr, indices, t_indptr, t_indices, 0.f, dropout_p,
                            seed, offset, draw_dev, loss_out, dZ, lddz, workspace, workspace_bytes, counts_dev, stream);
}

extern "C" int gae_decoder_bce(const float *Z, float *mask, int64_t ldz, int64_t n, int64_t d,
                               const int32_t *indptr, const int32_t *indices, const int32_t *t_indptr,
                               const int32_t *t_indices, float pos_weight, float dropout_p, uint64_t seed,
                               uint64_t offset, uint64_t *draw_dev, float *loss_out, float *dZ, int64_t lddz,
                               void *workspace, int64_t workspace_bytes, void *stream)
{
    return gae_decoder_bce_rows(Z, mask, ldz, n, d, 0, n, indptr, indices, t_indptr, t_indices, pos_weight, dropout_p,
                                seed, offset, draw_dev, loss_out, dZ, lddz, workspace, workspace_bytes, stream);
}

// Where a PRODUCER kernel (gae_x_gcn_layer_fused_prep: the last encoder layer) puts what bce_prepare_kernel would
// compute, inside a workspace of gae_decoder_bce_workspace_bytes(n, n, d) bytes: gae_x_decoder_bce_prepared then starts
// at the dense kernel.
extern "C" int gae_x_decoder_bce_prep_layout(int64_t n, int64_t d, void *workspace, int64_t workspace_bytes,
                                           gae_bce_prep *out)
{
    GAE_REQUIRE(out && workspace, GAE_E_NULL, "gae_x_decoder_bce_prep_layout: NULL pointer");
    GAE_REQUIRE(n > 0 && d > 0 && d <= 64, GAE_E_RANGE, "gae_x_decoder_bce_prep_layout: n, d out of range");
    BcePlan p;
    bce_plan(n, n, d, true, p);
    GAE_REQUIRE(workspace_bytes >= p.total_bytes && gae::aligned16(workspace), GAE_E_WORKSPACE,
                "gae_x_decoder_bce_prep_layout: workspace %lld < %lld bytes (or not 16-byte aligned)",
                (long long)workspace_bytes, (long long)p.total_bytes);
    char *w = static_cast<char *>(workspace) + p.o_bytes;
    out->Zt = reinterpret_cast<float *>(w); w += p.zt_bytes;
    out->Zhi = reinterpret_cast<uint16_t *>(w); w += p.zh_bytes;
    out->Zlo = reinterpret_cast<uint16_t *>(w); w += p.zh_bytes;
    out->colsum_partial = reinterpret_cast<double *>(w); w += p.cs_bytes;
    out->scal = reinterpret_cast<double *>(w + 2 * p.DP * 8 + ((p.DP * 4 + 7) & ~7));
    out->all_pairs = bce_all_pairs(p, n, n);
    out->max_blocks = p.cs_bytes / (2 * p.DP * 8);
    out->DP = p.DP;
    out->reserved = 0;
    return GAE_OK;
}

extern "C" int gae_x_decoder_bce_prepared(float *mask, int64_t ldz, int64_t n, int64_t d, const int32_t *indptr,
                                        const int32_t *indices, const int32_t *t_indptr, const int32_t *t_indices,
                                        float pos_weight, const int64_t *counts_dev, float dropout_p,
                                        uint64_t *draw_dev, int64_t n_prep_blocks, float *loss_out, float *dZ,
                                        int64_t lddz, void *workspace, int64_t workspace_bytes, void *stream)
{
    GAE_REQUIRE(n_prep_blocks >= 1, GAE_E_RANGE, "gae_x_decoder_bce_prepared: n_prep_blocks must be what the producer reported");
    return decoder_bce_impl(nullptr, mask, ldz, n, d, 0, n, indptr, indices, t_indptr, t_indices, pos_weight, dropout_p,
                            0, 0, draw_dev, loss_out, dZ, lddz, workspace, workspace_bytes, counts_dev, stream,
                            n_prep_blocks);
}
