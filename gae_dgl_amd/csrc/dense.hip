// Dense contractions of the hot path on the fp32 matrix cores of gfx950.
//
//   K3/K5  Y  = act(M W^T + b)        NodeApplyModule.forward  gae_dgl/gae.py:13-16
//   K4     dW = dYm^T M, db, dM = dYm W   (autograd of the above)
//   K6     inverted-dropout mask         InnerProductDecoder    gae_dgl/gae.py:70
//   K7     logits = Zt Zt^T              InnerProductDecoder    gae_dgl/gae.py:71
//          dZ = ((G + G^T) Zt) (.) mask  (its autograd, dense parity path)
//
// All products use v_mfma_f32_32x32x2_f32: f32 in / f32 accumulate, exact f32
// (bitwise a k-ordered fmaf chain), which keeps the 1e-5 parity budget.
// Operand map of the instruction (wave64, lane l):
//   A[i = l & 31][k = l >> 5],  B[k = l >> 5][j = l & 31],
//   D reg r -> row (r & 3) + 8 (r >> 2) + 4 (l >> 5), col l & 31.
#include <string.h>

#include "common.h"

namespace {

using gae::kWave;
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { PRO_NONE = 0, PRO_RELU_MASK = 1, PRO_MUL_MASK = 2 };
gae::Knob g_linear_depth{3};  // "linear_depth": k-steps of loads in flight + 1 in that kernel (3, 4, 5)
gae::Knob g_linear_nw{0};     // "linear_nw": waves per block of that kernel (0 = 4, 8)
gae::Knob g_linear_f32x16{1}; // "linear_f32x16": exact-fp32 forward Linear through the 64-byte-piece loader (0 = gemm_stream_kernel)
gae::Knob g_gemm_stream{1};  // tuning knob: 0 = LDS-tiled kernel for every shape
gae::Knob g_gemm_rows{1};    // "gemm_rows": 0 = never use gemm_rows_kernel, 1 = operands of >= kGemmRowsMinN rows, 2 = wherever it applies
constexpr int64_t kGemmRowsMinN = 1 << 18;
gae::Knob g_linear_wlds{1};  // tuning knob: 0 = never use linear_fwd_wlds_kernel, 1 = where measured faster, 2 = wherever it applies
gae::Knob g_linear_bf16{0};  // tuning knob: 0 = exact fp32 forward Linear (default: embeddings within 2e-7 of fp64 instead of
                        // 7e-6, tools/encode_error.py), 1 = bf16 x 3 forward where measured faster (Pubmed L1 15.4 -> 13.0 us),
                        // 2 = wherever it applies
gae::Knob g_atb_bf16{1};     // "atb_bf16", weight-gradient products (dW kernel where the layout allows; gae_gcn2_bwd_dense): 0 = exact fp32
                             // MFMAs, 1 = three bf16 pieces per operand, six pairs (fp32-grade; default), 2 = two pieces, three pairs (16 bits)
gae::Knob g_atb_rows{0};     // tuning knob: rows per block (= per partial) of the dW kernel; 0 = auto (atb_plan)

// ---------------------------------------------------------------------------
// out[n, J] = epi( proA(A)[n, K] * proB(B) )      B given as [J, K] (BT) or [K, J]
// Block = 4 waves, 128 rows x (NT * 32) columns, K staged through LDS in tiles
// of 32 (row stride 33 floats: conflict-free ds_read_b32 fragment reads).
// ---------------------------------------------------------------------------
constexpr int BM = 128, KT = 32, LDS_LD = KT + 1;

template <int NT, bool BT, int PRO_A, bool MASK_B, bool VEC_A>
__global__ __launch_bounds__(256) void gemm_kernel(
    const float *__restrict__ A, int64_t lda, const float *__restrict__ Amask, int64_t ldam,
    const float *__restrict__ B, int64_t ldb, const float *__restrict__ Bmask, int64_t ldbm,
    const float *__restrict__ bias, int act, float *__restrict__ out, int64_t ldo, int64_t n, int K, int64_t J)
{
    __shared__ float As[BM * LDS_LD];
    __shared__ float Bs[NT * 32 * LDS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row0 = int64_t(blockIdx.x) * BM;
    const int64_t col0 = int64_t(blockIdx.y) * (NT * 32);

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    for (int k0 = 0; k0 < K; k0 += KT) {
        // ---- stage A tile [128][32]
        if (VEC_A) {
            const int c4 = (tid & 7) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = (tid >> 3) + 32 * i;
                const int64_t gr = row0 + r;
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (gr < n) {
                    const int k = k0 + c4;
                    if (k + 4 <= K) {
                        const float4 t = *reinterpret_cast<const float4 *>(A + gr * lda + k);
                        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
                        if (PRO_A != PRO_NONE) {
                            const float4 m = *reinterpret_cast<const float4 *>(Amask + gr * ldam + k);
                            if (PRO_A == PRO_RELU_MASK) {
                                v[0] = m.x > 0.f ? v[0] : 0.f; v[1] = m.y > 0.f ? v[1] : 0.f;
                                v[2] = m.z > 0.f ? v[2] : 0.f; v[3] = m.w > 0.f ? v[3] : 0.f;
                            } else {
                                v[0] *= m.x; v[1] *= m.y; v[2] *= m.z; v[3] *= m.w;
                            }
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (k + q < K) {
                                float a = A[gr * lda + k + q];
                                if (PRO_A == PRO_RELU_MASK) a = Amask[gr * ldam + k + q] > 0.f ? a : 0.f;
                                if (PRO_A == PRO_MUL_MASK) a *= Amask[gr * ldam + k + q];
                                v[q] = a;
                            }
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) As[r * LDS_LD + c4 + q] = v[q];
            }
        } else {
            const int c = tid & 31;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int r = (tid >> 5) + 8 * i;
                const int64_t gr = row0 + r;
                float a = 0.f;
                if (gr < n && k0 + c < K) {
                    a = A[gr * lda + k0 + c];
                    if (PRO_A == PRO_RELU_MASK) a = Amask[gr * ldam + k0 + c] > 0.f ? a : 0.f;
                    if (PRO_A == PRO_MUL_MASK) a *= Amask[gr * ldam + k0 + c];
                }
                As[r * LDS_LD + c] = a;
            }
        }
        // ---- stage B tile as Bs[j][k]
        if (BT) {
            const int c = tid & 31;
#pragma unroll
            for (int i = 0; i < NT * 4; ++i) {
                const int j = (tid >> 5) + 8 * i;
                const int64_t gj = col0 + j;
                float b = 0.f;
                if (gj < J && k0 + c < K) {
                    b = B[gj * ldb + k0 + c];
                    if (MASK_B) b *= Bmask[gj * ldbm + k0 + c];
                }
                Bs[j * LDS_LD + c] = b;
            }
        } else {
            // B[k][j]: consecutive lanes read consecutive j (coalesced), write transposed
            constexpr int JW = NT * 32;
#pragma unroll
            for (int i = 0; i < (JW * KT) / 256; ++i) {
                const int idx = tid + 256 * i;
                const int j = idx % JW, k = idx / JW;
                const int64_t gj = col0 + j;
                float b = 0.f;
                if (gj < J && k0 + k < K) {
                    b = B[int64_t(k0 + k) * ldb + gj];
                    if (MASK_B) b *= Bmask[int64_t(k0 + k) * ldbm + gj];
                }
                Bs[j * LDS_LD + k] = b;
            }
        }
        __syncthreads();
        const float *ap = As + (wave * 32 + (lane & 31)) * LDS_LD + (lane >> 5);
        const float *bp = Bs + (lane & 31) * LDS_LD + (lane >> 5);
#pragma unroll
        for (int kk = 0; kk < KT / 2; ++kk) {
            const float a = ap[2 * kk];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float b = bp[t * 32 * LDS_LD + 2 * kk];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // ---- epilogue
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int64_t col = col0 + t * 32 + (lane & 31);
        if (col >= J) continue;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = row0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < n) {
                float y = acc[t][r] + bv;
                if (act == GAE_ACT_RELU) y = y > 0.f ? y : 0.f;
                out[row * ldo + col] = y;
            }
        }
    }
}

template <int NT, bool BT, int PRO_A, bool MASK_B>
int launch_gemm(const float *A, int64_t lda, const float *Amask, int64_t ldam, const float *B, int64_t ldb,
                const float *Bmask, int64_t ldbm, const float *bias, int act, float *out, int64_t ldo, int64_t n,
                int K, int64_t J, hipStream_t s)
{
    const dim3 grid(unsigned((n + BM - 1) / BM), unsigned((J + NT * 32 - 1) / (NT * 32)));
    bool vec = (lda % 4 == 0) && gae::aligned16(A);
    if (PRO_A != PRO_NONE) vec = vec && (ldam % 4 == 0) && gae::aligned16(Amask);
    if (vec)
        hipLaunchKernelGGL((gemm_kernel<NT, BT, PRO_A, MASK_B, true>), grid, dim3(256), 0, s, A, lda, Amask, ldam, B,
                           ldb, Bmask, ldbm, bias, act, out, ldo, n, K, J);
    else
        hipLaunchKernelGGL((gemm_kernel<NT, BT, PRO_A, MASK_B, false>), grid, dim3(256), 0, s, A, lda, Amask, ldam,
                           B, ldb, Bmask, ldbm, bias, act, out, ldo, n, K, J);
    GAE_CHECK_LAUNCH("gemm_kernel");
    return GAE_OK;
}

template <int NT, bool BT, int PRO_A>
int launch_gemm_stream(const float *A, int64_t lda, const float *Amask, int64_t ldam, const float *B, int64_t ldb,
                       const float *bias, int act, float *out, int64_t ldo, int64_t n, int K, int64_t J, hipStream_t s,
                       float *split_ws, int64_t split_ws_floats);

template <bool BT, int PRO_A, bool MASK_B>
int dispatch_gemm(const float *A, int64_t lda, const float *Amask, int64_t ldam, const float *B, int64_t ldb,
                  const float *Bmask, int64_t ldbm, const float *bias, int act, float *out, int64_t ldo, int64_t n,
                  int K, int64_t J, hipStream_t s, float *split_ws = nullptr, int64_t split_ws_floats = 0)
{
    if (!MASK_B && J <= 128 && g_gemm_stream) {
        if (J <= 32)
            return launch_gemm_stream<1, BT, PRO_A>(A, lda, Amask, ldam, B, ldb, bias, act, out, ldo, n, K, J, s, split_ws,
                                                    split_ws_floats);
        if (J <= 64)
            return launch_gemm_stream<2, BT, PRO_A>(A, lda, Amask, ldam, B, ldb, bias, act, out, ldo, n, K, J, s, split_ws,
                                                    split_ws_floats);
        return launch_gemm_stream<4, BT, PRO_A>(A, lda, Amask, ldam, B, ldb, bias, act, out, ldo, n, K, J, s, split_ws,
                                                split_ws_floats);
    }
    if (J <= 32)
        return launch_gemm<1, BT, PRO_A, MASK_B>(A, lda, Amask, ldam, B, ldb, Bmask, ldbm, bias, act, out, ldo, n, K, J, s);
    if (J <= 64)
        return launch_gemm<2, BT, PRO_A, MASK_B>(A, lda, Amask, ldam, B, ldb, Bmask, ldbm, bias, act, out, ldo, n, K, J, s);
    return launch_gemm<4, BT, PRO_A, MASK_B>(A, lda, Amask, ldam, B, ldb, Bmask, ldbm, bias, act, out, ldo, n, K, J, s);
}

// ---------------------------------------------------------------------------
// Tall-skinny variant (J <= 128, the Linear layers): out[n, J] = epi(proA(A) * B).
// A is streamed exactly once: a block owns 32 rows, its 4 waves split K into
// contiguous quarters and read their A slab straight into MFMA fragments
// (lane (i, h) loads the 16 bytes A[i][k + 4h .. k + 4h + 3]; the 4 values feed
// 4 consecutive MFMAs, B uses the same k permutation), no LDS staging and no
// barrier in the K loop; the 4 partial accumulators meet in LDS once, in fixed
// order.  n / 32 blocks keep all CUs busy where 128-row tiles would not.
// ---------------------------------------------------------------------------
template <int NT, bool BT, int PRO_A, bool AVEC, bool BVEC>
__global__ __launch_bounds__(256) void gemm_stream_kernel(
    const float *__restrict__ A, int64_t lda, const float *__restrict__ Amask, int64_t ldam,
    const float *__restrict__ B, int64_t ldb, const float *__restrict__ bias, int act, float *__restrict__ out,
    int64_t ldo, int64_t n, int K, int J, int kb_per_split, int64_t split_stride)
{
    __shared__ float red[4 * NT * 16 * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int64_t row = int64_t(blockIdx.x) * 32 + i;
    const bool rv = row < n;
    // split-K over blockIdx.y (operands with few row tiles and a long K): this block owns k-blocks [kbs0, kbs1)
    // and writes its partial product to its own slab of `out`
    const int kblocks = (K + 7) / 8;
    const int kbs0 = blockIdx.y * kb_per_split, kbs1 = min(kbs0 + kb_per_split, kblocks);
    const int per = (kbs1 - kbs0 + 3) / 4;
    const int kb0 = kbs0 + wave * per, kb1 = min(kb0 + per, kbs1);
    out += blockIdx.y * split_stride;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const float *ap = A + row * lda;
    const float *mp = PRO_A != PRO_NONE ? Amask + row * ldam : nullptr;
    // Branch-free loads (indices clamped into range, zeroing by selects in compute()): see atb_partial_kernel.
    struct Stage { float a[4], m[4]; float b[NT][4]; };
    const int64_t rowc = rv ? row : n - 1;
    const float *apc = A + rowc * lda;
    const float *mpc = PRO_A != PRO_NONE ? Amask + rowc * ldam : nullptr;
    constexpr bool a4 = AVEC, b4 = BVEC;                // compile-time: a runtime choice between two load styles makes
                                                        // hipcc branch around the loads and drain vmcnt at the joins
    const int K4 = (K + 3) & ~3;                        // avec: lda % 4 == 0 and lda >= K, so [K, K4) is row padding
    auto load = [&](Stage &st, int kb) {
        const int k = kb * 8 + 4 * h;
        const int kc = k <= K4 - 4 ? k : K4 - 4;        // in-bounds start of a 4-wide vector load (== k when any
                                                        // of its elements is < K, so positions stay aligned)
        if (a4) {
            const float4 t4 = *reinterpret_cast<const float4 *>(apc + kc);
            st.a[0] = t4.x; st.a[1] = t4.y; st.a[2] = t4.z; st.a[3] = t4.w;
            if (PRO_A != PRO_NONE) {
                const float4 m4 = *reinterpret_cast<const float4 *>(mpc + kc);
                st.m[0] = m4.x; st.m[1] = m4.y; st.m[2] = m4.z; st.m[3] = m4.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int kq = k + q < K ? k + q : K - 1;
                st.a[q] = apc[kq];
                if (PRO_A != PRO_NONE) st.m[q] = mpc[kq];
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int j = t * 32 + i;
            const int jc = j < J ? j : J - 1;
            if (BT) {
                const float *bp = B + int64_t(jc) * ldb;
                if (b4) {
                    const float4 t4 = *reinterpret_cast<const float4 *>(bp + kc);
                    st.b[t][0] = t4.x; st.b[t][1] = t4.y; st.b[t][2] = t4.z; st.b[t][3] = t4.w;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) st.b[t][q] = bp[k + q < K ? k + q : K - 1];
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) st.b[t][q] = B[int64_t(k + q < K ? k + q : K - 1) * ldb + jc];
            }
        }
    };
    auto compute = [&](const Stage &st, int kb) {
        const int k = kb * 8 + 4 * h;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const bool jv = t * 32 + i < J;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool kv = kb < kb1 && k + q < K;
                float av = (rv && kv) ? st.a[q] : 0.f;
                if (PRO_A == PRO_RELU_MASK) av = st.m[q] > 0.f ? av : 0.f;
                if (PRO_A == PRO_MUL_MASK) av *= st.m[q];
                const float bv = (jv && kv) ? st.b[t][q] : 0.f;
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
            }
        }
    };
    // two k-blocks in flight ahead of the MFMAs (register double buffering, no LDS, no barrier)
    if (kb0 < kb1) {
        Stage s0, s1, s2;
        int kb = kb0;
#define GAE_PIN() __builtin_amdgcn_sched_barrier(0)
        load(s0, kb); load(s1, kb + 1); GAE_PIN();
        while (true) {
            load(s2, kb + 2); GAE_PIN(); compute(s0, kb); GAE_PIN(); if (++kb >= kb1) break;
            load(s0, kb + 2); GAE_PIN(); compute(s1, kb); GAE_PIN(); if (++kb >= kb1) break;
            load(s1, kb + 2); GAE_PIN(); compute(s2, kb); GAE_PIN(); if (++kb >= kb1) break;
        }
#undef GAE_PIN
    }
    // ---- fixed-order cross-wave reduction: red[wave][t][r][lane]
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave * NT + t) * 16 + r) * 64 + lane] = acc[t][r];
    __syncthreads();
    // wave w finalises accumulator registers 4w .. 4w+3  (rows 8w + {0..3} + 4h)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = t * 32 + i;
        const float bv = (bias && col < J) ? bias[col] : 0.f;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = wave * 4 + rr;
            float y = red[((0 * NT + t) * 16 + r) * 64 + lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) y += red[((w * NT + t) * 16 + r) * 64 + lane];
            const int64_t orow = int64_t(blockIdx.x) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (orow < n && col < J) {
                y += bv;
                if (act == GAE_ACT_RELU) y = y > 0.f ? y : 0.f;
                out[orow * ldo + col] = y;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Very tall operands with short rows (K <= 64, millions of rows: the RMAT layers 32 -> 32 -> 16, forward and dM):
// gemm_stream_kernel splits K over the 4 waves of a block, i.e. for K = 32 a block of 256 threads moves 4 KB of A,
// fetches as many bytes of B again and pays a barrier plus a 16 KB LDS reduction -- 2.9 TB/s on 2^24 rows where a
// device copy of the same bytes reaches 5.0.  Here a WAVE owns whole 32-row tiles: the B fragments of the whole K
// range are loaded once per wave and stay in registers over `tiles_per_wave` tiles, A is read with one 16-byte load
// per lane and k-block (all of a tile's loads issued together, the next tile's before this tile's MFMAs), no LDS, no
// barrier.  Products are the same exact-fp32 MFMAs; a row's k terms are added in ONE chain in k order (the split
// form adds four quarter chains): within the tolerance contract, not bit-identical to gemm_stream_kernel.
// ---------------------------------------------------------------------------
template <int NT, int KB, bool BT, int PRO_A, bool AVEC>
__global__ __launch_bounds__(256) void gemm_rows_kernel(
    const float *__restrict__ A, int64_t lda, const float *__restrict__ Amask, int64_t ldam,
    const float *__restrict__ B, int64_t ldb, const float *__restrict__ bias, int act, float *__restrict__ out,
    int64_t ldo, int64_t n, int K, int J, int tiles_per_wave)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int K4 = (K + 3) & ~3;
    float b[NT][KB][4], bv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int j = t * 32 + i;
        const int jc = j < J ? j : J - 1;
        bv[t] = (bias && j < J) ? bias[jc] : 0.f;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = kb * 8 + 4 * h + q;
                const int kc = k < K ? k : K - 1;
                const float v = BT ? B[int64_t(jc) * ldb + kc] : B[int64_t(kc) * ldb + jc];
                b[t][kb][q] = (k < K && j < J) ? v : 0.f;
            }
    }
    struct Stage { float a[KB][4], m[KB][4]; };
    auto load = [&](Stage &st, int64_t row0) {
        const int64_t row = row0 + i;
        const int64_t rowc = row < n ? row : n - 1;
        const float *ap = A + rowc * lda;
        const float *mp = PRO_A != PRO_NONE ? Amask + rowc * ldam : nullptr;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const int k = kb * 8 + 4 * h;
            if (AVEC) {
                const int kc = k <= K4 - 4 ? k : K4 - 4;
                const float4 t4 = *reinterpret_cast<const float4 *>(ap + kc);
                st.a[kb][0] = t4.x; st.a[kb][1] = t4.y; st.a[kb][2] = t4.z; st.a[kb][3] = t4.w;
                if (PRO_A != PRO_NONE) {
                    const float4 m4 = *reinterpret_cast<const float4 *>(mp + kc);
                    st.m[kb][0] = m4.x; st.m[kb][1] = m4.y; st.m[kb][2] = m4.z; st.m[kb][3] = m4.w;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int kq = k + q < K ? k + q : K - 1;
                    st.a[kb][q] = ap[kq];
                    if (PRO_A != PRO_NONE) st.m[kb][q] = mp[kq];
                }
            }
        }
    };
    auto tile = [&](const Stage &st, int64_t row0) {
        const bool rv = row0 + i < n;
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool kv = kb * 8 + 4 * h + q < K;
                float av = (rv && kv) ? st.a[kb][q] : 0.f;
                if (PRO_A == PRO_RELU_MASK) av = st.m[kb][q] > 0.f ? av : 0.f;
                if (PRO_A == PRO_MUL_MASK) av *= st.m[kb][q];
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[t][kb][q], acc[t], 0, 0, 0);
            }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int col = t * 32 + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t orow = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (orow < n && col < J) {
                    float y = acc[t][r] + bv[t];
                    if (act == GAE_ACT_RELU) y = y > 0.f ? y : 0.f;
                    out[orow * ldo + col] = y;
                }
            }
        }
    };
    const int64_t t0 = (int64_t(blockIdx.x) * 4 + wave) * tiles_per_wave;
    const int64_t n_tiles = (n + 31) / 32;
    if (t0 >= n_tiles) return;
    const int64_t t1 = t0 + tiles_per_wave < n_tiles ? t0 + tiles_per_wave : n_tiles;
    Stage s0, s1;
    load(s0, t0 * 32);
    for (int64_t tt = t0; tt < t1; tt += 2) {
        if (tt + 1 < t1) load(s1, (tt + 1) * 32);
        tile(s0, tt * 32);
        if (tt + 1 >= t1) break;
        if (tt + 2 < t1) load(s0, (tt + 2) * 32);
        tile(s1, (tt + 1) * 32);
    }
}

// ---------------------------------------------------------------------------
// Forward Linear with the weight slice in LDS: out[n, J <= 32] = act(A[n, K] W[J, K]^T + b).
// In gemm_stream_kernel the W fragments are fetched from global memory in the same "32 rows x 32 bytes" pattern as
// the A fragments, i.e. HALF of the kernel's line requests go to the (L2-resident) 64 KB weight matrix, and line
// requests per CU are what bounds it.  Here a block of 8 waves owns 64 rows (2 row tiles x 4 quarters of the
// block's K range); each quarter's slice of W (32 x <= 128 floats) is loaded ONCE with whole-line loads into LDS
// and read from there by both row tiles.  Split-K over blockIdx.y as in gemm_stream_kernel.
// ---------------------------------------------------------------------------
constexpr int kWq = 128;          // floats of K per wave quarter (16 k-blocks of 8)
constexpr int kLdw = kWq + 4;     // LDS row stride of a W slice (floats)

template <bool AVEC, bool WVEC>
__global__ __launch_bounds__(512) void linear_fwd_wlds_kernel(
    const float *__restrict__ A, int64_t lda, const float *__restrict__ W, int64_t ldw,
    const float *__restrict__ bias, int act, float *__restrict__ out, int64_t ldo, int64_t n, int K, int J,
    int kb_per_split, int64_t split_stride)
{
    __shared__ __attribute__((aligned(16))) float ws[4][32 * kLdw];     // W slices, one per K quarter; later: red
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int quarter = wave & 3, rt = wave >> 2;
    const int i = lane & 31, h = lane >> 5;
    const int64_t row = int64_t(blockIdx.x) * 64 + rt * 32 + i;
    const bool rv = row < n;
    const int kblocks = (K + 7) / 8;
    const int kbs0 = blockIdx.y * kb_per_split, kbs1 = min(kbs0 + kb_per_split, kblocks);
    const int per = (kbs1 - kbs0 + 3) / 4;                       // <= 16 (launcher)
    const int kb0 = kbs0 + quarter * per, kb1 = min(kb0 + per, kbs1);
    out += blockIdx.y * split_stride;
    const int K4 = (K + 3) & ~3;

    // ---- stage this quarter's W slice: waves (quarter, 0) and (quarter, 1) load 16 rows each, 2 rows x 512 B per
    // instruction; columns past K and rows past J are staged as zeros
    {
        float *dst = ws[quarter];
        const int c = (lane & 31) * 4, jl = lane >> 5;
        const int k = kb0 * 8 + c;
        const int kc = k <= K4 - 4 ? k : K4 - 4;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int j = rt * 16 + it * 2 + jl;
            const int jc = j < J ? j : J - 1;
            float4 v;
            if (WVEC) {
                v = *reinterpret_cast<const float4 *>(W + int64_t(jc) * ldw + kc);
            } else {                                  // odd K: the rows of W are not 16-byte aligned
                const float *wr = W + int64_t(jc) * ldw;
                v.x = wr[k + 0 < K ? k + 0 : K - 1]; v.y = wr[k + 1 < K ? k + 1 : K - 1];
                v.z = wr[k + 2 < K ? k + 2 : K - 1]; v.w = wr[k + 3 < K ? k + 3 : K - 1];
            }
            const bool jv = j < J && kb0 < kb1;
            v.x = (jv && k + 0 < K) ? v.x : 0.f; v.y = (jv && k + 1 < K) ? v.y : 0.f;
            v.z = (jv && k + 2 < K) ? v.z : 0.f; v.w = (jv && k + 3 < K) ? v.w : 0.f;
            *reinterpret_cast<float4 *>(dst + j * kLdw + c) = v;
        }
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int64_t rowc = rv ? row : n - 1;
    const float *apc = A + rowc * lda;
    struct Stage { float a[4]; };
    auto load = [&](Stage &st, int kb) {
        const int k = kb * 8 + 4 * h;
        if (AVEC) {
            const int kc = k <= K4 - 4 ? k : K4 - 4;
            const float4 t4 = *reinterpret_cast<const float4 *>(apc + kc);
            st.a[0] = t4.x; st.a[1] = t4.y; st.a[2] = t4.z; st.a[3] = t4.w;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) st.a[q] = apc[k + q < K ? k + q : K - 1];
        }
    };
    __syncthreads();                                  // the quarter's slice is complete (two waves wrote it)
    const float *wq = ws[quarter] + i * kLdw + 4 * h;
    auto compute = [&](const Stage &st, int kb) {
        const float4 b4 = *reinterpret_cast<const float4 *>(wq + (kb - kb0) * 8);   // zero beyond K / J
        const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
        const int k = kb * 8 + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool kv = kb < kb1 && k + q < K;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32((rv && kv) ? st.a[q] : 0.f, kv ? bv[q] : 0.f, acc, 0, 0, 0);
        }
    };
    if (kb0 < kb1) {
        Stage s0, s1, s2;
        int kb = kb0;
#define GAE_PIN() __builtin_amdgcn_sched_barrier(0)
        load(s0, kb); load(s1, kb + 1); GAE_PIN();
        while (true) {
            load(s2, kb + 2); GAE_PIN(); compute(s0, kb); GAE_PIN(); if (++kb >= kb1) break;
            load(s0, kb + 2); GAE_PIN(); compute(s1, kb); GAE_PIN(); if (++kb >= kb1) break;
            load(s1, kb + 2); GAE_PIN(); compute(s2, kb); GAE_PIN(); if (++kb >= kb1) break;
        }
#undef GAE_PIN
    }
    // ---- fixed-order reduction over the 4 quarters of each row tile; red[rt][quarter][r][lane] aliases ws
    __syncthreads();
    float *red = &ws[0][0];
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((rt * 4 + quarter) * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
    const float bv = (bias && i < J) ? bias[i] : 0.f;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {                  // wave (rt, quarter) finalises registers 4 quarter ..
        const int r = quarter * 4 + rr;
        float y = red[((rt * 4 + 0) * 16 + r) * 64 + lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) y += red[((rt * 4 + w) * 16 + r) * 64 + lane];
        const int64_t orow = int64_t(blockIdx.x) * 64 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (orow < n && i < J) {
            y += bv;
            if (act == GAE_ACT_RELU) y = y > 0.f ? y : 0.f;
            out[orow * ldo + i] = y;
        }
    }
}

// ---------------------------------------------------------------------------
// Forward Linear with bf16 x 3 products: out[n, J <= 32] = act(A[n, K] W[J, K]^T + b), 16-byte aligned rows of A.
// v_mfma_f32_16x16x16_bf16 wants lane (row = l15, k = 4 g + r): ONE float4 A[row][k0 + 4 g ..] per lane and 16-k
// step, i.e. an instruction reads 64 contiguous bytes of 16 rows -- half as many requests per 128-byte line as the
// 32-byte pieces of gemm_stream_kernel's fp32 fragments (4.3 L1->L2 requests per line there) -- and the matrix
// pipe does the products in a fifth of the fp32 MFMA time.  A block owns 32 rows (2 row tiles); its 4 waves split
// the block's K range (split-K over blockIdx.y as in gemm_stream_kernel) and meet in LDS in fixed order.
// ---------------------------------------------------------------------------
// EXACT = true: the same loader (64-byte row pieces: 2 requests per 128-byte line instead of the 4.3 of
// gemm_stream_kernel's 32-byte pieces) feeding v_mfma_f32_16x16x4_f32 -- exact fp32 products; the 4 values a lane
// holds go to 4 consecutive MFMAs, each of which contracts the k positions {4 g + r} of the 16-k step.
template <bool WVEC, bool EXACT = false, int NW = 4, int DEPTH = 3>
__global__ __launch_bounds__(64 * NW) void linear_fwd_pieces_kernel(
    const float *__restrict__ A, int64_t lda, const float *__restrict__ W, int64_t ldw,
    const float *__restrict__ bias, int act, float *__restrict__ out, int64_t ldo, int64_t n, int K, int J,
    int ks_per_split, int64_t split_stride)
{
    __shared__ float red[NW][16 * 64];      // NW waves split the block's K range
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int64_t row0 = int64_t(blockIdx.x) * 32;
    const int ksteps = (K + 15) / 16;
    const int kss0 = blockIdx.y * ks_per_split, kss1 = min(kss0 + ks_per_split, ksteps);
    const int per = (kss1 - kss0 + NW - 1) / NW;
    const int ks0 = kss0 + wave * per, ks1 = min(ks0 + per, kss1);
    out += blockIdx.y * split_stride;
    const int K4 = (K + 3) & ~3;

    gae::v4f acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = gae::v4f{0.f, 0.f, 0.f, 0.f};
    const float *ap[2], *wp[2];
    bool rv[2], jv[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int64_t r = row0 + 16 * q + l15;
        rv[q] = r < n;
        ap[q] = A + (rv[q] ? r : n - 1) * lda;
        const int j = 16 * q + l15;
        jv[q] = j < J;
        wp[q] = W + int64_t(jv[q] ? j : J - 1) * ldw;
    }
    struct Stage { float4 a[2], w[2]; };
    auto load = [&](Stage &st, int ks) {
        const int k = ks * 16 + 4 * g;
        const int kc = k <= K4 - 4 ? k : K4 - 4;          // in-bounds start of the vector loads (rows padded to K4)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            st.a[q] = *reinterpret_cast<const float4 *>(ap[q] + kc);
            if (WVEC) {
                st.w[q] = *reinterpret_cast<const float4 *>(wp[q] + kc);
            } else {
                // odd K: rows of W are only 4-byte aligned.  gfx950 loads an unaligned dwordx4; the last group of a
                // row starts at K - 4 instead (never past the end of W) and is shifted into place by selects
                typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
                const int ku = k <= K - 4 ? k : K - 4;
                const f4u l = *reinterpret_cast<const f4u *>(wp[q] + ku);
                const int sh = k - ku;                    // 0 except in the tail group (then 1..3; >= 4: all masked)
                st.w[q].x = sh == 0 ? l[0] : sh == 1 ? l[1] : sh == 2 ? l[2] : l[3];
                st.w[q].y = sh == 0 ? l[1] : sh == 1 ? l[2] : l[3];
                st.w[q].z = sh == 0 ? l[2] : l[3];
                st.w[q].w = l[3];
            }
        }
    };
    auto compute = [&](const Stage &st, int ks) {
        const int k = ks * 16 + 4 * g;
        const bool live = ks < ks1;
        const bool kv0 = live && k + 0 < K, kv1 = live && k + 1 < K, kv2 = live && k + 2 < K, kv3 = live && k + 3 < K;
        gae::v4s ah[2], al[2], wh[2], wl[2];
        gae::v4f avq[2], wvq[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const gae::v4f av = {(rv[q] && kv0) ? st.a[q].x : 0.f, (rv[q] && kv1) ? st.a[q].y : 0.f,
                                 (rv[q] && kv2) ? st.a[q].z : 0.f, (rv[q] && kv3) ? st.a[q].w : 0.f};
            const gae::v4f wv = {(jv[q] && kv0) ? st.w[q].x : 0.f, (jv[q] && kv1) ? st.w[q].y : 0.f,
                                 (jv[q] && kv2) ? st.w[q].z : 0.f, (jv[q] && kv3) ? st.w[q].w : 0.f};
            avq[q] = av; wvq[q] = wv;
            if (!EXACT) {
                gae::split_bf16x4(av, ah[q], al[q]);
                gae::split_bf16x4(wv, wh[q], wl[q]);
            }
        }
        if (EXACT) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(avq[mt][r], wvq[nt][r], acc[mt][nt], 0, 0, 0);
            return;
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al[mt], wh[nt], acc[mt][nt], 0, 0, 0);
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah[mt], wl[nt], acc[mt][nt], 0, 0, 0);
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah[mt], wh[nt], acc[mt][nt], 0, 0, 0);
            }
    };
    if (ks0 < ks1) {
        // ring of DEPTH stages: DEPTH - 1 k-steps of loads are in flight ahead of the MFMAs (a wave lives as long as
        // the launch -- 2.4 waves per SIMD on Pubmed -- so its time is steps x latency / depth)
        Stage st[DEPTH];
        int ks = ks0;
#define GAE_PIN() __builtin_amdgcn_sched_barrier(0)
#pragma unroll
        for (int d = 0; d < DEPTH - 1; ++d) load(st[d], ks + d);
        GAE_PIN();
        bool more = true;
        while (more) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                if (more) {
                    load(st[(d + DEPTH - 1) % DEPTH], ks + DEPTH - 1); GAE_PIN(); compute(st[d], ks); GAE_PIN();
                    more = ++ks < ks1;
                }
            }
        }
#undef GAE_PIN
    }
    // ---- fixed-order cross-wave reduction: red[wave][(mt, nt, r)][lane]; wave w finalises (mt, nt) = (w >> 1, w & 1)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][((mt * 2 + nt) * 4 + r) * 64 + lane] = acc[mt][nt][r];
    __syncthreads();
    if (wave >= 4) return;
    const int mt = wave >> 1, nt = wave & 1;
    const int j = 16 * nt + l15;                    // acc[mt][nt][r] = out[row0 + 16 mt + 4 g + r][16 nt + l15]
    const float bv = (bias && j < J) ? bias[j] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float y = red[0][((mt * 2 + nt) * 4 + r) * 64 + lane];
#pragma unroll
        for (int w = 1; w < NW; ++w) y += red[w][((mt * 2 + nt) * 4 + r) * 64 + lane];
        const int64_t orow = row0 + 16 * mt + 4 * g + r;
        if (orow < n && j < J) {
            y += bv;
            if (act == GAE_ACT_RELU) y = y > 0.f ? y : 0.f;
            out[orow * ldo + j] = y;
        }
    }
}

// out[e] = act(bias[e % J] + sum_s partial[s][e])  -- the second pass of the split-K forward Linear
__global__ __launch_bounds__(256) void split_reduce_kernel(const float *__restrict__ partial, int splits,
                                                           int64_t n_elems, int J, const float *__restrict__ bias,
                                                           int act, float *__restrict__ out, int64_t ldo)
{
    const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (e >= n_elems) return;
    float v[8];
    float y = 0.f;
    int sp = 0;
    for (; sp + 8 <= splits; sp += 8) {     // 8 independent loads per trip, added in split order
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = partial[(sp + u) * n_elems + e];
#pragma unroll
        for (int u = 0; u < 8; ++u) y += v[u];
    }
    for (; sp < splits; ++sp) y += partial[sp * n_elems + e];
    const int64_t r = e / J;
    const int c = int(e - r * J);
    if (bias) y += bias[c];
    if (act == GAE_ACT_RELU) y = y > 0.f ? y : 0.f;
    out[r * ldo + c] = y;
}

// split-K factor of the tall-skinny product: 1 unless the operand has too few 32-row tiles to keep every CU
// pulling on HBM (a CU sustains ~10 GB/s of line misses: Cora's 85 tiles alone reach 0.8 TB/s)
inline int gemm_stream_splits(int64_t n, int K)
{
    const int64_t tiles = (n + 31) / 32;
    const int kblocks = (K + 7) / 8;
    if (tiles >= 384 || kblocks < 64) return 1;
    int64_t sp = (768 + tiles - 1) / tiles;
    if (sp > kblocks / 16) sp = kblocks / 16;        // at least 16 k-blocks (4 per wave) per split
    if (sp > 32) sp = 32;
    return sp < 1 ? 1 : int(sp);
}

template <int NT, bool BT, int PRO_A>
int launch_gemm_stream(const float *A, int64_t lda, const float *Amask, int64_t ldam, const float *B, int64_t ldb,
                       const float *bias, int act, float *out, int64_t ldo, int64_t n, int K, int64_t J, hipStream_t s,
                       float *split_ws, int64_t split_ws_floats)
{
    bool avec = (lda % 4 == 0) && gae::aligned16(A) && K >= 1;
    if (PRO_A != PRO_NONE) avec = avec && (ldam % 4 == 0) && gae::aligned16(Amask);
    const bool bvec = BT && (ldb % 4 == 0) && gae::aligned16(B) && K >= 4 && (K % 4 == 0);
    // very tall operands with short rows: a wave owns whole row tiles, B stays in registers (gemm_rows_kernel)
    if (g_gemm_rows && NT <= 2 && K >= 1 && K <= 64 && n >= (g_gemm_rows > 1 ? 1 : kGemmRowsMinN)) {
        const int64_t tiles = (n + 31) / 32;
        int tpw = int(tiles / (int64_t(256) * 4 * 8));       // >= 8 waves per SIMD's worth of tiles before a wave takes two
        tpw = tpw < 1 ? 1 : tpw > 8 ? 8 : tpw;
        const dim3 grid(unsigned((tiles + 4 * tpw - 1) / (4 * tpw)));
#define GAE_GR(KBV, AV)                                                                                            \
    hipLaunchKernelGGL((gemm_rows_kernel<NT, KBV, BT, PRO_A, AV>), grid, dim3(256), 0, s, A, lda, Amask, ldam, B,  \
                       ldb, bias, act, out, ldo, n, K, int(J), tpw)
        const bool av = avec && lda >= ((K + 3) & ~3) && (PRO_A == PRO_NONE || ldam >= ((K + 3) & ~3));
        if (K <= 32) { if (av) GAE_GR(4, true); else GAE_GR(4, false); }
        else { if (av) GAE_GR(8, true); else GAE_GR(8, false); }
#undef GAE_GR
        GAE_CHECK_LAUNCH("gemm_rows_kernel");
        return GAE_OK;
    }
    int splits = split_ws ? gemm_stream_splits(n, K) : 1;
    if (splits > 1 && split_ws_floats < int64_t(splits) * n * J) splits = 1;
    const int kblocks = (K + 7) / 8;
    const int kbps = (kblocks + splits - 1) / splits;
    // forward Linear with bf16 x 3 products and 64-byte row pieces (linear_fwd_pieces_kernel)
    // (short unaligned rows of W keep gemm_stream_kernel: ZINC K = 39 measured 10.3 -> 13.2 us; knob 2 forces it)
    const bool wvec_ok = (ldb % 4 == 0) && gae::aligned16(B) && ldb >= ((K + 3) & ~3);
    const bool exact16 = g_linear_f32x16 && g_linear_bf16 == 0 && K >= 128;   // exact twin of the same loader
    if (NT == 1 && BT && PRO_A == PRO_NONE &&
        (exact16 || g_linear_bf16 > 1 || (g_linear_bf16 == 1 && (wvec_ok || K >= 128))) &&
        avec && lda >= ((K + 3) & ~3) && K >= 16 && n > 0) {
        const bool wvec = wvec_ok;
        const int ksteps = (K + 15) / 16;
        const int kspp = (ksteps + splits - 1) / splits;
        float *dst = splits > 1 ? split_ws : out;
        const dim3 grid(unsigned((n + 31) / 32), unsigned(splits));
#define GAE_LB(WV, EX, NW)                                                                                         \
    if (g_linear_depth == 5)                                                                                       \
        hipLaunchKernelGGL((linear_fwd_pieces_kernel<WV, EX, NW, 5>), grid, dim3(64 * NW), 0, s, A, lda, B, ldb,     \
                           splits > 1 ? nullptr : bias, splits > 1 ? int(GAE_ACT_IDENTITY) : act, dst,             \
                           splits > 1 ? J : ldo, n, K, int(J), kspp, n * J);                                       \
    else if (g_linear_depth == 4)                                                                                  \
        hipLaunchKernelGGL((linear_fwd_pieces_kernel<WV, EX, NW, 4>), grid, dim3(64 * NW), 0, s, A, lda, B, ldb,     \
                           splits > 1 ? nullptr : bias, splits > 1 ? int(GAE_ACT_IDENTITY) : act, dst,             \
                           splits > 1 ? J : ldo, n, K, int(J), kspp, n * J);                                       \
    else                                                                                                           \
    hipLaunchKernelGGL((linear_fwd_pieces_kernel<WV, EX, NW>), grid, dim3(64 * NW), 0, s, A, lda, B, ldb,            \
                       splits > 1 ? nullptr : bias, splits > 1 ? int(GAE_ACT_IDENTITY) : act, dst,                 \
                       splits > 1 ? J : ldo, n, K, int(J), kspp, n * J)
        const bool nw8 = g_linear_nw == 8;       // 8 waves per block: measured equal or slower on every layer shape
        if (exact16 && wvec) { if (nw8) GAE_LB(true, true, 8); else GAE_LB(true, true, 4); }
        else if (exact16) { if (nw8) GAE_LB(false, true, 8); else GAE_LB(false, true, 4); }
        else if (wvec) GAE_LB(true, false, 4);
        else GAE_LB(false, false, 4);
#undef GAE_LB
        GAE_CHECK_LAUNCH("linear_fwd_pieces_kernel");
        if (splits > 1) {
            const int64_t ne = n * J;
            hipLaunchKernelGGL(split_reduce_kernel, dim3(unsigned((ne + 255) / 256)), dim3(256), 0, s, split_ws, splits,
                               ne, int(J), bias, act, out, ldo);
            GAE_CHECK_LAUNCH("split_reduce_kernel");
        }
        return GAE_OK;
    }
    // forward Linear (W given as [J <= 32][K], no mask) with the weight slices in LDS (linear_fwd_wlds_kernel):
    // measured faster only for very long K under split-K (Citeseer 3327 x 3703: 32.7 -> 24.3 us); slower on
    // Pubmed (15.2 -> 16.7 us), Cora (15.2 -> 17.6 us) and the narrow ZINC layers, which keep gemm_stream_kernel
    if (NT == 1 && BT && PRO_A == PRO_NONE && kbps <= 64 && n > 0 &&
        (g_linear_wlds > 1 ? K >= 32 : (g_linear_wlds == 1 && splits > 1 && K >= 2048))) {
        float *dst = splits > 1 ? split_ws : out;
        const dim3 grid(unsigned((n + 63) / 64), unsigned(splits));
        const bool wvec = (ldb % 4 == 0) && gae::aligned16(B);
#define GAE_WL(AV, WV)                                                                                             \
    hipLaunchKernelGGL((linear_fwd_wlds_kernel<AV, WV>), grid, dim3(512), 0, s, A, lda, B, ldb,                    \
                       splits > 1 ? nullptr : bias, splits > 1 ? int(GAE_ACT_IDENTITY) : act, dst,                 \
                       splits > 1 ? J : ldo, n, K, int(J), kbps, n * J)
        if (avec && wvec) GAE_WL(true, true);
        else if (avec) GAE_WL(true, false);
        else if (wvec) GAE_WL(false, true);
        else GAE_WL(false, false);
#undef GAE_WL
        GAE_CHECK_LAUNCH("linear_fwd_wlds_kernel");
        if (splits > 1) {
            const int64_t ne = n * J;
            hipLaunchKernelGGL(split_reduce_kernel, dim3(unsigned((ne + 255) / 256)), dim3(256), 0, s, split_ws, splits,
                               ne, int(J), bias, act, out, ldo);
            GAE_CHECK_LAUNCH("split_reduce_kernel");
        }
        return GAE_OK;
    }
    const dim3 grid(unsigned((n + 31) / 32), unsigned(splits));
    float *dst = splits > 1 ? split_ws : out;
    const int64_t ldd = splits > 1 ? J : ldo;
    const float *bias1 = splits > 1 ? nullptr : bias;
    const int act1 = splits > 1 ? GAE_ACT_IDENTITY : act;
#define GAE_GS(AV, BV)                                                                                              \
    hipLaunchKernelGGL((gemm_stream_kernel<NT, BT, PRO_A, AV, BV>), grid, dim3(256), 0, s, A, lda, Amask, ldam, B, ldb, \
                       bias1, act1, dst, ldd, n, K, int(J), kbps, n * J)
    if (avec && bvec) GAE_GS(true, true);
    else if (avec) GAE_GS(true, false);
    else if (bvec) GAE_GS(false, true);
    else GAE_GS(false, false);
#undef GAE_GS
    GAE_CHECK_LAUNCH("gemm_stream_kernel");
    if (splits > 1) {
        const int64_t ne = n * J;
        hipLaunchKernelGGL(split_reduce_kernel, dim3(unsigned((ne + 255) / 256)), dim3(256), 0, s, split_ws, splits, ne,
                           int(J), bias, act, out, ldo);
        GAE_CHECK_LAUNCH("split_reduce_kernel");
    }
    return GAE_OK;
}

// ---------------------------------------------------------------------------
// out[O, I] (+)= sum_r proP(P)[r, O]^T Q[r, I]   -- reduction over rows.
// Both operands are read in their natural row-major layout straight into the
// MFMA fragments (lane = column, k = row parity): no LDS in the row loop.  A block
// of NW waves owns one row slot and one 32 x (IT*32) output tile: the waves split
// the slot's rows (NW = 8: twice the loads in flight per partial of the 4-wave
// form, and a second wave per SIMD whose MFMAs cover the other's memory waits),
// meet in LDS in a fixed tree order and write ONE partial; a second kernel sums
// the partials in slot order (deterministic, no float atomics).
// ---------------------------------------------------------------------------
template <int IT, int PRO_P, bool QVEC, int NW>
__global__ __launch_bounds__(NW * 64) void atb_partial_kernel(
    const float *__restrict__ P, int64_t ldp, const float *__restrict__ Pmask, int64_t ldpm,
    const float *__restrict__ Q, int64_t ldq, int64_t n, int O, int I, int64_t rows_per_slot,
    float *__restrict__ partial, int64_t slot_stride, int64_t colsum_offset)
{
    // IT == 4: lane j owns the 4 adjacent columns 4j..4j+3 of a 128-column group (one float4 per row, tile t =
    //          column 4j + t);  IT == 1: lane j owns column j of a 32-column group (narrow outputs).
    static_assert(IT == 4 || IT == 1, "IT is 1 or 4");
    static_assert(!QVEC || IT == 4, "the float4 path needs IT == 4");
    static_assert(NW == 4 || NW == 8, "4 or 8 waves");
    constexpr int PARK = IT * 16 * 64 + 64;          // accumulators + column sums of one parked wave
    __shared__ float red[NW / 2][PARK];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t slot = blockIdx.x;                 // one partial per BLOCK: its waves split the slot's rows
    const int j = lane & 31, h = lane >> 5;
    const int o = blockIdx.z * 32 + j;
    const int cb = blockIdx.y * (IT * 32) + (IT == 4 ? 4 * j : j);   // tile t of this lane is column cb + t
    const int64_t rows_per_wave = rows_per_slot / NW;                 // multiple of 8
    const int64_t r_begin = slot * rows_per_slot + wave * rows_per_wave;
    int64_t r_end = r_begin + rows_per_wave;
    if (r_end > n) r_end = n;

    f32x16 acc[IT];
#pragma unroll
    for (int t = 0; t < IT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float csum = 0.f;

    // Loads are BRANCH-FREE (row / column indices are clamped into range, out-of-range values are zeroed by
    // selects in compute()): hipcc otherwise branches around every guarded load and drains vmcnt(0) at each
    // join, which serialises the whole pipeline (cdna guide, "load everything, select afterwards").
    struct Stage { float a[4], m[4]; float b[4][IT]; };
    const int64_t r_last = r_end - 1;                       // r_begin < r_end whenever anything is loaded
    const int oc = o < O ? o : O - 1;
    const int I4 = (I + 3) & ~3;                            // QVEC: ldq >= I4, columns [I, I4) are row padding
    const int cbc = QVEC ? (cb + 4 <= I4 ? cb : I4 - 4) : 0;
    auto load = [&](Stage &st, int64_t r0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int64_t r = r0 + 2 * u + h;
            r = r < r_last ? r : r_last;
            st.a[u] = P[r * ldp + oc];
            if (PRO_P != PRO_NONE) st.m[u] = Pmask[r * ldpm + oc];
            if (QVEC) {
                const float4 q4 = *reinterpret_cast<const float4 *>(Q + r * ldq + cbc);
                st.b[u][0] = q4.x; st.b[u][1] = q4.y; st.b[u][2] = q4.z; st.b[u][3] = q4.w;
            } else {
#pragma unroll
                for (int t = 0; t < IT; ++t) {
                    const int c = cb + t < I ? cb + t : (I > 0 ? I - 1 : 0);
                    st.b[u][t] = I > 0 ? Q[r * ldq + c] : 0.f;
                }
            }
        }
    };
    auto compute = [&](const Stage &st, int64_t r0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool rv = r0 + 2 * u + h < r_end;
            float av = (rv && o < O) ? st.a[u] : 0.f;
            if (PRO_P == PRO_RELU_MASK) av = st.m[u] > 0.f ? av : 0.f;
            if (PRO_P == PRO_MUL_MASK) av *= st.m[u];
            csum += av;
#pragma unroll
            for (int t = 0; t < IT; ++t) {
                const float bv = (rv && cb + t < I) ? st.b[u][t] : 0.f;
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
            }
        }
    };
    if (r_begin < r_end) {
        // ring of DEPTH stages of 8 rows: DEPTH - 1 stages of loads are in flight ahead of the MFMAs.  Narrow
        // operands (IT == 1: <= 192 bytes per row) get a deeper ring (more bytes in flight per wave): dW of a 16 x 32
        // layer on 2^24 rows 991 -> 936 us, of the 32 x 32 layer with its mask operand 1130 -> 1103 us.
        constexpr int DEPTH = IT == 1 ? 6 : 3;
        Stage st[DEPTH];
        int64_t r0 = r_begin;
        // sched_barrier pins the issue order: without it the machine scheduler sinks every load next to its
        // consumer and the prefetch distance collapses to zero
#define GAE_PIN() __builtin_amdgcn_sched_barrier(0)
#pragma unroll
        for (int d = 0; d < DEPTH - 1; ++d) load(st[d], r0 + 8 * d);
        GAE_PIN();
        bool more = true;
        while (more) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                if (more) {
                    load(st[(d + DEPTH - 1) % DEPTH], r0 + 8 * (DEPTH - 1)); GAE_PIN(); compute(st[d], r0); GAE_PIN();
                    r0 += 8;
                    more = r0 < r_end;
                }
            }
        }
#undef GAE_PIN
    }
    // ---- block reduction, fixed tree order: (w) += (w + half) for half = NW/2, NW/4, ... 1
    csum += __shfl_down(csum, 32, 64);
#pragma unroll
    for (int half = NW / 2; half >= 1; half >>= 1) {
        if (wave >= half && wave < 2 * half) {
            float *rp = red[wave - half];
#pragma unroll
            for (int t = 0; t < IT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) rp[(t * 16 + r) * 64 + lane] = acc[t][r];
            rp[IT * 16 * 64 + lane] = csum;
        }
        __syncthreads();
        if (wave < half) {
            const float *rp = red[wave];
#pragma unroll
            for (int t = 0; t < IT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] += rp[(t * 16 + r) * 64 + lane];
            csum += rp[IT * 16 * 64 + lane];
        }
        if (half > 1) __syncthreads();
    }
    if (wave != 0) return;
    // partial[slot][O][I] (+ [O] column sums at colsum_offset); lane j holds columns cb..cb+IT-1 of row oo
    float *pp = partial + slot * slot_stride;
    const bool full = blockIdx.z * 32 + 32 <= O && (QVEC ? cb + 4 <= I && (I & 3) == 0 : false);
    if (full) {      // whole tile inside the output: unconditional 16-byte stores
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int oo = blockIdx.z * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            *reinterpret_cast<float4 *>(pp + int64_t(oo) * I + cb) =
                make_float4(acc[0][r], acc[IT > 1 ? 1 : 0][r], acc[IT > 2 ? 2 : 0][r], acc[IT > 3 ? 3 : 0][r]);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int oo = blockIdx.z * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (oo >= O) continue;
            float *dst = pp + int64_t(oo) * I + cb;
#pragma unroll
            for (int t = 0; t < IT; ++t)
                if (cb + t < I) dst[t] = acc[t][r];
        }
    }
    if (colsum_offset >= 0 && blockIdx.y == 0 && h == 0 && o < O) pp[colsum_offset + o] = csum;
}

// ---------------------------------------------------------------------------
// The same reduction with bf16 x 3 products (gae::split_bf16x4) on v_mfma_f32_16x16x16_bf16, for O <= 32 and
// 16-byte aligned rows of Q: on gfx950 the fp32 MFMA of atb_partial_kernel occupies the fp32 FMA lanes for
// ~7 us of the Pubmed layer-1 launch and does not overlap the wave's own loads; the bf16 matrix pipe does the
// same products in a fifth of the time, the operand split costs ~2.5 VALU ops per element.  The dropped lo.lo
// term is 2^-16 relative (same scheme as the fused loss).
// A block of 8 waves owns one row slot and 64 columns: lane (l15, g) of a wave loads, per step of 16 rows, the
// float4 Q[row 4g + r][c0 + 4 l15 ..] for r = 0..3 -- column 4 l15 + t belongs to output tile t -- and
// P[row][16 mt + l15]; accumulators acc[mt][t][r] = out[16 mt + 4 g + r][c0 + 4 l15 + t].
// ---------------------------------------------------------------------------
// P3 (knob atb_bf16 = 1, the default): three bf16 pieces per operand and the six piece pairs down to 2^-24
// (gae::split_bf16x4_3) -- an fp32-grade product; the two-piece form (knob 2: 16 mantissa bits per operand) left
// 1e-4 .. 3e-4 of the gradient's scale on operands whose large columns cancel, ten times the exact fp32 kernel's
// error (tests/test_gpu_wgrad_condition.py).
template <int PRO_P, bool P3>
__global__ __launch_bounds__(512) void atb_bf16_kernel(
    const float *__restrict__ P, int64_t ldp, const float *__restrict__ Pmask, int64_t ldpm,
    const float *__restrict__ Q, int64_t ldq, int64_t n, int O, int I, int64_t rows_per_slot,
    float *__restrict__ partial, int64_t slot_stride, int64_t colsum_offset)
{
    constexpr int NW = 8;
    constexpr int PARK = 34 * 64;                    // 32 accumulator + 2 column-sum floats per lane
    __shared__ float red[NW / 2][PARK];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int64_t slot = blockIdx.x;
    const int c0 = blockIdx.y * 64, col = c0 + 4 * l15;
    const int64_t rows_per_wave = rows_per_slot / NW;                 // multiple of 16
    const int64_t r_begin = slot * rows_per_slot + wave * rows_per_wave;
    int64_t r_end = r_begin + rows_per_wave;
    if (r_end > n) r_end = n;

    gae::v4f acc[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[mt][t] = gae::v4f{0.f, 0.f, 0.f, 0.f};
    float csum[2] = {0.f, 0.f};

    struct Stage { float a[2][4], m[2][4]; float4 b[4]; };
    const int64_t r_last = r_end - 1;
    const int I4 = (I + 3) & ~3;
    const int colc = col + 4 <= I4 ? col : I4 - 4;
    int oc[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) oc[mt] = 16 * mt + l15 < O ? 16 * mt + l15 : O - 1;
    auto load = [&](Stage &st, int64_t r0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int64_t row = r0 + 4 * g + r;
            row = row < r_last ? row : r_last;
            st.b[r] = *reinterpret_cast<const float4 *>(Q + row * ldq + colc);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                st.a[mt][r] = P[row * ldp + oc[mt]];
                if (PRO_P != PRO_NONE) st.m[mt][r] = Pmask[row * ldpm + oc[mt]];
            }
        }
    };
    auto compute = [&](const Stage &st, int64_t r0) {
        bool rv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) rv[r] = r0 + 4 * g + r < r_end;
        gae::v4s ah[2], al[2], bh[4], bl[4], am[P3 ? 2 : 1], bm[P3 ? 4 : 1];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            gae::v4f av;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = (rv[r] && 16 * mt + l15 < O) ? st.a[mt][r] : 0.f;
                if (PRO_P == PRO_RELU_MASK) v = st.m[mt][r] > 0.f ? v : 0.f;
                if (PRO_P == PRO_MUL_MASK) v *= st.m[mt][r];
                av[r] = v;
                csum[mt] += v;
            }
            if constexpr (P3) gae::split_bf16x4_3(av, ah[mt], am[mt], al[mt]);
            else gae::split_bf16x4(av, ah[mt], al[mt]);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float b0 = t == 0 ? st.b[0].x : t == 1 ? st.b[0].y : t == 2 ? st.b[0].z : st.b[0].w;
            const float b1 = t == 0 ? st.b[1].x : t == 1 ? st.b[1].y : t == 2 ? st.b[1].z : st.b[1].w;
            const float b2 = t == 0 ? st.b[2].x : t == 1 ? st.b[2].y : t == 2 ? st.b[2].z : st.b[2].w;
            const float b3 = t == 0 ? st.b[3].x : t == 1 ? st.b[3].y : t == 2 ? st.b[3].z : st.b[3].w;
            const bool cv = col + t < I;
            const gae::v4f bv = {(rv[0] && cv) ? b0 : 0.f, (rv[1] && cv) ? b1 : 0.f, (rv[2] && cv) ? b2 : 0.f,
                                 (rv[3] && cv) ? b3 : 0.f};
            if constexpr (P3) gae::split_bf16x4_3(bv, bh[t], bm[t], bl[t]);
            else gae::split_bf16x4(bv, bh[t], bl[t]);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                gae::v4f c = acc[mt][t];             // smallest terms first
                c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al[mt], bh[t], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah[mt], bl[t], c, 0, 0, 0);
                if constexpr (P3) {
                    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(am[mt], bm[t], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(am[mt], bh[t], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah[mt], bm[t], c, 0, 0, 0);
                }
                acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah[mt], bh[t], c, 0, 0, 0);
            }
    };
    if (r_begin < r_end) {
        Stage s0, s1, s2;
        int64_t r0 = r_begin;
#define GAE_PIN() __builtin_amdgcn_sched_barrier(0)
        load(s0, r0); load(s1, r0 + 16); GAE_PIN();
        while (true) {
            load(s2, r0 + 32); GAE_PIN(); compute(s0, r0); GAE_PIN(); r0 += 16; if (r0 >= r_end) break;
            load(s0, r0 + 32); GAE_PIN(); compute(s1, r0); GAE_PIN(); r0 += 16; if (r0 >= r_end) break;
            load(s1, r0 + 32); GAE_PIN(); compute(s2, r0); GAE_PIN(); r0 += 16; if (r0 >= r_end) break;
        }
#undef GAE_PIN
    }
    // column sums of P: this lane summed rows 4 g + r of column 16 mt + l15 -> add the 4 row groups
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        csum[mt] += __shfl_xor(csum[mt], 16, 64);
        csum[mt] += __shfl_xor(csum[mt], 32, 64);
    }
    // ---- block reduction, fixed tree order: (w) += (w + half) for half = 4, 2, 1
#pragma unroll
    for (int half = NW / 2; half >= 1; half >>= 1) {
        if (wave >= half && wave < 2 * half) {
            float *rp = red[wave - half];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) rp[((mt * 4 + t) * 4 + r) * 64 + lane] = acc[mt][t][r];
                rp[(32 + mt) * 64 + lane] = csum[mt];
            }
        }
        __syncthreads();
        if (wave < half) {
            const float *rp = red[wave];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[mt][t][r] += rp[((mt * 4 + t) * 4 + r) * 64 + lane];
                csum[mt] += rp[(32 + mt) * 64 + lane];
            }
        }
        if (half > 1) __syncthreads();
    }
    if (wave != 0) return;
    float *pp = partial + slot * slot_stride;
    const bool vec_store = (I & 3) == 0 && col + 4 <= I;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oo = 16 * mt + 4 * g + r;
            if (oo >= O) continue;
            float *dst = pp + int64_t(oo) * I + col;
            if (vec_store) {
                *reinterpret_cast<float4 *>(dst) = make_float4(acc[mt][0][r], acc[mt][1][r], acc[mt][2][r], acc[mt][3][r]);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (col + t < I) dst[t] = acc[mt][t][r];
            }
        }
    if (colsum_offset >= 0 && blockIdx.y == 0 && g == 0) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
            if (16 * mt + l15 < O) pp[colsum_offset + 16 * mt + l15] = csum[mt];
    }
}

// out[e] = sum_slot partial[slot * slot_stride + e]  (+ optional accumulate into out, optional mask multiply);
// elements e >= n_first go to out2[e - n_first] (the column sums parked behind a slot's tile).
// 64 elements per block of 16 waves; wave q takes slots q, q + 16, ... (independent loads, all in flight) and the
// waves meet in LDS in fixed order.
__global__ __launch_bounds__(1024) void reduce_slots_kernel(const float *__restrict__ partial, int64_t n_slots,
                                                            int64_t slot_stride, int64_t n_elems,
                                                            float *__restrict__ out, int64_t out_cols, int64_t ldo,
                                                            int accumulate, const float *__restrict__ mulmask,
                                                            int64_t ldmask, float *__restrict__ out2, int64_t n_first)
{
    __shared__ float red[16][64];
    const int el = threadIdx.x & 63, q = threadIdx.x >> 6;
    for (int64_t base = int64_t(blockIdx.x) * 64; base < n_elems; base += int64_t(gridDim.x) * 64) {
        const int64_t e = base + el;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (e < n_elems) {
            int64_t k = q;
            for (; k + 48 < n_slots; k += 64) {
                const float v0 = partial[k * slot_stride + e], v1 = partial[(k + 16) * slot_stride + e];
                const float v2 = partial[(k + 32) * slot_stride + e], v3 = partial[(k + 48) * slot_stride + e];
                s0 += v0; s1 += v1; s2 += v2; s3 += v3;
            }
            for (; k < n_slots; k += 16) s0 += partial[k * slot_stride + e];
        }
        red[q][el] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (q == 0 && e < n_elems) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 16; ++w) s += red[w][el];
            if (out2 && e >= n_first) {
                out2[e - n_first] = s;
            } else {
                const int64_t r = e / out_cols, c = e - r * out_cols;
                float *op = out + r * ldo + c;
                if (accumulate) s += *op;
                if (mulmask) s *= mulmask[r * ldmask + c];
                *op = s;
            }
        }
        __syncthreads();
    }
}

// dst[i, k] = (dst[i, k] + sum_slot partial[slot][k][i]) * mask[i, k]   (a slot holds [d][n] at slot_stride floats)
__global__ __launch_bounds__(256) void reduce_slots_transposed_kernel(const float *__restrict__ partial,
                                                                      int64_t n_slots, int64_t slot_stride,
                                                                      int64_t d, int64_t n,
                                                                      float *__restrict__ dst, int64_t ldd,
                                                                      const float *__restrict__ mask, int64_t ldmask)
{
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    const int64_t total = n * d;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t k = e / n, i = e - k * n;  // consecutive threads -> consecutive i (coalesced partial reads)
        float s = 0.f;
        for (int64_t q = 0; q < n_slots; ++q) s += partial[q * slot_stride + e];
        float v = dst[i * ldd + k] + s;
        if (mask) v *= mask[i * ldmask + k];
        dst[i * ldd + k] = v;
    }
}

struct AtbPlan {
    int64_t n_slots, rows_per_slot, blocks, slot_stride;   // slot_stride: floats per partial (tile + column sums)
};

// shared by the workspace query and the launcher: one slot (= one partial) per block of 8 waves.
// Rows per slot: so that the launch is about 0.8 blocks per CU -- ONE round of 512-thread blocks (measured on
// every layer shape of the BASELINE configs: a second, partial round costs more than the longer row loop).
AtbPlan atb_plan(int64_t n, int64_t O, int64_t I)
{
    AtbPlan p;
    const bool bf16 = g_atb_bf16 && O <= 32 && I > 32;                  // atb_bf16_kernel: 64 columns per block
    const int64_t cgroups = I <= 32 ? (I + 31) / 32 : bf16 ? (I + 63) / 64 : (I + 127) / 128;
    const int64_t tiles = (cgroups > 0 ? cgroups : 1) * ((O + 31) / 32 > 0 ? (O + 31) / 32 : 1);   // blocks per slot
    int64_t rows = g_atb_rows > 0 ? g_atb_rows : (n * tiles + 207) / 208;
    rows = (rows + 127) / 128 * 128;                 // 8 waves x a multiple of 16 rows (8 for the fp32 kernel)
    if (rows < 128) rows = 128;
    int64_t want = (n + rows - 1) / rows;
    int64_t cap = (int64_t(32) << 20) / (O * I > 0 ? O * I : 1);  // <= 128 MiB of partials
    if (cap > 4096) cap = 4096;
    if (cap < 1) cap = 1;
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    int64_t rps = (n + want - 1) / want;
    rps = (rps + 127) / 128 * 128;
    if (rps < 128) rps = 128;
    p.rows_per_slot = rps;
    p.n_slots = p.blocks = (n + rps - 1) / rps > 0 ? (n + rps - 1) / rps : 1;
    p.slot_stride = (O * I + O + 3) / 4 * 4;         // [O][I] tile, then [O] column sums; 16-byte aligned slots
    return p;
}

template <int PRO_P>
int launch_atb(const float *P, int64_t ldp, const float *Pmask, int64_t ldpm, const float *Q, int64_t ldq,
               int64_t n, int O, int I, float *partial, bool colsum, const AtbPlan &pl, hipStream_t s)
{
    const bool narrow = I <= 32;                     // one 32-column tile: no 4-tile float4 mapping needed
    // bf16 x 3 matrix-core products when the layout allows float4 loads of Q (atb_bf16_kernel)
    if (g_atb_bf16 && !narrow && O <= 32 && (ldq % 4 == 0) && gae::aligned16(Q) && ldq >= ((I + 3) & ~3) && I >= 4 &&
        gae::aligned16(partial) && pl.rows_per_slot % 128 == 0) {
        const dim3 grid(unsigned(pl.blocks), unsigned((I + 63) / 64), 1);
        if (g_atb_bf16 == 2)
            hipLaunchKernelGGL((atb_bf16_kernel<PRO_P, false>), grid, dim3(512), 0, s, P, ldp, Pmask, ldpm, Q, ldq, n, O, I,
                               pl.rows_per_slot, partial, pl.slot_stride, colsum ? int64_t(O) * I : int64_t(-1));
        else
            hipLaunchKernelGGL((atb_bf16_kernel<PRO_P, true>), grid, dim3(512), 0, s, P, ldp, Pmask, ldpm, Q, ldq, n, O, I,
                               pl.rows_per_slot, partial, pl.slot_stride, colsum ? int64_t(O) * I : int64_t(-1));
        GAE_CHECK_LAUNCH("atb_bf16_kernel");
        return GAE_OK;
    }
    const int cols_per_block = narrow ? 32 : 128;
    const unsigned gy = I > 0 ? unsigned((I + cols_per_block - 1) / cols_per_block) : 1u;  // I == 0: column sums only
    const dim3 grid(unsigned(pl.blocks), gy, unsigned((O + 31) / 32));
    // float4 loads of Q: 16-byte aligned rows that reach the next multiple of 4 columns (row padding past I is
    // read but never used)
    const bool qvec = !narrow && (ldq % 4 == 0) && gae::aligned16(Q) && ldq >= ((I + 3) & ~3) && I >= 4 &&
                      gae::aligned16(partial);
    const int64_t cso = colsum ? int64_t(O) * I : -1;
#define GAE_ATB(IT, QV)                                                                                             \
    hipLaunchKernelGGL((atb_partial_kernel<IT, PRO_P, QV, 8>), grid, dim3(512), 0, s, P, ldp, Pmask, ldpm, Q, ldq, n, \
                       O, I, pl.rows_per_slot, partial, pl.slot_stride, cso)
    if (narrow) GAE_ATB(1, false);
    else if (qvec) GAE_ATB(4, true);
    else GAE_ATB(4, false);
#undef GAE_ATB
    GAE_CHECK_LAUNCH("atb_partial_kernel");
    return GAE_OK;
}

// out[r, c] = sum_slot partial[slot][r * out_cols + c] for the first n_first elements of a slot, out2[.] for the rest
int launch_reduce(const float *partial, int64_t n_slots, int64_t slot_stride, int64_t n_elems, float *out,
                  int64_t out_cols, int64_t ldo, int accumulate, const float *mulmask, int64_t ldmask, float *out2,
                  int64_t n_first, hipStream_t s)
{
    int64_t g = (n_elems + 63) / 64;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(reduce_slots_kernel, dim3(unsigned(g)), dim3(1024), 0, s, partial, n_slots, slot_stride, n_elems,
                       out, out_cols, ldo, accumulate, mulmask, ldmask, out2, n_first);
    GAE_CHECK_LAUNCH("reduce_slots_kernel");
    return GAE_OK;
}

// ---------------------------------------------------------------------------
// Philox4x32-10 counter RNG -> inverted dropout multiplier
// ---------------------------------------------------------------------------
using gae::philox_round;

__global__ __launch_bounds__(256) void dropout_mask_kernel(float *__restrict__ mask, int64_t n, float p, float scale,
                                                           uint64_t seed, uint64_t offset,
                                                           const uint64_t *__restrict__ draw_dev)
{
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    const int64_t nquad = (n + 3) / 4;
    // device-side draw counter: graph replays advance the stream.  The draw index goes into the HIGH counter words
    // (a stream of its own per draw), not into the element counter: draws over tensors of different sizes
    // (inductive batches) would otherwise overlap.
    const uint64_t draw = draw_dev ? *draw_dev : 0;
    for (int64_t q = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; q < nquad; q += stride) {
        uint32_t c[4];
        gae::philox4x32_10(offset + uint64_t(q), draw, seed, c);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t e = q * 4 + i;
            if (e < n) mask[e] = gae::dropout_multiplier(c[i], p, scale);
        }
    }
}


// standard normal noise: Philox4x32-10 + Box-Muller, 4 values per counter (VGAE reparameterisation)
__global__ __launch_bounds__(256) void normal_noise_kernel(float *__restrict__ out, int64_t n, uint64_t seed,
                                                           uint64_t offset, const uint64_t *__restrict__ draw_dev)
{
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    const int64_t nquad = (n + 3) / 4;
    const uint64_t draw = draw_dev ? *draw_dev : 0;     // one stream per draw (word 3), tag in word 2
    for (int64_t q = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; q < nquad; q += stride) {
        const uint64_t ctr = offset + uint64_t(q);
        uint32_t c[4] = {uint32_t(ctr), uint32_t(ctr >> 32), 0x6e6f726du ^ uint32_t(draw >> 32), uint32_t(draw)};   // stream tag differs from dropout
        uint32_t k0 = uint32_t(seed), k1 = uint32_t(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            philox_round(c, k0, k1);
            k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
        float z[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float u1 = (float(c[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);     // (0, 1)
            const float u2 = float(c[2 * h + 1] >> 8) * (1.0f / 16777216.0f);          // [0, 1)
            const float rad = sqrtf(-2.0f * __logf(u1));
            float sn, cs;
            __sincosf(6.28318530717958648f * u2, &sn, &cs);
            z[2 * h] = rad * cs; z[2 * h + 1] = rad * sn;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (q * 4 + i < n) out[q * 4 + i] = z[i];
    }
}

// VGAE head (Kipf & Welling 2016; README.md:58 cites the paper, the reference has no code for it):
//   z = mu + eps * exp(logstd);   KL = -(0.5 / N) * mean_i sum_j (1 + 2 logstd - mu^2 - exp(2 logstd))
// forward writes z and per-block fp64 partial sums of the KL bracket; backward adds the KL gradient to
// the gradient arriving through z.
__global__ __launch_bounds__(256) void vgae_head_fwd_kernel(const float *__restrict__ mu, const float *__restrict__ ls,
                                                            int64_t ldm, int d, const float *__restrict__ eps,
                                                            int64_t n_elems, float *__restrict__ z,
                                                            double *__restrict__ partial)
{
    __shared__ double red[4];
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    double acc = 0.0;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < n_elems; e += stride) {
        const int64_t i = e / d, at = i * ldm + (e - i * d);        // mu / log sigma rows are ldm floats apart
        const float m = mu[at], l = ls[at];
        const float s = __expf(l);
        z[e] = fmaf(eps[e], s, m);
        acc += double(1.0f + 2.0f * l - m * m - s * s);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// N(0, 1) quad q of a draw: exactly normal_noise_kernel's values (same Philox stream, same Box-Muller)
__device__ __forceinline__ void normal_quad(int64_t q, uint64_t seed, uint64_t offset, uint64_t draw, float (&z)[4])
{
    const uint64_t ctr = offset + uint64_t(q);
    uint32_t c[4] = {uint32_t(ctr), uint32_t(ctr >> 32), 0x6e6f726du ^ uint32_t(draw >> 32), uint32_t(draw)};
    uint32_t k0 = uint32_t(seed), k1 = uint32_t(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float u1 = (float(c[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);     // (0, 1)
        const float u2 = float(c[2 * h + 1] >> 8) * (1.0f / 16777216.0f);          // [0, 1)
        const float rad = sqrtf(-2.0f * __logf(u1));
        float sn, cs;
        __sincosf(6.28318530717958648f * u2, &sn, &cs);
        z[2 * h] = rad * cs; z[2 * h + 1] = rad * sn;
    }
}

// VGAE head + everything between it and the loss's dense kernel in ONE launch (d = 16): the noise of this draw (unless
// given), z = mu + eps exp(log sigma), the KL partial of the block, and the prepare step of the fused loss on z (no
// dropout: Zt = z, its bf16 hi / lo, fp64 column sums per block of 64 rows) -- replaces normal_noise_kernel,
// vgae_head_fwd_kernel and bce_prepare_kernel.  Thread = 4 consecutive elements of a row, block = 64 rows.
__global__ __launch_bounds__(256) void vgae_head_prep_kernel(const float *__restrict__ mu, const float *__restrict__ ls,
                                                             int64_t ldm, float *__restrict__ eps, int draw_eps,
                                                             uint64_t seed, uint64_t offset,
                                                             const uint64_t *__restrict__ draw_dev, int64_t n,
                                                             float *__restrict__ z, float *__restrict__ Zt,
                                                             unsigned short *__restrict__ Zhi,
                                                             unsigned short *__restrict__ Zlo,
                                                             double *__restrict__ colsum_partial,
                                                             double *__restrict__ kl_partial)
{
    __shared__ double red[4][17];
    const int64_t q = int64_t(blockIdx.x) * 256 + threadIdx.x;       // quad id: row q / 4, columns 4 (q % 4) ..
    const int64_t i = q >> 2;
    const int k4 = int(q & 3) * 4;
    const bool in = i < n;
    float ev[4] = {0.f, 0.f, 0.f, 0.f}, zv[4] = {0.f, 0.f, 0.f, 0.f};
    double kl = 0.0;
    if (in) {
        const float4 m = *reinterpret_cast<const float4 *>(mu + i * ldm + k4);
        const float4 l = *reinterpret_cast<const float4 *>(ls + i * ldm + k4);
        if (draw_eps) {
            normal_quad(q, seed, offset, draw_dev ? *draw_dev : 0, ev);
            *reinterpret_cast<float4 *>(eps + q * 4) = make_float4(ev[0], ev[1], ev[2], ev[3]);
        } else {
            const float4 e4 = *reinterpret_cast<const float4 *>(eps + q * 4);
            ev[0] = e4.x; ev[1] = e4.y; ev[2] = e4.z; ev[3] = e4.w;
        }
        const float mm[4] = {m.x, m.y, m.z, m.w}, ll[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float sg = __expf(ll[c]);
            zv[c] = fmaf(ev[c], sg, mm[c]);
            kl += double(1.0f + 2.0f * ll[c] - mm[c] * mm[c] - sg * sg);
        }
        const float4 z4 = make_float4(zv[0], zv[1], zv[2], zv[3]);
        *reinterpret_cast<float4 *>(z + q * 4) = z4;
        *reinterpret_cast<float4 *>(Zt + q * 4) = z4;
        unsigned short hi[4], lo[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            hi[c] = gae::f32_to_bf16(zv[c]);
            lo[c] = gae::f32_to_bf16(zv[c] - gae::bf16_to_f32(hi[c]));
        }
        *reinterpret_cast<uint2 *>(Zhi + q * 4) = make_uint2(hi[0] | (unsigned(hi[1]) << 16), hi[2] | (unsigned(hi[3]) << 16));
        *reinterpret_cast<uint2 *>(Zlo + q * 4) = make_uint2(lo[0] | (unsigned(lo[1]) << 16), lo[2] | (unsigned(lo[3]) << 16));
    }
    // column sums over the block's 64 rows (lanes with the same columns are 4 apart) and the block's KL partial,
    // fixed order: butterfly inside the wave, then the 4 waves
    double cs[4] = {double(zv[0]), double(zv[1]), double(zv[2]), double(zv[3])};
#pragma unroll
    for (int off = 4; off < 64; off <<= 1) {
#pragma unroll
        for (int c = 0; c < 4; ++c) cs[c] += __shfl_xor(cs[c], off, 64);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) kl += __shfl_xor(kl, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < 4) {
#pragma unroll
        for (int c = 0; c < 4; ++c) red[wave][lane * 4 + c] = cs[c];
    }
    if (lane == 0) red[wave][16] = kl;
    __syncthreads();
    if (threadIdx.x < 16) {
        const double t = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
        colsum_partial[(int64_t(blockIdx.x) * 2 + 0) * 16 + threadIdx.x] = t;
        colsum_partial[(int64_t(blockIdx.x) * 2 + 1) * 16 + threadIdx.x] = t;
    }
    if (threadIdx.x == 16) kl_partial[blockIdx.x] = ((red[0][16] + red[1][16]) + red[2][16]) + red[3][16];
}

__global__ __launch_bounds__(256) void vgae_kl_finalize_kernel(const double *__restrict__ partial, int n_partial,
                                                               double scale, float *__restrict__ kl_out)
{
    __shared__ double red[256];
    double s = 0.0;
    for (int k = threadIdx.x; k < n_partial; k += 256) s += partial[k];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (int(threadIdx.x) < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) *kl_out = float(red[0] * scale);
}

// dmu = dz + gkl * mu / N^2 ;  dls = dz * eps * exp(ls) + gkl * (exp(2 ls) - 1) / N^2
__global__ __launch_bounds__(256) void vgae_head_bwd_kernel(const float *__restrict__ dz, const float *__restrict__ mu,
                                                            const float *__restrict__ ls, const float *__restrict__ eps,
                                                            const float *__restrict__ gkl_dev, float inv_n2,
                                                            int64_t n_elems, float *__restrict__ dmu,
                                                            float *__restrict__ dls, int64_t ldm, int d)
{
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    const float gk = (gkl_dev ? *gkl_dev : 1.0f) * inv_n2;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < n_elems; e += stride) {
        const int64_t i = e / d, at = i * ldm + (e - i * d);        // mu / log sigma / their gradients: ldm floats apart
        const float m = mu[at], l = ls[at], s = __expf(l);
        const float g = dz ? dz[e] : 0.f;
        dmu[at] = fmaf(gk, m, g);
        dls[at] = fmaf(g * eps[e], s, gk * (s * s - 1.0f));
    }
}

} // namespace

namespace gae {
Knob *dense_knob(const char *name)
{
    if (strcmp(name, "gemm_stream") == 0) return &g_gemm_stream;
    if (strcmp(name, "gemm_rows") == 0) return &g_gemm_rows;
    if (strcmp(name, "linear_f32x16") == 0) return &g_linear_f32x16;
    if (strcmp(name, "linear_nw") == 0) return &g_linear_nw;
    if (strcmp(name, "linear_depth") == 0) return &g_linear_depth;
    if (strcmp(name, "atb_rows") == 0) return &g_atb_rows;
    if (strcmp(name, "atb_bf16") == 0) return &g_atb_bf16;
    if (strcmp(name, "linear_bf16") == 0) return &g_linear_bf16;
    if (strcmp(name, "linear_wlds") == 0) return &g_linear_wlds;
    return nullptr;
}
} // namespace gae

// ===========================================================================
namespace gae {
// xw.hip: the stream family for wide inputs / narrow outputs (W stationary in registers, X read once)
bool xw_usable(const void *X, int64_t ldx, int64_t n, int64_t K, int64_t J, int elem);
int64_t xw_fwd_workspace_bytes(int64_t n, int64_t K, int64_t J, int elem);
int xw_fwd_launch(const void *X, int64_t ldx, int64_t n, int K, int elem, const float *W, int64_t ldw, const float *bias,
                  int J, int act, float *out, int64_t ldo, void *ws, int64_t ws_bytes, hipStream_t s, bool keep_splits);
int64_t xtg_workspace_bytes(int64_t n, int64_t K, int elem);
int xtg_launch(const void *X, int64_t ldx, int64_t n, int K, int elem, const float *G, int64_t ldg, const float *Gmask,
               int64_t ldgm, const float *D, int64_t ldd, const float *Dmask, int64_t lddm, int J, float *dW,
               int64_t lddw, float *db, void *ws, int64_t ws_bytes, hipStream_t s);
}

extern "C" int64_t gae_linear_fwd_workspace_bytes(int64_t n, int64_t f_in, int64_t f_out)
{
    if (n < 0 || f_in < 0 || f_out < 0 || f_in >= (1 << 24)) return GAE_E_SIZE;
    const int sp = f_out <= 128 ? gemm_stream_splits(n, int(f_in)) : 1;
    const int64_t old = sp > 1 ? (int64_t(sp) * n * f_out * 4 + 255) / 256 * 256 : 0;
    const int64_t xw = (f_out >= 1 && f_out <= 32 && f_in >= 64 && n > 0) ? gae::xw_fwd_workspace_bytes(n, f_in, f_out, 4) : 0;
    return old > xw ? old : xw;
}

extern "C" int gae_linear_fwd(const float *M, int64_t ldm, int64_t n, int64_t f_in, const float *W, const float *b,
                              int64_t f_out, int act, float *Y, int64_t ldy, void *workspace,
                              int64_t workspace_bytes, void *stream)
{
    GAE_REQUIRE(n >= 0 && f_in >= 0 && f_out >= 0, GAE_E_SIZE, "gae_linear_fwd: negative size");
    GAE_REQUIRE(f_in < (1 << 24) && f_out < (1 << 24), GAE_E_SIZE, "gae_linear_fwd: feature width too large");
    GAE_REQUIRE(ldm >= f_in && ldy >= f_out, GAE_E_SIZE, "gae_linear_fwd: leading dimension too small");
    GAE_REQUIRE(act == GAE_ACT_IDENTITY || act == GAE_ACT_RELU, GAE_E_DTYPE, "gae_linear_fwd: activation %d", act);
    if (n == 0 || f_out == 0) return GAE_OK;
    GAE_REQUIRE(Y && (f_in == 0 || (M && W)), GAE_E_NULL, "gae_linear_fwd: NULL pointer");
    GAE_REQUIRE(!workspace || gae::aligned16(workspace), GAE_E_ALIGN, "gae_linear_fwd: workspace not 16-byte aligned");
    // wide input, narrow output (layer 1 in the reference's order, (A X) W^T): the stream family of xw.hip
    if (gae::xw_usable(M, ldm, n, f_in, f_out, 4) &&
        (gae::xw_fwd_workspace_bytes(n, f_in, f_out, 4) == 0 ||
         (workspace && workspace_bytes >= gae::xw_fwd_workspace_bytes(n, f_in, f_out, 4))))
        return gae::xw_fwd_launch(M, ldm, n, int(f_in), 4, W, f_in, b, int(f_out), act, Y, ldy, workspace, workspace_bytes,
                                  gae::as_stream(stream), false);
    return dispatch_gemm<true, PRO_NONE, false>(M, ldm, nullptr, 0, W, f_in, nullptr, 0, b, act, Y, ldy, n, int(f_in),
                                                f_out, gae::as_stream(stream), static_cast<float *>(workspace),
                                                workspace ? workspace_bytes / 4 : 0);
}

static int64_t align256(int64_t x) { return (x + 255) / 256 * 256; }

extern "C" int64_t gae_linear_bwd_workspace_bytes(int64_t n, int64_t f_in, int64_t f_out)
{
    if (n < 0 || f_in < 0 || f_out < 0) return GAE_E_SIZE;
    const AtbPlan pl = atb_plan(n, f_out, f_in);
    return align256(pl.n_slots * pl.slot_stride * 4) + 256;
}

extern "C" int gae_x_linear_bwd_partials(const float *dY, int64_t lddy, const float *Y, int64_t ldy, int act, const float *M,
                                       int64_t ldm, int64_t n, int64_t f_in, int64_t f_out, int want_dW, int want_db,
                                       void *workspace, int64_t workspace_bytes, int64_t *layout_out, void *stream)
{
    GAE_REQUIRE(n > 0 && f_in >= 0 && f_out > 0 && layout_out, GAE_E_SIZE, "gae_x_linear_bwd_partials: bad sizes");
    GAE_REQUIRE(f_in < (1 << 24) && f_out < (1 << 24), GAE_E_SIZE, "gae_x_linear_bwd_partials: feature width too large");
    GAE_REQUIRE(act == GAE_ACT_IDENTITY || act == GAE_ACT_RELU, GAE_E_DTYPE, "gae_x_linear_bwd_partials: activation %d", act);
    GAE_REQUIRE(dY && lddy >= f_out && (want_dW || want_db), GAE_E_NULL, "gae_x_linear_bwd_partials: dY missing");
    GAE_REQUIRE(act != GAE_ACT_RELU || (Y && ldy >= f_out), GAE_E_NULL, "gae_x_linear_bwd_partials: RELU needs Y");
    GAE_REQUIRE(!want_dW || (M && ldm >= f_in && f_in > 0), GAE_E_NULL, "gae_x_linear_bwd_partials: dW needs M");
    const AtbPlan pl = atb_plan(n, f_out, f_in);
    const int64_t need = align256(pl.n_slots * pl.slot_stride * 4);
    GAE_REQUIRE(workspace && workspace_bytes >= need && gae::aligned16(workspace), GAE_E_WORKSPACE,
                "gae_x_linear_bwd_partials: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
    const bool relu = act == GAE_ACT_RELU;
    const float *Q = want_dW ? M : dY;
    const int64_t ldq = want_dW ? ldm : lddy;
    const int I = want_dW ? int(f_in) : 0;
    float *partial = static_cast<float *>(workspace);
    hipStream_t s = gae::as_stream(stream);
    const int rc = relu ? launch_atb<PRO_RELU_MASK>(dY, lddy, Y, ldy, Q, ldq, n, int(f_out), I, partial, want_db != 0, pl, s)
                        : launch_atb<PRO_NONE>(dY, lddy, nullptr, 0, Q, ldq, n, int(f_out), I, partial, want_db != 0, pl, s);
    layout_out[0] = pl.n_slots; layout_out[1] = pl.slot_stride; layout_out[2] = f_out * int64_t(I);
    return rc;
}

extern "C" int gae_linear_bwd(const float *dY, int64_t lddy, const float *Y, int64_t ldy, int act, const float *M,
                              int64_t ldm, const float *W, int64_t n, int64_t f_in, int64_t f_out, float *dW,
                              float *db, float *dM, int64_t lddm, void *workspace, int64_t workspace_bytes,
                              void *stream)
{
    GAE_REQUIRE(n >= 0 && f_in >= 0 && f_out >= 0, GAE_E_SIZE, "gae_linear_bwd: negative size");
    GAE_REQUIRE(f_in < (1 << 24) && f_out < (1 << 24), GAE_E_SIZE, "gae_linear_bwd: feature width too large");
    GAE_REQUIRE(act == GAE_ACT_IDENTITY || act == GAE_ACT_RELU, GAE_E_DTYPE, "gae_linear_bwd: activation %d", act);
    GAE_REQUIRE(lddy >= f_out, GAE_E_SIZE, "gae_linear_bwd: lddy < f_out");
    GAE_REQUIRE(act != GAE_ACT_RELU || (Y && ldy >= f_out), GAE_E_NULL, "gae_linear_bwd: RELU needs Y");
    hipStream_t s = gae::as_stream(stream);
    if (f_out == 0 || (f_in == 0 && !db)) return GAE_OK;
    GAE_REQUIRE(n == 0 || dY, GAE_E_NULL, "gae_linear_bwd: dY is NULL");
    const bool relu = act == GAE_ACT_RELU;
    // (dW = dYm^T M stays with atb_bf16_kernel: measured against xw.hip's gae_xw_wgrad on the same operands it is the
    //  faster one -- Pubmed 17.3 vs 20.0 us, Cora 12.3 vs 16.4 -- bf16 x 3 products leave the fp32 lanes free)
    if (dW || db) {
        GAE_REQUIRE(!dW || (ldm >= f_in && (n == 0 || M)), GAE_E_NULL, "gae_linear_bwd: dW needs M");
        const AtbPlan pl = atb_plan(n, f_out, f_in);
        const int64_t need = align256(pl.n_slots * pl.slot_stride * 4);
        GAE_REQUIRE(workspace && workspace_bytes >= need, GAE_E_WORKSPACE,
                    "gae_linear_bwd: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
        GAE_REQUIRE(gae::aligned16(workspace), GAE_E_ALIGN, "gae_linear_bwd: workspace not 16-byte aligned");
        float *partial = static_cast<float *>(workspace);
        // dW = dYm^T M : P = dY [n, f_out] (O = f_out), Q = M [n, f_in] (I = f_in); the column sums of dYm (= db)
        // ride behind each slot's tile and come out of the same reduction launch
        const float *Q = dW ? M : dY;  // db only: any readable Q, I = 0 tiles skipped
        const int64_t ldq = dW ? ldm : lddy;
        const int I = dW ? int(f_in) : 0;
        int rc = relu ? launch_atb<PRO_RELU_MASK>(dY, lddy, Y, ldy, Q, ldq, n, int(f_out), I, partial, db != nullptr, pl, s)
                      : launch_atb<PRO_NONE>(dY, lddy, nullptr, 0, Q, ldq, n, int(f_out), I, partial, db != nullptr, pl, s);
        if (rc) return rc;
        // the slots' tiles and column sums are added in the library's one order for partial lists (gae::sum_partials):
        // gae_adam_step's deferred reduction of the same slots gives the same bits
        const int64_t n_tile = f_out * I;
        gae::PartialList la{}, lb{};
        if (dW) la = gae::PartialList{partial, dW, n_tile, pl.n_slots, pl.slot_stride, n_tile, n_tile, n_tile};
        if (db) lb = gae::PartialList{partial + n_tile, db, f_out, pl.n_slots, pl.slot_stride, f_out, f_out, f_out};
        rc = gae::launch_partials_reduce(la, lb, s);
        if (rc) return rc;
    }
    if (dM && n > 0) {
        GAE_REQUIRE(lddm >= f_in && W, GAE_E_NULL, "gae_linear_bwd: dM needs W and lddm >= f_in");
        // dM[n, f_in] = dYm[n, f_out] * W[f_out, f_in]   (B given as [K, J])
        if (relu)
            return dispatch_gemm<false, PRO_RELU_MASK, false>(dY, lddy, Y, ldy, W, f_in, nullptr, 0, nullptr,
                                                              GAE_ACT_IDENTITY, dM, lddm, n, int(f_out), f_in, s);
        return dispatch_gemm<false, PRO_NONE, false>(dY, lddy, nullptr, 0, W, f_in, nullptr, 0, nullptr,
                                                     GAE_ACT_IDENTITY, dM, lddm, n, int(f_out), f_in, s);
    }
    return GAE_OK;
}

extern "C" int gae_dropout_mask(float *mask, int64_t n_elems, float p, uint64_t seed, uint64_t offset,
                                const uint64_t *draw_dev, void *stream)
{
    GAE_REQUIRE(n_elems >= 0, GAE_E_SIZE, "gae_dropout_mask: negative size");
    GAE_REQUIRE(p >= 0.f && p < 1.f, GAE_E_RANGE, "gae_dropout_mask: p = %g outside [0, 1)", double(p));
    if (n_elems == 0) return GAE_OK;
    GAE_REQUIRE(mask, GAE_E_NULL, "gae_dropout_mask: mask is NULL");
    int64_t g = ((n_elems + 3) / 4 + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(unsigned(g)), dim3(256), 0, gae::as_stream(stream), mask, n_elems, p,
                       1.0f / (1.0f - p), seed, offset, draw_dev);
    GAE_CHECK_LAUNCH("dropout_mask_kernel");
    return GAE_OK;
}

extern "C" int gae_decoder_dense(const float *Z, const float *mask, int64_t ldz, int64_t n, int64_t d, float *out,
                                 int64_t ldo, void *stream)
{
    GAE_REQUIRE(n >= 0 && d >= 0, GAE_E_SIZE, "gae_decoder_dense: negative size");
    GAE_REQUIRE(d < (1 << 24), GAE_E_SIZE, "gae_decoder_dense: d too large");
    GAE_REQUIRE(ldz >= d && ldo >= n, GAE_E_SIZE, "gae_decoder_dense: leading dimension too small");
    if (n == 0) return GAE_OK;
    GAE_REQUIRE(out && (d == 0 || Z), GAE_E_NULL, "gae_decoder_dense: NULL pointer");
    hipStream_t s = gae::as_stream(stream);
    if (mask)
        return launch_gemm<4, true, PRO_MUL_MASK, true>(Z, ldz, mask, ldz, Z, ldz, mask, ldz, nullptr,
                                                        GAE_ACT_IDENTITY, out, ldo, n, int(d), n, s);
    return launch_gemm<4, true, PRO_NONE, false>(Z, ldz, nullptr, 0, Z, ldz, nullptr, 0, nullptr, GAE_ACT_IDENTITY,
                                                 out, ldo, n, int(d), n, s);
}

extern "C" int64_t gae_decoder_dense_bwd_workspace_bytes(int64_t n, int64_t d)
{
    if (n < 0 || d < 0) return GAE_E_SIZE;
    const AtbPlan pl = atb_plan(n, d, n);
    return align256(pl.n_slots * pl.slot_stride * 4) + 256;
}

extern "C" int gae_decoder_dense_bwd(const float *G, int64_t ldg, const float *Z, const float *mask, int64_t ldz,
                                     int64_t n, int64_t d, float *dZ, int64_t lddz, void *workspace,
                                     int64_t workspace_bytes, void *stream)
{
    GAE_REQUIRE(n >= 0 && d >= 0, GAE_E_SIZE, "gae_decoder_dense_bwd: negative size");
    GAE_REQUIRE(n < (1 << 24) && d < (1 << 24), GAE_E_SIZE, "gae_decoder_dense_bwd: too large for the dense path");
    GAE_REQUIRE(ldg >= n && ldz >= d && lddz >= d, GAE_E_SIZE, "gae_decoder_dense_bwd: leading dimension too small");
    if (n == 0 || d == 0) return GAE_OK;
    GAE_REQUIRE(G && Z && dZ, GAE_E_NULL, "gae_decoder_dense_bwd: NULL pointer");
    hipStream_t s = gae::as_stream(stream);
    const AtbPlan pl = atb_plan(n, d, n);
    const int64_t need = align256(pl.n_slots * pl.slot_stride * 4);
    GAE_REQUIRE(workspace && workspace_bytes >= need, GAE_E_WORKSPACE,
                "gae_decoder_dense_bwd: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
    int rc;
    // term 1: dZ = G Zt              (A = G [n, n], B = Zt given as [K = n, J = d])
    if (mask)
        rc = dispatch_gemm<false, PRO_NONE, true>(G, ldg, nullptr, 0, Z, ldz, mask, ldz, nullptr, GAE_ACT_IDENTITY, dZ,
                                                  lddz, n, int(n), d, s);
    else
        rc = dispatch_gemm<false, PRO_NONE, false>(G, ldg, nullptr, 0, Z, ldz, nullptr, 0, nullptr, GAE_ACT_IDENTITY,
                                                   dZ, lddz, n, int(n), d, s);
    if (rc) return rc;
    // term 2: (G^T Zt)^T [d, n] = Zt^T G   (P = Z with the mask folded in, Q = G), partials per row slot
    float *partial = static_cast<float *>(workspace);
    if (mask)
        rc = launch_atb<PRO_MUL_MASK>(Z, ldz, mask, ldz, G, ldg, n, int(d), int(n), partial, false, pl, s);
    else
        rc = launch_atb<PRO_NONE>(Z, ldz, nullptr, 0, G, ldg, n, int(d), int(n), partial, false, pl, s);
    if (rc) return rc;
    // dZ[i, k] = (dZ[i, k] + sum_slot partial[slot][k][i]) * mask[i, k]
    int64_t g = (n * d + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(reduce_slots_transposed_kernel, dim3(unsigned(g)), dim3(256), 0, s, partial, pl.n_slots,
                       pl.slot_stride, d, n, dZ, lddz, mask, ldz);
    GAE_CHECK_LAUNCH("reduce_slots_transposed_kernel");
    return GAE_OK;
}

extern "C" int gae_normal_noise(float *out, int64_t n_elems, uint64_t seed, uint64_t offset, const uint64_t *draw_dev,
                                void *stream)
{
    GAE_REQUIRE(n_elems >= 0, GAE_E_SIZE, "gae_normal_noise: negative size");
    if (n_elems == 0) return GAE_OK;
    GAE_REQUIRE(out, GAE_E_NULL, "gae_normal_noise: out is NULL");
    int64_t g = ((n_elems + 3) / 4 + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(normal_noise_kernel, dim3(unsigned(g)), dim3(256), 0, gae::as_stream(stream), out, n_elems, seed,
                       offset, draw_dev);
    GAE_CHECK_LAUNCH("normal_noise_kernel");
    return GAE_OK;
}

static int vgae_blocks(int64_t n_elems)
{
    int64_t g = (n_elems + 255) / 256;
    if (g > 1024) g = 1024;
    if (g < 1) g = 1;
    return int(g);
}

extern "C" int64_t gae_vgae_head_workspace_bytes(int64_t n_elems) { return n_elems < 0 ? GAE_E_SIZE : 1024 * 8 + 256; }

extern "C" int gae_vgae_head_fwd(const float *mu, const float *logstd, int64_t ldm, const float *eps, int64_t n, int64_t d,
                                 float *z, float *kl_out, void *workspace, int64_t workspace_bytes, void *stream)
{
    GAE_REQUIRE(n > 0 && d > 0 && ldm >= d && d < (1 << 24), GAE_E_SIZE, "gae_vgae_head_fwd: needs n, d > 0 and ldm >= d");
    GAE_REQUIRE(mu && logstd && eps && z && kl_out && workspace, GAE_E_NULL, "gae_vgae_head_fwd: NULL pointer");
    GAE_REQUIRE(workspace_bytes >= 1024 * 8, GAE_E_WORKSPACE, "gae_vgae_head_fwd: workspace too small");
    hipStream_t s = gae::as_stream(stream);
    const int g = vgae_blocks(n * d);
    double *partial = static_cast<double *>(workspace);
    hipLaunchKernelGGL(vgae_head_fwd_kernel, dim3(g), dim3(256), 0, s, mu, logstd, ldm, int(d), eps, n * d, z, partial);
    GAE_CHECK_LAUNCH("vgae_head_fwd_kernel");
    // KL = -(0.5 / N) * (1 / N) * sum_ij (...)
    hipLaunchKernelGGL(vgae_kl_finalize_kernel, dim3(1), dim3(256), 0, s, partial, g, -0.5 / (double(n) * double(n)),
                       kl_out);
    GAE_CHECK_LAUNCH("vgae_kl_finalize_kernel");
    return GAE_OK;
}

// see include/gae_hip.h
extern "C" int gae_x_vgae_head_prep(const float *mu, const float *logstd, int64_t ldm, float *eps, int draw_eps,
                                  uint64_t seed, uint64_t offset, const uint64_t *draw_dev, int64_t n, int64_t d, float *z,
                                  const gae_bce_prep *prep, double *kl_partial, int64_t kl_capacity,
                                  int64_t *n_blocks_out, void *stream)
{
    GAE_REQUIRE(d == 16, GAE_E_RANGE, "gae_x_vgae_head_prep: the fused form takes d = 16 (got %lld): use gae_vgae_head_fwd", (long long)d);
    GAE_REQUIRE(n > 0 && ldm >= d && ldm % 4 == 0, GAE_E_SIZE, "gae_x_vgae_head_prep: needs n > 0 and rows of whole 16-byte vectors");
    GAE_REQUIRE(mu && logstd && eps && z && prep && kl_partial && n_blocks_out, GAE_E_NULL, "gae_x_vgae_head_prep: NULL pointer");
    GAE_REQUIRE(gae::aligned16(mu) && gae::aligned16(logstd) && gae::aligned16(eps) && gae::aligned16(z), GAE_E_ALIGN,
                "gae_x_vgae_head_prep: operands must be 16-byte aligned");
    const int64_t blocks = (n + 63) / 64;
    GAE_REQUIRE(prep->DP == 16 && blocks <= prep->max_blocks && blocks <= kl_capacity && prep->Zt && prep->Zhi &&
                    prep->Zlo && prep->colsum_partial,
                GAE_E_WORKSPACE, "gae_x_vgae_head_prep: layout / KL buffer too small for %lld blocks", (long long)blocks);
    *n_blocks_out = blocks;
    hipLaunchKernelGGL(vgae_head_prep_kernel, dim3(unsigned(blocks)), dim3(256), 0, gae::as_stream(stream), mu, logstd, ldm,
                       eps, draw_eps, seed, offset, draw_dev, n, z, prep->Zt, prep->Zhi, prep->Zlo, prep->colsum_partial,
                       kl_partial);
    GAE_CHECK_LAUNCH("vgae_head_prep_kernel");
    return GAE_OK;
}

extern "C" int gae_vgae_head_bwd(const float *dz, const float *mu, const float *logstd, int64_t ldm, const float *eps,
                                 const float *gkl_dev, int64_t n, int64_t d, float *dmu, float *dlogstd, void *stream)
{
    GAE_REQUIRE(n > 0 && d > 0 && ldm >= d && d < (1 << 24), GAE_E_SIZE, "gae_vgae_head_bwd: needs n, d > 0 and ldm >= d");
    GAE_REQUIRE(mu && logstd && eps && dmu && dlogstd, GAE_E_NULL, "gae_vgae_head_bwd: NULL pointer");
    hipLaunchKernelGGL(vgae_head_bwd_kernel, dim3(vgae_blocks(n * d)), dim3(256), 0, gae::as_stream(stream), dz, mu,
                       logstd, eps, gkl_dev, float(1.0 / (double(n) * double(n))), n * d, dmu, dlogstd, ldm, int(d));
    GAE_CHECK_LAUNCH("vgae_head_bwd_kernel");
    return GAE_OK;
}
