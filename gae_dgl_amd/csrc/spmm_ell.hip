// K1/K2, third kernel family: "ell" -- the packed-neighbour-table form of the sparse aggregation
//     M = diag(rs) A diag(cs) H        (gae_dgl/gae.py:18-19,28: update_all(copy_src, sum); backward on A^T)
// written for the smallest possible instruction stream per gathered byte.
//
// Why: rocprofv3 (profiles/r02_spmm_diag.md) showed the row-group kernel of spmm.hip INSTRUCTION-ISSUE bound on
// the launches that matter (Pubmed F = 500: SIMDs 80 % busy, average L1->L2 read latency only 415 cycles, TLB
// clean, HBM/fabric at 60 %): per neighbour row it spent a ds_bpermute, a 64-bit address computation, exec-mask
// juggling for the predicated load, a zero fill for the empty slots and four scalar adds.  Here a neighbour costs
//     v_mov_dpp row_newbcast (1-2)  +  v_add_u32  +  buffer_load_dwordx4  +  2 v_pk_add_f32
//   * the lane that holds slot k of the table turns the column id into the BYTE OFFSET of that row of H once;
//     empty slots (-1 and the markers) become an offset behind the buffer: the raw-buffer bounds check makes the
//     load return zeros without touching memory -- no predication, no branches, no zero fill, and adding +0.0f
//     leaves every sum bit-identical to the CSR-order sum of spmm.hip / oracle/spmm_ref.c;
//   * slot k reaches the lanes of its group with one DPP row_newbcast (two bank-masked ones for 8-lane groups);
//   * all NB (4 or 8) neighbour rows of the RPG rows a lane group owns are requested before the first is used;
//     waves leave the slot loop together as soon as no row of the wave has a neighbour left (scalar branch).
// Ownership, XCD-aware block mapping, feature tiles and store policies are those of spmm_rowgroup2_kernel.
#include <string.h>

#include <utility>

#include "common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int32_t kEllOverflow = -2;   // last slot: the row continues in the CSR arrays
constexpr int32_t kEllSkip = -3;       // slot 0: heavy row of the skew plan (segment kernels produce it)

template <int CTRL, int BANK>
__device__ __forceinline__ unsigned dpp_mov(unsigned old, unsigned v)
{
    return unsigned(__builtin_amdgcn_update_dpp(int(old), int(v), CTRL, 0xF, BANK, false));
}

// value of lane (group base + K) of the caller's 16-lane row, for groups of LP16 lanes inside the row
template <int LP16, int K>
__device__ __forceinline__ unsigned bcast_slot(unsigned v)
{
    constexpr int ROW_NEWBCAST = 0x150;
    if constexpr (LP16 == 16) {
        return dpp_mov<ROW_NEWBCAST + K, 0xF>(v, v);
    } else if constexpr (LP16 == 8) {
        const unsigned t = dpp_mov<ROW_NEWBCAST + K, 0x3>(v, v);          // banks 0,1 = lanes 0..7
        return dpp_mov<ROW_NEWBCAST + 8 + K, 0xC>(t, v);                  // banks 2,3 = lanes 8..15
    } else {
        static_assert(LP16 == 4, "groups of 4, 8 or 16 lanes per DPP row");
        unsigned t = dpp_mov<ROW_NEWBCAST + K, 0x1>(v, v);
        t = dpp_mov<ROW_NEWBCAST + 4 + K, 0x2>(t, v);
        t = dpp_mov<ROW_NEWBCAST + 8 + K, 0x4>(t, v);
        return dpp_mov<ROW_NEWBCAST + 12 + K, 0x8>(t, v);
    }
}

// lanes of a wave that hold slot K (in their register K / LP16)
template <int LP16, int K>
constexpr unsigned long long holder_mask()
{
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l)
        if ((l & (LP16 - 1)) == (K % LP16)) m |= 1ull << l;
    return m;
}

template <typename T>
struct Vec16;
template <>
struct Vec16<float> {
    static constexpr int NV = 4;
    static __device__ __forceinline__ void add(float (&acc)[4], const u32x4 raw)
    {
        const f32x4 v = __builtin_bit_cast(f32x4, raw);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += v[i];
    }
    static __device__ __forceinline__ void fma(float (&acc)[4], const u32x4 raw, float c)
    {
        const f32x4 v = __builtin_bit_cast(f32x4, raw);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = fmaf(c, v[i], acc[i]);
    }
    static __device__ __forceinline__ u32x4 pack(const float (&acc)[4])
    {
        return __builtin_bit_cast(u32x4, (f32x4){acc[0], acc[1], acc[2], acc[3]});
    }
};
template <>
struct Vec16<unsigned short> {   // bf16 storage, fp32 accumulation
    static constexpr int NV = 8;
    static __device__ __forceinline__ void add(float (&acc)[8], const u32x4 raw)
    {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[2 * i] += __uint_as_float(raw[i] << 16);
            acc[2 * i + 1] += __uint_as_float(raw[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ void fma(float (&acc)[8], const u32x4 raw, float c)
    {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[2 * i] = fmaf(c, __uint_as_float(raw[i] << 16), acc[2 * i]);
            acc[2 * i + 1] = fmaf(c, __uint_as_float(raw[i] & 0xffff0000u), acc[2 * i + 1]);
        }
    }
    static __device__ __forceinline__ u32x4 pack(const float (&acc)[8])
    {
        u32x4 w;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            w[i] = unsigned(gae::f32_to_bf16(acc[2 * i])) | (unsigned(gae::f32_to_bf16(acc[2 * i + 1])) << 16);
        return w;
    }
};

}  // namespace
namespace gae {
gae::Knob g_spmm_ell_rpg{0};      // rows per lane group of the ell kernels: 0 = auto (1)
// "ell_side": how the side work of the fused layer (gae_x_gcn_layer_fused_wgrad) forms its outer product.  Bit 3: MFMA
// operands straight from global memory, no LDS, no barrier, when the output is at most 4 tiles of 16 x 16 (Pubmed:
// launch 9.0 us; gather alone 7.5); bit 2: on the matrix cores from LDS tiles (9.6 us; the form for larger outputs);
// neither: scalar LDS loop (10.4 us); bit 1: tile loads without a division per element (off: 11.9 us).  Default: all.
gae::Knob g_ell_side{14};
}
namespace {

struct EllArgs {
    const int32_t *indptr, *indices, *ell;
    const void *H;
    void *M;
    const float *row_scale, *col_scale;
    int64_t n_rows, ldm;          // ldm in elements
    unsigned ldh_bytes, h_bytes;  // row pitch and total size of H in bytes (h_bytes = n_cols * ldh_bytes < 2^32 - 2^16)
    unsigned n_cols, F, nvec;     // nvec = ceil(F / NV)
    unsigned n_row_blocks, n_ftiles, tile_vecs;   // tile_vecs <= LPR: 16-byte vectors per feature tile
    int xcd_tiled, store_pad, store_mode;
    // fused dense epilogue (EPI_J > 0):  Y = act(M W^T + b), W given as W[o * w_so + k * w_sk]
    const float *W, *bias;
    float *Y;
    int64_t ldy;
    int J, w_so, w_sk, act, store_m;
    // plain kernels (EPI_J = 0), fp32: M = act(aggregate + ep_bias) (ep_bias may be NULL; ep_act = GAE_ACT_*), and --
    // MASKED instantiations -- the gathered rows are H (.) [Hmask > 0] (Hmask: same layout as H): the two halves
    // around the dense product of a transform-first GCN layer (gae_spmm_csr_epilogue)
    const float *ep_bias;
    int ep_act;
    const void *Hmask;
    // GATHER MODE 2 (split-sum): H is the first of n_splits partial matrices split_bytes apart (what gae_xw_fwd leaves
    // when it splits a long f_in over thread blocks); a gathered row is the sum of its n_splits partial rows, added in
    // split order -- the value the stand-alone split reduction would have stored, without that launch
    int n_splits;
    unsigned split_bytes, empty_off;   // empty_off: byte offset of an empty table slot (behind every operand)
    // two-matrix form of the fused layer (gae_x_gcn_layer_fused2): stored rows >= w_split of the weight come from W2
    // (same strides), outputs >= w_split of the bias from bias2 -- two heads on one aggregate, one launch
    const float *W2, *bias2;
    int w_split, w_t;             // w_t: W is addressed transposed (the stored row is the input index k, not the output o)
    // side work of a fused-layer launch on the block's OWN rows (gae_x_gcn_layer_fused_wgrad): the weight gradient of the
    // layer whose backward this launch is, sw_part[block][o][i] = sum over the block's rows r of P[r][o] Q[r][i]
    // (P = the gathered operand dY, Q = the forward's stored aggregate) and, behind it, the column sums of P (db) --
    // per-block partial sums in row order, added up later (gae_adam_step's deferred reduction or a reduction launch)
    const float *sw_P, *sw_Q;
    float *sw_part;
    int64_t sw_ldp, sw_ldq, sw_stride;
    int sw_O, sw_I, sw_variant;
    // prepare step of the fused loss in the epilogue of the layer that produces Z (gae_x_gcn_layer_fused_prep, EPI_J =
    // 16): Zt = Y (.) dropout mask padded to 16 columns, its bf16 hi / lo, per-block fp64 column sums -- what
    // bce_prepare_kernel (decoder_bce.hip) computes from the stored Z
    float *pz_t;
    unsigned short *pz_hi, *pz_lo;
    double *pz_cs, *pz_scal;
    float *pz_mask;
    int64_t pz_ldmask;
    float pz_drop_p, pz_drop_scale;
    uint64_t pz_seed, pz_offset;
    const uint64_t *pz_draw;
    const int64_t *pz_counts;
    double pz_all_pairs;
};

template <typename T>
__device__ __forceinline__ void store16(T *p, const u32x4 w, int mode)
{
    if (mode == 2) {
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
    } else if (mode == 1) {
        __builtin_nontemporal_store(w, reinterpret_cast<u32x4 *>(p));
    } else {
        *reinterpret_cast<u32x4 *>(p) = w;
    }
}
__device__ __forceinline__ void store_elem(float *p, float v) { *p = v; }
__device__ __forceinline__ void store_elem(unsigned short *p, float v) { *p = gae::f32_to_bf16(v); }

// byte offsets of slots B0 + U... for the lanes of a group (compile-time slot numbers: the DPP control word is an
// immediate)
template <int LP16, int B0, int NREG, int... U>
__device__ __forceinline__ void slot_offsets(const unsigned (&off)[NREG], unsigned add, unsigned (&out)[sizeof...(U)],
                                             std::integer_sequence<int, U...>)
{
    ((out[U] = bcast_slot<LP16, (B0 + U) % LP16>(off[(B0 + U) / LP16]) + add), ...);
}

// v (.) [m > 0] on four fp32 values (the ReLU gate of a backward gather: v = dY, m = Y)
__device__ __forceinline__ u32x4 relu_gate(u32x4 v, const u32x4 m)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = __uint_as_float(m[i]) > 0.f ? v[i] : 0u;
    return v;
}

// Slots [B0, B0 + N) of the RPG rows, straight-line: every load is issued before the first add.  Rows of the wave
// with fewer neighbours get zeros from the bounds check.
template <typename T, int LP16, int RPG, int NREG, int B0, int N, bool SCALED, int MODE>
__device__ __forceinline__ void ell_slots(__amdgpu_buffer_rsrc_t rs_h, __amdgpu_buffer_rsrc_t rs_c,
                                          __amdgpu_buffer_rsrc_t rs_m,
                                          const unsigned (&off)[RPG][NREG], const unsigned (&coff)[RPG][NREG],
                                          unsigned lane_off, bool live, float (&acc)[RPG][Vec16<T>::NV], int n_splits,
                                          unsigned split_bytes)
{
    constexpr bool MASKED = MODE == 1;
    unsigned vo[RPG][N], co[RPG][N];
#pragma unroll
    for (int r = 0; r < RPG; ++r) {
        slot_offsets<LP16, B0>(off[r], lane_off, vo[r], std::make_integer_sequence<int, N>{});
        if (SCALED) slot_offsets<LP16, B0>(coff[r], 0u, co[r], std::make_integer_sequence<int, N>{});
    }
    if (live) {       // lanes beyond the row's last vector take no part in the memory traffic
        u32x4 raw[RPG][N], msk[RPG][MASKED ? N : 1];
        float cs[RPG][N];
#pragma unroll
        for (int r = 0; r < RPG; ++r)
#pragma unroll
            for (int u = 0; u < N; ++u) {
                raw[r][u] = __builtin_amdgcn_raw_buffer_load_b128(rs_h, vo[r][u], 0, 0);
                if (MASKED) msk[r][u] = __builtin_amdgcn_raw_buffer_load_b128(rs_m, vo[r][u], 0, 0);
                if (SCALED) cs[r][u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_c, co[r][u], 0, 0));
            }
        if constexpr (MODE == 2) {
            for (int sp = 1; sp < n_splits; ++sp) {          // + the row's other partials, in split order
                u32x4 more[RPG][N];
#pragma unroll
                for (int r = 0; r < RPG; ++r)
#pragma unroll
                    for (int u = 0; u < N; ++u)
                        more[r][u] = __builtin_amdgcn_raw_buffer_load_b128(rs_h, vo[r][u] + unsigned(sp) * split_bytes, 0, 0);
#pragma unroll
                for (int r = 0; r < RPG; ++r)
#pragma unroll
                    for (int u = 0; u < N; ++u)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            raw[r][u][i] = __float_as_uint(__uint_as_float(raw[r][u][i]) + __uint_as_float(more[r][u][i]));
            }
        }
#pragma unroll
        for (int r = 0; r < RPG; ++r)
#pragma unroll
            for (int u = 0; u < N; ++u) {
                if (MASKED) raw[r][u] = relu_gate(raw[r][u], msk[r][u]);
                if (SCALED) Vec16<T>::fma(acc[r], raw[r][u], cs[r][u]);
                else Vec16<T>::add(acc[r], raw[r][u]);
            }
    }
}

// One batch: slots [B0, B0 + NB).  Only as many slots as the longest row of the WAVE fills are touched (scalar
// count from the ballots, one straight-line body per count): a bounds-checked load that returns zeros still costs
// its address-unit cycles (no-edge launch of the Pubmed shape: 14.2 us with 8 such loads per row, 9 us without).
template <typename T, int LP16, int RPG, int NREG, int B0, int NB, bool SCALED, int MODE, int... U>
__device__ __forceinline__ void ell_batch(__amdgpu_buffer_rsrc_t rs_h, __amdgpu_buffer_rsrc_t rs_c,
                                          __amdgpu_buffer_rsrc_t rs_m,
                                          const unsigned (&off)[RPG][NREG], const unsigned (&coff)[RPG][NREG],
                                          const unsigned long long (&valid)[RPG][NREG], unsigned lane_off, bool live,
                                          float (&acc)[RPG][Vec16<T>::NV], int n_splits, unsigned split_bytes,
                                          std::integer_sequence<int, U...>)
{
    int cnt = 0;       // slots of this batch that hold a neighbour in some row of the wave (they fill from the left)
    (([&] {
         bool any = false;
#pragma unroll
         for (int r = 0; r < RPG; ++r) any = any || (valid[r][(B0 + U) / LP16] & holder_mask<LP16, B0 + U>()) != 0;
         cnt += any ? 1 : 0;
     }()),
     ...);
    cnt = __builtin_amdgcn_readfirstlane(cnt);
    switch (cnt) {
#define GAE_ELL_CASE(N)                                                                                      \
    case N:                                                                                                  \
        if constexpr (N <= NB)                                                                               \
            ell_slots<T, LP16, RPG, NREG, B0, (N <= NB ? N : 1), SCALED, MODE>(rs_h, rs_c, rs_m, off, coff, lane_off, live, acc, n_splits, split_bytes); \
        break;
        GAE_ELL_CASE(1) GAE_ELL_CASE(2) GAE_ELL_CASE(3) GAE_ELL_CASE(4)
        GAE_ELL_CASE(5) GAE_ELL_CASE(6) GAE_ELL_CASE(7) GAE_ELL_CASE(8)
#undef GAE_ELL_CASE
    default: break;
    }
}

// Edges [c0, e1) of a row (those behind its table slots), gathered by the whole wave: 64 column ids per coalesced load (the next 64
// requested before the rows of these are), lane group g of the 64 / LPR takes ids g, g + G, ... of the chunk -- up to
// 8 rows of H in flight per lane --, the groups' partial sums meet in a butterfly over the lane bits above the group.
// Every lane returns the row's sum for ITS 16-byte vector (lanes that are not `live` return zeros).  The order of the
// additions differs from the CSR order of the table slots: a long row's sum agrees with the oracle's to rounding,
// not to the bit.
template <typename T, int LPR, bool SCALED, int MODE, typename Args>
__device__ __forceinline__ void ell_long_row(const Args &a, __amdgpu_buffer_rsrc_t rs_h, __amdgpu_buffer_rsrc_t rs_c,
                                             __amdgpu_buffer_rsrc_t rs_m, int c0, int e1, int lane, unsigned lane_off,
                                             bool live, float (&part)[Vec16<T>::NV])
{
    constexpr int NV = Vec16<T>::NV, G = 64 / LPR, U = LPR;   // U ids of a chunk per group
    constexpr int UB = U < 8 ? U : 8;
#pragma unroll
    for (int i = 0; i < NV; ++i) part[i] = 0.f;
    const int gi = lane / LPR;
    // (requesting the first 64 ids in front of the slot batches -- the row's edge range loaded next to its table row --
    //  was measured: no gain, Pubmed with hubs 0.2253 -> 0.2279 ms per step)
    int32_t ids = (c0 + lane < e1) ? a.indices[c0 + lane] : -1;
    while (c0 < e1) {
        const int cnt = min(64, e1 - c0);                                  // scalar
        const int32_t cur = ids;
        if (c0 + 64 < e1) ids = (c0 + 64 + lane < e1) ? a.indices[c0 + 64 + lane] : -1;
#pragma unroll
        for (int u0 = 0; u0 < U; u0 += UB) {
            if (u0 * G < cnt) {                                            // scalar
                unsigned vo[UB], co[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int32_t j = __shfl(cur, gi + (u0 + u) * G);
                    vo[u] = unsigned(j) < a.n_cols ? unsigned(j) * a.ldh_bytes + lane_off : a.empty_off;
                    co[u] = min(unsigned(j), a.n_cols) * 4u;
                }
                if (live) {
                    u32x4 raw[UB], msk[MODE == 1 ? UB : 1];
                    float cs[UB];
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        raw[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_h, vo[u], 0, 0);
                        if (MODE == 1) msk[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_m, vo[u], 0, 0);
                        if (SCALED) cs[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_c, co[u], 0, 0));
                    }
                    // (every request above is issued before the first sum below: left alone, the scheduler pulls the
                    //  additions up between the loads and waits for the first before it issues the second -- three
                    //  round trips per chunk instead of one)
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (MODE == 2) {
                        for (int sp = 1; sp < a.n_splits; ++sp) {
#pragma unroll
                            for (int u = 0; u < UB; ++u) {
                                const u32x4 more = __builtin_amdgcn_raw_buffer_load_b128(
                                    rs_h, vo[u] + unsigned(sp) * a.split_bytes, 0, 0);
#pragma unroll
                                for (int i = 0; i < 4; ++i)
                                    raw[u][i] = __float_as_uint(__uint_as_float(raw[u][i]) + __uint_as_float(more[i]));
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        if (MODE == 1) raw[u] = relu_gate(raw[u], msk[u]);
                        if (SCALED) Vec16<T>::fma(part, raw[u], cs[u]);
                        else Vec16<T>::add(part, raw[u]);
                    }
                }
            }
        }
        c0 += 64;
    }
#pragma unroll
    for (int sft = LPR; sft < 64; sft <<= 1)
#pragma unroll
        for (int i = 0; i < NV; ++i) part[i] += __shfl_xor(part[i], sft);
}

// One wave = 64 / LPR lane groups x RPG rows of one feature tile.  (A persistent variant -- waves looping over
// their items with the next item's table rows prefetched -- was measured and is 10 % SLOWER on every shape: the
// stores of item i sit in front of the gathers of item i + 1 in the wave's in-order memory queue.)
// sum over the LPR lanes of a group (8 or 16 lanes inside a 16-lane DPP row), every lane gets the total
template <int LPR>
__device__ __forceinline__ float group_allreduce(float p)
{
    constexpr int QUAD_1032 = 0xB1, QUAD_2301 = 0x4E, ROW_MIRROR = 0x140, ROW_HALF_MIRROR = 0x141;
    p += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p), QUAD_1032, 0xF, 0xF, false));
    p += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p), QUAD_2301, 0xF, 0xF, false));
    p += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p), ROW_HALF_MIRROR, 0xF, 0xF, false));
    if constexpr (LPR == 16)
        p += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p), ROW_MIRROR, 0xF, 0xF, false));
    return p;
}

// EPI_J = 0: plain aggregation.  EPI_J = 16 / 32 (fp32, the whole row in one lane group: F <= 4 LPR <= 64): the
// GCN layer in one pass -- M = aggregate (stored only if a.store_m: the backward's dW needs it), then
// Y = act(M W^T + b) from the row still in registers: every lane multiplies its 4 features with its 4 columns of
// the weight rows (LDS copy, loaded once per block), a DPP butterfly adds the LPR lanes, lane l keeps outputs
// l J / LPR ....  NodeApplyModule after update_all (gae.py:28-29) without the round trip of M through HBM and
// without a second launch.
template <typename T, int LPR, int RPG, int W, int NB, bool SCALED, int EPI_J = 0, int MODE = 0>
__global__ __launch_bounds__(256) void spmm_ell_kernel(const EllArgs a)
{
    constexpr bool MASKED = MODE == 1;             // MODE: 0 plain gather, 1 ReLU-gated gather, 2 split-sum gather
    static_assert(MODE == 0 || sizeof(T) == 4, "the gated and split-sum gathers are fp32 forms");
    constexpr int NV = Vec16<T>::NV;
    constexpr int LDW = 68;                         // floats per LDS weight row: 64 columns + 4 (bank spread)
    __shared__ __attribute__((aligned(16))) float Ws[EPI_J > 0 ? EPI_J * LDW : 4];
    // the weights travel to LDS through registers: requested here, written (and the block synchronised) only in
    // front of the epilogue, so their round trip hides behind the table load and the gathers of this short block
    constexpr int WREG = EPI_J > 0 ? EPI_J * 64 / 256 : 1;
    float wreg[WREG];
    if constexpr (EPI_J > 0) {
        static_assert(sizeof(T) == 4 && RPG == 1 && (LPR == 8 || LPR == 16), "epilogue: fp32 rows of one lane group");
#pragma unroll
        for (int q = 0; q < WREG; ++q) {
            const int idx = threadIdx.x + 256 * q, o = idx >> 6, k = idx & 63;
            const bool in = o < a.J && unsigned(k) < a.F;
            const int srow = a.w_t ? k : o;                                   // row of the matrix as it is stored
            const bool second = a.W2 != nullptr && srow >= a.w_split;
            const int oo = (second && !a.w_t) ? o - a.w_split : o, kk = (second && a.w_t) ? k - a.w_split : k;
            const float *wb = second ? a.W2 : a.W;
            const float w = wb[in ? int64_t(oo) * a.w_so + int64_t(kk) * a.w_sk : 0];      // branch-free load
            wreg[q] = in ? w : 0.f;
        }
    }
    constexpr int LP16 = LPR < 16 ? LPR : 16;      // lanes of one group inside a 16-lane DPP row
    constexpr int NREG = (W + LP16 - 1) / LP16;    // table registers per lane and row
    constexpr int GPB = 256 / LPR, RPB = GPB * RPG;
    static_assert(W % NB == 0 && NB % 4 == 0, "slots are consumed in whole batches");
    unsigned blk, ftile;
    if (a.xcd_tiled) {
        // XCD x (= blockIdx % 8) owns feature tiles x, x + 8, ... and sweeps all row blocks of a tile before the
        // next: the tile's slice of H stays in that XCD's private L2 while it is gathered
        const unsigned xcd = blockIdx.x % gae::kNumXcd, k = blockIdx.x / gae::kNumXcd;
        const unsigned tl = k / a.n_row_blocks;
        blk = k - tl * a.n_row_blocks;
        ftile = xcd + gae::kNumXcd * tl;
        if (ftile >= a.n_ftiles) return;
    } else {
        blk = gae::xcd_remap(blockIdx.x, gridDim.x);
        ftile = blockIdx.y;
    }
    const int lane = threadIdx.x & 63;
    const int lig = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    const unsigned fvec = ftile * a.tile_vecs + lig;   // this lane's 16-byte vector of the row
    const bool live = unsigned(lig) < a.tile_vecs && fvec < a.nvec;
    const unsigned lane_off = fvec * 16u;

    // side work (EPI kernels): the block's own rows of P and Q are requested here, used behind the epilogue
    constexpr int SW_ROWS = 256 / LPR;                         // = rows per block of the EPI kernels (RPG = 1)
    constexpr int SWP = EPI_J > 0 ? SW_ROWS * 32 / 256 : 1, SWQ = EPI_J > 0 ? SW_ROWS * 64 / 256 : 1;
    __shared__ float SwP[EPI_J > 0 ? SW_ROWS * 32 : 1], SwQ[EPI_J > 0 ? SW_ROWS * 64 : 1];
    float swp[SWP], swq[SWQ];
    constexpr int JPL_K = EPI_J > 0 ? EPI_J / LPR : 1;        // outputs a lane keeps of its row
    // (prepare epilogue) the two device-side scalars it needs, requested HERE: a load issued where they are used would
    // stall every wave at the end of the kernel for a full round trip
    int64_t pz_n_valid = a.n_rows;
    uint64_t pz_draw_idx = 0;
    if constexpr (EPI_J == 16) {
        if (a.pz_t != nullptr) {
            if (a.pz_counts) pz_n_valid = a.pz_counts[0];
            if (a.pz_drop_p > 0.f && a.pz_draw) pz_draw_idx = *a.pz_draw;
        }
    }
    double pz_sum[JPL_K];                                      // (prepare epilogue) this lane's Zt values, for the column sums
    float pz_m[JPL_K];                                         // ... and its dropout multipliers
#pragma unroll
    for (int q = 0; q < JPL_K; ++q) pz_m[q] = 1.f;
#pragma unroll
    for (int q = 0; q < JPL_K; ++q) pz_sum[q] = 0.0;
    // direct form of the side work (at most 4 output tiles: one per wave): every lane requests the MFMA operands of
    // its wave's tile HERE, straight from P / Q (4 rows x 16 columns per instruction: whole 64-byte pieces of rows) --
    // no LDS staging, no barrier; they arrive while the wave gathers
    constexpr int SW_K = EPI_J > 0 ? SW_ROWS / 4 : 1;
    float sda[SW_K], sdb[SW_K];
    bool sw_direct = false;
    int sd_mt = 0, sd_nt = 0;
    bool sd_ones = false, sd_live = false;
    if constexpr (EPI_J > 0) {
        if (a.sw_part != nullptr && (a.sw_variant & 8)) {
            const int mt_n = (a.sw_O + 15) >> 4, nt_n = (a.sw_I + 15) >> 4;
            sw_direct = mt_n * nt_n + mt_n <= 4;
            if (sw_direct) {
                const int tile = 3 - int(threadIdx.x >> 6);       // tile ids: column sums (db) first, then products
                sd_live = tile < mt_n * nt_n + mt_n;
                sd_ones = tile < mt_n;
                const int pt = tile - mt_n;
                sd_mt = sd_ones ? tile : pt / nt_n;
                sd_nt = sd_ones ? 0 : pt - sd_mt * nt_n;
                const int l15 = threadIdx.x & 15, g4 = (threadIdx.x & 63) >> 4;
                const int o = sd_mt * 16 + l15, i = sd_nt * 16 + l15;
#pragma unroll
                for (int st = 0; st < SW_K; ++st) {
                    const int64_t rw = int64_t(blk) * SW_ROWS + 4 * st + g4;
                    const bool okr = sd_live && rw < a.n_rows;
                    const bool oka = okr && o < a.sw_O, okb = okr && !sd_ones && i < a.sw_I;
                    const float av = a.sw_P[oka ? rw * a.sw_ldp + o : 0];
                    const float bv = a.sw_Q[okb ? rw * a.sw_ldq + i : 0];
                    sda[st] = oka ? av : 0.f;
                    sdb[st] = sd_ones ? 1.f : (okb ? bv : 0.f);
                }
            }
        }
    }
    if constexpr (EPI_J > 0) {
        if (a.sw_part != nullptr && !sw_direct) {              // block-uniform
            // element idx of the block's [rows][width] tile; rows that are stored end to end (ld == width: the usual
            // case) are one contiguous piece of memory -- no division per element
            auto tile_load = [&](const float *base, int64_t ld, int width, int idx) {
                int64_t off;
                bool ok;
                if (ld == width && (a.sw_variant & 2)) {
                    off = int64_t(blk) * SW_ROWS * width + idx;
                    ok = idx < SW_ROWS * width && off < a.n_rows * width;
                } else {
                    const int r = idx / width, c = idx - r * width;
                    const int64_t rw = int64_t(blk) * SW_ROWS + r;
                    off = rw * ld + c;
                    ok = r < SW_ROWS && rw < a.n_rows;
                }
                const float v = base[ok ? off : 0];
                return ok ? v : 0.f;
            };
#pragma unroll
            for (int q = 0; q < SWP; ++q) swp[q] = tile_load(a.sw_P, a.sw_ldp, a.sw_O, threadIdx.x + 256 * q);
#pragma unroll
            for (int q = 0; q < SWQ; ++q) swq[q] = tile_load(a.sw_Q, a.sw_ldq, a.sw_I, threadIdx.x + 256 * q);
        }
    }

    __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.H), 0, int(a.h_bytes), 0x00020000);
    __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(SCALED ? a.col_scale : nullptr), 0, SCALED ? int(a.n_cols * 4u) : 0, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(MASKED ? a.Hmask : nullptr), 0,
                                                                    MASKED ? int(a.h_bytes) : 0, 0x00020000);

    int64_t row[RPG];
    unsigned off[RPG][NREG], coff[RPG][NREG];
    unsigned long long valid[RPG][NREG], skipm[RPG], ovfm[RPG];
    float acc[RPG][NV];
#pragma unroll
    for (int r = 0; r < RPG; ++r) {
        row[r] = int64_t(blk) * RPB + r * GPB + grp;
#pragma unroll
        for (int i = 0; i < NV; ++i) acc[r][i] = 0.f;
        // ---- the table row: slot q * LP16 + (lane % LP16) sits in register q (every 16-lane row of a wide group
        //      holds its own copy: the re-read is served by the same L1 line)
        int32_t first = -1, last = -1;
        int32_t jraw[NREG];
#pragma unroll
        for (int q = 0; q < NREG; ++q) {
            const int s = q * LP16 + (lane & (LP16 - 1));
            jraw[q] = -1;
            if (row[r] < a.n_rows && s < W) jraw[q] = a.ell[row[r] * W + s];
        }
        if constexpr (EPI_J == 16) {
            // (prepare epilogue) this lane's dropout multipliers, computed while the table row is on its way: the
            // Philox rounds would otherwise sit at the very end of the kernel, behind the gather
            if (a.pz_t != nullptr && r == 0) {
                const bool draw = a.pz_drop_p > 0.f;
                uint32_t c[4] = {0u, 0u, 0u, 0u};
                int64_t c_of = -1;                                        // the counter c belongs to
#pragma unroll
                for (int q = 0; q < JPL_K; ++q) {
                    const int o = lig * JPL_K + q;
                    float m = 1.f;
                    if (row[r] < a.n_rows && o < a.J) {
                        if (row[r] >= pz_n_valid) {       // padding row of a fixed-capacity batch: zero row of Zt, zero mask
                            m = 0.f;
                            if (draw) a.pz_mask[row[r] * a.pz_ldmask + o] = 0.f;
                        } else if (draw) {
                            // the Philox stream of gae_dropout_mask: element e = i d + k of the [n, d] mask; a lane's
                            // outputs are adjacent elements and mostly share one counter (4 per draw)
                            const int64_t e = row[r] * a.J + o;
                            if ((e >> 2) != c_of) {
                                c_of = e >> 2;
                                gae::philox4x32_10(a.pz_offset + uint64_t(c_of), pz_draw_idx, a.pz_seed, c);
                            }
                            const uint32_t bits = (e & 2) ? ((e & 1) ? c[3] : c[2]) : ((e & 1) ? c[1] : c[0]);
                            m = gae::dropout_multiplier(bits, a.pz_drop_p, a.pz_drop_scale);
                            a.pz_mask[row[r] * a.pz_ldmask + o] = m;
                        } else if (a.pz_mask) {
                            m = a.pz_mask[row[r] * a.pz_ldmask + o];
                        }
                    }
                    pz_m[q] = m;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NREG; ++q) {
            const int s = q * LP16 + (lane & (LP16 - 1));
            const int32_t j = jraw[q];
            off[r][q] = unsigned(j) < a.n_cols ? unsigned(j) * a.ldh_bytes : a.empty_off;   // empty / marker -> behind the buffer
            if (SCALED) coff[r][q] = min(unsigned(j), a.n_cols) * 4u;
            valid[r][q] = __builtin_amdgcn_ballot_w64(j >= 0);
            if (s == 0) first = j;
            if (s == W - 1) last = j;
        }
        skipm[r] = __builtin_amdgcn_ballot_w64(first == kEllSkip);
        ovfm[r] = __builtin_amdgcn_ballot_w64(last == kEllOverflow);
    }

    // ---- slot batches; `more` is a scalar: no row of this wave has a neighbour in slot B0
#define GAE_ELL_BATCH(B0)                                                                                          \
    if constexpr ((B0) < W) {                                                                                      \
        bool more = (B0) == 0;                                                                                     \
        _Pragma("unroll") for (int r = 0; r < RPG; ++r)                                                            \
            more = more || (valid[r][(B0) / LP16] & holder_mask<LP16, (B0)>()) != 0;                               \
        if (!more) goto slots_done;                                                                                \
        ell_batch<T, LP16, RPG, NREG, (B0), NB, SCALED, MODE>(rs_h, rs_c, rs_m, off, coff, valid, lane_off, live, acc, \
                                                              a.n_splits, a.split_bytes,                           \
                                                        std::make_integer_sequence<int, NB>{});                    \
    }
    GAE_ELL_BATCH(0)
    GAE_ELL_BATCH(NB)
    GAE_ELL_BATCH(2 * NB)
    GAE_ELL_BATCH(3 * NB)
#undef GAE_ELL_BATCH
slots_done:
    // ---- rows longer than the table continue from the CSR arrays, the WHOLE WAVE on one such row at a time
    //      (ell_long_row): the real citation graphs have hubs of 100 - 170 neighbours
#pragma unroll
    for (int r = 0; r < RPG; ++r) {
        const int holder0 = lane & ~(LPR - 1);                            // lane holding slot 0 of this group
        if ((skipm[r] >> holder0) & 1ull) row[r] = a.n_rows;              // heavy row: not produced here
        // one bit per lane group whose row continues (the lane of its first DPP row that holds slot W - 1)
        unsigned long long todo = ovfm[r] & holder_mask<LPR, (W - 1) % LP16>();
        while (todo != 0) {                                               // scalar: almost never entered
            const int hl = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int g0 = hl & ~(LPR - 1);
            const int rw = __builtin_amdgcn_readlane(int(row[r]), g0);    // (tables exist for < 2^31 rows)
            // both ends of the row's edge range in ONE round trip (lanes 0 and 1; two scalar loads came out one
            // behind the other)
            const int32_t ends = lane < 2 ? a.indptr[rw + lane] : 0;
            float part[NV];
            ell_long_row<T, LPR, SCALED, MODE>(a, rs_h, rs_c, rs_m, __builtin_amdgcn_readlane(ends, 0) + (W - 1),
                                               __builtin_amdgcn_readlane(ends, 1), lane, lane_off, live, part);
            if ((lane & ~(LPR - 1)) == g0) {
#pragma unroll
                for (int i = 0; i < NV; ++i) acc[r][i] += part[i];
            }
        }
    }
    if (EPI_J == 0 && !live) return;
    if constexpr (EPI_J > 0) {
#pragma unroll
        for (int q = 0; q < WREG; ++q) {
            const int idx = threadIdx.x + 256 * q;
            Ws[(idx >> 6) * LDW + (idx & 63)] = wreg[q];
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < RPG; ++r) {
        if (EPI_J == 0 && row[r] >= a.n_rows) continue;
        if (SCALED) {
            const float rs = row[r] < a.n_rows ? a.row_scale[row[r]] : 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) acc[r][i] *= rs;
        }
        if constexpr (EPI_J == 0 && sizeof(T) == 4) {
            if (a.ep_bias != nullptr || a.ep_act != GAE_ACT_IDENTITY) {     // scalar test: plain launches skip it
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const unsigned c = fvec * NV + i;
                    float y = acc[r][i] + ((a.ep_bias != nullptr && live && c < a.F) ? a.ep_bias[c] : 0.f);
                    if (a.ep_act == GAE_ACT_RELU) y = fmaxf(y, 0.f);
                    acc[r][i] = y;
                }
            }
        }
        if (live && row[r] < a.n_rows && (EPI_J == 0 || a.store_m)) {
            T *mp = static_cast<T *>(a.M) + row[r] * a.ldm + int64_t(fvec) * NV;
            if ((fvec + 1) * NV <= a.F || a.store_pad) {
                store16(mp, Vec16<T>::pack(acc[r]), a.store_mode);
            } else {
#pragma unroll
                for (int i = 0; i < NV; ++i)
                    if (fvec * NV + i < a.F) store_elem(mp + i, acc[r][i]);
            }
        }
        if constexpr (EPI_J > 0) {
            // dead lanes (beyond the row's last vector) and rows beyond n_rows carry zeros and take part in the
            // butterfly; weight columns >= F are zero in LDS
            constexpr int JPL = EPI_J / LPR;          // outputs kept per lane
            float yv[JPL];
#pragma unroll
            for (int q = 0; q < JPL; ++q) yv[q] = 0.f;
#pragma unroll
            for (int j = 0; j < EPI_J; ++j) {
                const f32x4 w = *reinterpret_cast<const f32x4 *>(&Ws[j * LDW + 4 * lig]);
                float p = fmaf(acc[r][3], w[3], fmaf(acc[r][2], w[2], fmaf(acc[r][1], w[1], acc[r][0] * w[0])));
                p = group_allreduce<LPR>(p);
                if (j / JPL == lig) yv[j % JPL] = p;
            }
            if (row[r] < a.n_rows) {
#pragma unroll
                for (int q = 0; q < JPL; ++q) {
                    const int o = lig * JPL + q;
                    if (o < a.J) {
                        float y = yv[q] + (a.bias ? ((a.bias2 != nullptr && o >= a.w_split) ? a.bias2[o - a.w_split] : a.bias[o]) : 0.f);
                        if (a.act == GAE_ACT_RELU) y = fmaxf(y, 0.f);
                        a.Y[row[r] * a.ldy + o] = y;
                        yv[q] = y;
                    }
                }
            }
            if constexpr (EPI_J == 16) {
                if (a.pz_t != nullptr) {                                  // block-uniform
                    const bool in = row[r] < a.n_rows;
#pragma unroll
                    for (int q = 0; q < JPL; ++q) {
                        const int o = lig * JPL + q;
                        const float v = (in && o < a.J) ? yv[q] * pz_m[q] : 0.f;      // (multipliers: see the table load)
                        if (in) {
                            a.pz_t[row[r] * 16 + o] = v;
                            const unsigned short hi = gae::f32_to_bf16(v);
                            a.pz_hi[row[r] * 16 + o] = hi;
                            a.pz_lo[row[r] * 16 + o] = gae::f32_to_bf16(v - gae::bf16_to_f32(hi));
                        }
                        pz_sum[q] = double(v);
                    }
                }
            }
        }
    }
    if constexpr (EPI_J == 16) {
        if (a.pz_t != nullptr) {
            // column sums of the block's rows, fixed order: butterfly over the rows of a wave (lanes with the same
            // outputs are LPR apart), then the 4 waves in order
            double *cred = reinterpret_cast<double *>(SwQ);              // [wave][16]
#pragma unroll
            for (int q = 0; q < JPL_K; ++q) {
#pragma unroll
                for (int off = LPR; off < 64; off <<= 1) pz_sum[q] += __shfl_xor(pz_sum[q], off, 64);
                if ((threadIdx.x & 63) < LPR) cred[(threadIdx.x >> 6) * 16 + lig * JPL_K + q] = pz_sum[q];
            }
            __syncthreads();
            if (threadIdx.x < 16) {
                const double t = ((cred[threadIdx.x] + cred[16 + threadIdx.x]) + cred[32 + threadIdx.x]) + cred[48 + threadIdx.x];
                a.pz_cs[(int64_t(blk) * 2 + 0) * 16 + threadIdx.x] = t;                        // all rows
                a.pz_cs[(int64_t(blk) * 2 + 1) * 16 + threadIdx.x] = t;                        // ... = the row window
            }
            if (a.pz_counts && blk == 0 && threadIdx.x == 0) {
                const double nv = double(a.pz_counts[0]), ev = double(a.pz_counts[1]);
                a.pz_scal[0] = ev > 0 ? (nv * nv - ev) / ev : 0.0;      // pos_weight (train_inductive.py:46)
                a.pz_scal[1] = nv > 0 ? 1.0 / (nv * nv) : 0.0;          // mean over the N^2 real pairs
                a.pz_scal[2] = a.pz_all_pairs - nv * nv;                // evaluated pairs with a zero operand
            }
        }
    }
    if constexpr (EPI_J > 0) {
        if (a.sw_part != nullptr && sw_direct) {
            if (sd_live) {                                        // wave-uniform
                const int O = a.sw_O, I = a.sw_I;
                const int l15 = threadIdx.x & 15, g4 = (threadIdx.x & 63) >> 4;
                typedef float f32x4_t __attribute__((ext_vector_type(4)));
                f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int st = 0; st < SW_K; ++st) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(sda[st], sdb[st], acc, 0, 0, 0);
                float *pp = a.sw_part + int64_t(blk) * a.sw_stride;
                const int i = sd_nt * 16 + l15;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int oo = sd_mt * 16 + 4 * g4 + q;
                    if (sd_ones) { if (oo < O && l15 == 0) pp[O * I + oo] = acc[q]; }
                    else if (oo < O && i < I) pp[oo * I + i] = acc[q];
                }
            }
        } else if (a.sw_part != nullptr) {
            const int O = a.sw_O, I = a.sw_I;
#pragma unroll
            for (int q = 0; q < SWP; ++q) SwP[threadIdx.x + 256 * q] = swp[q];       // [r][O], rows end to end
#pragma unroll
            for (int q = 0; q < SWQ; ++q) SwQ[threadIdx.x + 256 * q] = swq[q];       // [r][I]
            __syncthreads();
            float *pp = a.sw_part + int64_t(blk) * a.sw_stride;
            if (a.sw_variant & 4) {
                // the outer product on the matrix cores: out [O x I] = P^T [O x rows] Q [rows x I], 16 x 16 output
                // tiles dealt to the 4 waves, K = the block's rows in steps of 4 (v_mfma_f32_16x16x4_f32: exact fp32
                // products, one fixed accumulation order)
                const int wv = threadIdx.x >> 6, l15 = threadIdx.x & 15, g4 = (threadIdx.x & 63) >> 4;
                const int mt_n = (O + 15) >> 4, nt_n = (I + 15) >> 4;
                // tiles mt_n * nt_n .. + mt_n - 1: the column sums of P (db) as P^T times a column of ones; the
                // highest waves take them first (they have the fewest product tiles)
                for (int tile = 3 - wv; tile < mt_n * nt_n + mt_n; tile += 4) {
                    const bool ones = tile < mt_n;                        // (tile ids: column sums first, then products)
                    const int pt = tile - mt_n;
                    const int mt = ones ? tile : pt / nt_n, nt = ones ? 0 : pt - mt * nt_n;
                    const int o = mt * 16 + l15, i = nt * 16 + l15;
                    typedef float f32x4_t __attribute__((ext_vector_type(4)));
                    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int st = 0; st < SW_ROWS / 4; ++st) {
                        const int r = 4 * st + g4;
                        const float av = o < O ? SwP[r * O + o] : 0.f;
                        const float bv = ones ? 1.f : (i < I ? SwQ[r * I + i] : 0.f);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int oo = mt * 16 + 4 * g4 + q;
                        if (ones) { if (oo < O && l15 == 0) pp[O * I + oo] = acc[q]; }
                        else if (oo < O && i < I) pp[oo * I + i] = acc[q];
                    }
                }
            } else {
                for (int e = threadIdx.x; e < O * I; e += 256) {
                    const int o = e / I, i = e - o * I;
                    float t = 0.f;
#pragma unroll 8
                    for (int r = 0; r < SW_ROWS; ++r) t = fmaf(SwP[r * O + o], SwQ[r * I + i], t);   // row order
                    pp[e] = t;
                }
            }
            if (!(a.sw_variant & 4) && int(threadIdx.x) < O) {
                float t = 0.f;
                for (int r = 0; r < SW_ROWS; ++r) t += SwP[r * O + threadIdx.x];
                pp[O * I + threadIdx.x] = t;
            }
        }
    }
}

template <typename T, int LPR, int RPG, int W, int NB>
int launch_ell(const EllArgs &a, bool scaled, hipStream_t s)
{
    constexpr int RPB = (256 / LPR) * RPG;
    EllArgs b = a;
    b.n_row_blocks = unsigned((a.n_rows + RPB - 1) / RPB);
    b.n_ftiles = (a.nvec + a.tile_vecs - 1) / a.tile_vecs;
    const dim3 grid = a.xcd_tiled
                          ? dim3(gae::kNumXcd * ((b.n_ftiles + gae::kNumXcd - 1) / gae::kNumXcd) * b.n_row_blocks)
                          : dim3(b.n_row_blocks, b.n_ftiles);
    if (scaled) hipLaunchKernelGGL((spmm_ell_kernel<T, LPR, RPG, W, NB, true>), grid, dim3(256), 0, s, b);
    else hipLaunchKernelGGL((spmm_ell_kernel<T, LPR, RPG, W, NB, false>), grid, dim3(256), 0, s, b);
    GAE_CHECK_LAUNCH("spmm_ell_kernel");
    return GAE_OK;
}

template <int LPR, int W, int NB, int EPI_J>
int launch_ell_epi(const EllArgs &a, bool scaled, hipStream_t s)
{
    constexpr int RPB = 256 / LPR;
    EllArgs b = a;
    b.n_row_blocks = unsigned((a.n_rows + RPB - 1) / RPB);
    b.n_ftiles = 1;
    const dim3 grid(b.n_row_blocks, 1);
    if (scaled) hipLaunchKernelGGL((spmm_ell_kernel<float, LPR, 1, W, NB, true, EPI_J>), grid, dim3(256), 0, s, b);
    else hipLaunchKernelGGL((spmm_ell_kernel<float, LPR, 1, W, NB, false, EPI_J>), grid, dim3(256), 0, s, b);
    GAE_CHECK_LAUNCH("spmm_ell_kernel (fused layer)");
    return GAE_OK;
}

template <int LPR, int EPI_J>
int launch_ell_epi_w(const EllArgs &a, int W, bool scaled, hipStream_t s)
{
    if (W == 4) return launch_ell_epi<LPR, 4, 4, EPI_J>(a, scaled, s);
    if (W == 8) return launch_ell_epi<LPR, 8, 8, EPI_J>(a, scaled, s);
    return launch_ell_epi<LPR, 16, 8, EPI_J>(a, scaled, s);
}

template <typename T, int LPR, int RPG>
int launch_ell_w(const EllArgs &a, int W, bool scaled, hipStream_t s)
{
    if (W == 4) return launch_ell<T, LPR, RPG, 4, 4>(a, scaled, s);
    if (W == 8) return launch_ell<T, LPR, RPG, 8, 8>(a, scaled, s);
    return launch_ell<T, LPR, RPG, 16, 8>(a, scaled, s);
}

template <typename T>
int launch_ell_t(const EllArgs &a, int lpr, int rpg, int W, bool scaled, hipStream_t s)
{
#define GAE_ELL_L(LPR)                                                                   \
    if (lpr == LPR) {                                                                    \
        if (rpg >= 2 && sizeof(T) == 4) return launch_ell_w<T, LPR, 2>(a, W, scaled, s); \
        return launch_ell_w<T, LPR, 1>(a, W, scaled, s);                                 \
    }
    GAE_ELL_L(8)
    GAE_ELL_L(16)
    GAE_ELL_L(32)
    GAE_ELL_L(64)
#undef GAE_ELL_L
    gae::set_error("spmm_ell: unsupported lane-group width %d", lpr);
    return GAE_E_RANGE;
}

} // namespace

namespace gae {

Knob *spmm_ell_knob(const char *name)
{
    if (strcmp(name, "spmm_ell_rpg") == 0) return &g_spmm_ell_rpg;
    if (strcmp(name, "ell_side") == 0) return &g_ell_side;
    return nullptr;
}

// Can the ell family run this launch?  (16-byte vector layout is the caller's precondition.)
bool spmm_ell_usable(int64_t n_cols, int64_t ldh, int elem, int ell_width, int tile_vecs)
{
    const int64_t h_bytes = n_cols * ldh * elem;
    return (ell_width == 4 || ell_width == 8 || ell_width == 16) && tile_vecs > 4 && tile_vecs <= 64 &&
           h_bytes + (int64_t(1) << 16) < (int64_t(1) << 32) && n_cols > 0;
}

// tile_vecs: 16-byte vectors of one feature tile (1 .. 64; the lane group is the next power of two, at least 8);
// xcd_tiled as GAE_SPMM_TILE decided
int spmm_ell_launch(const int32_t *indptr, const int32_t *indices, const int32_t *ell, int ell_width, int64_t n_rows,
                    int64_t n_cols, const void *H, int64_t ldh, void *M, int64_t ldm, int F, int dtype,
                    const float *rs, const float *cs, int tile_vecs, int xcd_tiled, int store_pad, int store_mode,
                    hipStream_t s)
{
    int lanes_per_row = 8;
    while (lanes_per_row < tile_vecs) lanes_per_row *= 2;
    const int elem = dtype == GAE_F32 ? 4 : 2, nv = 16 / elem;
    EllArgs a{};
    a.indptr = indptr; a.indices = indices; a.ell = ell;
    a.H = H; a.M = M; a.row_scale = rs; a.col_scale = cs;
    a.n_rows = n_rows; a.ldm = ldm;
    a.ldh_bytes = unsigned(ldh * elem);
    a.h_bytes = unsigned(n_cols * ldh * elem);
    a.empty_off = a.h_bytes; a.n_splits = 1;
    a.n_cols = unsigned(n_cols); a.F = unsigned(F); a.nvec = unsigned((F + nv - 1) / nv);
    a.tile_vecs = unsigned(tile_vecs);
    a.xcd_tiled = xcd_tiled; a.store_pad = store_pad; a.store_mode = store_mode;
    const int rpg = g_spmm_ell_rpg > 0 ? g_spmm_ell_rpg : 1;
    const bool scaled = rs != nullptr;
    if (dtype == GAE_F32) return launch_ell_t<float>(a, lanes_per_row, rpg, ell_width, scaled, s);
    return launch_ell_t<unsigned short>(a, lanes_per_row, 1, ell_width, scaled, s);
}

} // namespace gae

// side work of a fused-layer launch (EllArgs::sw_*)
struct FusedSide { const float *P, *Q; float *part; int64_t ldp, ldq, stride; int O, I; };
// prepare step of the loss in the epilogue (EllArgs::pz_*)
struct FusedPrep {
    const gae_bce_prep *lay;
    float *mask; int64_t ldmask; float drop_p; uint64_t seed, offset; const uint64_t *draw; const int64_t *counts;
};

// GCN.forward (gae_dgl/gae.py:26-31) in one launch: update_all(copy_src, sum) and NodeApplyModule (Linear + bias +
// activation) -- see the EPI_J form of spmm_ell_kernel.
static int gcn_layer_fused_impl(const int32_t *indptr, const int32_t *indices, int64_t n_rows, int64_t n_cols,
                                const float *H, int64_t ldh, float *M, int64_t ldm, int64_t F, const float *row_scale,
                                const float *col_scale, const gae_spmm_plan *plan, const float *W, int64_t w_stride_out,
                                int64_t w_stride_in, const float *bias, int64_t J, int act, float *Y, int64_t ldy,
                                const float *W2, const float *bias2, int64_t w_split, int w_transposed, void *stream,
                                const FusedSide *side = nullptr, const FusedPrep *prep = nullptr);

extern "C" int gae_gcn_layer_fused(const int32_t *indptr, const int32_t *indices, int64_t n_rows, int64_t n_cols,
                                   const float *H, int64_t ldh, float *M, int64_t ldm, int64_t F,
                                   const float *row_scale, const float *col_scale, const gae_spmm_plan *plan,
                                   const float *W, int64_t w_stride_out, int64_t w_stride_in, const float *bias,
                                   int64_t J, int act, float *Y, int64_t ldy, void *stream)
{
    return gcn_layer_fused_impl(indptr, indices, n_rows, n_cols, H, ldh, M, ldm, F, row_scale, col_scale, plan, W,
                                w_stride_out, w_stride_in, bias, J, act, Y, ldy, nullptr, nullptr, 0, 0, stream);
}

// Two GCN heads on ONE aggregate in one launch (VGAE: mu and log sigma, gae_dgl_amd/vgae.py): the weight is the
// stack [W; W2] along its STORED rows (both matrices share the strides; w_split rows belong to W), the bias [b; b2].
// Forward (w_transposed = 0): Y[:, :w_split] = act(M W^T + b), Y[:, w_split:] = act(M W2^T + b2).  Backward of
// identity heads (w_transposed = 1, strides swapped as in gae_gcn_layer_fused): dH = (A^T dY) [W; W2].
extern "C" int gae_x_gcn_layer_fused2(const int32_t *indptr, const int32_t *indices, int64_t n_rows, int64_t n_cols,
                                    const float *H, int64_t ldh, float *M, int64_t ldm, int64_t F,
                                    const float *row_scale, const float *col_scale, const gae_spmm_plan *plan,
                                    const float *W, const float *W2, int64_t w_split, int w_transposed,
                                    int64_t w_stride_out, int64_t w_stride_in, const float *bias, const float *bias2,
                                    int64_t J, int act, float *Y, int64_t ldy, void *stream)
{
    GAE_REQUIRE(W2 != nullptr && w_split >= 1, GAE_E_NULL, "gae_x_gcn_layer_fused2: W2 / w_split missing");
    GAE_REQUIRE(w_split < (w_transposed ? F : J), GAE_E_RANGE, "gae_x_gcn_layer_fused2: w_split outside the stacked rows");
    GAE_REQUIRE((bias == nullptr) == (bias2 == nullptr), GAE_E_NULL, "gae_x_gcn_layer_fused2: give both biases or none");
    return gcn_layer_fused_impl(indptr, indices, n_rows, n_cols, H, ldh, M, ldm, F, row_scale, col_scale, plan, W,
                                w_stride_out, w_stride_in, bias, J, act, Y, ldy, W2, bias2, w_split, w_transposed, stream);
}

static int gcn_layer_fused_impl(const int32_t *indptr, const int32_t *indices, int64_t n_rows, int64_t n_cols,
                                const float *H, int64_t ldh, float *M, int64_t ldm, int64_t F, const float *row_scale,
                                const float *col_scale, const gae_spmm_plan *plan, const float *W, int64_t w_stride_out,
                                int64_t w_stride_in, const float *bias, int64_t J, int act, float *Y, int64_t ldy,
                                const float *W2, const float *bias2, int64_t w_split, int w_transposed, void *stream,
                                const FusedSide *side, const FusedPrep *prep)
{
    GAE_REQUIRE(n_rows >= 0 && n_cols >= 0, GAE_E_SIZE, "gae_gcn_layer_fused: negative size");
    GAE_REQUIRE(F >= 1 && F <= 64 && J >= 1 && J <= 32, GAE_E_RANGE,
                "gae_gcn_layer_fused: needs 1 <= F <= 64 and 1 <= J <= 32 (got %lld, %lld): use gae_spmm_csr + gae_linear_fwd",
                (long long)F, (long long)J);
    GAE_REQUIRE(act == GAE_ACT_IDENTITY || act == GAE_ACT_RELU, GAE_E_RANGE, "gae_gcn_layer_fused: act %d", act);
    GAE_REQUIRE((row_scale == nullptr) == (col_scale == nullptr), GAE_E_NULL,
                "gae_gcn_layer_fused: row_scale and col_scale must both be given or both be NULL");
    GAE_REQUIRE(plan && plan->ell && plan->n_heavy == 0 && plan->vh_n_virtual == 0 &&
                    (plan->ell_width == 4 || plan->ell_width == 8 || plan->ell_width == GAE_SPMM_ELL_WIDTH),
                GAE_E_RANGE, "gae_gcn_layer_fused: needs a plan with a packed neighbour table and no heavy or XCD-pinned rows");
    if (n_rows == 0) return GAE_OK;
    GAE_REQUIRE(indptr && H && W && Y, GAE_E_NULL, "gae_gcn_layer_fused: NULL pointer");
    GAE_REQUIRE(ldh >= F && ldh % 4 == 0 && gae::aligned16(H), GAE_E_ALIGN,
                "gae_gcn_layer_fused: rows of H must be whole 16-byte vectors");
    GAE_REQUIRE(!M || (ldm >= F && ldm % 4 == 0 && gae::aligned16(M)), GAE_E_ALIGN,
                "gae_gcn_layer_fused: rows of M must be whole 16-byte vectors");
    GAE_REQUIRE(ldy >= J, GAE_E_SIZE, "gae_gcn_layer_fused: ldy < J");
    GAE_REQUIRE(n_cols > 0 && n_cols * ldh * 4 + (int64_t(1) << 16) < (int64_t(1) << 32), GAE_E_SIZE,
                "gae_gcn_layer_fused: H larger than a raw buffer resource addresses");
    EllArgs a{};
    a.indptr = indptr; a.indices = indices; a.ell = plan->ell;
    a.H = H; a.M = M; a.row_scale = row_scale; a.col_scale = col_scale;
    a.n_rows = n_rows; a.ldm = M ? ldm : 0;
    a.ldh_bytes = unsigned(ldh * 4);
    a.h_bytes = unsigned(n_cols * ldh * 4);
    a.empty_off = a.h_bytes; a.n_splits = 1;
    a.n_cols = unsigned(n_cols); a.F = unsigned(F); a.nvec = unsigned((F + 3) / 4);
    a.tile_vecs = a.nvec;
    a.xcd_tiled = 0; a.store_pad = 0; a.store_mode = 0;
    a.W = W; a.bias = bias; a.Y = Y; a.ldy = ldy; a.J = int(J); a.w_so = int(w_stride_out); a.w_sk = int(w_stride_in);
    a.act = act; a.store_m = M != nullptr;
    a.W2 = W2; a.bias2 = bias2; a.w_split = int(w_split); a.w_t = w_transposed;
    if (side) {
        a.sw_P = side->P; a.sw_Q = side->Q; a.sw_part = side->part;
        a.sw_ldp = side->ldp; a.sw_ldq = side->ldq; a.sw_stride = side->stride; a.sw_O = side->O; a.sw_I = side->I;
        a.sw_variant = int(gae::g_ell_side);
    }
    if (prep) {
        a.pz_t = prep->lay->Zt; a.pz_hi = prep->lay->Zhi; a.pz_lo = prep->lay->Zlo; a.pz_cs = prep->lay->colsum_partial;
        a.pz_scal = prep->lay->scal; a.pz_all_pairs = prep->lay->all_pairs;
        a.pz_mask = prep->mask; a.pz_ldmask = prep->ldmask; a.pz_drop_p = prep->drop_p;
        a.pz_drop_scale = 1.0f / (1.0f - prep->drop_p);
        a.pz_seed = prep->seed; a.pz_offset = prep->offset; a.pz_draw = prep->draw; a.pz_counts = prep->counts;
    }
    hipStream_t s = gae::as_stream(stream);
    const bool scaled = row_scale != nullptr;
    const int ew = plan->ell_width;
    if (a.nvec <= 8) return J <= 16 ? launch_ell_epi_w<8, 16>(a, ew, scaled, s) : launch_ell_epi_w<8, 32>(a, ew, scaled, s);
    return J <= 16 ? launch_ell_epi_w<16, 16>(a, ew, scaled, s) : launch_ell_epi_w<16, 32>(a, ew, scaled, s);
}

// gae_gcn_layer_fused on the LAST encoder layer of a training step, with the prepare step of the fused loss in its
// epilogue (see gae_x_decoder_bce_prep_layout in include/gae_hip.h).
extern "C" int gae_x_gcn_layer_fused_prep(const int32_t *indptr, const int32_t *indices, int64_t n, const float *H,
                                        int64_t ldh, float *M, int64_t ldm, int64_t F, const float *row_scale,
                                        const float *col_scale, const gae_spmm_plan *plan, const float *W,
                                        int64_t w_stride_out, int64_t w_stride_in, const float *bias, int64_t J,
                                        float *Z, int64_t ldz, const gae_bce_prep *prep, float *mask, int64_t ldmask,
                                        float dropout_p, uint64_t seed, uint64_t offset, const uint64_t *draw_dev,
                                        const int64_t *counts_dev, int64_t *n_prep_blocks_out, void *stream)
{
    GAE_REQUIRE(prep && n_prep_blocks_out, GAE_E_NULL, "gae_x_gcn_layer_fused_prep: NULL pointer");
    GAE_REQUIRE(J >= 1 && J <= 16 && prep->DP == 16, GAE_E_RANGE,
                "gae_x_gcn_layer_fused_prep: the embedding must be at most 16 wide (got %lld)", (long long)J);
    GAE_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, GAE_E_RANGE, "gae_x_gcn_layer_fused_prep: dropout_p outside [0, 1)");
    GAE_REQUIRE(dropout_p == 0.f || mask, GAE_E_NULL, "gae_x_gcn_layer_fused_prep: dropout_p > 0 needs the mask output buffer");
    GAE_REQUIRE(!mask || ldmask >= J, GAE_E_SIZE, "gae_x_gcn_layer_fused_prep: ldmask < J");
    GAE_REQUIRE(n >= 1, GAE_E_SIZE, "gae_x_gcn_layer_fused_prep: empty graph");
    const int64_t rpb = (F + 3) / 4 <= 8 ? 32 : 16, blocks = (n + rpb - 1) / rpb;
    GAE_REQUIRE(blocks <= prep->max_blocks && prep->Zt && prep->Zhi && prep->Zlo && prep->colsum_partial, GAE_E_WORKSPACE,
                "gae_x_gcn_layer_fused_prep: layout not from gae_x_decoder_bce_prep_layout for this n");
    *n_prep_blocks_out = blocks;
    FusedPrep fp{prep, mask, ldmask, dropout_p, seed, offset, draw_dev, counts_dev};
    return gcn_layer_fused_impl(indptr, indices, n, n, H, ldh, M, ldm, F, row_scale, col_scale, plan, W, w_stride_out,
                                w_stride_in, bias, J, GAE_ACT_IDENTITY, Z, ldz, nullptr, nullptr, 0, 0, stream, nullptr, &fp);
}

// The identity-activation backward of gae_gcn_layer_fused in ONE launch: dH = (A^T dY) W (the fused kernel on the CSR
// of A^T, W addressed transposed) and -- side work of the same blocks on their own rows -- the partial sums of
// dW = dY^T M and db = colsum(dY) (M = the aggregate the forward stored).  rows per block: 32 (F <= 32) or 16.
static int64_t fused_wgrad_blocks(int64_t n_rows, int64_t F) { const int64_t rpb = (F + 3) / 4 <= 8 ? 32 : 16; return (n_rows + rpb - 1) / rpb; }

extern "C" int64_t gae_x_gcn_layer_fused_wgrad_workspace_bytes(int64_t n_rows, int64_t F, int64_t I)
{
    if (n_rows < 0 || F < 1 || F > 32 || I < 1 || I > 64) return GAE_E_SIZE;
    return fused_wgrad_blocks(n_rows, F) * (F * I + F) * 4 + 256;
}

static int fused_wgrad_impl(const int32_t *t_indptr, const int32_t *t_indices, int64_t n, const float *dY, int64_t lddy,
                            int64_t F, const float *row_scale, const float *col_scale, const gae_spmm_plan *plan_t,
                            const float *W, const float *W2, int64_t w_split, int64_t ldw, int64_t I, float *dH,
                            int64_t lddh, const float *M, int64_t ldm, float *dW, float *db, void *workspace,
                            int64_t workspace_bytes, int64_t *layout_out, void *stream)
{
    GAE_REQUIRE(F >= 1 && F <= 32 && I >= 1 && I <= 32, GAE_E_RANGE,
                "gae_x_gcn_layer_fused_wgrad: needs 1 <= f_out <= 32 and 1 <= f_in <= 32 (got %lld, %lld)", (long long)F,
                (long long)I);
    const int64_t need = gae_x_gcn_layer_fused_wgrad_workspace_bytes(n, F, I);
    GAE_REQUIRE(need >= 0, GAE_E_SIZE, "gae_x_gcn_layer_fused_wgrad: negative size");
    const int64_t blocks = fused_wgrad_blocks(n, F), stride = F * I + F;
    if (layout_out) { layout_out[0] = blocks; layout_out[1] = stride; layout_out[2] = F * I; }
    if (n == 0) return GAE_OK;
    GAE_REQUIRE(M && workspace && workspace_bytes >= need && gae::aligned16(workspace), GAE_E_WORKSPACE,
                "gae_x_gcn_layer_fused_wgrad: M / workspace missing or smaller than %lld bytes", (long long)need);
    GAE_REQUIRE(ldm >= I && ldw >= I && lddy >= F, GAE_E_SIZE, "gae_x_gcn_layer_fused_wgrad: leading dimension too small");
    GAE_REQUIRE(!W2 || (w_split >= 1 && w_split < F), GAE_E_RANGE, "gae_x_gcn_layer_fused2_wgrad: w_split outside (0, f_out)");
    FusedSide side{dY, M, static_cast<float *>(workspace), lddy, ldm, stride, int(F), int(I)};
    // W [F][I] as nn.Linear stores it is the transposed weight of this launch: output j <- W[k][j]
    int rc = gcn_layer_fused_impl(t_indptr, t_indices, n, n, dY, lddy, nullptr, 0, F, row_scale, col_scale, plan_t, W, 1,
                                  ldw, nullptr, I, GAE_ACT_IDENTITY, dH, lddh, W2, nullptr, W2 ? w_split : 0, 1, stream,
                                  &side);
    if (rc || (!dW && !db)) return rc;
    gae::PartialList la{}, lb{};
    const float *part = static_cast<const float *>(workspace);
    if (dW) la = gae::PartialList{part, dW, F * I, blocks, stride, F * I, F * I, F * I};
    if (db) lb = gae::PartialList{part + F * I, db, F, blocks, stride, F, F, F};
    return gae::launch_partials_reduce(la, lb, gae::as_stream(stream));
}

extern "C" int gae_x_gcn_layer_fused_wgrad(const int32_t *t_indptr, const int32_t *t_indices, int64_t n, const float *dY,
                                         int64_t lddy, int64_t F, const float *row_scale, const float *col_scale,
                                         const gae_spmm_plan *plan_t, const float *W, int64_t ldw, int64_t I, float *dH,
                                         int64_t lddh, const float *M, int64_t ldm, float *dW, float *db,
                                         void *workspace, int64_t workspace_bytes, int64_t *layout_out, void *stream)
{
    return fused_wgrad_impl(t_indptr, t_indices, n, dY, lddy, F, row_scale, col_scale, plan_t, W, nullptr, 0, ldw, I, dH,
                            lddh, M, ldm, dW, db, workspace, workspace_bytes, layout_out, stream);
}

// ... of gae_x_gcn_layer_fused2 (two identity heads on one aggregate): dY = [dY1 | dY2] ([n, f_out], f_out = d1 + d2),
// the weight is the stack [W; W2] along its stored rows (w_split = d1 rows in W, same ldw), dW [f_out, f_in] stacked
// alike (rows < w_split = dW1), db [f_out].
extern "C" int gae_x_gcn_layer_fused2_wgrad(const int32_t *t_indptr, const int32_t *t_indices, int64_t n, const float *dY,
                                          int64_t lddy, int64_t F, const float *row_scale, const float *col_scale,
                                          const gae_spmm_plan *plan_t, const float *W, const float *W2, int64_t w_split,
                                          int64_t ldw, int64_t I, float *dH, int64_t lddh, const float *M, int64_t ldm,
                                          float *dW, float *db, void *workspace, int64_t workspace_bytes,
                                          int64_t *layout_out, void *stream)
{
    GAE_REQUIRE(W2 != nullptr, GAE_E_NULL, "gae_x_gcn_layer_fused2_wgrad: W2 is NULL");
    return fused_wgrad_impl(t_indptr, t_indices, n, dY, lddy, F, row_scale, col_scale, plan_t, W, W2, w_split, ldw, I, dH,
                            lddh, M, ldm, dW, db, workspace, workspace_bytes, layout_out, stream);
}

namespace {
template <int LPR, int W, int NB, int MODE>
int launch_ell_mode(const EllArgs &a, bool scaled, hipStream_t s)
{
    constexpr int RPB = 256 / LPR;
    EllArgs b = a;
    b.n_row_blocks = unsigned((a.n_rows + RPB - 1) / RPB);
    b.n_ftiles = 1;
    const dim3 grid(b.n_row_blocks, 1);
    if (scaled) hipLaunchKernelGGL((spmm_ell_kernel<float, LPR, 1, W, NB, true, 0, MODE>), grid, dim3(256), 0, s, b);
    else hipLaunchKernelGGL((spmm_ell_kernel<float, LPR, 1, W, NB, false, 0, MODE>), grid, dim3(256), 0, s, b);
    GAE_CHECK_LAUNCH("spmm_ell_kernel (gated / split-sum gather)");
    return GAE_OK;
}
template <int LPR, int MODE>
int launch_ell_mode_w(const EllArgs &a, int W, bool scaled, hipStream_t s)
{
    if (W == 4) return launch_ell_mode<LPR, 4, 4, MODE>(a, scaled, s);
    if (W == 8) return launch_ell_mode<LPR, 8, 8, MODE>(a, scaled, s);
    return launch_ell_mode<LPR, 16, 8, MODE>(a, scaled, s);
}
} // namespace

// The sparse half of a TRANSFORM-FIRST GCN layer, act(A (H W^T) + b) instead of act((A H) W^T + b) (gae_dgl/gae.py:
// 26-31; same value up to fp32 rounding -- the aggregation then runs at the OUTPUT width and nothing of the input
// width is written):
//   forward   Y  = act(diag(rs) A diag(cs) P + b)               P = X W^T from gae_xw_fwd, Hmask = NULL
//   backward  G  = diag(rs) A^T diag(cs) (dY (.) [Y > 0])        Hmask = Y (the layer's output), bias = NULL, identity
// in one launch of the packed-table kernel (bias + activation at store time; the ReLU gate applied to the gathered
// rows).  Needs a plan with a packed neighbour table and no heavy / XCD-pinned rows; F <= 64, fp32, rows of whole
// 16-byte vectors; Hmask has the layout of H.  Sums run in CSR order (bit-identical to gae_spmm_csr on the same rows).
extern "C" int gae_spmm_csr_epilogue(const int32_t *indptr, const int32_t *indices, int64_t n_rows, int64_t n_cols,
                                     const float *H, int64_t ldh, const float *Hmask, float *Y, int64_t ldy, int64_t F,
                                     const float *row_scale, const float *col_scale, const gae_spmm_plan *plan,
                                     const float *bias, int act, int64_t n_splits, int64_t split_stride, void *stream)
{
    GAE_REQUIRE(n_splits >= 1 && (n_splits == 1 || (split_stride >= n_cols * ldh && Hmask == nullptr)), GAE_E_RANGE,
                "gae_spmm_csr_epilogue: n_splits >= 1; split partials need split_stride >= n_cols * ldh and no Hmask");
    GAE_REQUIRE(n_splits == 1 || n_splits * split_stride * 4 < (int64_t(1) << 27), GAE_E_SIZE,
                "gae_spmm_csr_epilogue: split partials larger than 128 MiB");
    GAE_REQUIRE(n_rows >= 0 && n_cols >= 0, GAE_E_SIZE, "gae_spmm_csr_epilogue: negative size");
    GAE_REQUIRE(F >= 1 && F <= 64, GAE_E_RANGE, "gae_spmm_csr_epilogue: needs 1 <= F <= 64 (got %lld)", (long long)F);
    GAE_REQUIRE(act == GAE_ACT_IDENTITY || act == GAE_ACT_RELU, GAE_E_RANGE, "gae_spmm_csr_epilogue: act %d", act);
    GAE_REQUIRE((row_scale == nullptr) == (col_scale == nullptr), GAE_E_NULL,
                "gae_spmm_csr_epilogue: row_scale and col_scale must both be given or both be NULL");
    GAE_REQUIRE(plan && plan->ell && plan->n_heavy == 0 && plan->vh_n_virtual == 0 &&
                    (plan->ell_width == 4 || plan->ell_width == 8 || plan->ell_width == GAE_SPMM_ELL_WIDTH),
                GAE_E_RANGE, "gae_spmm_csr_epilogue: needs a plan with a packed neighbour table and no heavy or XCD-pinned rows");
    if (n_rows == 0) return GAE_OK;
    GAE_REQUIRE(indptr && H && Y, GAE_E_NULL, "gae_spmm_csr_epilogue: NULL pointer");
    GAE_REQUIRE(ldh >= F && ldh % 4 == 0 && gae::aligned16(H) && (!Hmask || gae::aligned16(Hmask)), GAE_E_ALIGN,
                "gae_spmm_csr_epilogue: rows of H (and Hmask) must be whole 16-byte vectors");
    GAE_REQUIRE(ldy >= (F + 3) / 4 * 4 && ldy % 4 == 0 && gae::aligned16(Y), GAE_E_ALIGN,
                "gae_spmm_csr_epilogue: rows of Y must be whole 16-byte vectors");
    GAE_REQUIRE(n_cols > 0 && n_cols * ldh * 4 + (int64_t(1) << 16) < (int64_t(1) << 32), GAE_E_SIZE,
                "gae_spmm_csr_epilogue: H larger than a raw buffer resource addresses");
    EllArgs a{};
    a.indptr = indptr; a.indices = indices; a.ell = plan->ell;
    a.H = H; a.M = Y; a.row_scale = row_scale; a.col_scale = col_scale;
    a.n_rows = n_rows; a.ldm = ldy;
    a.ldh_bytes = unsigned(ldh * 4);
    a.h_bytes = unsigned(n_cols * ldh * 4);
    a.n_cols = unsigned(n_cols); a.F = unsigned(F); a.nvec = unsigned((F + 3) / 4);
    a.tile_vecs = a.nvec;
    a.xcd_tiled = 0; a.store_pad = 1; a.store_mode = 0;     // (rows of Y are whole vectors: checked above)
    a.empty_off = a.h_bytes; a.n_splits = 1;
    a.ep_bias = bias; a.ep_act = act; a.Hmask = Hmask;
    hipStream_t s = gae::as_stream(stream);
    const bool scaled = row_scale != nullptr;
    const int ew = plan->ell_width;
    if (n_splits > 1) {
        a.n_splits = int(n_splits); a.split_bytes = unsigned(split_stride * 4);
        a.h_bytes = unsigned(((n_splits - 1) * split_stride + n_cols * ldh) * 4);     // the buffer covers every partial
        a.empty_off = 0xF0000000u;          // + (n_splits - 1) split_bytes stays behind the buffer (< 2^27 checked above)
        return a.nvec <= 8 ? launch_ell_mode_w<8, 2>(a, ew, scaled, s) : launch_ell_mode_w<16, 2>(a, ew, scaled, s);
    }
    if (Hmask != nullptr)
        return a.nvec <= 8 ? launch_ell_mode_w<8, 1>(a, ew, scaled, s) : launch_ell_mode_w<16, 1>(a, ew, scaled, s);
    return a.nvec <= 8 ? launch_ell_w<float, 8, 1>(a, ew, scaled, s) : launch_ell_w<float, 16, 1>(a, ew, scaled, s);
}
