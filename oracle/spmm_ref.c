/* CPU restatement (plain C) of the sparse aggregation on the hot path of
 * shionhonda/gae-dgl.  TEST INFRASTRUCTURE ONLY: built by
 * __graft_entry__.build() into oracle/libgae_oracle.so, called only from
 * tests/, smoke() and bench.py's cpu_baseline leg.
 *
 * The reference executes this op inside the third-party package `dgl`
 * (gae_dgl/gae.py:18-19,28: update_all(copy_src('h','m'), sum('m','h'))),
 * which is not vendored and not version-pinned (README.md:10-16).  DGL's CPU
 * backend runs a row-parallel CSR traversal (OpenMP over destination rows,
 * sequential sum over a row's in-edges); that published algorithm is what is
 * written here.  Backward (autograd of gae.py:28 via train_inductive.py:51)
 * is the same traversal on the CSR of A^T.
 *
 * Parity pin: checked in tests/test_oracle_golden.py against the vectors the
 * reference's own gae.py produced (tests/golden/), through the Python oracle.
 */
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* M[v,:] = rs[v] * sum_{e in row v} cs[idx[e]] * H[idx[e],:]   (rs/cs may be NULL) */
void oracle_spmm_csr_f32(int64_t n_rows, const int32_t *indptr, const int32_t *indices,
                         const float *H, int64_t ldh, float *M, int64_t ldm, int64_t F,
                         const float *row_scale, const float *col_scale)
{
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t v = 0; v < n_rows; ++v) {
        float *out = M + v * ldm;
        memset(out, 0, (size_t)F * sizeof(float));
        for (int32_t e = indptr[v]; e < indptr[v + 1]; ++e) {
            const int32_t u = indices[e];
            const float *h = H + (int64_t)u * ldh;
            if (col_scale) {
                const float c = col_scale[u];
                for (int64_t f = 0; f < F; ++f) out[f] += c * h[f];
            } else {
                for (int64_t f = 0; f < F; ++f) out[f] += h[f];
            }
        }
        if (row_scale) {
            const float r = row_scale[v];
            for (int64_t f = 0; f < F; ++f) out[f] *= r;
        }
    }
}

/* Same traversal with a double accumulator and double output: the reference for rows so long (RMAT hubs: 10^5
 * in-edges) that the order of an fp32 sum matters more than the 1e-5 tolerance. */
void oracle_spmm_csr_f32_acc64(int64_t n_rows, const int32_t *indptr, const int32_t *indices,
                               const float *H, int64_t ldh, double *M, int64_t ldm, int64_t F)
{
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t v = 0; v < n_rows; ++v) {
        double *out = M + v * ldm;
        for (int64_t f = 0; f < F; ++f) out[f] = 0.0;
        for (int32_t e = indptr[v]; e < indptr[v + 1]; ++e) {
            const float *h = H + (int64_t)indices[e] * ldh;
            for (int64_t f = 0; f < F; ++f) out[f] += (double)h[f];
        }
    }
}

/* Y = act(M W^T + b), W [F_out, F_in] row-major (nn.Linear, gae.py:10,14-15) */
void oracle_linear_f32(int64_t n, const float *M, int64_t F_in, const float *W, const float *b,
                       int64_t F_out, int relu, float *Y)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        for (int64_t o = 0; o < F_out; ++o) {
            float acc = 0.f;
            for (int64_t k = 0; k < F_in; ++k) acc += M[i * F_in + k] * W[o * F_in + k];
            acc += b[o];
            Y[i * F_out + o] = (relu && acc < 0.f) ? 0.f : acc;
        }
    }
}

void oracle_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
