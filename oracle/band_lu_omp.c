/* CPU BASELINE LEG — TEST / BENCH INFRASTRUCTURE, NOT PRODUCT CODE (only bench.py's cpu_baseline leg and tests/ load it).
 *
 * A multi-core direct solve of one Newton system of the 3D path, so that `cpu_baseline` has a leg that uses the box's cores: the
 * reference's linear solver is MUMPS (3D/MPNP_CO2ER_pore.py:792), a threaded multifrontal LU that is not in this image; SciPy's
 * SuperLU is serial and LAPACK's dgbsv on the scalar band does 3.6e11 flops without getting faster on more cores.  This is the
 * algorithm of the library's own GPU fallback (csrc/gmpnp_band_lu.h) on the host: block-banded LU in the slab order of the
 * vertices (node blocks of NF x NF, half-bandwidth b blocks), right-looking, pivoting inside the diagonal blocks only, the
 * rank-NF update of the (<= b)^2 window behind each pivot shared out over the threads (OpenMP), one round of iterative refinement
 * left to the caller.  Storage: blk[(i * (2b+1) + (j - i + b)) * NF*NF + r * NF + c] = A(i*NF + r, j*NF + c).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NF 9
#define BS (NF * NF)

static int invert_block(const double* a, double* inv)
{ /* Gauss-Jordan with partial pivoting on [a | I] */
    double w[NF][2 * NF];
    for (int r = 0; r < NF; r++)
        for (int c = 0; c < NF; c++) { w[r][c] = a[r * NF + c]; w[r][NF + c] = (r == c) ? 1.0 : 0.0; }
    for (int k = 0; k < NF; k++) {
        int p = k; double best = fabs(w[k][k]);
        for (int r = k + 1; r < NF; r++) if (fabs(w[r][k]) > best) { best = fabs(w[r][k]); p = r; }
        if (!(best > 0.0)) return 1;
        if (p != k) for (int c = 0; c < 2 * NF; c++) { double t = w[k][c]; w[k][c] = w[p][c]; w[p][c] = t; }
        const double ip = 1.0 / w[k][k];
        for (int c = 0; c < 2 * NF; c++) w[k][c] *= ip;
        for (int r = 0; r < NF; r++)
            if (r != k) { const double f = w[r][k]; if (f != 0.0) for (int c = 0; c < 2 * NF; c++) w[r][c] -= f * w[k][c]; }
    }
    for (int r = 0; r < NF; r++) for (int c = 0; c < NF; c++) inv[r * NF + c] = w[r][NF + c];
    return 0;
}

static inline void mm_sub(double* restrict c, const double* restrict a, const double* restrict b)
{ /* c -= a b, NF x NF */
    for (int r = 0; r < NF; r++) {
        double acc[NF];
        for (int j = 0; j < NF; j++) acc[j] = c[r * NF + j];
        for (int k = 0; k < NF; k++) { const double ar = a[r * NF + k]; for (int j = 0; j < NF; j++) acc[j] -= ar * b[k * NF + j]; }
        for (int j = 0; j < NF; j++) c[r * NF + j] = acc[j];
    }
}

int band_lu_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Factor in place and solve: rhs (n*NF) is overwritten with the solution.  Returns 0, or k+1 if diagonal block k is singular. */
int band_lu_solve(int n, int b, double* blk, double* rhs, int threads)
{
    const size_t W = (size_t)(2 * b + 1);
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    int fail = 0;
    for (int k = 0; k < n && !fail; k++) {
        double dinv[BS], tmp[BS], tv[NF];
        double* dk = blk + ((size_t)k * W + b) * BS;
        if (invert_block(dk, dinv)) { fail = k + 1; break; }
        const int hi = (k + b < n - 1) ? k + b : n - 1;
        /* U row: A_kj <- Dinv A_kj ; rhs_k <- Dinv rhs_k */
#pragma omp parallel for schedule(static) private(tmp) if (hi - k > 8)
        for (int j = k + 1; j <= hi; j++) {
            double* akj = blk + ((size_t)k * W + (size_t)(j - k + b)) * BS;
            memcpy(tmp, akj, sizeof tmp);
            for (int r = 0; r < NF; r++)
                for (int c = 0; c < NF; c++) { double s = 0.0; for (int m = 0; m < NF; m++) s += dinv[r * NF + m] * tmp[m * NF + c]; akj[r * NF + c] = s; }
        }
        for (int r = 0; r < NF; r++) { double s = 0.0; for (int m = 0; m < NF; m++) s += dinv[r * NF + m] * rhs[(size_t)k * NF + m]; tv[r] = s; }
        memcpy(rhs + (size_t)k * NF, tv, sizeof tv);
        memcpy(dk, dinv, sizeof dinv); /* keep the inverse: not needed again, but the block is the factor's diagonal */
        /* window update: rows i in (k, hi], columns j in (k, hi] */
        /* (row, quarter of the columns) pairs, so that a box with more threads than window rows still has work for all of them */
        const int m = hi - k;
#pragma omp parallel for collapse(2) schedule(static) if (m > 8)
        for (int i = k + 1; i <= hi; i++)
            for (int q = 0; q < 4; q++) {
                const double* lik = blk + ((size_t)i * W + (size_t)(k - i + b)) * BS;
                const double* urow = blk + ((size_t)k * W + (size_t)b) * BS; /* A_kj at urow + (j-k)*BS */
                double* arow = blk + ((size_t)i * W + (size_t)(k - i + b)) * BS; /* A_ij at arow + (j-k)*BS */
                const int j0 = k + 1 + (int)((long)m * q / 4), j1 = k + 1 + (int)((long)m * (q + 1) / 4);
                for (int j = j0; j < j1; j++) mm_sub(arow + (size_t)(j - k) * BS, lik, urow + (size_t)(j - k) * BS);
                if (q == 0) {
                    double* ri = rhs + (size_t)i * NF;
                    const double* rk = rhs + (size_t)k * NF;
                    for (int r = 0; r < NF; r++) { double s = 0.0; for (int mm = 0; mm < NF; mm++) s += lik[r * NF + mm] * rk[mm]; ri[r] -= s; }
                }
            }
    }
    if (fail) return fail;
    /* back substitution: x_k = y_k - sum_{j > k} U_kj x_j */
    for (int k = n - 1; k >= 0; k--) {
        const int hi = (k + b < n - 1) ? k + b : n - 1;
        double* xk = rhs + (size_t)k * NF;
        for (int j = k + 1; j <= hi; j++) {
            const double* ukj = blk + ((size_t)k * W + (size_t)(j - k + b)) * BS;
            const double* xj = rhs + (size_t)j * NF;
            for (int r = 0; r < NF; r++) { double s = 0.0; for (int m = 0; m < NF; m++) s += ukj[r * NF + m] * xj[m]; xk[r] -= s; }
        }
    }
    return 0;
}
