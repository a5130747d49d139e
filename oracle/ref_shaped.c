/* TEST INFRASTRUCTURE ONLY -- never linked, imported or executed by the product path.
 *
 * "Reference-shaped" CPU restatement of egobox-gp's fixed-theta likelihood in the shape of the reference's DEFAULT
 * build (no `blas` feature): single thread, the (n(n-1)/2, d) difference table materialised, the kernel evaluated
 * over the table, R filled by scatter, an UNBLOCKED Cholesky and scalar triangular solves.  It exists for two
 * things (SURVEY 8d "CPU baseline (i)"):
 *   1. a second, independent checker of oracle/gp_oracle.py (different language, no LAPACK) -- tests/test_oracle_golden.py;
 *   2. the single-thread baseline `bench.py` reports beside the LAPACK-backed one.
 *
 * Follows (reference checkout paths):
 *   normalize                      crates/gp/src/utils.rs:45-54        (sample std, ddof = 1, zero std -> 1)
 *   DiffMatrix::_cross_diff        crates/gp/src/utils.rs:80-104       (pairs (k,i), k < i, k-major; |x_k - x_i|)
 *   SquaredExponentialCorr::value  crates/gp/src/correlation_models.rs:91-104
 *   AbsoluteExponentialCorr::value crates/gp/src/correlation_models.rs:185-196
 *   Matern32Corr / Matern52Corr    crates/gp/src/correlation_models.rs:326-353, 497-523   (w = identity)
 *   reduced_likelihood             crates/gp/src/algorithm.rs:988-1056 (constant mean: F = ones, p = 1)
 * The factorisation lives in the un-vendored crate linfa-linalg 0.2.1 (`cholesky()`, `solve_triangular`), absent
 * from the reference checkout: restated here as the textbook row-oriented (Cholesky-Banachiewicz) algorithm with
 * a not-positive-definite error on a negative pivot, and plain forward / backward substitution.
 *
 * Build: make -C oracle  ->  oracle/lib/libref_shaped.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { REF_OK = 0, REF_NOT_POSITIVE_DEFINITE = 1, REF_NOMEM = -1, REF_BAD_ARG = -2 };
enum { CORR_SQEXP = 0, CORR_ABSEXP = 1, CORR_MATERN32 = 2, CORR_MATERN52 = 3 };

/* utils.rs:45-54 */
static void normalize_cols(const double *a, int64_t n, int64_t d, double *out, double *mean, double *std) {
    for (int64_t j = 0; j < d; j++) {
        double s = 0.0;
        for (int64_t i = 0; i < n; i++) s += a[i * d + j];
        const double mu = s / (double)n;
        double v = 0.0;
        for (int64_t i = 0; i < n; i++) {
            const double c = a[i * d + j] - mu;
            v += c * c;
        }
        double sd = (n > 1) ? sqrt(v / (double)(n - 1)) : 0.0;
        if (sd == 0.0) sd = 1.0;
        mean[j] = mu;
        std[j] = sd;
        for (int64_t i = 0; i < n; i++) out[i * d + j] = (a[i * d + j] - mu) / sd;
    }
}

/* correlation_models.rs: one row of the difference table -> r */
static double corr_row(int corr, const double *dij, const double *theta, int64_t d) {
    switch (corr) {
    case CORR_SQEXP: {
        double s = 0.0;
        for (int64_t l = 0; l < d; l++) s += theta[l] * theta[l] * dij[l] * dij[l];
        return exp(-0.5 * s);
    }
    case CORR_ABSEXP: {
        double s = 0.0;
        for (int64_t l = 0; l < d; l++) s += theta[l] * dij[l];
        return exp(-s);
    }
    case CORR_MATERN32: {
        const double q = sqrt(3.0);
        double a = 1.0, s = 0.0;
        for (int64_t l = 0; l < d; l++) {
            const double t = theta[l] * dij[l];
            a *= 1.0 + q * t;
            s += t;
        }
        return a * exp(-q * s);
    }
    default: {
        const double q = sqrt(5.0);
        double a = 1.0, s = 0.0;
        for (int64_t l = 0; l < d; l++) {
            const double t = theta[l] * dij[l];
            a *= 1.0 + q * t + (5.0 / 3.0) * t * t;
            s += t;
        }
        return a * exp(-q * s);
    }
    }
}

/* Fixed-theta likelihood of a constant-mean GP, the reference's default-build dataflow.
 * x (n,d) row-major, y (n), theta (d).  out[0] = reduced likelihood, out[1] = sigma2 (original y units),
 * out[2] = beta, out[3] = smallest Cholesky pivot; r_chol_out (n*n, optional) receives the lower factor,
 * gamma_out (n, optional) gamma.  Returns REF_OK, REF_NOT_POSITIVE_DEFINITE, or a negative error. */
int ref_shaped_likelihood(const double *x, const double *y, int64_t n, int64_t d, const double *theta, int corr,
                          double nugget, double *out, double *r_chol_out, double *gamma_out) {
    if (!x || !y || !theta || !out || n < 2 || d < 1 || corr < 0 || corr > 3) return REF_BAD_ARG;
    const int64_t pairs = n * (n - 1) / 2;
    double *xn = (double *)malloc(sizeof(double) * (size_t)(n * d));
    double *yn = (double *)malloc(sizeof(double) * (size_t)n);
    double *xm = (double *)malloc(sizeof(double) * (size_t)(2 * d));
    double *D = (double *)malloc(sizeof(double) * (size_t)(pairs * d));       /* the (pairs, d) table */
    int64_t *idx = (int64_t *)malloc(sizeof(int64_t) * (size_t)(pairs * 2));  /* its (pairs, 2) index table */
    double *r = (double *)malloc(sizeof(double) * (size_t)pairs);
    double *R = (double *)malloc(sizeof(double) * (size_t)(n * n));
    double *ft = (double *)malloc(sizeof(double) * (size_t)n);
    double *yt = (double *)malloc(sizeof(double) * (size_t)n);
    int rc = REF_OK;
    if (!xn || !yn || !xm || !D || !idx || !r || !R || !ft || !yt) {
        rc = REF_NOMEM;
        goto done;
    }
    double ymean, ystd;
    normalize_cols(x, n, d, xn, xm, xm + d);
    normalize_cols(y, n, 1, yn, &ymean, &ystd);

    /* DiffMatrix: utils.rs:80-104 */
    {
        int64_t row = 0;
        for (int64_t k = 0; k < n - 1; k++)
            for (int64_t i = k + 1; i < n; i++, row++) {
                idx[2 * row] = k;
                idx[2 * row + 1] = i;
                for (int64_t l = 0; l < d; l++) D[row * d + l] = fabs(xn[k * d + l] - xn[i * d + l]);
            }
    }
    /* value over the table */
    for (int64_t row = 0; row < pairs; row++) r[row] = corr_row(corr, D + row * d, theta, d);
    /* R = I (1 + nugget), scatter into both triangles: algorithm.rs:997-1001 */
    memset(R, 0, sizeof(double) * (size_t)(n * n));
    for (int64_t i = 0; i < n; i++) R[i * n + i] = 1.0 + nugget;
    for (int64_t row = 0; row < pairs; row++) {
        R[idx[2 * row] * n + idx[2 * row + 1]] = r[row];
        R[idx[2 * row + 1] * n + idx[2 * row]] = r[row];
    }
    /* unblocked row-oriented Cholesky, lower factor in place */
    double min_pivot = INFINITY;
    for (int64_t j = 0; j < n; j++) {
        double dsum = 0.0;
        for (int64_t k = 0; k < j; k++) {
            double s = 0.0;
            for (int64_t i = 0; i < k; i++) s += R[k * n + i] * R[j * n + i];
            s = (R[j * n + k] - s) / R[k * n + k];
            R[j * n + k] = s;
            dsum += s * s;
        }
        const double piv = R[j * n + j] - dsum;
        if (!(piv > 0.0)) {
            rc = REF_NOT_POSITIVE_DEFINITE;
            goto done;
        }
        R[j * n + j] = sqrt(piv);
        if (R[j * n + j] < min_pivot) min_pivot = R[j * n + j];
        for (int64_t k = j + 1; k < n; k++) R[j * n + k] = 0.0;
    }
    /* ft = C^-1 1, yt = C^-1 y  (forward substitution) */
    for (int64_t i = 0; i < n; i++) {
        double sf = 1.0, sy = yn[i];
        for (int64_t k = 0; k < i; k++) {
            sf -= R[i * n + k] * ft[k];
            sy -= R[i * n + k] * yt[k];
        }
        ft[i] = sf / R[i * n + i];
        yt[i] = sy / R[i * n + i];
    }
    /* p = 1: QR(ft) = (ft / |ft|, |ft|); beta = Rq^-1 Q^T yt; the cond check of a 1x1 factor never fires */
    double ff = 0.0, fy = 0.0;
    for (int64_t i = 0; i < n; i++) {
        ff += ft[i] * ft[i];
        fy += ft[i] * yt[i];
    }
    const double beta = fy / ff;
    double rho2 = 0.0, logdet = 0.0;
    for (int64_t i = 0; i < n; i++) {
        yt[i] -= ft[i] * beta; /* rho */
        rho2 += yt[i] * yt[i];
        logdet += log10(R[i * n + i]);
    }
    logdet *= 2.0 / (double)n;
    const double sigma2 = rho2 / (double)n;
    out[0] = -(double)n * (log10(sigma2) + logdet); /* algorithm.rs:1039-1043 */
    out[1] = sigma2 * ystd * ystd;
    out[2] = beta;
    out[3] = min_pivot;
    if (gamma_out) { /* gamma = C^-T rho (backward substitution) */
        for (int64_t i = n - 1; i >= 0; i--) {
            double s = yt[i];
            for (int64_t k = i + 1; k < n; k++) s -= R[k * n + i] * gamma_out[k];
            gamma_out[i] = s / R[i * n + i];
        }
    }
    if (r_chol_out) memcpy(r_chol_out, R, sizeof(double) * (size_t)(n * n));
done:
    free(xn); free(yn); free(xm); free(D); free(idx); free(r); free(R); free(ft); free(yt);
    return rc;
}
