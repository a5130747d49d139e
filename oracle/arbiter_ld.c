/* TEST INFRASTRUCTURE ONLY -- never linked, imported or executed by the product path.
 *
 * Extended-precision ARBITER for the ill-posed parity cases: the fixed-theta reduced likelihood of a constant-mean GP
 * computed END TO END in x87 `long double` (64-bit mantissa, eps = 1.08e-19) -- normalisation, differences, kernel
 * (expl), correlation matrix, Cholesky, solves, GLS, log10 -- from the same double inputs the GPU path and the
 * LAPACK-backed oracle get.  Where cond(R) ~ 1 / nugget two double-precision evaluations of the same likelihood
 * differ by 1e-6 .. 1e-2; this one is ~2000x closer to the exact value than either
 * and says which of them is.  tests/golden/make_arbiter.py runs it once and commits its answers
 * (tests/golden/arbiter.json); the GPU tests then assert |gpu - truth| <= c |lapack - truth|.
 *
 * Follows the same reference lines as oracle/ref_shaped.c (crates/gp/src/utils.rs:45-54, correlation_models.rs:91-104,
 * 185-196, 326-353, 497-523, algorithm.rs:988-1056); the factorisation (un-vendored linfa-linalg / LAPACK in the
 * reference) is the column-oriented Cholesky-Crout form, rows of a column in parallel (OpenMP).
 *
 * Build: make -C oracle  ->  oracle/lib/libarbiter_ld.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef long double ld;
enum { ARB_OK = 0, ARB_NOT_POSITIVE_DEFINITE = 1, ARB_NOMEM = -1, ARB_BAD_ARG = -2 };

static ld corr_ld(int corr, const ld *a, const ld *b, const ld *theta, int64_t d) {
    ld s = 0.0L, prod = 1.0L;
    switch (corr) {
    case 0:
        for (int64_t l = 0; l < d; l++) {
            const ld t = theta[l] * (a[l] - b[l]);
            s += t * t;
        }
        return expl(-0.5L * s);
    case 1:
        for (int64_t l = 0; l < d; l++) s += theta[l] * fabsl(a[l] - b[l]);
        return expl(-s);
    case 2: {
        const ld q = sqrtl(3.0L);
        for (int64_t l = 0; l < d; l++) {
            const ld t = theta[l] * fabsl(a[l] - b[l]);
            prod *= 1.0L + q * t;
            s += t;
        }
        return prod * expl(-q * s);
    }
    default: {
        const ld q = sqrtl(5.0L);
        for (int64_t l = 0; l < d; l++) {
            const ld t = theta[l] * fabsl(a[l] - b[l]);
            prod *= 1.0L + q * t + (5.0L / 3.0L) * t * t;
            s += t;
        }
        return prod * expl(-q * s);
    }
    }
}

/* x (n,d) row-major, y (n), theta (d), all double.  out[0] = likelihood, out[1] = sigma2 (original y units), out[2] = beta,
 * out[3] = smallest pivot (diagonal of C), out[4] = the likelihood's low part (likelihood_ld - (double)likelihood_ld). */
int arbiter_ld_likelihood(const double *x, const double *y, int64_t n, int64_t d, const double *theta, int corr,
                          double nugget, double *out) {
    if (!x || !y || !theta || !out || n < 2 || d < 1 || corr < 0 || corr > 3) return ARB_BAD_ARG;
    ld *xn = (ld *)malloc(sizeof(ld) * (size_t)(n * d));
    ld *yn = (ld *)malloc(sizeof(ld) * (size_t)n);
    ld *th = (ld *)malloc(sizeof(ld) * (size_t)d);
    ld *R = (ld *)malloc(sizeof(ld) * (size_t)(n * n));
    ld *ft = (ld *)malloc(sizeof(ld) * (size_t)n);
    ld *yt = (ld *)malloc(sizeof(ld) * (size_t)n);
    int rc = ARB_OK;
    if (!xn || !yn || !th || !R || !ft || !yt) {
        rc = ARB_NOMEM;
        goto done;
    }
    for (int64_t l = 0; l < d; l++) th[l] = theta[l];
    /* utils.rs:45-54 in long double: the exact mean / sample std of the double inputs */
    for (int64_t j = 0; j < d; j++) {
        ld s = 0.0L, v = 0.0L;
        for (int64_t i = 0; i < n; i++) s += x[i * d + j];
        const ld mu = s / (ld)n;
        for (int64_t i = 0; i < n; i++) v += ((ld)x[i * d + j] - mu) * ((ld)x[i * d + j] - mu);
        ld sd = sqrtl(v / (ld)(n - 1));
        if (sd == 0.0L) sd = 1.0L;
        for (int64_t i = 0; i < n; i++) xn[i * d + j] = ((ld)x[i * d + j] - mu) / sd;
    }
    ld ystd;
    {
        ld s = 0.0L, v = 0.0L;
        for (int64_t i = 0; i < n; i++) s += y[i];
        const ld mu = s / (ld)n;
        for (int64_t i = 0; i < n; i++) v += ((ld)y[i] - mu) * ((ld)y[i] - mu);
        ystd = sqrtl(v / (ld)(n - 1));
        if (ystd == 0.0L) ystd = 1.0L;
        for (int64_t i = 0; i < n; i++) yn[i] = ((ld)y[i] - mu) / ystd;
    }
    /* R (lower triangle) = kernel, diagonal 1 + nugget: algorithm.rs:997-1001 */
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t i = 0; i < n; i++) {
        for (int64_t j = 0; j < i; j++) R[i * n + j] = corr_ld(corr, xn + i * d, xn + j * d, th, d);
        R[i * n + i] = 1.0L + (ld)nugget;
    }
    /* Cholesky-Crout, lower factor in place: column j = (A[:, j] - L[:, :j] L[j, :j]^T) / L[j][j] */
    ld min_pivot = INFINITY;
    for (int64_t j = 0; j < n; j++) {
        ld dsum = 0.0L;
        for (int64_t k = 0; k < j; k++) dsum += R[j * n + k] * R[j * n + k];
        const ld piv = R[j * n + j] - dsum;
        if (!(piv > 0.0L)) {
            rc = ARB_NOT_POSITIVE_DEFINITE;
            goto done;
        }
        const ld ljj = sqrtl(piv);
        R[j * n + j] = ljj;
        if (ljj < min_pivot) min_pivot = ljj;
        const ld *rj = R + j * n;
#pragma omp parallel for schedule(static) if (n - j > 256)
        for (int64_t i = j + 1; i < n; i++) {
            const ld *ri = R + i * n;
            ld s = 0.0L;
            for (int64_t k = 0; k < j; k++) s += ri[k] * rj[k];
            R[i * n + j] = (ri[j] - s) / ljj;
        }
    }
    for (int64_t i = 0; i < n; i++) {
        ld sf = 1.0L, sy = yn[i];
        for (int64_t k = 0; k < i; k++) {
            sf -= R[i * n + k] * ft[k];
            sy -= R[i * n + k] * yt[k];
        }
        ft[i] = sf / R[i * n + i];
        yt[i] = sy / R[i * n + i];
    }
    ld ff = 0.0L, fy = 0.0L;
    for (int64_t i = 0; i < n; i++) {
        ff += ft[i] * ft[i];
        fy += ft[i] * yt[i];
    }
    const ld beta = fy / ff;
    ld rho2 = 0.0L, logdet = 0.0L;
    for (int64_t i = 0; i < n; i++) {
        const ld rho = yt[i] - ft[i] * beta;
        rho2 += rho * rho;
        logdet += log10l(R[i * n + i]);
    }
    logdet *= 2.0L / (ld)n;
    const ld sigma2 = rho2 / (ld)n;
    const ld lkh = -(ld)n * (log10l(sigma2) + logdet); /* algorithm.rs:1039-1043 */
    out[0] = (double)lkh;
    out[1] = (double)(sigma2 * ystd * ystd);
    out[2] = (double)beta;
    out[3] = (double)min_pivot;
    out[4] = (double)(lkh - (ld)(double)lkh);
done:
    free(xn); free(yn); free(th); free(R); free(ft); free(yt);
    return rc;
}
