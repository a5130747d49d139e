/* TEST INFRASTRUCTURE ONLY -- never linked, imported or executed by the product path.
 *
 * Threaded (OpenMP, all host cores) half of the "blas-feature-shaped" CPU baseline (SURVEY 8d (ii);
 * crates/gp/Cargo.toml:20 `blas` feature = ndarray-linalg / LAPACK): the correlation-matrix build.  With the
 * `blas` feature the reference's O(n^3) part is a multithreaded LAPACK dpotrf (crates/gp/src/algorithm.rs:1077);
 * the baseline takes that from scipy/OpenBLAS (oracle/cpu_baseline.py).  Its O(n^2 d) part -- DiffMatrix
 * (crates/gp/src/utils.rs:80-104), CorrelationModel::value (crates/gp/src/correlation_models.rs:91-104, 185-196,
 * 326-353, 497-523; w = identity) and the scatter into both triangles (crates/gp/src/algorithm.rs:997-1001) -- is
 * restated here fused (no (pairs, d) table) and parallel over rows, i.e. the most favourable CPU form, so that the
 * GPU/CPU ratio bench.py prints is not inflated by a slow host loop.
 *
 * Build: make -C oracle  ->  oracle/lib/libblas_shaped.so
 */
#include <math.h>
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { CORR_SQEXP = 0, CORR_ABSEXP = 1, CORR_MATERN32 = 2, CORR_MATERN52 = 3 };

int blas_shaped_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static double corr_pair(int corr, const double *a, const double *b, const double *theta, int64_t d) {
    switch (corr) {
    case CORR_SQEXP: { /* correlation_models.rs:97-101 */
        double s = 0.0;
        for (int64_t l = 0; l < d; l++) {
            const double t = theta[l] * (a[l] - b[l]);
            s += t * t;
        }
        return exp(-0.5 * s);
    }
    case CORR_ABSEXP: { /* :191-193 */
        double s = 0.0;
        for (int64_t l = 0; l < d; l++) s += theta[l] * fabs(a[l] - b[l]);
        return exp(-s);
    }
    case CORR_MATERN32: { /* :333-349 */
        const double s3 = sqrt(3.0);
        double pa = 1.0, sb = 0.0;
        for (int64_t l = 0; l < d; l++) {
            const double t = theta[l] * fabs(a[l] - b[l]);
            pa *= 1.0 + s3 * t;
            sb += t;
        }
        return pa * exp(-s3 * sb);
    }
    default: { /* Matern-5/2 :505-519 */
        const double s5 = sqrt(5.0);
        double pa = 1.0, sb = 0.0;
        for (int64_t l = 0; l < d; l++) {
            const double t = theta[l] * fabs(a[l] - b[l]);
            pa *= 1.0 + s5 * t + (5.0 / 3.0) * t * t;
            sb += t;
        }
        return pa * exp(-s5 * sb);
    }
    }
}

/* r (n x n, row-major, BOTH triangles, diagonal 1 + nugget) from normalised x (n x d) */
int blas_shaped_corr_matrix(int corr, const double *xn, int64_t n, int64_t d, const double *theta, double nugget,
                            double *r) {
    if (!xn || !theta || !r || n < 1 || d < 1 || corr < 0 || corr > 3) return -2;
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t i = 0; i < n; i++) {
        const double *a = xn + i * d;
        for (int64_t j = 0; j < i; j++) {
            const double v = corr_pair(corr, a, xn + j * d, theta, d);
            r[i * n + j] = v;
            r[j * n + i] = v;
        }
        r[i * n + i] = 1.0 + nugget;
    }
    return 0;
}
