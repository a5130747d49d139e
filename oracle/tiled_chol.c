/* TEST INFRASTRUCTURE ONLY -- never linked, imported or executed by the product path.
 *
 * A HARDER CPU denominator for bench.py's cpu_baseline leg (round 5): the reference's `blas` feature factors with one
 * multithreaded LAPACK dpotrf (crates/gp/src/algorithm.rs:1077), and OpenBLAS' threaded dpotrf stops scaling at ~16 of
 * the GPU box's 256 hardware threads (0.12 of the host's FP64 peak).  This is the standard fix: a tiled right-looking
 * Cholesky whose tile kernels are SINGLE-threaded BLAS / LAPACK calls (dpotrf, dtrsm, dsyrk, dgemm on nb x nb tiles)
 * scheduled as OpenMP tasks with data dependences -- what PLASMA / a task-parallel runtime would do for the same call.
 * The BLAS is the one scipy already loaded (its path is passed in and dlopen'ed: no link-time dependency).
 *
 * Build: make -C oracle  ->  oracle/lib/libtiled_chol.so
 */
#include <dlfcn.h>
#include <stdint.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef void (*dpotrf_t)(const char *, const int *, double *, const int *, int *);
typedef void (*dtrsm_t)(const char *, const char *, const char *, const char *, const int *, const int *, const double *,
                        const double *, const int *, double *, const int *);
typedef void (*dsyrk_t)(const char *, const char *, const int *, const int *, const double *, const double *, const int *,
                        const double *, double *, const int *);
typedef void (*dgemm_t)(const char *, const char *, const int *, const int *, const int *, const double *, const double *,
                        const int *, const double *, const int *, const double *, double *, const int *);
typedef void (*setthr_t)(int);

static void *sym2(void *h, const char *a, const char *b) {
    void *p = dlsym(h, a);
    return p ? p : dlsym(h, b);
}

/* a: n x n, column-major, lower triangle referenced, factored in place (A = L L^T).  Returns LAPACK's info (0, or the
 * 1-based index of the first non-positive pivot), -1 when the BLAS could not be loaded, -2 for bad arguments.
 * (The tiles of a symmetric matrix's row-major buffer are the tiles of its column-major reading: callers with a C-order
 * symmetric matrix pass it as it is.) */
int tiled_potrf(const char *blas_path, double *a, int64_t n64, int64_t nb64, int threads) {
    if (!blas_path || !a || n64 < 1 || nb64 < 8 || n64 > 2147483647) return -2;
    void *h = dlopen(blas_path, RTLD_NOW | RTLD_GLOBAL);
    if (!h) return -1;
    dpotrf_t potrf = (dpotrf_t)sym2(h, "scipy_dpotrf_", "dpotrf_");
    dtrsm_t trsm = (dtrsm_t)sym2(h, "scipy_dtrsm_", "dtrsm_");
    dsyrk_t syrk = (dsyrk_t)sym2(h, "scipy_dsyrk_", "dsyrk_");
    dgemm_t gemm = (dgemm_t)sym2(h, "scipy_dgemm_", "dgemm_");
    setthr_t setthr = (setthr_t)sym2(h, "scipy_openblas_set_num_threads", "openblas_set_num_threads");
    if (!potrf || !trsm || !syrk || !gemm) return -1;
    if (setthr) setthr(1); /* every tile kernel runs in the thread of the task that calls it */
    const int n = (int)n64, nb = (int)nb64, nt = (n + nb - 1) / nb;
    char *dep = (char *)calloc((size_t)nt * nt, 1); /* dependence proxies, one per tile */
    int first_info = 0;
    const double one = 1.0, mone = -1.0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#define TILE(i, j) (a + (size_t)(j) * nb * n + (size_t)(i) * nb)
#define ROWS(i) ((i) == nt - 1 ? n - (i) * nb : nb)
#pragma omp parallel
#pragma omp single
    {
        for (int k = 0; k < nt; k++) {
            const int kb = ROWS(k);
#pragma omp task depend(inout : dep[k * nt + k]) firstprivate(k, kb) shared(first_info)
            {
                int info = 0;
                potrf("L", &kb, TILE(k, k), &n, &info);
                if (info != 0) {
#pragma omp critical
                    if (first_info == 0 || k * nb + info < first_info) first_info = k * nb + info;
                }
            }
            for (int i = k + 1; i < nt; i++) {
                const int ib = ROWS(i);
#pragma omp task depend(in : dep[k * nt + k]) depend(inout : dep[i * nt + k]) firstprivate(i, k, ib, kb)
                trsm("R", "L", "T", "N", &ib, &kb, &one, TILE(k, k), &n, TILE(i, k), &n);
            }
            for (int i = k + 1; i < nt; i++) {
                const int ib = ROWS(i);
#pragma omp task depend(in : dep[i * nt + k]) depend(inout : dep[i * nt + i]) firstprivate(i, k, ib, kb)
                syrk("L", "N", &ib, &kb, &mone, TILE(i, k), &n, &one, TILE(i, i), &n);
                for (int j = k + 1; j < i; j++) {
                    const int jb = ROWS(j);
#pragma omp task depend(in : dep[i * nt + k], dep[j * nt + k]) depend(inout : dep[i * nt + j]) firstprivate(i, j, k, ib, jb, kb)
                    gemm("N", "T", &ib, &jb, &kb, &mone, TILE(i, k), &n, TILE(j, k), &n, &one, TILE(i, j), &n);
                }
            }
        }
    }
#undef TILE
#undef ROWS
    free(dep);
    return first_info;
}
