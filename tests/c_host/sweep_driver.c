/* A compiled (plain C99) host of the multi-GPU boundary at world = 1: draws an RCCL unique id, creates the sweep
 * (egx_sweep_create builds a one-rank RCCL communicator inside libegx_gp_hip.so), runs egx_sweep_likelihood on a
 * handful of candidates -- one of them NaN, one at theta = 1e-3 (R ~ all ones, positive definite by its nugget alone:
 * the CPU oracle, oracle/gp_oracle.py::likelihood_at on these inputs, factors it: status 0, likelihood 2081.17, a sum of
 * logarithms of hundreds of noise-level pivots that is only reproducible to ~1e-2 between factorisations; the explicit-
 * inverse triangular solves of round 1 lost a pivot there) -- and checks the gathered (likelihood, status)
 * pairs against egx_gp_likelihood_batch on a plain handle.  What a Rust `extern "C"` shim behind the rayon multistart
 * (crates/gp/src/algorithm.rs:928-945) would do on every rank.  Exit code 0 = all good. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "egx_gp.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int32_t rc_ = (call);                                                    \
        if (rc_ != EGX_SUCCESS) {                                                \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, egx_last_error());     \
            return 10 + rc_;                                                     \
        }                                                                        \
    } while (0)

enum { N = 700, D = 3, K = 7 };

int main(void) {
    if (egx_device_count() < 1) {
        fprintf(stderr, "no HIP device\n");
        return 2;
    }
    static double x[N * D], y[N];
    unsigned long long s = 88172645463325252ULL; /* xorshift64: deterministic inputs */
    for (int i = 0; i < N * D; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        x[i] = (double)(s >> 11) / 9007199254740992.0;
    }
    for (int i = 0; i < N; i++) y[i] = sin(4.0 * x[i * D]) + x[i * D + 1] * x[i * D + 2];
    double thetas[K * D];
    for (int k = 0; k < K; k++)
        for (int j = 0; j < D; j++) thetas[k * D + j] = 0.4 + 0.35 * k + 0.1 * j;
    thetas[2 * D + 1] = NAN;                              /* status 4 */
    for (int j = 0; j < D; j++) thetas[5 * D + j] = 1e-3; /* R ~ all ones: numerically rank deficient, status 0 (oracle) */

    unsigned char id[EGX_SWEEP_ID_BYTES];
    CHECK(egx_sweep_unique_id(id));
    egx_gp_config cfg;
    egx_gp_config_default(&cfg);
    cfg.corr = EGX_CORR_MATERN52;
    egx_sweep *sw = NULL;
    CHECK(egx_sweep_create(&cfg, x, y, N, D, id, 0, 1, &sw));
    int32_t rank = -1, world = -1, rccl_ranks = -1, ver = 0;
    int64_t ngather = -1;
    CHECK(egx_sweep_info(sw, &rank, &world, &rccl_ranks, &ver, &ngather));
    int ok = rank == 0 && world == 1 && rccl_ranks == 1 && ver > 0 && ngather == 0;
    double lk[K], lk_ref[K];
    int32_t st[K], st_ref[K];
    CHECK(egx_sweep_likelihood(sw, thetas, K, D, lk, st));
    CHECK(egx_sweep_info(sw, NULL, NULL, NULL, NULL, &ngather));
    ok = ok && ngather == 1;

    egx_gp *gp = NULL;
    CHECK(egx_gp_create(&cfg, x, y, N, D, &gp));
    CHECK(egx_gp_likelihood_batch(gp, thetas, K, D, lk_ref, st_ref));
    for (int k = 0; k < K; k++) {
        const int same = st[k] == st_ref[k] && (lk[k] == lk_ref[k] || (isinf(lk[k]) && isinf(lk_ref[k])));
        if (!same) fprintf(stderr, "candidate %d: sweep (%g, %d) vs batch (%g, %d)\n", k, lk[k], st[k], lk_ref[k], st_ref[k]);
        ok = ok && same;
    }
    ok = ok && st[2] == EGX_STATUS_NAN_THETA && st[0] == EGX_STATUS_OK;
    if (!(st[5] == EGX_STATUS_OK && fabs(lk[5] / 2081.1671241723093 - 1.0) < 1e-2)) {
        fprintf(stderr, "candidate 5 (theta = 1e-3): (%.13g, %d), oracle (2081.1671241723093, 0)\n", lk[5], st[5]);
        ok = 0;
    }
    /* the winning candidate is finalized on the sweep's own handle */
    int best = -1;
    for (int k = 0; k < K; k++)
        if (st[k] == EGX_STATUS_OK && (best < 0 || lk[k] > lk[best])) best = k;
    CHECK(egx_gp_finalize(egx_sweep_handle(sw), thetas + best * D, D));
    double pred = 0.0;
    CHECK(egx_gp_predict(egx_sweep_handle(sw), x, 1, &pred));
    ok = ok && fabs(pred - y[0]) < 1e-6;
    /* the generic gather used by the mixture-of-experts path */
    double send[3] = {1.0, 2.0, 3.0}, recv[3] = {0, 0, 0};
    CHECK(egx_sweep_allgather(sw, send, 3, recv));
    ok = ok && recv[0] == 1.0 && recv[1] == 2.0 && recv[2] == 3.0;
    egx_gp_destroy(gp);
    egx_sweep_destroy(sw);
    if (!ok) {
        fprintf(stderr, "FAILED\n");
        return 1;
    }
    printf("OK sweep world=1 rccl_version=%d best=%d lkh=%.12g\n", ver, best, lk[best]);
    return 0;
}
