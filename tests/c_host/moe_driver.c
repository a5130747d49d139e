/* A compiled (plain C99) host of the mixture-of-experts entry point at world = 1: three experts (egx_gp_create +
 * egx_gp_finalize), responsibilities from a closed form, and egx_moe_predict_valvar in both recombinations
 *   - without a sweep handle (single process), and
 *   - through a sweep handle that owns a ONE-RANK RCCL communicator (the same all-gather code path as world > 1),
 * against the recombination formulas of crates/moe/src/algorithm.rs:411-423, 670-685 (smooth: sum_e p_e y_e,
 * sum_e p_e^2 v_e) and :879-935 (hard: the expert of argmax_e p_e) applied to the experts' own egx_gp_predict_valvar
 * outputs.  What a Rust `extern "C"` shim behind GpMixture::predict / predict_var would call.  Exit code 0 = all good. */
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

enum { N = 500, D = 2, K = 3, M = 777 };

static unsigned long long rng_state = 88172645463325252ULL;
static double urand(void) {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (double)(rng_state >> 11) / 9007199254740992.0;
}

int main(void) {
    if (egx_device_count() < 1) {
        fprintf(stderr, "no HIP device\n");
        return 2;
    }
    static double x[K][N * D], y[K][N], xq[M * D], probas[M * K];
    static double ye[K][M], ve[K][M], val[M], var[M], val2[M], var2[M];
    egx_gp *experts[K];
    int32_t ids[K];
    egx_gp_config cfg;
    egx_gp_config_default(&cfg);
    cfg.corr = EGX_CORR_MATERN52;
    const double theta[D] = {1.3, 0.9};
    for (int e = 0; e < K; e++) {
        for (int i = 0; i < N; i++) {
            x[e][i * D] = e + urand();          /* cluster e lives on [e, e + 1] x [0, 1] */
            x[e][i * D + 1] = urand();
            y[e][i] = sin(3.0 * x[e][i * D]) + (e + 1) * x[e][i * D + 1];
        }
        CHECK(egx_gp_create(&cfg, x[e], y[e], N, D, &experts[e]));
        CHECK(egx_gp_finalize(experts[e], theta, D));
        ids[e] = e;
    }
    for (int a = 0; a < M; a++) {
        xq[a * D] = 3.0 * urand();
        xq[a * D + 1] = urand();
        double s = 0.0;
        for (int e = 0; e < K; e++) {  /* soft responsibilities around the cluster centres */
            const double dx = xq[a * D] - (e + 0.5);
            probas[a * K + e] = exp(-2.0 * dx * dx);
            s += probas[a * K + e];
        }
        for (int e = 0; e < K; e++) probas[a * K + e] /= s;
    }
    for (int e = 0; e < K; e++) CHECK(egx_gp_predict_valvar(experts[e], xq, M, ye[e], ve[e]));

    unsigned char id[EGX_SWEEP_ID_BYTES];
    CHECK(egx_sweep_unique_id(id));
    egx_sweep *sw = NULL;
    CHECK(egx_sweep_create(&cfg, x[0], y[0], N, D, id, 0, 1, &sw));
    int ok = 1;
    for (int smooth = 0; smooth <= 1; smooth++) {
        CHECK(egx_moe_predict_valvar(NULL, experts, ids, K, K, probas, xq, M, D, smooth, val, var));
        CHECK(egx_moe_predict_valvar(sw, experts, ids, K, K, probas, xq, M, D, smooth, val2, var2));
        double worst = 0.0;
        for (int a = 0; a < M; a++) {
            double wy = 0.0, wv = 0.0;
            if (smooth) {
                for (int e = 0; e < K; e++) {
                    wy += probas[a * K + e] * ye[e][a];
                    wv += probas[a * K + e] * probas[a * K + e] * ve[e][a];
                }
            } else {
                int best = 0;
                for (int e = 1; e < K; e++)
                    if (probas[a * K + e] > probas[a * K + best]) best = e;
                wy = ye[best][a];
                wv = ve[best][a];
            }
            /* routed subsets run through the same kernels in another batch composition: equal to rounding */
            const double ey = fabs(val[a] - wy) / (1.0 + fabs(wy)), ev = fabs(var[a] - wv) / (1e-9 + fabs(wv));
            if (ey > worst) worst = ey;
            if (ev > 1e-6 && fabs(var[a] - wv) > 1e-12 && ev > worst) worst = ev;
            if (val[a] != val2[a] || var[a] != var2[a]) {
                fprintf(stderr, "smooth %d point %d: with / without the communicator differ\n", smooth, a);
                ok = 0;
            }
        }
        if (worst > 1e-9) {
            fprintf(stderr, "smooth %d: worst deviation from the recombination formula %.3e\n", smooth, worst);
            ok = 0;
        }
    }
    /* only one of the outputs; a rank that owns no expert; bad arguments */
    CHECK(egx_moe_predict_valvar(NULL, experts, ids, K, K, probas, xq, M, D, 1, val2, NULL));
    for (int a = 0; a < M; a++) ok = ok && val2[a] == val[a];
    ok = ok && egx_moe_predict_valvar(NULL, experts, ids, K - 1, K, probas, xq, M, D, 1, val, var) == EGX_ERR_INVALID_VALUE;
    ok = ok && egx_moe_predict_valvar(NULL, experts, ids, K, K, probas, xq, M, D, 1, NULL, NULL) == EGX_ERR_INVALID_VALUE;
    egx_sweep_destroy(sw);
    for (int e = 0; e < K; e++) egx_gp_destroy(experts[e]);
    if (!ok) return 1;
    printf("OK moe recombination, %d experts, %d points, both recombinations, with and without a one-rank communicator\n", K, M);
    return 0;
}
