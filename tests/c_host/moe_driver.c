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
    /* the x-gradients of both recombinations (crates/moe/src/algorithm.rs:691-783 smooth, :942-1010 hard) against the same
     * formulas applied to the experts' own egx_gp_predict_valvar_gradients outputs; p' of the closed-form responsibilities */
    {
        static double gye[K][M * D], gve[K][M * D], dprobas[M * K * D], gval[M * D], gvar[M * D], gval2[M * D], gvar2[M * D];
        for (int e = 0; e < K; e++) CHECK(egx_gp_predict_valvar_gradients(experts[e], xq, M, gye[e], gve[e]));
        for (int a = 0; a < M; a++) {
            /* p_e = g_e / s, g_e = exp(-2 (x0 - c_e)^2):  d p_e / d x0 = p_e (-4 (x0 - c_e) - sum_f p_f (-4 (x0 - c_f))), d / d x1 = 0 */
            double mean_dl = 0.0;
            for (int e = 0; e < K; e++) mean_dl += probas[a * K + e] * (-4.0 * (xq[a * D] - (e + 0.5)));
            for (int e = 0; e < K; e++) {
                dprobas[(a * K + e) * D] = probas[a * K + e] * (-4.0 * (xq[a * D] - (e + 0.5)) - mean_dl);
                dprobas[(a * K + e) * D + 1] = 0.0;
            }
        }
        for (int smooth = 0; smooth <= 1; smooth++) {
            CHECK(egx_moe_predict_valvar_gradients(NULL, experts, ids, K, K, probas, dprobas, xq, M, D, smooth, gval, gvar));
            CHECK(egx_moe_predict_valvar_gradients(sw, experts, ids, K, K, probas, dprobas, xq, M, D, smooth, gval2, gvar2));
            double worst = 0.0, scale_y = 1e-300, scale_v = 1e-300;
            for (int e = 0; e < K; e++)
                for (int i = 0; i < M * D; i++) {
                    if (fabs(gye[e][i]) > scale_y) scale_y = fabs(gye[e][i]);
                    if (fabs(gve[e][i]) > scale_v) scale_v = fabs(gve[e][i]);
                }
            for (int a = 0; a < M; a++)
                for (int j = 0; j < D; j++) {
                    double wy = 0.0, wv = 0.0;
                    if (smooth) {
                        for (int e = 0; e < K; e++) {
                            const double p = probas[a * K + e], pp = dprobas[(a * K + e) * D + j];
                            wy += p * gye[e][a * D + j] + pp * ye[e][a];
                            wv += p * p * gve[e][a * D + j] + 2.0 * p * pp * ve[e][a];
                        }
                    } else {
                        int best = 0;
                        for (int e = 1; e < K; e++)
                            if (probas[a * K + e] > probas[a * K + best]) best = e;
                        wy = gye[best][a * D + j];
                        wv = gve[best][a * D + j];
                    }
                    const double ey = fabs(gval[a * D + j] - wy) / scale_y, ev = fabs(gvar[a * D + j] - wv) / scale_v;
                    if (ey > worst) worst = ey;
                    if (ev > worst) worst = ev;
                    if (gval[a * D + j] != gval2[a * D + j] || gvar[a * D + j] != gvar2[a * D + j]) ok = 0;
                }
            if (worst > 1e-8) {
                fprintf(stderr, "gradients, smooth %d: worst deviation from the recombination formula %.3e\n", smooth, worst);
                ok = 0;
            }
        }
        ok = ok && egx_moe_predict_valvar_gradients(NULL, experts, ids, K, K, probas, NULL, xq, M, D, 1, gval, gvar) == EGX_ERR_INVALID_VALUE;
        CHECK(egx_moe_predict_valvar_gradients(NULL, experts, ids, K, K, probas, NULL, xq, M, D, 0, gval2, NULL));
    }
    /* only one of the outputs; a rank that owns no expert; bad arguments */
    CHECK(egx_moe_predict_valvar(NULL, experts, ids, K, K, probas, xq, M, D, 1, val2, NULL));
    for (int a = 0; a < M; a++) ok = ok && val2[a] == val[a];
    ok = ok && egx_moe_predict_valvar(NULL, experts, ids, K - 1, K, probas, xq, M, D, 1, val, var) == EGX_ERR_INVALID_VALUE;
    ok = ok && egx_moe_predict_valvar(NULL, experts, ids, K, K, probas, xq, M, D, 1, NULL, NULL) == EGX_ERR_INVALID_VALUE;
    egx_sweep_destroy(sw);
    for (int e = 0; e < K; e++) egx_gp_destroy(experts[e]);
    if (!ok) return 1;
    printf("OK moe recombination (values, variances, x-gradients), %d experts, %d points, both recombinations, with and "
           "without a one-rank communicator\n", K, M);
    return 0;
}
