/* A compiled (plain C99) host of the drop-in boundary: no Python, no torch -- what a Rust `extern "C"` shim would do.
 * Fits the reference's notebook problem (doc/Gpx_Tutorial.ipynb cells 9-14: 5 points, constant mean, squared
 * exponential) at its printed theta and checks the printed likelihood / variance and the Python test pins
 * (python/egobox/tests/test_gpmix.py:37-53), then a sparse FITC model on the same data.  Exit code 0 = all good. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "egx_gp.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int32_t rc_ = (call);                                                    \
        if (rc_ != EGX_SUCCESS) {                                                \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, egx_last_error());     \
            return 10 + rc_;                                                     \
        }                                                                        \
    } while (0)

static int close_to(double a, double b, double tol, const char *what) {
    if (fabs(a - b) <= tol) return 1;
    fprintf(stderr, "%s: got %.15g, want %.15g (tol %g)\n", what, a, b, tol);
    return 0;
}

int main(void) {
    const double xt[5] = {0.0, 1.0, 2.0, 3.0, 4.0}, yt[5] = {0.0, 1.0, 1.5, 0.9, 1.0};
    const double theta = 1.83209405; /* printed to 8 digits in the notebook */
    if (egx_device_count() < 1) {
        fprintf(stderr, "no HIP device\n");
        return 2;
    }
    egx_gp *gp = NULL;
    CHECK(egx_gp_create(NULL, xt, yt, 5, 1, &gp)); /* NULL config = Kriging defaults */
    double lkh = 0.0;
    int32_t status = -1;
    CHECK(egx_gp_likelihood(gp, &theta, 1, &lkh, &status));
    int ok = status == EGX_STATUS_OK && close_to(lkh, 0.5781740714613353, 1e-12, "likelihood");
    CHECK(egx_gp_finalize(gp, &theta, 1));
    double sigma2 = 0.0, lk2 = 0.0;
    egx_gp_inner_view view = {0};
    view.sigma2 = &sigma2;
    view.likelihood = &lk2;
    CHECK(egx_gp_get_inner(gp, &view));
    ok &= close_to(sigma2, 0.30494058899172644, 1e-8 * 0.305, "variance") && lk2 == lkh;
    const double xq[2] = {1.0, 1.1};
    double y[2], v[2], gy[2], gv[2];
    CHECK(egx_gp_predict_valvar(gp, xq, 2, y, v));
    CHECK(egx_gp_predict_valvar_gradients(gp, xq, 2, gy, gv));
    ok &= close_to(y[0], 1.0, 1e-7, "predict(1.0)") && close_to(v[0], 0.0, 1e-7, "var(1.0)");
    ok &= close_to(y[1], 1.1163, 1e-3, "predict(1.1)") && close_to(v[1], 0.0, 1e-3, "var(1.1)");
    ok &= close_to(gy[1], 1.1204, 1e-3, "dy/dx(1.1)") && close_to(gv[1], 0.0145, 1e-3, "dvar/dx(1.1)");
    /* error channel: wrong theta length */
    const double th3[3] = {1.0, 1.0, 1.0};
    ok &= egx_gp_finalize(gp, th3, 3) == EGX_ERR_INVALID_VALUE;
    egx_gp_destroy(gp);

    /* sparse GP on the same points, all of them inducing points: nu = noise, the noisy full GP */
    egx_sgp *sgp = NULL;
    CHECK(egx_sgp_create(NULL, xt, yt, 5, 1, xt, 5, &sgp));
    double slk = 0.0;
    CHECK(egx_sgp_likelihood(sgp, &theta, 1, 0.3, 1e-4, &slk, &status));
    ok &= status == EGX_STATUS_OK && isfinite(slk);
    CHECK(egx_sgp_finalize(sgp, &theta, 1, 0.3, 1e-4));
    CHECK(egx_sgp_predict(sgp, xq, 2, y));
    CHECK(egx_sgp_predict_var(sgp, xq, 2, v));
    ok &= close_to(y[0], 1.0, 5e-3, "sparse predict(1.0)") && v[0] > 0.0 && v[0] < 1e-2;
    egx_sgp_destroy(sgp);
    printf("%s: likelihood %.13f variance %.15f predict(1.1) %.6f\n", ok ? "OK" : "FAILED", lkh, sigma2, y[1]);
    return ok ? 0 : 1;
}
