// C++17 host tests written the way the reference writes its own (crates/gp/src/algorithm.rs):
//   test_gp!($regr, $corr)        :1239-1303   default fit on 5 points, predict / predict_var within epsilon = 0.5
//   test_bug_var_derivatives      :1723-1797   fixed theta, d var / dx against central differences (1e-5)
//   golden A                      doc/Gpx_Tutorial.ipynb cell 14 (theta printed, likelihood / variance to full precision)
//   theta0 length check           :829-838     (a panic in the reference, InvalidValueError here)
//   the round-3 members of the mirror (value + variance gradients in one pass, likelihood batch, mixture recombination, pool)
//   the round-5 members: fit_group (models of one shape fitted in lock-step), schedule()
// through include/egx_gp.hpp -> C ABI -> GPU.  Exit code = number of failed checks.
#include <cmath>
#include <cstdio>
#include <vector>

#include "egx_gp.hpp"

using namespace egobox;
static int failures = 0;
#define EXPECT(cond)                                                        \
    do {                                                                    \
        if (!(cond)) {                                                      \
            std::fprintf(stderr, "%s:%d: EXPECT(%s) failed\n", __FILE__, __LINE__, #cond); \
            failures++;                                                     \
        }                                                                   \
    } while (0)

static void test_gp(Mean regr, Corr corr) {
    const double xt[5] = {0.0, 1.0, 2.0, 3.0, 4.0}, yt[5] = {0.0, 1.0, 1.5, 0.9, 1.0};
    if (regr == Mean::Quadratic) {
        // p = 3 basis columns on 5 points is fine; the reference runs it too
    }
    auto gp = GaussianProcess::params(regr, corr).theta_init({0.1}).fit(xt, 5, 1, yt);
    const double xq[2] = {1.0, 3.5};
    auto yvals = gp.predict(xq, 2);
    EXPECT(std::fabs(yvals[0] - 1.0) <= 0.5 && std::fabs(yvals[1] - 0.9) <= 0.5);
    auto yvars = gp.predict_var(xq, 2);
    EXPECT(std::fabs(yvars[0] - 0.0) <= 0.5 && std::fabs(yvars[1] - 0.1) <= 0.5);
    std::vector<double> xplot(100);
    for (int i = 0; i < 100; i++) xplot[i] = 4.0 * i / 99.0;
    auto vals = gp.predict(xplot.data(), 100);
    auto vars = gp.predict_var(xplot.data(), 100);
    for (int i = 0; i < 100; i++) EXPECT(std::isfinite(vals[i]) && vars[i] >= 0.0);
    EXPECT(gp.theta().size() == 1 && gp.theta()[0] >= 1e-2 && gp.theta()[0] <= 1e1);
    EXPECT(gp.dims().first == 1 && gp.dims().second == 1);
}

static void test_golden_a() {
    const double xt[5] = {0.0, 1.0, 2.0, 3.0, 4.0}, yt[5] = {0.0, 1.0, 1.5, 0.9, 1.0};
    auto gp = Kriging::params().theta_tuning(ThetaTuning::Fixed({1.83209405})).fit(xt, 5, 1, yt);
    EXPECT(std::fabs(gp.likelihood() - 0.5781740714613353) <= 1e-12);
    EXPECT(std::fabs(gp.variance() - 0.30494058899172644) <= 1e-8 * 0.305);
    const double x11 = 1.1;
    EXPECT(std::fabs(gp.predict(&x11, 1)[0] - 1.1163) <= 1e-3);
    EXPECT(std::fabs(gp.predict_gradients(&x11, 1)[0] - 1.1204) <= 1e-3);
    EXPECT(std::fabs(gp.predict_var_gradients(&x11, 1)[0] - 0.0145) <= 1e-3);
    // the tuned default fit must do at least as well as the reference's printed optimum
    auto tuned = Kriging::params().fit(xt, 5, 1, yt);
    EXPECT(tuned.likelihood() >= 0.5781740714613353 - 1e-3);
}

static void test_bug_var_derivatives() {
    const double xt[24] = {6.875, -4.375, -3.125, 1.875, 1.875, -1.875, -4.375, 3.125, 8.125, 9.375, 4.375, 4.375,
                           0.625, 0.625,  9.375,  6.875, 5.625, 8.125,  -0.625, -3.125, 3.125, 5.625, -1.875, -0.625};
    const double yt[12] = {2.43286801,  13.10840811, 5.32908578,  17.81862219, 74.08849877, 39.68137781,
                           14.96009727, 63.17475741, 61.26331775, -7.46009727, 44.39159189, 2.17091422};
    auto gp = Kriging::params()
                  .theta_tuning(ThetaTuning::Fixed({std::sqrt(2. * 0.0437386), std::sqrt(2. * 0.00697978)}))
                  .fit(xt, 12, 2, yt);
    const double e = 5e-6, xa = -1.3, xb = 2.5;
    const double x[10] = {xa, xb, xa + e, xb, xa - e, xb, xa, xb + e, xa, xb - e};
    auto y_pred = gp.predict_var(x, 5);
    auto y_deriv = gp.predict_var_gradients(x, 1);
    EXPECT(std::fabs(y_deriv[0] - (y_pred[1] - y_pred[2]) / (2. * e)) <= 1e-5);
    EXPECT(std::fabs(y_deriv[1] - (y_pred[3] - y_pred[4]) / (2. * e)) <= 1e-5);
}

static void test_errors() {
    const double xt[5] = {0.0, 1.0, 2.0, 3.0, 4.0}, yt[5] = {0.0, 1.0, 1.5, 0.9, 1.0};
    bool thrown = false;
    try {
        Kriging::params().theta_init({0.1, 0.2}).fit(xt, 5, 1, yt);
    } catch (const InvalidValueError &) {
        thrown = true;
    }
    EXPECT(thrown);
    thrown = false;
    try {
        Kriging::params().fit(xt, 1, 1, yt);  // one training point
    } catch (const InvalidValueError &) {
        thrown = true;
    }
    EXPECT(thrown);
    // ThetaTuning::Partial keeps the inactive component
    const double x2[12] = {0.1, 0.9, 0.4, 0.2, 0.8, 0.7, 0.3, 0.5, 0.95, 0.05, 0.6, 0.35};
    const double y2[6] = {0.3, 0.1, 0.9, 0.4, 0.7, 0.5};
    auto gp = GaussianProcess::params(Mean::Constant, Corr::Matern52)
                  .theta_tuning(ThetaTuning::Partial({0.7, 1.3}, {{1e-2, 1e1}}, {0}))
                  .n_start(2)
                  .fit(x2, 6, 2, y2);
    EXPECT(gp.theta()[1] == 1.3);
}

// The round-3 entry points through the C++ mirror: one-pass value + variance gradients, the likelihood batch with its
// lock-step width, training_data, the mixture recombination of crates/moe/src/algorithm.rs:411-423, 670-685, 879-935 applied
// to the experts' own outputs, and the resource pool.
static void test_round3_mirror() {
    const double xt[24] = {6.875, -4.375, -3.125, 1.875, 1.875, -1.875, -4.375, 3.125, 8.125, 9.375, 4.375, 4.375,
                           0.625, 0.625,  9.375,  6.875, 5.625, 8.125,  -0.625, -3.125, 3.125, 5.625, -1.875, -0.625};
    const double yt[12] = {2.43286801,  13.10840811, 5.32908578,  17.81862219, 74.08849877, 39.68137781,
                           14.96009727, 63.17475741, 61.26331775, -7.46009727, 44.39159189, 2.17091422};
    auto a = Kriging::params().theta_tuning(ThetaTuning::Fixed({0.3, 0.12})).n_workspaces(4).fit(xt, 12, 2, yt);
    auto b = GaussianProcess::params(Mean::Linear, Corr::Matern52).theta_tuning(ThetaTuning::Fixed({0.2, 0.4})).fit(xt, 12, 2, yt);
    const double xq[6] = {-1.3, 2.5, 4.0, 4.0, 0.0, 7.5};
    auto gy = a.predict_gradients(xq, 3);
    auto gv = a.predict_var_gradients(xq, 3);
    auto both = a.predict_valvar_gradients(xq, 3);
    for (int i = 0; i < 6; i++) {
        EXPECT(std::fabs(both.first[i] - gy[i]) <= 1e-12 * (1.0 + std::fabs(gy[i])));
        EXPECT(std::fabs(both.second[i] - gv[i]) <= 1e-12 * (1.0 + std::fabs(gv[i])));
    }
    // likelihood batch: the fitted theta among the candidates gives the fitted likelihood, any lock-step width the same bits
    const double thetas[8] = {0.3, 0.12, 0.5, 0.5, 0.05, 0.9, 2.0, 0.01};
    EXPECT(a.set_lockstep(1) == 1);
    auto l1 = a.likelihood_batch(thetas, 4, 2);
    EXPECT(a.set_lockstep(4) == 4);
    auto l4 = a.likelihood_batch(thetas, 4, 2);
    for (int c = 0; c < 4; c++) EXPECT(l1.second[c] == l4.second[c] && l1.first[c] == l4.first[c]);
    EXPECT(l1.second[0] == EGX_STATUS_OK && std::fabs(l1.first[0] - a.likelihood()) <= 1e-12 * std::fabs(a.likelihood()));
    // likelihoods + theta-gradients of the same candidates: the likelihoods are the batch's, the gradient of candidate 0 agrees
    // with a central difference of the likelihood, and a second call returns the same bits (fixed reduction order)
    auto g1 = a.likelihood_grad_batch(thetas, 4, 2);
    auto g2 = a.likelihood_grad_batch(thetas, 4, 2);
    for (int c = 0; c < 4; c++) {
        EXPECT(g1.status[c] == l1.second[c]);
        if (g1.status[c] == EGX_STATUS_OK) EXPECT(g1.likelihood[c] == l1.first[c] && g1.likelihood[c] == g2.likelihood[c]);
        for (int l = 0; l < 2; l++) EXPECT(g1.gradient[2 * c + l] == g2.gradient[2 * c + l]);
    }
    {
        const double e = 1e-6, pm[4] = {0.3 + e, 0.12, 0.3 - e, 0.12};
        auto lpm = a.likelihood_batch(pm, 2, 2);
        const double fd = (lpm.first[0] - lpm.first[1]) / (2 * e);
        EXPECT(std::fabs(g1.gradient[0] - fd) <= 1e-4 * (1.0 + std::fabs(fd)));
    }
    a.shrink(2);
    auto l2 = a.likelihood_batch(thetas, 4, 2);  // on the workspaces that are left
    for (int c = 0; c < 4; c++) EXPECT(l2.second[c] == l1.second[c] && (l2.second[c] != EGX_STATUS_OK || l2.first[c] == l1.first[c]));
    auto td = a.training_data(12);
    for (int i = 0; i < 24; i++) EXPECT(td.first[i] == xt[i]);
    for (int i = 0; i < 12; i++) EXPECT(td.second[i] == yt[i]);
    // mixture of the two experts
    const double probas[6] = {0.7, 0.3, 0.2, 0.8, 0.5, 0.5};
    auto ya = a.predict_valvar(xq, 3), yb = b.predict_valvar(xq, 3);
    {   // x-gradients of the hard mixture = the routed experts' own gradients
        auto hg = moe_predict_valvar_gradients({&a, &b}, probas, nullptr, xq, 3, 2, false);
        auto ga = a.predict_valvar_gradients(xq, 3), gb = b.predict_valvar_gradients(xq, 3);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 2; j++) {
                const bool first = probas[2 * i] >= probas[2 * i + 1];
                const double wy = (first ? ga.first : gb.first)[2 * i + j], wv = (first ? ga.second : gb.second)[2 * i + j];
                EXPECT(std::fabs(hg.first[2 * i + j] - wy) <= 1e-10 * (1.0 + std::fabs(wy)));   // (a routed subset is another batch)
                EXPECT(std::fabs(hg.second[2 * i + j] - wv) <= 1e-10 * (1.0 + std::fabs(wv)));
            }
    }
    auto smooth = moe_predict_valvar({&a, &b}, probas, xq, 3, 2, true);
    auto hard = moe_predict_valvar({&a, &b}, probas, xq, 3, 2, false);
    for (int i = 0; i < 3; i++) {
        const double pa = probas[2 * i], pb = probas[2 * i + 1];
        const double sv = pa * ya.first[i] + pb * yb.first[i], svar = pa * pa * ya.second[i] + pb * pb * yb.second[i];
        EXPECT(std::fabs(smooth.first[i] - sv) <= 1e-12 * (1.0 + std::fabs(sv)));
        EXPECT(std::fabs(smooth.second[i] - svar) <= 1e-12 * (1.0 + std::fabs(svar)));
        const bool first = pa >= pb;  // argmax, first maximum
        EXPECT(std::fabs(hard.first[i] - (first ? ya.first[i] : yb.first[i])) <= 1e-12 * (1.0 + std::fabs(sv)));
        EXPECT(std::fabs(hard.second[i] - (first ? ya.second[i] : yb.second[i])) <= 1e-12 * (1.0 + std::fabs(svar)));
    }
    const PoolStats before = pool_stats();
    {
        auto c = Kriging::params().theta_tuning(ThetaTuning::Fixed({0.3, 0.12})).fit(xt, 12, 2, yt);
    }  // destroyed: its resources are cached
    {
        auto c = Kriging::params().theta_tuning(ThetaTuning::Fixed({0.3, 0.12})).fit(xt, 12, 2, yt);
    }
    const PoolStats after = pool_stats();
    EXPECT(after.hits > before.hits && after.cached_bytes > 0);
    EXPECT(trim() > 0 && pool_stats().cached_bytes == 0);
}

// Round 5 through the mirror: fit_group (egx_gp_create_group + egx_gp_finalize_multi: the expert loop of
// crates/moe/src/algorithm.rs:167-177 in lock-step) gives every model bit for bit what `fit` gives it alone, with its fitted
// scalars in place; schedule() reports how a handle factors.
static void test_round5_mirror() {
    const int k = 3, n = 40, d = 2;
    std::vector<double> xs((size_t)k * n * d), ys((size_t)k * n);
    for (int e = 0; e < k; e++)
        for (int i = 0; i < n; i++) {
            const double u = (i * 0.6180339887 + 0.13 * e), v = (i * 0.7548776662 + 0.29 * e);
            const double x0 = 10.0 * (u - std::floor(u)) - 5.0, x1 = 10.0 * (v - std::floor(v));
            xs[((size_t)e * n + i) * d] = x0, xs[((size_t)e * n + i) * d + 1] = x1;
            ys[(size_t)e * n + i] = std::sin(0.7 * x0) * std::cos(0.4 * x1) + 0.1 * e * x0;
        }
    const auto params = GaussianProcess::params(Mean::Linear, Corr::Matern52).theta_tuning(ThetaTuning::Fixed({0.3, 0.2}));
    auto group = params.fit_group(xs.data(), ys.data(), n, d, k);
    EXPECT((int)group.size() == k);
    const double xq[6] = {-1.3, 2.5, 4.0, 4.0, 0.0, 7.5};
    for (int e = 0; e < k; e++) {
        auto lone = params.fit(xs.data() + (size_t)e * n * d, n, d, ys.data() + (size_t)e * n);
        EXPECT(group[(size_t)e].theta().size() == 2 && group[(size_t)e].theta()[0] == 0.3 && group[(size_t)e].theta()[1] == 0.2);
        EXPECT(group[(size_t)e].likelihood() == lone.likelihood() && group[(size_t)e].variance() == lone.variance());
        auto pg = group[(size_t)e].predict_valvar(xq, 3), pl = lone.predict_valvar(xq, 3);
        for (int i = 0; i < 3; i++) EXPECT(pg.first[i] == pl.first[i] && pg.second[i] == pl.second[i]);
        const auto sg = group[(size_t)e].schedule(), sl = lone.schedule();
        EXPECT(sg == sl && sg[2] == 1 && sg[3] == 1);  // 128 columns: the whole factorisation is one chain launch
    }
    // round 6: a TUNED fit_group (ThetaTuning::Full, the default: egx_gp_fit_multi -- all members' COBYLA machines in lock-step)
    // leaves every member where the tuned `fit` of its training set on one workspace leaves it: the same theta*, bit for bit
    auto tuned = GaussianProcess::params(Mean::Constant, Corr::SquaredExponential).n_start(3).fit_group(xs.data(), ys.data(), n, d, k);
    for (int32_t e = 0; e < k; e++) {
        auto lone = GaussianProcess::params(Mean::Constant, Corr::SquaredExponential).n_start(3).fit(xs.data() + (size_t)e * n * d, n, d, ys.data() + (size_t)e * n);
        EXPECT(tuned[(size_t)e].theta() == lone.theta() && tuned[(size_t)e].likelihood() == lone.likelihood());
        EXPECT(tuned[(size_t)e].n_evals() == lone.n_evals() && lone.n_evals() > 4);
    }
    bool threw = false;
    try {
        (void)GaussianProcess::params(Mean::Constant, Corr::SquaredExponential)
            .theta_tuning(ThetaTuning::Partial({0.3, 0.3}, {{0.01, 10.0}}, {0}))
            .fit_group(xs.data(), ys.data(), n, d, k);  // Partial: not for groups
    } catch (const InvalidValueError &) {
        threw = true;
    }
    EXPECT(threw);
}

int main() {
    if (egx_device_count() < 1) {
        std::fprintf(stderr, "no HIP device\n");
        return 99;
    }
    for (Mean m : {Mean::Constant, Mean::Linear, Mean::Quadratic})
        for (Corr c : {Corr::SquaredExponential, Corr::AbsoluteExponential, Corr::Matern32, Corr::Matern52}) test_gp(m, c);
    test_golden_a();
    test_bug_var_derivatives();
    test_errors();
    test_round3_mirror();
    test_round5_mirror();
    std::printf("%s (%d failed checks)\n", failures ? "FAILED" : "OK", failures);
    return failures;
}
