// Optimiser drivers above the likelihood (GpValidParams::fit with ThetaTuning::Full / Partial,
// crates/gp/src/algorithm.rs:873-960, optimization.rs:26-169): multistart COBYLA (cobyla.h: Powell's method restated, all
// starts in lock-step through one likelihood batch per round),
// the new theta-gradient of the likelihood and a projected L-BFGS on it.
#include "gp_handle.h"

using namespace egx;

namespace egx {

// Multistart derivative-free fit over the ACTIVE theta components (all of them for ThetaTuning::Full; a subset for
// ThetaTuning::Partial, algorithm.rs:822-826, 873-960: the inactive components stay at theta_base), in two halves so that
// the starts can be SHARDED over the ranks of a sweep (egx_sweep_fit): fit_run_starts runs the COBYLA machines of the starts
// s with s mod world == rank, fit_reduce_finalize reduces all starts' results and keeps the winner's factor resident.
int fit_run_starts(egx_gp *gp, const double *theta_base /*h*/, const std::vector<int> &active,
                   const double *theta0s /*n_starts x k*/, int64_t n_starts, const double *lo, const double *hi,
                   int64_t bounds_len, int64_t max_eval, int rank, int world, std::vector<StartResult> &results) {
    if (!theta0s || !lo || !hi || n_starts < 1 || active.empty()) {
        set_error("NULL argument / no start point");
        return EGX_ERR_INVALID_VALUE;
    }
    const int hfull = gp->h;
    const int h = (int)active.size();  // optimised dimensions
    if (bounds_len != 1 && bounds_len != h) {  // algorithm.rs:901-912
        set_error("Bounds for theta should be either 1-dim or dim of xtrain (" + std::to_string(h) + "), got " +
                  std::to_string(bounds_len));
        return EGX_ERR_INVALID_VALUE;
    }
    std::vector<double> blo(h), bhi(h);
    for (int i = 0; i < h; i++) {
        const double l = lo[bounds_len == 1 ? 0 : i], u = hi[bounds_len == 1 ? 0 : i];
        if (!(l > 0.0) || !(u >= l)) {
            set_error("theta bounds must satisfy 0 < lo <= hi");
            return EGX_ERR_INVALID_VALUE;
        }
        blo[i] = std::log10(l);  // optimization.rs:32-35
        bhi[i] = std::log10(u);
    }
    std::unique_lock<std::shared_mutex> lock(gp->mu);
    EGX_RC(set_device(gp));
    // maxeval = clamp(10 h, GP_COBYLA_MIN_EVAL = 25, max_eval)  algorithm.rs:933-936
    int64_t per_start = 10 * (int64_t)h;
    if (per_start < 25) per_start = 25;
    if (max_eval >= 25 && per_start > max_eval) per_start = max_eval;
    for (int64_t s = 0; s < n_starts * h; s++)
        if (!(theta0s[s] > 0.0)) {
            set_error("theta start points must be > 0");
            return EGX_ERR_INVALID_VALUE;
        }
    results.assign((size_t)n_starts, StartResult{std::numeric_limits<double>::quiet_NaN(), std::vector<double>(h, 0.0), 0});
    gp->fitted = false;
    // COBYLA (cobyla.h), one machine per start, rhobeg 0.5 / ftol_rel 1e-4 (optimization.rs:16-24), all machines
    // advanced in LOCK-STEP: their trial points form one likelihood batch per round, pipelined over the
    // handle's workspaces (the reference runs the starts as rayon tasks, algorithm.rs:928-945)
    std::vector<int64_t> mine;
    for (int64_t s = rank; s < n_starts; s += world) mine.push_back(s);
    std::vector<CobylaBox> mach;
    mach.reserve(mine.size());
    for (int64_t s : mine) {
        std::vector<double> x0(h);
        for (int i = 0; i < h; i++) x0[i] = std::log10(theta0s[s * h + i]);
        mach.emplace_back(x0, blo, bhi, 0.5, 1e-4, per_start);
    }
    std::vector<double> thetas, lk, x;
    std::vector<int32_t> st;
    std::vector<size_t> who;
    for (;;) {
        thetas.clear();
        who.clear();
        for (size_t q = 0; q < mach.size(); q++)
            if (mach[q].ask(x)) {
                who.push_back(q);
                const size_t off = thetas.size();
                thetas.insert(thetas.end(), theta_base, theta_base + hfull);
                for (int i = 0; i < h; i++) thetas[off + active[i]] = std::pow(10.0, x[i]);
            }
        if (who.empty()) break;
        lk.assign(who.size(), 0.0);
        st.assign(who.size(), 0);
        EGX_RC(likelihood_batch_core(gp, thetas.data(), (int64_t)who.size(), hfull, lk.data(), st.data()));
        for (size_t q = 0; q < who.size(); q++) {
            const bool ok = st[q] == EGX_STATUS_OK && !std::isnan(lk[q]);
            mach[who[q]].tell(ok ? -lk[q] : std::numeric_limits<double>::infinity());  // algorithm.rs:893-896
        }
    }
    for (size_t q = 0; q < mach.size(); q++) {
        const CobylaBox &m = mach[q];
        double fb = m.best_f();
        if (std::isnan(fb) || fb >= 1e30) fb = std::numeric_limits<double>::infinity();  // optimization.rs:153-157
        results[(size_t)mine[q]] = StartResult{fb, m.best_x(), m.evals()};
    }
    return EGX_SUCCESS;
}

int fit_reduce_finalize(egx_gp *gp, const double *theta_base, const std::vector<int> &active, const double *theta0s,
                        const std::vector<StartResult> &results, int64_t *n_evals_out) {
    const int hfull = gp->h, h = (int)active.size();
    double best_f = std::numeric_limits<double>::infinity();
    std::vector<double> best_x(h, 0.0);
    int64_t evals = 0;
    for (size_t s = 0; s < results.size(); s++) {
        evals += results[s].evals;
        if (results[s].f < best_f) {  // algorithm.rs:942-945 reduce to min (first wins ties)
            best_f = results[s].f;
            best_x = results[s].x;
        }
    }
    if (n_evals_out) *n_evals_out = evals;
    std::vector<double> th(theta_base, theta_base + hfull);
    if (std::isfinite(best_f))
        for (int i = 0; i < h; i++) th[active[i]] = std::pow(10.0, best_x[i]);
    else  // every start failed: the reference falls through with ones (algorithm.rs:943) -> 10^1... keep start 0
        for (int i = 0; i < h; i++) th[active[i]] = theta0s[i];
    std::unique_lock<std::shared_mutex> lock(gp->mu);
    EGX_RC(set_device(gp));
    return do_finalize(gp, th.data(), hfull);
}

}  // namespace egx

extern "C" {

static int32_t fit_nm_core(egx_gp *gp, const double *theta_base /*h*/, const std::vector<int> &active,
                           const double *theta0s /*n_starts x k*/, int64_t n_starts, const double *lo,
                           const double *hi, int64_t bounds_len, int64_t max_eval, int64_t *n_evals_out) {
    std::vector<StartResult> results;
    EGX_RC(fit_run_starts(gp, theta_base, active, theta0s, n_starts, lo, hi, bounds_len, max_eval, 0, 1, results));
    return fit_reduce_finalize(gp, theta_base, active, theta0s, results, n_evals_out);
}

int32_t egx_gp_fit(egx_gp *gp, const double *theta0s, int64_t n_starts, const double *lo, const double *hi,
                   int64_t bounds_len, int64_t max_eval, int64_t *n_evals_out) {
    if (!gp || !theta0s || n_starts < 1) {
        set_error("NULL argument / no start point");
        return EGX_ERR_INVALID_VALUE;
    }
    std::vector<int> active(gp->h);
    for (int i = 0; i < gp->h; i++) active[i] = i;
    return fit_nm_core(gp, theta0s, active, theta0s, n_starts, lo, hi, bounds_len, max_eval, n_evals_out);
}

int32_t egx_gp_fit_partial(egx_gp *gp, const double *theta_init, const int64_t *active_idx, int64_t n_active,
                           const double *theta0s, int64_t n_starts, const double *lo, const double *hi,
                           int64_t bounds_len, int64_t max_eval, int64_t *n_evals_out) {
    if (!gp || !theta_init || !active_idx || n_active < 1 || n_active > gp->h) {
        set_error("NULL argument / bad active set");
        return EGX_ERR_INVALID_VALUE;
    }
    std::vector<int> active((size_t)n_active);
    for (int64_t i = 0; i < n_active; i++) {
        if (active_idx[i] < 0 || active_idx[i] >= gp->h || (i > 0 && active_idx[i] <= active_idx[i - 1])) {
            set_error("active theta components must be strictly increasing indices in [0, h)");
            return EGX_ERR_INVALID_VALUE;
        }
        active[(size_t)i] = (int)active_idx[i];
    }
    for (int i = 0; i < gp->h; i++)
        if (!(theta_init[i] > 0.0)) {
            set_error("theta_init must be > 0");
            return EGX_ERR_INVALID_VALUE;
        }
    return fit_nm_core(gp, theta_init, active, theta0s, n_starts, lo, hi, bounds_len, max_eval, n_evals_out);
}
}  // extern "C"

namespace egx {
// likelihood and dL/dtheta on workspace 0 (caller holds gp->mu and has set the device)
static int likelihood_grad_core(egx_gp *gp, const double *theta, int64_t theta_len, double *lkh, double *grad,
                                int32_t *status) {
    const int n = gp->n, n_pad = gp->n_pad, d = gp->d;
    std::vector<double> coef, thfull;
    int hcols = 1;
    EGX_RC(make_coef(gp, theta, theta_len, coef, hcols, &thfull));
    if (has_nan(theta, theta_len)) {
        *lkh = -std::numeric_limits<double>::infinity();
        *status = EGX_STATUS_NAN_THETA;
        for (int k = 0; k < gp->h; k++) grad[k] = 0.0;
        return EGX_SUCCESS;
    }
    gp->fitted = false;
    Workspace &w = gp->ws[0];
    EvalResult res;
    EGX_RC(enqueue_eval(gp, w, coef, hcols));
    EGX_RC(finish_eval(gp, w, res, true));
    *lkh = res.lkh;
    *status = res.status;
    if (res.status != EGX_STATUS_OK) {
        for (int k = 0; k < gp->h; k++) grad[k] = 0.0;
        return EGX_SUCCESS;
    }
    const size_t sq = (size_t)n_pad * n_pad;
    if (!gp->d_W) EGX_HIP_CHECK(hipMalloc(&gp->d_W, sizeof(double) * sq));
    if (!gp->d_Rinv) EGX_HIP_CHECK(hipMalloc(&gp->d_Rinv, sizeof(double) * sq));
    if (!gp->d_gout) EGX_HIP_CHECK(hipMalloc(&gp->d_gout, sizeof(double) * 2 * kMaxDim));
    // d_theta: the coefficient table (d x hcols) followed by |w| (d x h)
    if (!gp->d_theta) EGX_HIP_CHECK(hipMalloc(&gp->d_theta, sizeof(double) * 2 * kMaxDim * kMaxDim));
    // gamma = C^-T rho
    if (!res.rho_on_device) {
        std::memset(w.h_vec, 0, sizeof(double) * n_pad);
        std::memcpy(w.h_vec, res.rho.data(), sizeof(double) * n);
        EGX_HIP_CHECK(hipMemcpyAsync(w.d_rhs, w.h_vec, sizeof(double) * n_pad, hipMemcpyHostToDevice, w.stream));
    }
    EGX_RC(backward_solve(gp, w));
    // W = I * C^-T  (rows of the identity as right-hand sides), then -R^-1 = 0 - W W^T (lower tiles)
    gp->winv_epoch = ~(uint64_t)0;
    EGX_HIP_CHECK(hipMemsetAsync(gp->d_W, 0, sizeof(double) * sq, w.stream));
    {
        std::vector<double> ones(n_pad, 1.0);
        EGX_HIP_CHECK(hipMemcpy2DAsync(gp->d_W, sizeof(double) * (n_pad + 1), ones.data(), sizeof(double),
                                       sizeof(double), n_pad, hipMemcpyHostToDevice, w.stream));
        EGX_HIP_CHECK(hipStreamSynchronize(w.stream));
    }
    EGX_RC(launch_trsm_rows(w.stream, w.M, gp->ld, n_pad, w.dinv, gp->d_W, n_pad, n_pad, 1));
    EGX_HIP_CHECK(hipMemsetAsync(gp->d_Rinv, 0, sizeof(double) * sq, w.stream));
    EGX_RC(launch_gemm_nt_sub(w.stream, gp->d_Rinv, n_pad, gp->d_W, n_pad, gp->d_W, n_pad, n_pad, n_pad, n_pad, 1, 1));
    const int h = gp->h;
    const int nout = (hcols == 1) ? d : h;
    std::vector<double> wabs;
    {
        std::vector<double> up(coef);  // (d x hcols), then |w| (d x h) for the KPLS + Matern form
        if (hcols > 1) {
            wabs.resize((size_t)d * h);
            for (size_t e = 0; e < wabs.size(); e++) wabs[e] = std::fabs(gp->w_star[e]);
            up.insert(up.end(), wabs.begin(), wabs.end());
        }
        EGX_HIP_CHECK(hipMemcpyAsync(gp->d_theta, up.data(), sizeof(double) * up.size(), hipMemcpyHostToDevice, w.stream));
        EGX_HIP_CHECK(hipStreamSynchronize(w.stream));
    }
    EGX_RC(launch_grad_accum(w.stream, gp->corr, gp->d_xT, n_pad, n, d, gp->d_theta, hcols,
                             hcols > 1 ? gp->d_theta + (size_t)d * hcols : nullptr, nout, gp->d_Rinv, n_pad, w.d_vec,
                             gp->d_gout));
    std::vector<double> gout(2 * nout);
    EGX_HIP_CHECK(hipMemcpyAsync(gout.data(), gp->d_gout, sizeof(double) * 2 * nout, hipMemcpyDeviceToHost, w.stream));
    EGX_HIP_CHECK(hipStreamSynchronize(w.stream));
    // dL/dc_k = (1/ln10) [ gamma^T dR_k gamma / sigma2 - tr(R^-1 dR_k) ] ; d_Rinv holds -R^-1
    const double ln10 = std::log(10.0);
    std::vector<double> pk(nout);
    for (int k = 0; k < nout; k++) pk[k] = (gout[nout + k] / res.sigma2n + gout[k]) / ln10;
    if (!gp->has_w || hcols > 1) {
        for (int k = 0; k < h; k++) grad[k] = pk[k];  // w = I (c = theta), or KPLS + Matern (outputs are per theta)
    } else {
        // KPLS with a collapsed per-dimension coefficient c_j (make_coef): chain rule dc_j / dtheta_l
        //   sq-exp : c_j = sqrt(sum_l (theta_l w_jl)^2)  ->  theta_l w_jl^2 / c_j     (correlation_models.rs:97-98)
        //   abs-exp: c_j = sum_l theta_l |w_jl|           ->  |w_jl|                   (:191)
        const double *wm = gp->w_star.data();
        for (int l = 0; l < h; l++) {
            double sacc = 0.0;
            for (int j = 0; j < d; j++) {
                const double wjl = wm[(size_t)j * h + l];
                if (gp->corr == EGX_CORR_SQUARED_EXPONENTIAL) {
                    if (coef[j] > 0.0) sacc += pk[j] * thfull[l] * wjl * wjl / coef[j];
                } else {
                    sacc += pk[j] * std::fabs(wjl);
                }
            }
            grad[l] = sacc;
        }
    }
    return EGX_SUCCESS;
}
}  // namespace egx

extern "C" {

int32_t egx_gp_likelihood_grad(egx_gp *gp, const double *theta, int64_t theta_len, double *lkh, double *grad,
                               int32_t *status) {
    if (!gp || !theta || !lkh || !grad || !status) {
        set_error("NULL argument");
        return EGX_ERR_INVALID_VALUE;
    }
    std::unique_lock<std::shared_mutex> lock(gp->mu);
    EGX_RC(set_device(gp));
    return likelihood_grad_core(gp, theta, theta_len, lkh, grad, status);
}

/* Gradient-based alternative to egx_gp_fit (new: uses the theta-gradient the reference does not have).
 * Projected L-BFGS on x = log10(theta) inside the box, one run per start (sequential: the gradient scratch
 * is per handle), best start wins, then finalize.  max_iter bounds the iterations per start. */
int32_t egx_gp_fit_lbfgs(egx_gp *gp, const double *theta0s, int64_t n_starts, const double *lo, const double *hi,
                         int64_t bounds_len, int64_t max_iter, int64_t *n_evals_out) {
    if (!gp || !theta0s || !lo || !hi || n_starts < 1) {
        set_error("NULL argument / no start point");
        return EGX_ERR_INVALID_VALUE;
    }
    const int h = gp->h;
    if (bounds_len != 1 && bounds_len != h) {
        set_error("Bounds for theta should be either 1-dim or dim of xtrain (" + std::to_string(h) + "), got " +
                  std::to_string(bounds_len));
        return EGX_ERR_INVALID_VALUE;
    }
    std::vector<double> blo(h), bhi(h);
    for (int i = 0; i < h; i++) {
        const double l = lo[bounds_len == 1 ? 0 : i], u = hi[bounds_len == 1 ? 0 : i];
        if (!(l > 0.0) || !(u >= l)) {
            set_error("theta bounds must satisfy 0 < lo <= hi");
            return EGX_ERR_INVALID_VALUE;
        }
        blo[i] = std::log10(l);
        bhi[i] = std::log10(u);
    }
    for (int64_t s = 0; s < n_starts * h; s++)
        if (!(theta0s[s] > 0.0)) {
            set_error("theta start points must be > 0");
            return EGX_ERR_INVALID_VALUE;
        }
    if (max_iter < 1) max_iter = 50;
    std::unique_lock<std::shared_mutex> lock(gp->mu);
    EGX_RC(set_device(gp));
    const double ln10 = std::log(10.0), inf = std::numeric_limits<double>::infinity();
    int64_t evals = 0;
    int rc_inner = EGX_SUCCESS;
    // f(x) = -L(10^x), g = -dL/dx = -theta ln10 dL/dtheta
    auto fg = [&](const std::vector<double> &x, std::vector<double> &g) -> double {
        std::vector<double> th(h), gt(h);
        for (int i = 0; i < h; i++) th[i] = std::pow(10.0, x[i]);
        double lk = 0.0;
        int32_t st = 0;
        evals++;
        int rc = likelihood_grad_core(gp, th.data(), h, &lk, gt.data(), &st);
        if (rc) {
            rc_inner = rc;
            return inf;
        }
        if (st != EGX_STATUS_OK || !std::isfinite(lk)) return inf;
        for (int i = 0; i < h; i++) g[i] = -th[i] * ln10 * gt[i];
        return -lk;
    };
    auto clip = [&](std::vector<double> &x) {
        for (int i = 0; i < h; i++) x[i] = std::fmin(bhi[i], std::fmax(blo[i], x[i]));
    };
    double best_f = inf;
    std::vector<double> best_x(h, 0.0);
    const int mem = 8;
    for (int64_t s = 0; s < n_starts; s++) {
        std::vector<double> x(h), g(h), pg(h), d(h), xn(h), gn(h);
        for (int i = 0; i < h; i++) x[i] = std::log10(theta0s[s * h + i]);
        clip(x);
        double f = fg(x, g);
        if (rc_inner) return rc_inner;
        std::vector<std::vector<double>> S, Y;
        std::vector<double> rho;
        if (std::isfinite(f)) {
            for (int64_t it = 0; it < max_iter; it++) {
                double pgmax = 0.0;
                for (int i = 0; i < h; i++) {
                    const bool at_lo = x[i] <= blo[i] && g[i] > 0.0, at_hi = x[i] >= bhi[i] && g[i] < 0.0;
                    pg[i] = (at_lo || at_hi) ? 0.0 : g[i];
                    pgmax = std::fmax(pgmax, std::fabs(pg[i]));
                }
                if (pgmax <= 1e-5 * (1.0 + std::fabs(f))) break;
                // two-loop recursion on the projected gradient
                std::vector<double> q(pg), alpha(S.size());
                for (int j = (int)S.size() - 1; j >= 0; j--) {
                    double a = 0.0;
                    for (int i = 0; i < h; i++) a += S[j][i] * q[i];
                    a *= rho[j];
                    alpha[j] = a;
                    for (int i = 0; i < h; i++) q[i] -= a * Y[j][i];
                }
                if (!S.empty()) {
                    double sy = 0.0, yy = 0.0;
                    for (int i = 0; i < h; i++) {
                        sy += S.back()[i] * Y.back()[i];
                        yy += Y.back()[i] * Y.back()[i];
                    }
                    for (int i = 0; i < h; i++) q[i] *= sy / yy;
                }
                for (size_t j = 0; j < S.size(); j++) {
                    double b = 0.0;
                    for (int i = 0; i < h; i++) b += Y[j][i] * q[i];
                    b *= rho[j];
                    for (int i = 0; i < h; i++) q[i] += (alpha[j] - b) * S[j][i];
                }
                double dg = 0.0, dmax = 0.0;
                for (int i = 0; i < h; i++) {
                    d[i] = (pg[i] == 0.0) ? 0.0 : -q[i];
                    dg += d[i] * pg[i];
                    dmax = std::fmax(dmax, std::fabs(d[i]));
                }
                if (!(dg < 0.0)) {  // not a descent direction: steepest descent
                    dg = 0.0;
                    dmax = 0.0;
                    for (int i = 0; i < h; i++) {
                        d[i] = -pg[i];
                        dg += d[i] * pg[i];
                        dmax = std::fmax(dmax, std::fabs(d[i]));
                    }
                }
                double t = S.empty() ? std::fmin(1.0, 0.5 / dmax) : std::fmin(1.0, 1.0 / dmax);  // <= 1 decade per step
                double fnew = inf;
                bool ok = false;
                for (int ls = 0; ls < 12; ls++, t *= 0.5) {
                    for (int i = 0; i < h; i++) xn[i] = x[i] + t * d[i];
                    clip(xn);
                    fnew = fg(xn, gn);
                    if (rc_inner) return rc_inner;
                    double dec = 0.0;
                    for (int i = 0; i < h; i++) dec += pg[i] * (xn[i] - x[i]);
                    if (std::isfinite(fnew) && fnew <= f + 1e-4 * dec) {
                        ok = true;
                        break;
                    }
                }
                if (!ok) break;
                std::vector<double> sv(h), yv(h);
                double sy = 0.0;
                for (int i = 0; i < h; i++) {
                    sv[i] = xn[i] - x[i];
                    yv[i] = gn[i] - g[i];
                    sy += sv[i] * yv[i];
                }
                const double fprev = f;
                x = xn;
                g = gn;
                f = fnew;
                if (sy > 1e-12) {
                    S.push_back(sv);
                    Y.push_back(yv);
                    rho.push_back(1.0 / sy);
                    if ((int)S.size() > mem) {
                        S.erase(S.begin());
                        Y.erase(Y.begin());
                        rho.erase(rho.begin());
                    }
                }
                if (std::fabs(fprev - f) <= 1e-7 * (std::fabs(f) + 1e-300)) break;
            }
        }
        if (f < best_f) {
            best_f = f;
            best_x = x;
        }
    }
    if (n_evals_out) *n_evals_out = evals;
    std::vector<double> th(h);
    if (std::isfinite(best_f))
        for (int i = 0; i < h; i++) th[i] = std::pow(10.0, best_x[i]);
    else
        for (int i = 0; i < h; i++) th[i] = theta0s[i];
    return do_finalize(gp, th.data(), h);
}

// ---- kernel-level entry points ------------------------------------------------------------------
}  // extern "C"
