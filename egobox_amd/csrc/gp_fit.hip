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
// Both halves expect the CALLER to hold gp->mu exclusively (one critical section from the first evaluation to the
// finalized model, as the header documents for a handle: fit_nm_core below, egx_sweep_fit around its all-gather).
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
    EGX_RC(set_device(gp));
    return do_finalize(gp, th.data(), hfull);
}

}  // namespace egx

extern "C" {

static int32_t fit_nm_core(egx_gp *gp, const double *theta_base /*h*/, const std::vector<int> &active,
                           const double *theta0s /*n_starts x k*/, int64_t n_starts, const double *lo,
                           const double *hi, int64_t bounds_len, int64_t max_eval, int64_t *n_evals_out) {
    std::vector<StartResult> results;
    std::unique_lock<std::shared_mutex> lock(gp->mu);  // the whole fit is one critical section
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

// ---------------------------------------------------------------------------------------------
// K7, host side: likelihood AND dL/dtheta of k candidates (the reference has no gradient,
// algorithm.rs:880 ignores `_gradient`).  Per candidate, after the evaluation of the likelihood left the factor C in the
// workspace:   W = C^-T (rows of the identity through the forward block substitution, n^3/3 flop; round 4: it RIDES ALONG
//                                          the factorisation, group of panels by group of panels on a stream of its own --
//                                          PotrfInverse, egx_internal.h)
//              gamma = W rho                       (one pass over W)
//              -R^-1 = -(W W^T), lower triangle    (LDS-DMA stream kernel with per-tile K ranges, n^3/3 flop; written
//                                                   OVER the factor: no second n^2 buffer)
//              dL/dc_o = (1 / ln 10) sum_{i>j} 2 R_ij (gamma_i gamma_j / sigma2 - R^-1_ij) dlogR_ij/dc_o   (k_grad_accum)
// Round 4: everything is enqueued on the evaluation's stream behind the likelihood (round 3 had three blocking host
// round trips per gradient), the candidates of a batch are pipelined over the handle's workspaces like a likelihood
// batch, and a slot's candidates run every stage in LOCK-STEP (grid.z over consecutive workspaces and consecutive C^-T
// buffers of `slab_W`).  Every stage is deterministic and its kernel choice depends on one matrix' size only, so a
// candidate's (likelihood, gradient) are the same bits alone and in any batch.
// ---------------------------------------------------------------------------------------------
static int ensure_grad_scratch(egx_gp *gp, int ws_lo, int nuse) {
    const size_t sq = (size_t)gp->n_pad * gp->n_pad;
    if (gp->slab_W_count < nuse) {
        if (gp->slab_W) (void)hipFree(gp->slab_W);
        gp->slab_W = nullptr;
        gp->slab_W_count = 0;
        EGX_HIP_CHECK(dev_malloc(&gp->slab_W, sizeof(double) * sq * (size_t)nuse));
        gp->slab_W_count = nuse;
    }
    for (int i = ws_lo; i < ws_lo + nuse; i++) {
        Workspace &w = gp->ws[i];
        if (!w.d_gpart) EGX_HIP_CHECK(dev_malloc(&w.d_gpart, sizeof(double) * (size_t)grad_partial_doubles(gp->d)));
        if (!w.d_gout) EGX_HIP_CHECK(dev_malloc(&w.d_gout, sizeof(double) * (size_t)(gp->d + 64)));
        if (!w.h_gout) EGX_HIP_CHECK(hipHostMalloc(&w.h_gout, sizeof(double) * (size_t)(gp->d + 64), hipHostMallocDefault));
    }
    if (gp->has_w && !gp->d_wabs) {
        std::vector<double> wabs(gp->w_star.size());
        for (size_t e = 0; e < wabs.size(); e++) wabs[e] = std::fabs(gp->w_star[e]);
        EGX_HIP_CHECK(dev_malloc(&gp->d_wabs, sizeof(double) * wabs.size()));
        EGX_HIP_CHECK(hipMemcpy(gp->d_wabs, wabs.data(), sizeof(double) * wabs.size(), hipMemcpyHostToDevice));
    }
    return EGX_SUCCESS;
}

// the gradient stages of `count` consecutive workspaces from w0 (their factors in place, rho in d_rhs) on stream st;
// C^-T buffers slot0 .. of slab_W
// pre != 0: every candidate of the run has hcols == 1 and positive coefficients -- the trace kernel takes the prescaled form
static int enqueue_grad_run(egx_gp *gp, hipStream_t st, int w0, int count, int slot0, const double *inv_s2, int hcols, int pre) {
    const int n = gp->n, n_pad = gp->n_pad;
    const int64_t sq = (int64_t)n_pad * n_pad;
    Workspace &lead = gp->ws[w0];
    double *W0 = gp->slab_W + (int64_t)slot0 * sq;  // C^-T of these candidates: rode along their factorisation
    for (int j = 0; j < count; j++) {
        Workspace &w = gp->ws[w0 + j];
        EGX_RC(launch_uptri_gemv(st, W0 + (int64_t)j * sq, n_pad, n, w.d_rhs, w.d_vec));
    }
    GemmBatch gbm;
    gbm.count = count;
    gbm.sC = gp->stride_M;
    gbm.sA = gbm.sB = sq;
    EGX_RC(launch_syrk_uptri_neg(st, lead.M, gp->ld, W0, n_pad, n_pad, &gbm));
    const int nout = (hcols == 1) ? gp->d : gp->h;
    GradBatch gb;
    gb.count = count;
    for (int j = 0; j < count; j++) {
        Workspace &w = gp->ws[w0 + j];
        if (pre) EGX_RC(launch_scale_rows(st, gp->d_xT, n_pad, gp->d, w.d_coef, w.d_xs));  // (K1 may have taken its LDS-only form)
        gb.xs[j] = w.d_xs;
        gb.coef[j] = w.d_coef;
        gb.gamma[j] = w.d_vec;
        gb.rneg[j] = w.M;
        gb.inv_s2[j] = inv_s2[j];
        gb.part[j] = w.d_gpart;
        gb.out[j] = w.d_gout;
    }
    for (int j = count; j < kGradMaxBatch; j++) {
        gb.xs[j] = gb.coef[j] = gb.gamma[j] = gb.rneg[j] = nullptr;
        gb.inv_s2[j] = 0.0;
        gb.part[j] = gb.out[j] = nullptr;
    }
    EGX_RC(launch_grad_accum(st, gp->corr, gp->d_xT, n_pad, n, gp->d, hcols, hcols > 1 ? gp->d_wabs : nullptr, nout, gp->ld, gb,
                             pre));
    for (int j = 0; j < count; j++) {
        Workspace &w = gp->ws[w0 + j];
        EGX_HIP_CHECK(hipMemcpyAsync(w.h_gout, w.d_gout, sizeof(double) * nout, hipMemcpyDeviceToHost, st));
    }
    return EGX_SUCCESS;
}

// dL/dtheta (h) from the kernel's sums over the coefficients (nout): the chain rule of make_coef
static void grad_chain_rule(const egx_gp *gp, const double *gsum, int hcols, const std::vector<double> &coef,
                            const std::vector<double> &thfull, double *grad) {
    const int d = gp->d, h = gp->h;
    const double ln10 = std::log(10.0);
    if (!gp->has_w || hcols > 1) {
        for (int k = 0; k < h; k++) grad[k] = gsum[k] / ln10;  // w = I (c = theta), or KPLS + Matern (outputs are per theta)
        return;
    }
    // KPLS with a collapsed per-dimension coefficient c_j (make_coef): chain rule dc_j / dtheta_l
    //   sq-exp : c_j = sqrt(sum_l (theta_l w_jl)^2)  ->  theta_l w_jl^2 / c_j     (correlation_models.rs:97-98)
    //   abs-exp: c_j = sum_l theta_l |w_jl|           ->  |w_jl|                   (:191)
    const double *wm = gp->w_star.data();
    for (int l = 0; l < h; l++) {
        double sacc = 0.0;
        for (int j = 0; j < d; j++) {
            const double wjl = wm[(size_t)j * h + l], pk = gsum[j] / ln10;
            if (gp->corr == EGX_CORR_SQUARED_EXPONENTIAL) {
                if (coef[j] > 0.0) sacc += pk * thfull[l] * wjl * wjl / coef[j];
            } else {
                sacc += pk * std::fabs(wjl);
            }
        }
        grad[l] = sacc;
    }
}

// likelihood + gradient of the rows of thetas (k x theta_len): lkh[c], grad[c * h ..], status[c].  Caller holds gp->mu
// exclusively and has set the device.  A fitted model keeps workspace 0 as long as another workspace exists.
int likelihood_grad_batch_core(egx_gp *gp, const double *thetas, int64_t k, int64_t theta_len, double *lkh, double *grad,
                               int32_t *status) {
    if (k <= 0) return EGX_SUCCESS;
    const int h = gp->h;
    const int ws_lo = (gp->fitted && gp->ws.size() > 1) ? 1 : 0;
    const int nws = (int)gp->ws.size() - ws_lo;
    const int nuse = (int)std::min<int64_t>(nws, k);
    int B = gp->lockstep < 1 ? 1 : gp->lockstep;
    B = std::min(std::min(B, nuse), kGradMaxBatch);
    const int nslots = (nuse + B - 1) / B;
    EGX_RC(ensure_grad_scratch(gp, ws_lo, nuse));
    if (ws_lo == 0) gp->fitted = false;
    struct Slot {
        int phase = 0;  // 0 idle, 1 likelihood in flight, 2 gradient in flight
        std::vector<int64_t> cand;
        std::vector<std::vector<double>> coefs, thfull;
        std::vector<char> ok, pre;  // ok: has a gradient; pre: eligible for the prescaled trace kernel
        int hcols = 1;
    };
    std::vector<Slot> slots(nslots);
    for (auto &sl : slots) {
        sl.coefs.resize(B);
        sl.thfull.resize(B);
    }
    auto slot_cap = [&](int i) { return std::min(B, nuse - i * B); };
    int64_t next = 0;
    int busy = 0;
    auto run = [&]() -> int {
        for (int i = 0;; i = (i + 1) % nslots) {
            Slot &sl = slots[i];
            const int w0 = ws_lo + i * B;
            hipStream_t st = gp->ws[w0].stream;
            if (sl.phase == 2) {
                EGX_HIP_CHECK(hipStreamSynchronize(st));
                for (size_t j = 0; j < sl.cand.size(); j++)
                    if (sl.ok[j])
                        grad_chain_rule(gp, gp->ws[w0 + (int)j].h_gout, sl.hcols, sl.coefs[j], sl.thfull[j], grad + sl.cand[j] * h);
                sl.cand.clear();
                sl.phase = 0;
                busy--;
            } else if (sl.phase == 1) {
                // host half of the likelihoods (GLS, sigma2), then the gradient stages for the candidates that have one
                const int cnt = (int)sl.cand.size();
                std::vector<double> inv_s2(cnt, 0.0);
                sl.ok.assign(cnt, 0);
                sl.pre.assign(cnt, 0);
                for (int j = 0; j < cnt; j++) {
                    Workspace &w = gp->ws[w0 + j];
                    EvalResult res;
                    EGX_RC(finish_eval(gp, w, res, 2));
                    const int64_t c = sl.cand[j];
                    lkh[c] = res.lkh;
                    status[c] = res.status;
                    if (res.status != EGX_STATUS_OK || !(res.sigma2n > 0.0) || !std::isfinite(res.lkh)) continue;
                    sl.ok[j] = 1;
                    inv_s2[j] = 1.0 / res.sigma2n;
                    // the prescaled form divides by the coefficient once per output: a candidate's OWN coefficients decide
                    // (never its companions': the kernel choice must not depend on the batch)
                    sl.pre[j] = sl.hcols == 1 ? 1 : 0;
                    for (double cj : sl.coefs[j])
                        if (!(cj > 0.0) || !std::isfinite(cj)) sl.pre[j] = 0;
                    if (!res.rho_on_device) {  // host GLS: rho goes back (zero padded)
                        std::memset(w.h_vec, 0, sizeof(double) * gp->n_pad);
                        std::memcpy(w.h_vec, res.rho.data(), sizeof(double) * gp->n);
                        EGX_HIP_CHECK(hipMemcpyAsync(w.d_rhs, w.h_vec, sizeof(double) * gp->n_pad, hipMemcpyHostToDevice, st));
                    }
                }
                // maximal runs of consecutive candidates with a gradient (and the same form of the trace kernel): each run is one
                // lock-step launch sequence
                for (int j = 0; j < cnt;) {
                    if (!sl.ok[j]) {
                        j++;
                        continue;
                    }
                    int e = j;
                    while (e < cnt && sl.ok[e] && sl.pre[e] == sl.pre[j]) e++;
                    EGX_RC(enqueue_grad_run(gp, st, w0 + j, e - j, i * B + j, inv_s2.data() + j, sl.hcols, sl.pre[j]));
                    j = e;
                }
                sl.phase = 2;
                continue;
            }
            // idle: the next candidates (NaN thetas are answered at once, algorithm.rs:885-891)
            while (next < k && (int)sl.cand.size() < slot_cap(i)) {
                const int64_t c = next++;
                const double *th = thetas + c * theta_len;
                const size_t j = sl.cand.size();
                EGX_RC(make_coef(gp, th, theta_len, sl.coefs[j], sl.hcols, &sl.thfull[j]));
                for (int l = 0; l < h; l++) grad[c * h + l] = 0.0;
                if (has_nan(th, theta_len)) {
                    lkh[c] = -std::numeric_limits<double>::infinity();
                    status[c] = EGX_STATUS_NAN_THETA;
                    continue;
                }
                sl.cand.push_back(c);
            }
            if (!sl.cand.empty()) {
                EGX_RC(enqueue_eval_group(gp, w0, (int)sl.cand.size(), sl.coefs.data(), sl.hcols,
                                          gp->slab_W + (int64_t)(i * B) * gp->n_pad * gp->n_pad));
                sl.phase = 1;
                busy++;
            }
            if (next >= k && busy == 0) return EGX_SUCCESS;
        }
    };
    const int rc = run();
    if (rc) {  // leave no work behind that still writes into the workspaces
        const std::string msg = last_error_string();
        for (int i = 0; i < nslots; i++)
            if (slots[i].phase != 0) (void)hipStreamSynchronize(gp->ws[ws_lo + i * B].stream);
        (void)hipGetLastError();
        set_error(msg);
    }
    return rc;
}
}  // namespace egx

extern "C" {

int32_t egx_gp_likelihood_grad(egx_gp *gp, const double *theta, int64_t theta_len, double *lkh, double *grad,
                               int32_t *status) {
    if (!gp || !theta || !lkh || !grad || !status) {
        set_error("NULL argument");
        return EGX_ERR_INVALID_VALUE;
    }
    if (theta_len != 1 && theta_len != gp->h) {
        set_error("theta should be either 1-dim or dim of xtrain (w_star.ncols()), got " + std::to_string(theta_len));
        return EGX_ERR_INVALID_VALUE;
    }
    std::unique_lock<std::shared_mutex> lock(gp->mu);
    EGX_RC(set_device(gp));
    return likelihood_grad_batch_core(gp, theta, 1, theta_len, lkh, grad, status);
}

int32_t egx_gp_likelihood_grad_batch(egx_gp *gp, const double *thetas, int64_t k, int64_t theta_len, double *lkh,
                                     double *grads, int32_t *status) {
    if (!gp || k < 0 || (k > 0 && (!thetas || !lkh || !grads || !status))) {
        set_error("NULL argument");
        return EGX_ERR_INVALID_VALUE;
    }
    if (theta_len != 1 && theta_len != gp->h) {
        set_error("theta should be either 1-dim or dim of xtrain (w_star.ncols()), got " + std::to_string(theta_len));
        return EGX_ERR_INVALID_VALUE;
    }
    std::unique_lock<std::shared_mutex> lock(gp->mu);
    EGX_RC(set_device(gp));
    return likelihood_grad_batch_core(gp, thetas, k, theta_len, lkh, grads, status);
}

/* Gradient-based alternative to egx_gp_fit (new: uses the theta-gradient the reference does not have).
 * Projected L-BFGS on x = log10(theta) inside the box, one run per start; round 4: every start is an ask / tell machine
 * and ALL starts advance in lock-step -- a round's trial points (one per start that is still running) are ONE
 * likelihood + gradient batch (egx_gp_likelihood_grad_batch), as the COBYLA starts of egx_gp_fit are one likelihood
 * batch.  Each start walks exactly the points it walks alone.  Best start wins (the first on ties), then finalize.
 * max_iter bounds the iterations per start. */
namespace {
struct LbfgsStart {
    int h = 0;
    int64_t max_iter = 50;
    const std::vector<double> *blo = nullptr, *bhi = nullptr;
    std::vector<double> x, g, pg, dir, xn;
    std::vector<std::vector<double>> S, Y;
    std::vector<double> rho;
    double f = std::numeric_limits<double>::infinity(), t = 1.0;
    int64_t it = 0;
    int ls = 0;
    enum { INIT, LINE, DONE } st = INIT;
    static constexpr int mem = 8;

    void clip(std::vector<double> &v) const {
        for (int i = 0; i < h; i++) v[i] = std::fmin((*bhi)[i], std::fmax((*blo)[i], v[i]));
    }
    LbfgsStart(const double *theta0, int h_, const std::vector<double> &lo, const std::vector<double> &hi, int64_t max_it)
        : h(h_), max_iter(max_it), blo(&lo), bhi(&hi), x(h_), g(h_), pg(h_), dir(h_), xn(h_) {
        for (int i = 0; i < h; i++) x[i] = std::log10(theta0[i]);
        clip(x);
    }
    // the point to evaluate next (log10 theta), or false when this start is finished
    bool ask(std::vector<double> &out) const {
        if (st == DONE) return false;
        out = (st == INIT) ? x : xn;
        return true;
    }
    // start iteration `it` at (x, f, g): convergence test, two-loop recursion on the projected gradient, first trial step
    void begin_iteration() {
        if (it >= max_iter) {
            st = DONE;
            return;
        }
        double pgmax = 0.0;
        for (int i = 0; i < h; i++) {
            const bool at_lo = x[i] <= (*blo)[i] && g[i] > 0.0, at_hi = x[i] >= (*bhi)[i] && g[i] < 0.0;
            pg[i] = (at_lo || at_hi) ? 0.0 : g[i];
            pgmax = std::fmax(pgmax, std::fabs(pg[i]));
        }
        if (pgmax <= 1e-5 * (1.0 + std::fabs(f))) {
            st = DONE;
            return;
        }
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
            dir[i] = (pg[i] == 0.0) ? 0.0 : -q[i];
            dg += dir[i] * pg[i];
            dmax = std::fmax(dmax, std::fabs(dir[i]));
        }
        if (!(dg < 0.0)) {  // not a descent direction: steepest descent
            dmax = 0.0;
            for (int i = 0; i < h; i++) {
                dir[i] = -pg[i];
                dmax = std::fmax(dmax, std::fabs(dir[i]));
            }
        }
        t = S.empty() ? std::fmin(1.0, 0.5 / dmax) : std::fmin(1.0, 1.0 / dmax);  // <= 1 decade per step
        ls = 0;
        trial();
        st = LINE;
    }
    void trial() {
        for (int i = 0; i < h; i++) xn[i] = x[i] + t * dir[i];
        clip(xn);
    }
    // objective and gradient (in x) at the point ask() handed out; fv = +inf for a failed evaluation
    void tell(double fv, const std::vector<double> &gv) {
        if (st == INIT) {
            f = fv;
            g = gv;
            if (!std::isfinite(f)) {
                st = DONE;
                return;
            }
            begin_iteration();
            return;
        }
        if (st != LINE) return;
        double dec = 0.0;
        for (int i = 0; i < h; i++) dec += pg[i] * (xn[i] - x[i]);
        if (std::isfinite(fv) && fv <= f + 1e-4 * dec) {  // Armijo on the projected step
            std::vector<double> sv(h), yv(h);
            double sy = 0.0;
            for (int i = 0; i < h; i++) {
                sv[i] = xn[i] - x[i];
                yv[i] = gv[i] - g[i];
                sy += sv[i] * yv[i];
            }
            const double fprev = f;
            x = xn;
            g = gv;
            f = fv;
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
            it++;
            if (std::fabs(fprev - f) <= 1e-7 * (std::fabs(f) + 1e-300)) {
                st = DONE;
                return;
            }
            begin_iteration();
            return;
        }
        if (++ls >= 12) {  // the line search gave up: this start ends where it is
            st = DONE;
            return;
        }
        t *= 0.5;
        trial();
    }
};
}  // namespace

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
    gp->fitted = false;
    const double ln10 = std::log(10.0), inf = std::numeric_limits<double>::infinity();
    std::vector<LbfgsStart> mach;
    mach.reserve((size_t)n_starts);
    for (int64_t s = 0; s < n_starts; s++) mach.emplace_back(theta0s + s * h, h, blo, bhi, max_iter);
    int64_t evals = 0;
    std::vector<double> thetas, lk, gr, x, gx(h);
    std::vector<int32_t> stv;
    std::vector<size_t> who;
    for (;;) {
        thetas.clear();
        who.clear();
        for (size_t q = 0; q < mach.size(); q++)
            if (mach[q].ask(x)) {
                who.push_back(q);
                for (int i = 0; i < h; i++) thetas.push_back(std::pow(10.0, x[i]));
            }
        if (who.empty()) break;
        lk.assign(who.size(), 0.0);
        gr.assign(who.size() * (size_t)h, 0.0);
        stv.assign(who.size(), 0);
        evals += (int64_t)who.size();
        EGX_RC(likelihood_grad_batch_core(gp, thetas.data(), (int64_t)who.size(), h, lk.data(), gr.data(), stv.data()));
        // f(x) = -L(10^x), g = -dL/dx = -theta ln10 dL/dtheta
        for (size_t q = 0; q < who.size(); q++) {
            const bool ok = stv[q] == EGX_STATUS_OK && std::isfinite(lk[q]);
            for (int i = 0; i < h; i++) gx[i] = ok ? -thetas[q * h + i] * ln10 * gr[q * h + i] : 0.0;
            mach[who[q]].tell(ok ? -lk[q] : inf, gx);
        }
    }
    double best_f = inf;
    std::vector<double> best_x(h, 0.0);
    for (const auto &m : mach)
        if (m.f < best_f) {
            best_f = m.f;
            best_x = m.x;
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
