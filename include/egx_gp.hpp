// Header-only C++17 host API over the C ABI of egx_gp.h, shaped like the reference's Rust builder
// (crates/gp/src/parameters.rs:167-313, crates/gp/src/algorithm.rs:200-440):
//
//     auto gp = egobox::GaussianProcess::params(egobox::Mean::Constant, egobox::Corr::SquaredExponential)
//                   .theta_tuning(egobox::ThetaTuning::Fixed({1.83209405}))
//                   .fit(xt, n, d, yt);                     // Err(GpError::..) of the reference -> C++ exceptions
//     auto y = gp.predict(xq, m);  auto v = gp.predict_var(xq, m);  gp.theta(); gp.variance(); gp.likelihood();
//
// This image has no Rust toolchain; this is the compiled-language mirror of the reference interface (INTEGRATION.md
// shows the Rust `extern "C"` shim).  Everything numerical happens behind the C ABI on the GPU.
#ifndef EGX_GP_HPP
#define EGX_GP_HPP

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <memory>
#include <random>
#include <stdexcept>
#include <string>
#include <utility>
#include <array>
#include <vector>

#include "egx_gp.h"

namespace egobox {

enum class Mean { Constant = EGX_MEAN_CONSTANT, Linear = EGX_MEAN_LINEAR, Quadratic = EGX_MEAN_QUADRATIC };
enum class Corr {
    SquaredExponential = EGX_CORR_SQUARED_EXPONENTIAL,
    AbsoluteExponential = EGX_CORR_ABSOLUTE_EXPONENTIAL,
    Matern32 = EGX_CORR_MATERN32,
    Matern52 = EGX_CORR_MATERN52
};

// crates/gp/src/errors.rs:8-40
struct GpError : std::runtime_error {
    int rc;
    GpError(int rc_, const std::string &m) : std::runtime_error(m), rc(rc_) {}
};
struct InvalidValueError : GpError { using GpError::GpError; };
struct LinalgError : GpError { using GpError::GpError; };
struct LikelihoodComputationError : GpError { using GpError::GpError; };
struct PeerError : GpError { using GpError::GpError; };  // another rank of a collective failed (EGX_ERR_PEER)

inline void check(int32_t rc) {
    if (rc == EGX_SUCCESS) return;
    const std::string msg = egx_last_error();
    switch (rc) {
        case EGX_ERR_INVALID_VALUE: throw InvalidValueError(rc, msg);
        case EGX_ERR_LINALG: throw LinalgError(rc, msg);
        case EGX_ERR_LIKELIHOOD: throw LikelihoodComputationError(rc, msg);
        case EGX_ERR_PEER: throw PeerError(rc, msg);
        default: throw GpError(rc, msg);
    }
}

// crates/gp/src/parameters.rs:11-78
struct ThetaTuning {
    enum class Kind { Fixed, Full, Partial } kind = Kind::Full;
    std::vector<double> init{DEFAULT_INIT};
    std::vector<std::pair<double, double>> bounds{{DEFAULT_LO, DEFAULT_HI}};
    std::vector<int64_t> active;
    static constexpr double DEFAULT_INIT = 1e-1, DEFAULT_LO = 1e-2, DEFAULT_HI = 1e1;
    static ThetaTuning Fixed(std::vector<double> v) {
        ThetaTuning t;
        t.kind = Kind::Fixed;
        t.init = std::move(v);
        return t;
    }
    static ThetaTuning Full(std::vector<double> i, std::vector<std::pair<double, double>> b) {
        ThetaTuning t;
        t.init = std::move(i);
        t.bounds = std::move(b);
        return t;
    }
    static ThetaTuning Partial(std::vector<double> i, std::vector<std::pair<double, double>> b, std::vector<int64_t> a) {
        ThetaTuning t = Full(std::move(i), std::move(b));
        t.kind = Kind::Partial;
        t.active = std::move(a);
        return t;
    }
};

class GaussianProcess;

// GpParams / GpValidParams, parameters.rs:80-313
class GpParams {
  public:
    GpParams(Mean m, Corr c) : mean_(m), corr_(c) {}
    GpParams &theta_tuning(ThetaTuning t) { tuning_ = std::move(t); return *this; }
    GpParams &theta_init(std::vector<double> v) { tuning_.init = std::move(v); return *this; }
    GpParams &theta_bounds(std::vector<std::pair<double, double>> b) { tuning_.bounds = std::move(b); return *this; }
    GpParams &n_start(int n) { n_start_ = n; return *this; }
    GpParams &max_eval(int n) { max_eval_ = n; return *this; }
    GpParams &nugget(double v) { nugget_ = v; return *this; }
    GpParams &seed(uint64_t s) { seed_ = s; return *this; }
    GpParams &device(int dev) { device_ = dev; return *this; }
    GpParams &n_workspaces(int n) { n_workspaces_ = n; return *this; }
    // Fit<..>::fit, algorithm.rs:785-980.  x is (n x d) row-major, y has n entries.
    GaussianProcess fit(const double *x, int64_t n, int64_t d, const double *y) const;
    // `fit` (ThetaTuning::Fixed or ::Full) for k training sets of one shape at once: x (k x n x d), y (k x n).  The expert loop of
    // egobox-moe (crates/moe/src/algorithm.rs:167-177) with the experts factored in LOCK-STEP (egx_gp_create_group +
    // egx_gp_finalize_multi / egx_gp_fit_multi); each model is what `fit` returns for its training set on one workspace.
    std::vector<GaussianProcess> fit_group(const double *x, const double *y, int64_t n, int64_t d, int32_t k) const;

  private:
    Mean mean_;
    Corr corr_;
    ThetaTuning tuning_;
    int n_start_ = 10, max_eval_ = 50;  // GP_OPTIM_N_START, GP_COBYLA_MAX_EVAL (parameters.rs:107-121)
    double nugget_ = 100.0 * 2.220446049250313e-16;
    uint64_t seed_ = 42;
    int device_ = -1, n_workspaces_ = 1;
};

// GaussianProcess, algorithm.rs:174-440
class GaussianProcess {
  public:
    static GpParams params(Mean m, Corr c) { return GpParams(m, c); }

    std::vector<double> predict(const double *x, int64_t m) const {
        std::vector<double> y((size_t)m);
        check(egx_gp_predict(h_.get(), x, m, y.data()));
        return y;
    }
    std::vector<double> predict_var(const double *x, int64_t m) const {
        std::vector<double> v((size_t)m);
        check(egx_gp_predict_var(h_.get(), x, m, v.data()));
        return v;
    }
    std::pair<std::vector<double>, std::vector<double>> predict_valvar(const double *x, int64_t m) const {
        std::vector<double> y((size_t)m), v((size_t)m);
        check(egx_gp_predict_valvar(h_.get(), x, m, y.data(), v.data()));
        return {std::move(y), std::move(v)};
    }
    std::vector<double> predict_gradients(const double *x, int64_t m) const {  // (m x d) row-major
        std::vector<double> g((size_t)(m * d_));
        check(egx_gp_predict_gradients(h_.get(), x, m, g.data()));
        return g;
    }
    std::vector<double> predict_var_gradients(const double *x, int64_t m) const {
        std::vector<double> g((size_t)(m * d_));
        check(egx_gp_predict_var_gradients(h_.get(), x, m, g.data()));
        return g;
    }
    // algorithm.rs:711-727 -> the two (m x d) row-major gradients from one pass
    std::pair<std::vector<double>, std::vector<double>> predict_valvar_gradients(const double *x, int64_t m) const {
        std::vector<double> gy((size_t)(m * d_)), gv((size_t)(m * d_));
        check(egx_gp_predict_valvar_gradients(h_.get(), x, m, gy.data(), gv.data()));
        return {std::move(gy), std::move(gv)};
    }
    // reduced likelihood at k candidate thetas (k x theta_len row-major) -> (values, status per candidate): the evaluations
    // the reference's multistart closures make (algorithm.rs:880-897, 928-945), factored in lock-step groups on the GPU
    std::pair<std::vector<double>, std::vector<int32_t>> likelihood_batch(const double *thetas, int64_t k, int64_t theta_len) {
        std::vector<double> lk((size_t)k);
        std::vector<int32_t> st((size_t)k);
        check(egx_gp_likelihood_batch(h_.get(), thetas, k, theta_len, lk.data(), st.data()));
        return {std::move(lk), std::move(st)};
    }
    // the same with dL/dtheta (new: the reference's objective has no gradient, algorithm.rs:880): likelihoods (k), gradients
    // (k x h row-major), statuses (k); every stage in lock-step over a slot's candidates, bit for bit the single call
    struct LikelihoodGrad {
        std::vector<double> likelihood, gradient;
        std::vector<int32_t> status;
    };
    LikelihoodGrad likelihood_grad_batch(const double *thetas, int64_t k, int64_t theta_len) {
        int64_t h = 0;
        check(egx_gp_dims(h_.get(), nullptr, nullptr, nullptr, &h));
        LikelihoodGrad out;
        out.likelihood.resize((size_t)k);
        out.gradient.assign((size_t)(k * h), 0.0);
        out.status.resize((size_t)k);
        check(egx_gp_likelihood_grad_batch(h_.get(), thetas, k, theta_len, out.likelihood.data(), out.gradient.data(),
                                           out.status.data()));
        return out;
    }
    // a tuned model ran its multistart on several workspaces; the resident model needs one (plus one for likelihood
    // evaluations that must not un-fit it)
    void shrink(int32_t n_keep = 1) { check(egx_gp_shrink(h_.get(), n_keep)); }
    // { left-looking, left-looking rider, pipelined chain launches, whole-factorisation launch, panels per group, lock-step width,
    //   flow launch }
    std::array<int32_t, 7> schedule() const {
        std::array<int32_t, 7> s{};
        check(egx_gp_get_schedule(h_.get(), s.data(), 7));
        return s;
    }
    int32_t set_lockstep(int32_t width) {  // candidates per launch sequence (0 = the library's choice); returns the width in force
        check(egx_gp_set_lockstep(h_.get(), width));
        return egx_gp_get_lockstep(h_.get());
    }
    // GaussianProcess::training_data (algorithm.rs:969-978): the (n x d) inputs and (n) outputs the model was trained on
    std::pair<std::vector<double>, std::vector<double>> training_data(int64_t n) const {
        std::vector<double> x((size_t)(n * d_)), y((size_t)n);
        check(egx_gp_get_training_data(h_.get(), x.data(), y.data()));
        return {std::move(x), std::move(y)};
    }
    const std::vector<double> &theta() const { return theta_; }   // :413-416
    double variance() const { return sigma2_; }                   // :418-421
    double likelihood() const { return likelihood_; }             // :428-431
    std::pair<int64_t, int64_t> dims() const { return {d_, 1}; }  // :436-439
    int64_t n_evals() const { return n_evals_; }
    egx_gp *handle() const { return h_.get(); }

  private:
    friend class GpParams;
    struct Deleter {
        void operator()(egx_gp *p) const { egx_gp_destroy(p); }
    };
    std::unique_ptr<egx_gp, Deleter> h_;
    std::vector<double> theta_;
    double sigma2_ = 0.0, likelihood_ = 0.0;
    int64_t d_ = 0, n_evals_ = 0;
};

inline std::vector<GaussianProcess> GpParams::fit_group(const double *x, const double *y, int64_t n, int64_t d, int32_t k) const {
    if (tuning_.kind == ThetaTuning::Kind::Partial)
        throw InvalidValueError(EGX_ERR_INVALID_VALUE, "fit_group: ThetaTuning::Fixed or ::Full");
    egx_gp_config cfg;
    egx_gp_config_default(&cfg);
    cfg.corr = (int32_t)corr_;
    cfg.mean = (int32_t)mean_;
    cfg.nugget = nugget_;
    cfg.device = device_;
    std::vector<egx_gp *> raw((size_t)k, nullptr);
    check(egx_gp_create_group(&cfg, x, y, n, d, k, raw.data()));
    std::vector<GaussianProcess> out((size_t)k);
    for (int32_t j = 0; j < k; j++) {
        out[(size_t)j].h_.reset(raw[(size_t)j]);
        out[(size_t)j].d_ = d;
        out[(size_t)j].n_evals_ = 1;
    }
    int64_t hh = 0;
    check(egx_gp_dims(raw[0], nullptr, nullptr, nullptr, &hh));
    if (tuning_.init.size() != 1 && tuning_.init.size() != (size_t)hh)
        throw InvalidValueError(EGX_ERR_INVALID_VALUE, "Initial guess for theta should be either 1-dim or dim of xtrain, got " +
                                                           std::to_string(tuning_.init.size()));
    std::vector<double> thetas((size_t)k * (size_t)hh);
    for (size_t i = 0; i < thetas.size(); i++) thetas[i] = tuning_.init[tuning_.init.size() == 1 ? 0 : i % (size_t)hh];
    if (tuning_.kind == ThetaTuning::Kind::Fixed) {
        check(egx_gp_finalize_multi(raw.data(), k, thetas.data(), hh));
    } else {
        // ThetaTuning::Full for every member (round 6, egx_gp_fit_multi): the tuned fit the expert loop runs per cluster
        // (crates/moe/src/algorithm.rs:209-262 -> crates/gp/src/algorithm.rs:921-945), all models' COBYLA machines in lock-step;
        // the same starts for every model, as the reference's fits draw theirs from one fixed seed (optimization.rs:49-66)
        const size_t h = (size_t)hh;
        if (tuning_.bounds.size() != 1 && tuning_.bounds.size() != h)
            throw InvalidValueError(EGX_ERR_INVALID_VALUE, "Bounds for theta should be either 1-dim or dim of xtrain (" + std::to_string(h) +
                                                               "), got " + std::to_string(tuning_.bounds.size()));
        std::vector<double> lo(h), hi(h);
        for (size_t i = 0; i < h; i++) {
            const auto &b = tuning_.bounds[tuning_.bounds.size() == 1 ? 0 : i];
            lo[i] = b.first, hi[i] = b.second;
        }
        const size_t ns = (size_t)std::max(0, n_start_) + 1;
        std::vector<double> starts(ns * h);
        for (size_t i = 0; i < h; i++) starts[i] = thetas[i];
        std::mt19937_64 rng(seed_);  // (the draw of GpParams::fit: a fit_group member gets the starts its lone fit gets)
        std::uniform_real_distribution<double> u01(0.0, 1.0);
        for (size_t j = 0; j < h && ns > 1; j++) {
            std::vector<size_t> perm(ns - 1);
            for (size_t r = 0; r < ns - 1; r++) perm[r] = r;
            std::shuffle(perm.begin(), perm.end(), rng);
            for (size_t r = 0; r < ns - 1; r++) {
                const double u = ((double)perm[r] + u01(rng)) / (double)(ns - 1);
                starts[(r + 1) * h + j] = std::pow(10.0, std::log10(lo[j]) + u * (std::log10(hi[j]) - std::log10(lo[j])));
            }
        }
        std::vector<double> all((size_t)k * ns * h);
        for (int32_t j = 0; j < k; j++) std::copy(starts.begin(), starts.end(), all.begin() + (size_t)j * ns * h);
        std::vector<int64_t> ne((size_t)k, 0);
        check(egx_gp_fit_multi(raw.data(), k, all.data(), (int64_t)ns, lo.data(), hi.data(), (int64_t)h, max_eval_, ne.data()));
        for (int32_t j = 0; j < k; j++) out[(size_t)j].n_evals_ = ne[(size_t)j];
    }
    for (int32_t j = 0; j < k; j++) {  // theta() / variance() / likelihood() of every member, as `fit` leaves them
        GaussianProcess &gp = out[(size_t)j];
        gp.theta_.resize((size_t)hh);
        egx_gp_inner_view view{};
        view.theta = gp.theta_.data();
        view.sigma2 = &gp.sigma2_;
        view.likelihood = &gp.likelihood_;
        check(egx_gp_get_inner(raw[(size_t)j], &view));
    }
    return out;
}

inline GaussianProcess GpParams::fit(const double *x, int64_t n, int64_t d, const double *y) const {
    egx_gp_config cfg;
    egx_gp_config_default(&cfg);
    cfg.corr = (int32_t)corr_;
    cfg.mean = (int32_t)mean_;
    cfg.nugget = nugget_;
    cfg.device = device_;
    const bool tuned = tuning_.kind != ThetaTuning::Kind::Fixed;
    cfg.n_workspaces = tuned ? std::max(1, std::min(n_workspaces_, n_start_ + 1)) : std::max(1, n_workspaces_);
    egx_gp *raw = nullptr;
    check(egx_gp_create(&cfg, x, y, n, d, &raw));
    GaussianProcess gp;
    gp.h_.reset(raw);
    gp.d_ = d;
    int64_t hh = 0;
    check(egx_gp_dims(raw, nullptr, nullptr, nullptr, &hh));
    const size_t h = (size_t)hh;
    if (tuning_.init.size() != 1 && tuning_.init.size() != h)  // a panic in the reference, algorithm.rs:829-838
        throw InvalidValueError(EGX_ERR_INVALID_VALUE,
                                "Initial guess for theta should be either 1-dim or dim of xtrain (w_star.ncols()), got " +
                                    std::to_string(tuning_.init.size()));
    std::vector<double> theta0(h);
    for (size_t i = 0; i < h; i++) theta0[i] = tuning_.init[tuning_.init.size() == 1 ? 0 : i];
    if (!tuned) {
        check(egx_gp_finalize(raw, theta0.data(), (int64_t)h));
        gp.n_evals_ = 1;
    } else {
        if (tuning_.bounds.size() != 1 && tuning_.bounds.size() != h)  // algorithm.rs:901-912
            throw InvalidValueError(EGX_ERR_INVALID_VALUE, "Bounds for theta should be either 1-dim or dim of xtrain (" +
                                                               std::to_string(h) + "), got " +
                                                               std::to_string(tuning_.bounds.size()));
        std::vector<int64_t> active;
        if (tuning_.kind == ThetaTuning::Kind::Partial) active = tuning_.active;
        else for (size_t i = 0; i < h; i++) active.push_back((int64_t)i);
        std::sort(active.begin(), active.end());
        const size_t k = active.size();
        std::vector<double> lo(k), hi(k);
        for (size_t i = 0; i < k; i++) {
            const auto &b = tuning_.bounds[tuning_.bounds.size() == 1 ? 0 : (size_t)active[i]];
            lo[i] = b.first;
            hi[i] = b.second;
        }
        // prepare_multistart, optimization.rs:26-71: row 0 = the user's theta0, rows 1.. a Latin hypercube in log10
        // bounds (the reference's maximin-optimised, Xoshiro-seeded design is not reproducible here: parity-unpinned)
        const size_t ns = (size_t)std::max(0, n_start_) + 1;
        std::vector<double> starts(ns * k);
        for (size_t i = 0; i < k; i++) starts[i] = theta0[(size_t)active[i]];
        std::mt19937_64 rng(seed_);
        std::uniform_real_distribution<double> u01(0.0, 1.0);
        for (size_t j = 0; j < k && ns > 1; j++) {
            std::vector<size_t> perm(ns - 1);
            for (size_t r = 0; r < ns - 1; r++) perm[r] = r;
            std::shuffle(perm.begin(), perm.end(), rng);
            for (size_t r = 0; r < ns - 1; r++) {
                const double u = ((double)perm[r] + u01(rng)) / (double)(ns - 1);
                starts[(r + 1) * k + j] = std::pow(10.0, std::log10(lo[j]) + u * (std::log10(hi[j]) - std::log10(lo[j])));
            }
        }
        int64_t ne = 0;
        if (tuning_.kind == ThetaTuning::Kind::Partial)
            check(egx_gp_fit_partial(raw, theta0.data(), active.data(), (int64_t)k, starts.data(), (int64_t)ns, lo.data(),
                                     hi.data(), (int64_t)k, max_eval_, &ne));
        else
            check(egx_gp_fit(raw, starts.data(), (int64_t)ns, lo.data(), hi.data(), (int64_t)k, max_eval_, &ne));
        gp.n_evals_ = ne;
    }
    gp.theta_.resize(h);
    egx_gp_inner_view view{};
    view.theta = gp.theta_.data();
    view.sigma2 = &gp.sigma2_;
    view.likelihood = &gp.likelihood_;
    check(egx_gp_get_inner(raw, &view));
    return gp;
}

// GpMixture::predict / predict_var over fitted experts of THIS process (crates/moe/src/algorithm.rs:411-423, 670-685 smooth;
// :879-935 hard): probas (m x n_experts row-major) are the responsibilities, xq (m x d) in original units.
inline std::pair<std::vector<double>, std::vector<double>> moe_predict_valvar(const std::vector<const GaussianProcess *> &experts,
                                                                              const double *probas, const double *xq,
                                                                              int64_t m, int64_t d, bool smooth) {
    std::vector<egx_gp *> hs;
    std::vector<int32_t> ids;
    for (size_t e = 0; e < experts.size(); e++) {
        hs.push_back(experts[e]->handle());
        ids.push_back((int32_t)e);
    }
    std::vector<double> val((size_t)m), var((size_t)m);
    check(egx_moe_predict_valvar(nullptr, hs.data(), ids.data(), (int64_t)hs.size(), (int64_t)hs.size(), probas, xq, m, d,
                                 smooth ? 1 : 0, val.data(), var.data()));
    return {std::move(val), std::move(var)};
}

// GpMixture::predict_gradients / predict_var_gradients (crates/moe/src/algorithm.rs:691-783 smooth, :942-1010 hard) over
// fitted experts of THIS process: dprobas (m x n_experts x d, GaussianMixture::predict_probas_derivatives) is read by the
// smooth recombination only.  Returns (d val / dx, d var / dx), each (m x d) row-major.
inline std::pair<std::vector<double>, std::vector<double>> moe_predict_valvar_gradients(
    const std::vector<const GaussianProcess *> &experts, const double *probas, const double *dprobas, const double *xq, int64_t m,
    int64_t d, bool smooth) {
    std::vector<egx_gp *> hs;
    std::vector<int32_t> ids;
    for (size_t e = 0; e < experts.size(); e++) {
        hs.push_back(experts[e]->handle());
        ids.push_back((int32_t)e);
    }
    std::vector<double> gy((size_t)(m * d)), gv((size_t)(m * d));
    check(egx_moe_predict_valvar_gradients(nullptr, hs.data(), ids.data(), (int64_t)hs.size(), (int64_t)hs.size(), probas, dprobas,
                                           xq, m, d, smooth ? 1 : 0, gy.data(), gv.data()));
    return {std::move(gy), std::move(gv)};
}

// device resources destroyed models left in the library's pool (a model of the same shape created next reuses them)
inline int64_t trim() { return egx_trim(); }  // bytes freed
struct PoolStats { int64_t cached_bytes = 0, hits = 0, misses = 0; };
inline PoolStats pool_stats() {
    PoolStats s;
    egx_pool_stats(&s.cached_bytes, &s.hits, &s.misses);
    return s;
}

// chain launches that ran into their wait bound, and how many of those evaluations were run again by separate launches
struct ChainStats { int64_t aborted = 0, retried = 0; };
inline ChainStats chain_stats() {
    ChainStats s;
    egx_chain_stats(&s.aborted, &s.retried);
    return s;
}

// Kriging = constant mean + squared exponential, algorithm.rs:244-249
struct Kriging {
    static GpParams params() { return GpParams(Mean::Constant, Corr::SquaredExponential); }
};

}  // namespace egobox
#endif  // EGX_GP_HPP
