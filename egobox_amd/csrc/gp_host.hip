// C ABI implementation (include/egx_gp.h): handle lifetime, the likelihood / fit / predict
// drivers and the small host-side GLS algebra.  Mirrors, operation for operation,
// GpValidParams::fit and reduced_likelihood (crates/gp/src/algorithm.rs:785-1056) and
// GaussianProcess::predict* (:253-380) of the reference; the O(n^2 d) / O(n^3) / O(n^2 m) parts run
// in the HIP kernels of kernels_corr.hip / kernels_chol.hip.  There is no CPU fallback.
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <condition_variable>
#include <mutex>
#include <shared_mutex>
#include <thread>
#include <algorithm>
#include <vector>

#include "egx_internal.h"
#include "host_math.h"
#include "nelder_mead.h"

namespace egx {

static thread_local std::string g_last_error;
void set_error(const std::string &msg) { g_last_error = msg; }

#define EGX_RC(call)              \
    do {                          \
        int _rc = (call);         \
        if (_rc) return _rc;      \
    } while (0)

// Scoped device allocation for the temporaries of one call (freed on every exit path).
struct DevBuf {
    double *p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    int alloc(size_t n_doubles) {
        EGX_HIP_CHECK(hipMalloc(&p, sizeof(double) * (n_doubles ? n_doubles : 1)));
        return EGX_SUCCESS;
    }
};

struct Workspace {
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;  // look-ahead (panel) stream
    hipEvent_t ev_lu = nullptr, ev_panel = nullptr;
    double *M = nullptr;       // (m_tot x ld): correlation matrix / factor + appended RHS rows
    double *dinv = nullptr;    // (n_pad/64) x 64 x 64 inverses of the diagonal tiles
    double *dW = nullptr;      // (n_pad/256) x 256 x 256 transposed inverses of the diagonal blocks (lazy)
    double *d_coef = nullptr;  // d x hcols
    double *d_diag = nullptr;  // n
    double *d_vec = nullptr;   // n_pad (gamma)
    double *d_rhs = nullptr;   // n_pad (rho, destroyed by the back-substitution)
    int *d_info = nullptr;
    double *h_coef = nullptr;  // pinned
    double *h_rows = nullptr;  // pinned: q x n_pad solved RHS rows (ft^T, yt^T)
    double *h_diag = nullptr;  // pinned: n
    double *h_vec = nullptr;   // pinned: n_pad
    int *h_info = nullptr;     // pinned
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    GemmTrace trace;
};

struct EvalResult {
    double lkh = -std::numeric_limits<double>::infinity();
    int status = EGX_STATUS_OK;
    double sigma2n = 0.0;         // rho^2 / n in normalised units
    std::vector<double> beta;     // p
    std::vector<double> rho;      // n
    std::vector<double> ft;       // n x p row-major
    std::vector<double> ft_qr_r;  // p x p row-major
};

}  // namespace egx

using namespace egx;

struct egx_gp {
    int device = 0;
    int n = 0, d = 0, p = 0, h = 0, corr = 0, mean = 0;
    double nugget = 0.0;
    int n_pad = 0, rhs_pad = 0, m_tot = 0, q = 0;
    int64_t ld = 0;
    bool has_w = false;
    std::vector<double> w_star;  // d x h
    std::vector<double> x_raw, y_raw, xnorm, x_mean, x_std, ynorm, F;
    double y_mean = 0.0, y_std = 1.0;
    double *d_xT = nullptr;    // d x n_pad (k-major normalised inputs, zero padded)
    double *d_rhsT = nullptr;  // q x n_pad: columns of F then y (normalised), as rows
    std::vector<Workspace> ws;
    // exclusive for everything that touches the fitted state or all workspaces; SHARED for egx_gp_likelihood, whose
    // concurrent callers (the reference's rayon multistart closures, algorithm.rs:928-945) each take a workspace
    // from the pool below
    std::shared_mutex mu;
    std::mutex pool_mu;
    std::condition_variable pool_cv;
    std::vector<char> ws_busy;
    // fitted state (lives in ws[0])
    bool fitted = false;
    std::vector<double> theta;  // h
    double likelihood = 0.0, sigma2 = 0.0;
    std::vector<double> beta, gamma, ft, ft_qr_r;
    std::vector<double> fit_coef;
    int fit_hcols = 1;
    double *d_gamma = nullptr;  // n_pad
    double *d_fit_coef = nullptr;
    // gradient scratch (allocated on first use)
    double *d_W = nullptr, *d_Rinv = nullptr, *d_gout = nullptr, *d_theta = nullptr;
    // x-gradient state (lazy, per fitted factor): d_W = C^-T (shared with the theta-gradient scratch) and
    // -R^-1 F = -C^-T ft as an (n_pad x rhs_pad) matrix
    double *d_neg_invkf = nullptr;
    std::vector<double> h_neg_invkf;  // host copy (n x p) for the single-point path
    // device scratch of the single-point path, allocated once (a hipMalloc per call would cost more than the kernels)
    double *sp_R = nullptr, *sp_P = nullptr, *sp_y = nullptr, *sp_z = nullptr, *sp_wt = nullptr, *sp_out = nullptr,
           *sp_xq = nullptr;
    int sp_nsplit = 0;
    int small_var_calls = 0;  // single-point predict_var calls since the fit: the third one builds W = C^-T
    uint64_t fit_epoch = 0, winv_epoch = ~(uint64_t)0;
    egx_timings timings{};
};

namespace egx {

static int set_device(const egx_gp *gp) {
    EGX_HIP_CHECK(hipSetDevice(gp->device));
    return EGX_SUCCESS;
}

static void free_workspace(Workspace &w) {
    if (w.M) hipFree(w.M);
    if (w.dinv) hipFree(w.dinv);
    if (w.dW) hipFree(w.dW);
    if (w.d_coef) hipFree(w.d_coef);
    if (w.d_diag) hipFree(w.d_diag);
    if (w.d_vec) hipFree(w.d_vec);
    if (w.d_rhs) hipFree(w.d_rhs);
    if (w.d_info) hipFree(w.d_info);
    if (w.h_coef) hipHostFree(w.h_coef);
    if (w.h_rows) hipHostFree(w.h_rows);
    if (w.h_diag) hipHostFree(w.h_diag);
    if (w.h_vec) hipHostFree(w.h_vec);
    if (w.h_info) hipHostFree(w.h_info);
    for (auto &e : w.ev)
        if (e) hipEventDestroy(e);
    if (w.trace.ready)
        for (int i = 0; i < GemmTrace::kMax; i++) {
            hipEventDestroy(w.trace.e0[i]);
            hipEventDestroy(w.trace.e1[i]);
        }
    if (w.ev_lu) hipEventDestroy(w.ev_lu);
    if (w.ev_panel) hipEventDestroy(w.ev_panel);
    if (w.stream2) hipStreamDestroy(w.stream2);
    if (w.stream) hipStreamDestroy(w.stream);
    w = Workspace();
}

static int alloc_workspace(egx_gp *gp, Workspace &w) {
    EGX_HIP_CHECK(hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
    {
        const char *la = std::getenv("EGX_LOOKAHEAD");
        if (!la || la[0] != '0') {
            int lo = 0, hi = 0;
            EGX_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
            EGX_HIP_CHECK(hipStreamCreateWithPriority(&w.stream2, hipStreamNonBlocking, hi));
            EGX_HIP_CHECK(hipEventCreateWithFlags(&w.ev_lu, hipEventDisableTiming));
            EGX_HIP_CHECK(hipEventCreateWithFlags(&w.ev_panel, hipEventDisableTiming));
        }
    }
    EGX_HIP_CHECK(hipMalloc(&w.M, sizeof(double) * (size_t)gp->m_tot * gp->ld));
    EGX_HIP_CHECK(hipMalloc(&w.dinv, sizeof(double) * (size_t)(gp->n_pad / 64) * 4096));
    const int hmax = gp->has_w ? gp->h : 1;
    EGX_HIP_CHECK(hipMalloc(&w.d_coef, sizeof(double) * (size_t)gp->d * hmax));
    EGX_HIP_CHECK(hipMalloc(&w.d_diag, sizeof(double) * (size_t)gp->n_pad));
    EGX_HIP_CHECK(hipMalloc(&w.d_vec, sizeof(double) * (size_t)gp->n_pad));
    EGX_HIP_CHECK(hipMalloc(&w.d_rhs, sizeof(double) * (size_t)gp->n_pad));
    EGX_HIP_CHECK(hipMalloc(&w.d_info, sizeof(int)));
    EGX_HIP_CHECK(hipHostMalloc(&w.h_coef, sizeof(double) * (size_t)gp->d * hmax, hipHostMallocDefault));
    EGX_HIP_CHECK(hipHostMalloc(&w.h_rows, sizeof(double) * (size_t)gp->q * gp->n_pad, hipHostMallocDefault));
    EGX_HIP_CHECK(hipHostMalloc(&w.h_diag, sizeof(double) * (size_t)gp->n_pad, hipHostMallocDefault));
    EGX_HIP_CHECK(hipHostMalloc(&w.h_vec, sizeof(double) * (size_t)gp->n_pad, hipHostMallocDefault));
    EGX_HIP_CHECK(hipHostMalloc(&w.h_info, sizeof(int), hipHostMallocDefault));
    for (auto &e : w.ev) EGX_HIP_CHECK(hipEventCreate(&e));
    for (int i = 0; i < GemmTrace::kMax; i++) {
        EGX_HIP_CHECK(hipEventCreate(&w.trace.e0[i]));
        EGX_HIP_CHECK(hipEventCreate(&w.trace.e1[i]));
    }
    w.trace.ready = true;
    return EGX_SUCCESS;
}

// theta (len 1 or h) -> per-dimension coefficient table (d x hcols), see kernels_corr.hip.
// correlation_models.rs:97-98 (sq-exp theta_w), :191 (abs-exp), :333 / :505 (Matern theta_w).
static int make_coef(const egx_gp *gp, const double *theta, int64_t theta_len, std::vector<double> &coef,
                     int &hcols, std::vector<double> *theta_full) {
    if (theta_len != 1 && theta_len != gp->h) {
        set_error("theta should be either 1-dim or dim of xtrain (w_star.ncols()), got " + std::to_string(theta_len));
        return EGX_ERR_INVALID_VALUE;
    }
    std::vector<double> th(gp->h);
    for (int l = 0; l < gp->h; l++) th[l] = theta[theta_len == 1 ? 0 : l];
    if (theta_full) *theta_full = th;
    const int d = gp->d, h = gp->h;
    if (!gp->has_w) {
        hcols = 1;
        coef.assign(th.begin(), th.end());  // h == d
        return EGX_SUCCESS;
    }
    const double *w = gp->w_star.data();
    if (gp->corr == EGX_CORR_SQUARED_EXPONENTIAL) {
        hcols = 1;
        coef.assign(d, 0.0);
        for (int j = 0; j < d; j++) {
            double s = 0.0;
            for (int l = 0; l < h; l++) s += (th[l] * w[j * h + l]) * (th[l] * w[j * h + l]);
            coef[j] = std::sqrt(s);
        }
    } else if (gp->corr == EGX_CORR_ABSOLUTE_EXPONENTIAL) {
        hcols = 1;
        coef.assign(d, 0.0);
        for (int j = 0; j < d; j++) {
            double s = 0.0;
            for (int l = 0; l < h; l++) s += std::fabs(w[j * h + l]) * th[l];
            coef[j] = s;
        }
    } else {
        hcols = h;
        coef.assign((size_t)d * h, 0.0);
        for (int j = 0; j < d; j++)
            for (int l = 0; l < h; l++) coef[j * h + l] = th[l] * std::fabs(w[j * h + l]);
    }
    return EGX_SUCCESS;
}

// GPU half of one likelihood evaluation: R assembly + RHS rows, factorisation with fused forward
// solves, diagonal gather, async download of (diag C, ft^T, yt^T, info).  All asynchronous on w.stream.
static int enqueue_eval(egx_gp *gp, Workspace &w, const std::vector<double> &coef, int hcols) {
    std::memcpy(w.h_coef, coef.data(), sizeof(double) * coef.size());
    EGX_HIP_CHECK(hipMemcpyAsync(w.d_coef, w.h_coef, sizeof(double) * coef.size(), hipMemcpyHostToDevice, w.stream));
    EGX_HIP_CHECK(hipMemsetAsync(w.d_info, 0, sizeof(int), w.stream));
    EGX_HIP_CHECK(hipEventRecord(w.ev[0], w.stream));
    EGX_RC(launch_corr_sym(w.stream, gp->corr, gp->d_xT, gp->n_pad, gp->n, gp->d, w.d_coef, hcols, gp->nugget, w.M,
                           gp->ld, gp->n_pad));
    EGX_RC(launch_fill_rows(w.stream, w.M, gp->ld, gp->n_pad, gp->rhs_pad, gp->d_rhsT, gp->n_pad, gp->q, gp->n_pad));
    EGX_HIP_CHECK(hipEventRecord(w.ev[1], w.stream));
    EGX_RC(launch_potrf(w.stream, w.M, gp->ld, gp->n_pad, gp->m_tot, w.dinv, w.d_info, w.stream2, w.ev_lu,
                        w.ev_panel, &w.trace));
    EGX_HIP_CHECK(hipEventRecord(w.ev[2], w.stream));
    EGX_RC(launch_gather_diag(w.stream, w.M, gp->ld, gp->n, w.d_diag));
    EGX_HIP_CHECK(hipMemcpyAsync(w.h_diag, w.d_diag, sizeof(double) * gp->n, hipMemcpyDeviceToHost, w.stream));
    EGX_HIP_CHECK(hipMemcpyAsync(w.h_rows, w.M + (size_t)gp->n_pad * gp->ld, sizeof(double) * (size_t)gp->q * gp->n_pad,
                                 hipMemcpyDeviceToHost, w.stream));
    EGX_HIP_CHECK(hipMemcpyAsync(w.h_info, w.d_info, sizeof(int), hipMemcpyDeviceToHost, w.stream));
    EGX_HIP_CHECK(hipEventRecord(w.ev[3], w.stream));
    return EGX_SUCCESS;
}

// Host half: algorithm.rs:1007-1043 on the downloaded ft, yt, diag(C).
static int finish_eval(egx_gp *gp, Workspace &w, EvalResult &out, bool keep) {
    EGX_HIP_CHECK(hipStreamSynchronize(w.stream));
    const int n = gp->n, p = gp->p, n_pad = gp->n_pad;
    out = EvalResult();
    if (*w.h_info != 0) {
        out.status = EGX_STATUS_NOT_POSITIVE_DEFINITE;
        return EGX_SUCCESS;
    }
    for (int i = 0; i < n; i++)
        if (!(w.h_diag[i] > 0.0) || !std::isfinite(w.h_diag[i])) {
            out.status = EGX_STATUS_NOT_POSITIVE_DEFINITE;
            return EGX_SUCCESS;
        }
    // ft (column-major copy for the QR), yt
    std::vector<double> a((size_t)n * p), qty(n);
    for (int l = 0; l < p; l++) std::memcpy(&a[(size_t)l * n], w.h_rows + (size_t)l * n_pad, sizeof(double) * n);
    std::memcpy(qty.data(), w.h_rows + (size_t)p * n_pad, sizeof(double) * n);
    std::vector<double> yt(qty);
    std::vector<double> ftcm;
    ftcm = a;                     // keep ft (column-major) for rho
    hm::qr_apply(a, n, p, &qty);  // :1007  a -> R (upper), qty -> Q^T yt
    std::vector<double> R((size_t)p * p, 0.0);
    for (int i = 0; i < p; i++)
        for (int l = i; l < p; l++) R[(size_t)i * p + l] = (i < n) ? a[(size_t)l * n + i] : 0.0;
    // :1010-1027 conditioning of the p x p factor
    std::vector<double> sv = hm::singular_values(R, p);
    const double cond_ft = sv[p - 1] / sv[0];
    if (!(cond_ft >= 1e-10)) {
        std::vector<double> fcm((size_t)n * p);
        for (int l = 0; l < p; l++)
            for (int i = 0; i < n; i++) fcm[(size_t)l * n + i] = gp->F[(size_t)i * p + l];
        hm::qr_apply(fcm, n, p, nullptr);
        std::vector<double> RF((size_t)p * p, 0.0);
        for (int i = 0; i < p; i++)
            for (int l = i; l < p; l++) RF[(size_t)i * p + l] = (i < n) ? fcm[(size_t)l * n + i] : 0.0;
        std::vector<double> svf = hm::singular_values(RF, p);
        out.status = (svf[0] / svf[p - 1] > 1e15) ? EGX_STATUS_ILL_CONDITIONED_F : EGX_STATUS_ILL_CONDITIONED_FT;
        return EGX_SUCCESS;
    }
    // :1030 beta = R^-1 (Q^T yt)[:p]
    std::vector<double> beta(p);
    for (int i = p - 1; i >= 0; i--) {
        double s = qty[i];
        for (int l = i + 1; l < p; l++) s -= R[(size_t)i * p + l] * beta[l];
        beta[i] = s / R[(size_t)i * p + i];
    }
    // :1031-1032 rho = yt - ft beta
    std::vector<double> rho(yt);
    for (int l = 0; l < p; l++) {
        const double b = beta[l];
        const double *col = &ftcm[(size_t)l * n];
        for (int i = 0; i < n; i++) rho[i] -= col[i] * b;
    }
    double rho_sqr = 0.0;
    for (int i = 0; i < n; i++) rho_sqr += rho[i] * rho[i];
    // :1039-1043
    double slog = 0.0;
    for (int i = 0; i < n; i++) slog += std::log10(w.h_diag[i]);
    const double logdet = slog * 2.0 / (double)n;
    const double sigma2n = rho_sqr / (double)n;
    out.lkh = -(double)n * (std::log10(sigma2n) + logdet);
    out.sigma2n = sigma2n;
    out.status = EGX_STATUS_OK;
    if (keep) {
        out.beta = beta;
        out.rho = rho;
        out.ft_qr_r = R;
        out.ft.assign((size_t)n * p, 0.0);
        for (int l = 0; l < p; l++)
            for (int i = 0; i < n; i++) out.ft[(size_t)i * p + l] = ftcm[(size_t)l * n + i];
    }
    return EGX_SUCCESS;
}

static void record_timings(egx_gp *gp, Workspace &w, double host_ms, double solve_ms) {
    float t01 = 0, t12 = 0, t03 = 0;
    hipEventElapsedTime(&t01, w.ev[0], w.ev[1]);
    hipEventElapsedTime(&t12, w.ev[1], w.ev[2]);
    hipEventElapsedTime(&t03, w.ev[0], w.ev[3]);
    egx_timings &t = gp->timings;
    t.corr_build_ms = t01;
    t.potrf_ms = t12;
    t.potrf_syrk_ms = 0.0;
    t.syrk_launches = 0;
    t.syrk_flops = 0;
    double fl = 0.0;
    for (int i = 0; i < w.trace.used; i++) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, w.trace.e0[i], w.trace.e1[i]) == hipSuccess) {
            t.potrf_syrk_ms += ms;
            fl += w.trace.flops[i];
            t.syrk_launches++;
        }
    }
    t.syrk_flops = (int64_t)fl;
    t.solve_ms = solve_ms;
    t.host_ms = host_ms;
    t.total_ms = t03 + host_ms + solve_ms;
    const double nn = (double)gp->n;
    t.potrf_flops = (int64_t)(nn * nn * nn / 3.0);
    t.corr_bytes = (int64_t)(8.0 * nn * gp->d + 8.0 * nn * (nn + 1.0) / 2.0);
}

static bool has_nan(const double *theta, int64_t len) {
    for (int64_t i = 0; i < len; i++)
        if (std::isnan(theta[i])) return true;
    return false;
}

static int eval_one(egx_gp *gp, int widx, const double *theta, int64_t theta_len, EvalResult &res, bool keep) {
    std::vector<double> coef;
    int hcols = 1;
    EGX_RC(make_coef(gp, theta, theta_len, coef, hcols, nullptr));
    if (has_nan(theta, theta_len)) {  // algorithm.rs:885-891
        res = EvalResult();
        res.status = EGX_STATUS_NAN_THETA;
        return EGX_SUCCESS;
    }
    if (widx == 0) gp->fitted = false;
    Workspace &w = gp->ws[widx];
    EGX_RC(enqueue_eval(gp, w, coef, hcols));
    auto t0 = std::chrono::steady_clock::now();
    EGX_RC(finish_eval(gp, w, res, keep));
    // host_ms includes the wait for the stream; subtract GPU time below
    auto t1 = std::chrono::steady_clock::now();
    float gpu = 0;
    hipEventElapsedTime(&gpu, w.ev[0], w.ev[3]);
    double host_ms = std::chrono::duration<double, std::milli>(t1 - t0).count() - gpu;
    if (host_ms < 0) host_ms = 0;
    if (widx == 0) record_timings(gp, w, host_ms, 0.0);
    return EGX_SUCCESS;
}

// w.d_vec <- C^-T w.d_rhs  (block inverses are rebuilt: the factor in w.M has just changed)
static int backward_solve(egx_gp *gp, Workspace &w) {
    if (!w.dW)
        EGX_HIP_CHECK(hipMalloc(&w.dW, sizeof(double) * (size_t)((gp->n_pad + kNB - 1) / kNB) * 65536));
    EGX_RC(launch_block_inverse(w.stream, w.M, gp->ld, gp->n_pad, w.dinv, w.dW));
    EGX_RC(launch_trsv_t(w.stream, w.M, gp->ld, gp->n_pad, w.dW, w.d_rhs, w.d_vec));
    return EGX_SUCCESS;
}

static int do_finalize(egx_gp *gp, const double *theta, int64_t theta_len) {
    std::vector<double> coef, thfull;
    int hcols = 1;
    EGX_RC(make_coef(gp, theta, theta_len, coef, hcols, &thfull));
    if (has_nan(theta, theta_len)) {
        set_error("theta contains NaN");
        return EGX_ERR_INVALID_VALUE;
    }
    gp->fitted = false;
    Workspace &w = gp->ws[0];
    EvalResult res;
    EGX_RC(enqueue_eval(gp, w, coef, hcols));
    auto t0 = std::chrono::steady_clock::now();
    EGX_RC(finish_eval(gp, w, res, true));
    auto t1 = std::chrono::steady_clock::now();
    if (res.status == EGX_STATUS_NOT_POSITIVE_DEFINITE) {
        set_error("LinalgError: matrix is not positive definite (pivot " + std::to_string(*w.h_info) + ")");
        return EGX_ERR_LINALG;
    }
    if (res.status == EGX_STATUS_ILL_CONDITIONED_F) {
        set_error("LikelihoodComputation computation error: F is too ill conditioned. Poor combination of "
                  "regression model and observations.");
        return EGX_ERR_LIKELIHOOD;
    }
    if (res.status == EGX_STATUS_ILL_CONDITIONED_FT) {
        set_error("LikelihoodComputation computation error: ft is too ill conditioned, try another theta again");
        return EGX_ERR_LIKELIHOOD;
    }
    // gamma = C^-T rho   (algorithm.rs:1034)
    const int n = gp->n, n_pad = gp->n_pad;
    std::memset(w.h_vec, 0, sizeof(double) * n_pad);
    std::memcpy(w.h_vec, res.rho.data(), sizeof(double) * n);
    EGX_HIP_CHECK(hipMemcpyAsync(w.d_rhs, w.h_vec, sizeof(double) * n_pad, hipMemcpyHostToDevice, w.stream));
    EGX_HIP_CHECK(hipEventRecord(w.ev[4], w.stream));
    EGX_RC(backward_solve(gp, w));
    EGX_HIP_CHECK(hipMemcpyAsync(gp->d_gamma, w.d_vec, sizeof(double) * n_pad, hipMemcpyDeviceToDevice, w.stream));
    EGX_HIP_CHECK(hipMemcpyAsync(w.h_vec, w.d_vec, sizeof(double) * n_pad, hipMemcpyDeviceToHost, w.stream));
    EGX_HIP_CHECK(hipMemcpyAsync(gp->d_fit_coef, w.d_coef, sizeof(double) * coef.size(), hipMemcpyDeviceToDevice,
                                 w.stream));
    EGX_HIP_CHECK(hipStreamSynchronize(w.stream));
    auto t2 = std::chrono::steady_clock::now();
    gp->gamma.assign(w.h_vec, w.h_vec + n);
    gp->theta = thfull;
    gp->likelihood = res.lkh;
    gp->sigma2 = res.sigma2n * gp->y_std * gp->y_std;  // algorithm.rs:1048
    gp->beta = res.beta;
    gp->ft = res.ft;
    gp->ft_qr_r = res.ft_qr_r;
    gp->fit_coef = coef;
    gp->fit_hcols = hcols;
    gp->fitted = true;
    gp->fit_epoch++;
    gp->small_var_calls = 0;
    float gpu = 0;
    hipEventElapsedTime(&gpu, w.ev[0], w.ev[3]);
    double host_ms = std::chrono::duration<double, std::milli>(t1 - t0).count() - gpu;
    if (host_ms < 0) host_ms = 0;
    record_timings(gp, w, host_ms, std::chrono::duration<double, std::milli>(t2 - t1).count());
    return EGX_SUCCESS;
}

// ---- prediction -------------------------------------------------------------------------------
// Normalise a chunk of query points (algorithm.rs:254) and upload it k-major (d x m_pad, zero padded).
static int upload_queries(egx_gp *gp, const double *xq, int64_t m0, int m, int m_pad, std::vector<double> &xn,
                          DevBuf &d_xqT, hipStream_t s) {
    const int d = gp->d;
    xn.resize((size_t)m * d);
    std::vector<double> xt((size_t)d * m_pad, 0.0);
    for (int a = 0; a < m; a++)
        for (int j = 0; j < d; j++) {
            const double v = (xq[(size_t)(m0 + a) * d + j] - gp->x_mean[j]) / gp->x_std[j];
            xn[(size_t)a * d + j] = v;
            xt[(size_t)j * m_pad + a] = v;
        }
    EGX_RC(d_xqT.alloc(xt.size()));
    EGX_HIP_CHECK(hipMemcpyAsync(d_xqT.p, xt.data(), sizeof(double) * xt.size(), hipMemcpyHostToDevice, s));
    EGX_HIP_CHECK(hipStreamSynchronize(s));  // xt is a pageable buffer owned by this frame
    return EGX_SUCCESS;
}

static int predict_var_small(egx_gp *gp, const double *xq, int64_t m, double *vout);

static int predict_impl(egx_gp *gp, const double *xq, int64_t m, double *yout, double *vout) {
    if (!gp->fitted) {
        set_error("model is not fitted (call egx_gp_finalize or egx_gp_fit first)");
        return EGX_ERR_NOT_FITTED;
    }
    if (m < 0 || (m > 0 && !xq)) {
        set_error("bad query array");
        return EGX_ERR_INVALID_VALUE;
    }
    EGX_RC(set_device(gp));
    if (vout && !yout && m > 0 && m <= 8) {  // a few points at a time: EGO's inner loop
        const bool have_w = gp->winv_epoch == gp->fit_epoch && gp->d_W && gp->d_neg_invkf;
        if (have_w || ++gp->small_var_calls >= 3) return predict_var_small(gp, xq, m, vout);
    }
    Workspace &w = gp->ws[0];
    const int n = gp->n, n_pad = gp->n_pad, d = gp->d, p = gp->p;
    // chunk so that the (m_tile x n_pad) block of predict_var stays <= 1 GiB
    int64_t cap = ((int64_t)1 << 27) / n_pad / kTile * kTile;
    if (cap < kTile) cap = kTile;
    if (cap > 16384) cap = 16384;
    if (!vout) cap = 65536;
    std::vector<double> f(p), rhs(p), u(p), xn, racc, s0, sl;
    for (int64_t m0 = 0; m0 < m; m0 += cap) {
        const int mc = (int)((m - m0 < cap) ? (m - m0) : cap);
        const int m_pad = (int)round_up(mc, kTile);
        DevBuf d_xqT, d_racc, d_RT, d_s0, d_sl;
        EGX_RC(upload_queries(gp, xq, m0, mc, m_pad, xn, d_xqT, w.stream));
        // few queries: split the training range so that ~1024 workgroups exist (partial sums added below)
        int msplit = 1;
        if (m_pad / 64 < 1024) msplit = (1024 + m_pad / 64 - 1) / (m_pad / 64);
        if (msplit > n_pad / 64) msplit = n_pad / 64;
        {
            const int per = (n_pad / 64 + msplit - 1) / msplit;
            msplit = (n_pad / 64 + per - 1) / per;
        }
        if (yout) {
            racc.resize((size_t)msplit * m_pad);
            EGX_RC(d_racc.alloc((size_t)msplit * m_pad));
            EGX_RC(launch_predict_mean(w.stream, gp->corr, d_xqT.p, m_pad, m_pad, gp->d_xT, n_pad, n_pad, d,
                                       gp->d_fit_coef, gp->fit_hcols, gp->d_gamma, d_racc.p, msplit));
            EGX_HIP_CHECK(hipMemcpyAsync(racc.data(), d_racc.p, sizeof(double) * (size_t)msplit * m_pad,
                                         hipMemcpyDeviceToHost, w.stream));
        }
        if (vout) {
            s0.resize(m_pad);
            sl.resize((size_t)m_pad * p);
            EGX_RC(d_RT.alloc((size_t)m_pad * n_pad));
            EGX_RC(d_s0.alloc(m_pad));
            EGX_RC(d_sl.alloc((size_t)m_pad * p));
            // corr (m x n): algorithm.rs:372-380 ; rt = C^-1 corr^T: :337-350 (held transposed, row per query)
            EGX_RC(launch_cross_corr(w.stream, gp->corr, d_xqT.p, m_pad, m_pad, gp->d_xT, n_pad, n_pad, d,
                                     gp->d_fit_coef, gp->fit_hcols, d_RT.p, n_pad));
            EGX_RC(launch_trsm_rows(w.stream, w.M, gp->ld, n_pad, w.dinv, d_RT.p, n_pad, m_pad));
            // sum rt^2 and ft^T rt (:352): ft^T rows live below the factor in the workspace
            EGX_RC(launch_row_reduce(w.stream, d_RT.p, n_pad, m_pad, n, w.M + (size_t)n_pad * gp->ld, gp->ld, p,
                                     d_s0.p, d_sl.p));
            EGX_HIP_CHECK(hipMemcpyAsync(s0.data(), d_s0.p, sizeof(double) * m_pad, hipMemcpyDeviceToHost, w.stream));
            EGX_HIP_CHECK(hipMemcpyAsync(sl.data(), d_sl.p, sizeof(double) * (size_t)m_pad * p, hipMemcpyDeviceToHost,
                                         w.stream));
        }
        EGX_HIP_CHECK(hipStreamSynchronize(w.stream));
        for (int a = 0; a < mc; a++) {
            hm::regression_row(gp->mean, &xn[(size_t)a * d], d, f.data());
            if (yout) {
                double fb = 0.0, rg = 0.0;
                for (int l = 0; l < p; l++) fb += f[l] * gp->beta[l];
                for (int sp = 0; sp < msplit; sp++) rg += racc[(size_t)sp * m_pad + a];
                yout[m0 + a] = (fb + rg) * gp->y_std + gp->y_mean;  // algorithm.rs:260-262
            }
            if (vout) {
                // u = (Rq^T)^-1 (ft^T rt - f^T)   algorithm.rs:352-367 ; Rq^T lower triangular
                for (int l = 0; l < p; l++) rhs[l] = sl[(size_t)a * p + l] - f[l];
                double usq = 0.0;
                for (int i = 0; i < p; i++) {
                    double sacc = rhs[i];
                    for (int l = 0; l < i; l++) sacc -= gp->ft_qr_r[(size_t)l * p + i] * u[l];
                    u[i] = sacc / gp->ft_qr_r[(size_t)i * p + i];
                    usq += u[i] * u[i];
                }
                double mse = gp->sigma2 * (1.0 - s0[a] + usq);  // algorithm.rs:272-274
                vout[m0 + a] = (mse < 0.0) ? 0.0 : mse;         // :278
            }
        }
    }
    return EGX_SUCCESS;
}

// d_W <- C^-T (upper triangular, rows of the identity through the forward block substitution) and
// d_neg_invkf <- -C^-T [ft | yt] for the factor resident in workspace 0; cached per fitted state.
static int ensure_winv(egx_gp *gp) {
    if (gp->winv_epoch == gp->fit_epoch && gp->d_W && gp->d_neg_invkf) return EGX_SUCCESS;
    Workspace &w = gp->ws[0];
    const int n_pad = gp->n_pad;
    const size_t sq = (size_t)n_pad * n_pad;
    if (!gp->d_W) EGX_HIP_CHECK(hipMalloc(&gp->d_W, sizeof(double) * sq));
    if (!gp->d_neg_invkf) EGX_HIP_CHECK(hipMalloc(&gp->d_neg_invkf, sizeof(double) * (size_t)n_pad * gp->rhs_pad));
    EGX_HIP_CHECK(hipMemsetAsync(gp->d_W, 0, sizeof(double) * sq, w.stream));
    {
        std::vector<double> ones(n_pad, 1.0);
        EGX_HIP_CHECK(hipMemcpy2DAsync(gp->d_W, sizeof(double) * (n_pad + 1), ones.data(), sizeof(double),
                                       sizeof(double), n_pad, hipMemcpyHostToDevice, w.stream));
        EGX_HIP_CHECK(hipStreamSynchronize(w.stream));
    }
    EGX_RC(launch_trsm_rows(w.stream, w.M, gp->ld, n_pad, w.dinv, gp->d_W, n_pad, n_pad, 1));
    EGX_HIP_CHECK(hipMemsetAsync(gp->d_neg_invkf, 0, sizeof(double) * (size_t)n_pad * gp->rhs_pad, w.stream));
    // 0 - W [ft | yt]: the rows [ft | yt]^T sit below the factor; W upper triangular -> K range starts at the row tile
    EGX_RC(launch_gemm_nt_sub(w.stream, gp->d_neg_invkf, gp->rhs_pad, gp->d_W, n_pad, w.M + (size_t)n_pad * gp->ld,
                              gp->ld, n_pad, gp->rhs_pad, n_pad, 0, 1));
    {
        std::vector<double> tmp((size_t)n_pad * gp->rhs_pad);
        EGX_HIP_CHECK(hipMemcpyAsync(tmp.data(), gp->d_neg_invkf, sizeof(double) * tmp.size(), hipMemcpyDeviceToHost, w.stream));
        EGX_HIP_CHECK(hipStreamSynchronize(w.stream));
        gp->h_neg_invkf.resize((size_t)gp->n * gp->p);
        for (int i = 0; i < gp->n; i++)
            for (int l = 0; l < gp->p; l++) gp->h_neg_invkf[(size_t)i * gp->p + l] = tmp[(size_t)i * gp->rhs_pad + l];
    }
    gp->winv_epoch = gp->fit_epoch;
    return EGX_SUCCESS;
}

// Small batches (EGO's infill optimiser asks for one point at a time): per query two memory-bound passes over the cached
// W = C^-T instead of the batched block solves, then the x-gradient contraction with a per-training-point weight VECTOR.
static int small_path_buffers(egx_gp *gp) {
    if (gp->sp_R) return EGX_SUCCESS;
    const int n = gp->n, n_pad = gp->n_pad, d = gp->d;
    int nsplit = (n + 63) / 64;
    if (nsplit > 512) nsplit = 512;
    const int slabs = (n + 63) / 64, per = (slabs + nsplit - 1) / nsplit;
    gp->sp_nsplit = (slabs + per - 1) / per;
    EGX_HIP_CHECK(hipMalloc(&gp->sp_R, sizeof(double) * (size_t)kTile * n_pad));
    EGX_HIP_CHECK(hipMalloc(&gp->sp_P, sizeof(double) * (size_t)32 * n_pad));
    EGX_HIP_CHECK(hipMalloc(&gp->sp_y, sizeof(double) * n_pad));
    EGX_HIP_CHECK(hipMalloc(&gp->sp_z, sizeof(double) * n_pad));
    EGX_HIP_CHECK(hipMalloc(&gp->sp_wt, sizeof(double) * n_pad));
    EGX_HIP_CHECK(hipMalloc(&gp->sp_out, sizeof(double) * (size_t)gp->sp_nsplit * kTile * d));
    EGX_HIP_CHECK(hipMalloc(&gp->sp_xq, sizeof(double) * (size_t)d * kTile));
    return EGX_SUCCESS;
}

// normalised query a of xq, k-major with the other 127 slots zero, into the cached device slab; xn (d) on the host
static int small_path_query(egx_gp *gp, const double *xq, int64_t a, std::vector<double> &xn, std::vector<double> &slab) {
    const int d = gp->d;
    xn.resize(d);
    slab.assign((size_t)d * kTile, 0.0);
    for (int j = 0; j < d; j++) {
        xn[j] = (xq[(size_t)a * d + j] - gp->x_mean[j]) / gp->x_std[j];
        slab[(size_t)j * kTile] = xn[j];
    }
    EGX_HIP_CHECK(hipMemcpyAsync(gp->sp_xq, slab.data(), sizeof(double) * slab.size(), hipMemcpyHostToDevice, gp->ws[0].stream));
    return EGX_SUCCESS;
}

// y = C^-1 r, z = R^-1 r of ONE query (already in sp_xq) on the host; needs ensure_winv
static int small_path_solve(egx_gp *gp, std::vector<double> &y, std::vector<double> &z, bool want_z) {
    Workspace &w = gp->ws[0];
    const int n = gp->n, n_pad = gp->n_pad, d = gp->d;
    EGX_RC(launch_cross_corr(w.stream, gp->corr, gp->sp_xq, kTile, kTile, gp->d_xT, n_pad, n_pad, d, gp->d_fit_coef,
                             gp->fit_hcols, gp->sp_R, n_pad));
    EGX_RC(launch_uptri_solve_pair(w.stream, gp->d_W, n_pad, n, n_pad, gp->sp_R, gp->sp_P, gp->sp_y, gp->sp_z));
    y.resize(n_pad);
    EGX_HIP_CHECK(hipMemcpyAsync(y.data(), gp->sp_y, sizeof(double) * n_pad, hipMemcpyDeviceToHost, w.stream));
    if (want_z) {
        z.resize(n_pad);
        EGX_HIP_CHECK(hipMemcpyAsync(z.data(), gp->sp_z, sizeof(double) * n_pad, hipMemcpyDeviceToHost, w.stream));
    }
    EGX_HIP_CHECK(hipStreamSynchronize(w.stream));
    return EGX_SUCCESS;
}

static int xgrad_small(egx_gp *gp, const double *xq, int64_t m, double *gy, double *gv) {
    Workspace &w = gp->ws[0];
    const int n = gp->n, n_pad = gp->n_pad, d = gp->d, p = gp->p;
    if (gv) EGX_RC(ensure_winv(gp));
    EGX_RC(small_path_buffers(gp));
    const int m_pad = kTile;
    const int nblk = (n + 255) / 256;  // k_xgrad_point: one partial row of d sums per 256 training points
    std::vector<double> xn, slab, y, z, wt(n_pad, 0.0), part((size_t)nblk * d), f(p), a_vec(p), u(p), dd(p), df(d);
    auto reduce_out = [&](int k) {
        double sacc = 0.0;
        for (int sidx = 0; sidx < nblk; sidx++) sacc += part[(size_t)sidx * d + k];
        return sacc;
    };
    auto contract = [&](const double *weights) -> int {
        EGX_RC(launch_xgrad_point(w.stream, gp->corr, gp->sp_xq, m_pad, 1, gp->d_xT, n_pad, n, d, gp->d_fit_coef,
                                  gp->fit_hcols, weights, gp->sp_out));
        EGX_HIP_CHECK(hipMemcpyAsync(part.data(), gp->sp_out, sizeof(double) * (size_t)nblk * d, hipMemcpyDeviceToHost,
                                     w.stream));
        EGX_HIP_CHECK(hipStreamSynchronize(w.stream));
        return EGX_SUCCESS;
    };
    for (int64_t a = 0; a < m; a++) {
        EGX_RC(small_path_query(gp, xq, a, xn, slab));
        if (gy) {
            EGX_RC(contract(gp->d_gamma));
            hm::regression_jac_dot(gp->mean, xn.data(), d, gp->beta.data(), df.data());
            for (int k = 0; k < d; k++) gy[(size_t)a * d + k] = (df[k] + reduce_out(k)) * gp->y_std / gp->x_std[k];
        }
        if (gv) {
            EGX_RC(small_path_solve(gp, y, z, true));
            hm::regression_row(gp->mean, xn.data(), d, f.data());
            for (int l = 0; l < p; l++) {  // A = f - ft^T rt
                double sacc = 0.0;
                for (int i = 0; i < n; i++) sacc += gp->ft[(size_t)i * p + l] * y[i];
                a_vec[l] = f[l] - sacc;
            }
            for (int i = 0; i < p; i++) {
                double sacc = a_vec[i];
                for (int l = 0; l < i; l++) sacc -= gp->ft_qr_r[(size_t)l * p + i] * u[l];
                u[i] = sacc / gp->ft_qr_r[(size_t)i * p + i];
            }
            for (int i = p - 1; i >= 0; i--) {
                double sacc = u[i];
                for (int l = i + 1; l < p; l++) sacc -= gp->ft_qr_r[(size_t)i * p + l] * dd[l];
                dd[i] = sacc / gp->ft_qr_r[(size_t)i * p + i];
            }
            for (int i = 0; i < n; i++) {  // -(R^-1 r + R^-1 F D)_i
                double e = 0.0;
                for (int l = 0; l < p; l++) e += gp->h_neg_invkf[(size_t)i * p + l] * dd[l];
                wt[i] = -z[i] + e;
            }
            EGX_HIP_CHECK(hipMemcpyAsync(gp->sp_wt, wt.data(), sizeof(double) * n_pad, hipMemcpyHostToDevice, w.stream));
            EGX_RC(contract(gp->sp_wt));
            hm::regression_jac_dot(gp->mean, xn.data(), d, dd.data(), df.data());
            for (int k = 0; k < d; k++) gv[(size_t)a * d + k] = 2.0 * gp->sigma2 * (df[k] + reduce_out(k)) / gp->x_std[k];
        }
    }
    return EGX_SUCCESS;
}

// predict_var of a few points through the cached W (built on the third such call after a fit, or by any gradient call)
static int predict_var_small(egx_gp *gp, const double *xq, int64_t m, double *vout) {
    const int n = gp->n, d = gp->d, p = gp->p;
    EGX_RC(ensure_winv(gp));
    EGX_RC(small_path_buffers(gp));
    std::vector<double> xn, slab, y, z, f(p), u(p);
    for (int64_t a = 0; a < m; a++) {
        EGX_RC(small_path_query(gp, xq, a, xn, slab));
        EGX_RC(small_path_solve(gp, y, z, false));
        double s0 = 0.0;
        for (int i = 0; i < n; i++) s0 += y[i] * y[i];
        hm::regression_row(gp->mean, xn.data(), d, f.data());
        double usq = 0.0;
        for (int i = 0; i < p; i++) {  // u = (Rq^T)^-1 (ft^T rt - f)   algorithm.rs:352-367
            double sacc = -f[i];
            for (int t = 0; t < n; t++) sacc += gp->ft[(size_t)t * p + i] * y[t];
            for (int l = 0; l < i; l++) sacc -= gp->ft_qr_r[(size_t)l * p + i] * u[l];
            u[i] = sacc / gp->ft_qr_r[(size_t)i * p + i];
            usq += u[i] * u[i];
        }
        const double mse = gp->sigma2 * (1.0 - s0 + usq);
        vout[a] = (mse < 0.0) ? 0.0 : mse;
    }
    return EGX_SUCCESS;
}

// predict_gradients / predict_var_gradients (algorithm.rs:510-549, 555-617, 702-727), batched over the queries:
//   d mean / d x_k = (dF beta + sum_i gamma_i dr_i/dx_k) y_std / x_std_k
//   d var  / d x_k = 2 sigma2 / x_std_k * ( D^T dF_k - sum_i (R^-1 r + R^-1 F D)_i dr_i/dx_k ),  D = B^-1 A^T,
//   A = f(x)^T - r^T R^-1 F = f^T - rt^T ft,  B = F^T R^-1 F = Rq^T Rq   (Rq = ft_qr_r, so D = Rq^-1 Rq^-T A^T)
// The reference redoes R^-1 F and chol(B) for every query point; here they are per-fit state, the per-query
// R^-1 r = C^-T (C^-1 r) is the predict_var solve followed by one GEMM with the cached C^-T.
static int xgrad_impl(egx_gp *gp, const double *xq, int64_t m, double *gy, double *gv) {
    if (!gp->fitted) {
        set_error("model is not fitted (call egx_gp_finalize or egx_gp_fit first)");
        return EGX_ERR_NOT_FITTED;
    }
    if (m < 0 || (m > 0 && !xq)) {
        set_error("bad query array");
        return EGX_ERR_INVALID_VALUE;
    }
    EGX_RC(set_device(gp));
    if (m > 0 && m <= 8) return xgrad_small(gp, xq, m, gy, gv);
    Workspace &w = gp->ws[0];
    const int n = gp->n, n_pad = gp->n_pad, d = gp->d, p = gp->p, rp = gp->rhs_pad;
    if (gv) EGX_RC(ensure_winv(gp));
    int64_t cap = ((int64_t)1 << 27) / n_pad / kTile * kTile;
    if (cap < kTile) cap = kTile;
    if (cap > 16384) cap = 16384;
    if (!gv) cap = 65536;
    std::vector<double> xn, part, sl, f(p), a_vec(p), u(p), dd(p), dneg, df(d);
    for (int64_t m0 = 0; m0 < m; m0 += cap) {
        const int mc = (int)((m - m0 < cap) ? (m - m0) : cap);
        const int m_pad = (int)round_up(mc, kTile);
        // enough workgroups for small batches: split the training range (partial sums added on the host)
        int nsplit = 1;
        const int wgs = m_pad / 128;
        if (wgs < 512) nsplit = (512 + wgs - 1) / wgs;
        const int slabs = (n + 63) / 64;
        if (nsplit > slabs) nsplit = slabs;
        const int per = (slabs + nsplit - 1) / nsplit;
        nsplit = (slabs + per - 1) / per;
        DevBuf d_xqT, d_out, d_RT, d_s0, d_sl, d_Wt, d_D;
        EGX_RC(upload_queries(gp, xq, m0, mc, m_pad, xn, d_xqT, w.stream));
        const size_t out_sz = (size_t)nsplit * m_pad * d;
        EGX_RC(d_out.alloc(out_sz));
        part.resize(out_sz);
        auto reduce_out = [&](int a, int k) {
            double sacc = 0.0;
            for (int sidx = 0; sidx < nsplit; sidx++) sacc += part[((size_t)sidx * m_pad + a) * d + k];
            return sacc;
        };
        if (gy) {
            EGX_RC(launch_xgrad(w.stream, gp->corr, d_xqT.p, m_pad, m_pad, gp->d_xT, n_pad, n, d, gp->d_fit_coef,
                                gp->fit_hcols, gp->d_gamma, 0, 1, nsplit, d_out.p));
            EGX_HIP_CHECK(hipMemcpyAsync(part.data(), d_out.p, sizeof(double) * out_sz, hipMemcpyDeviceToHost, w.stream));
            EGX_HIP_CHECK(hipStreamSynchronize(w.stream));
            for (int a = 0; a < mc; a++) {
                hm::regression_jac_dot(gp->mean, &xn[(size_t)a * d], d, gp->beta.data(), df.data());
                for (int k = 0; k < d; k++)
                    gy[(size_t)(m0 + a) * d + k] = (df[k] + reduce_out(a, k)) * gp->y_std / gp->x_std[k];
            }
        }
        if (gv) {
            sl.resize((size_t)m_pad * p);
            EGX_RC(d_RT.alloc((size_t)m_pad * n_pad));
            EGX_RC(d_s0.alloc(m_pad));
            EGX_RC(d_sl.alloc((size_t)m_pad * p));
            EGX_RC(d_Wt.alloc((size_t)n_pad * m_pad));
            EGX_RC(d_D.alloc((size_t)m_pad * rp));
            EGX_RC(launch_cross_corr(w.stream, gp->corr, d_xqT.p, m_pad, m_pad, gp->d_xT, n_pad, n_pad, d,
                                     gp->d_fit_coef, gp->fit_hcols, d_RT.p, n_pad));
            EGX_RC(launch_trsm_rows(w.stream, w.M, gp->ld, n_pad, w.dinv, d_RT.p, n_pad, m_pad));
            EGX_RC(launch_row_reduce(w.stream, d_RT.p, n_pad, m_pad, n, w.M + (size_t)n_pad * gp->ld, gp->ld, p,
                                     d_s0.p, d_sl.p));
            EGX_HIP_CHECK(hipMemcpyAsync(sl.data(), d_sl.p, sizeof(double) * (size_t)m_pad * p, hipMemcpyDeviceToHost,
                                         w.stream));
            // -Z^T = 0 - C^-T rt  as an (n_pad x m_pad) matrix (W upper triangular: K range starts at the row tile)
            EGX_HIP_CHECK(hipMemsetAsync(d_Wt.p, 0, sizeof(double) * (size_t)n_pad * m_pad, w.stream));
            EGX_RC(launch_gemm_nt_sub(w.stream, d_Wt.p, m_pad, gp->d_W, n_pad, d_RT.p, n_pad, n_pad, m_pad, n_pad, 0, 1));
            EGX_HIP_CHECK(hipStreamSynchronize(w.stream));
            // D = B^-1 A^T per query (p x p work on the host), uploaded negated and zero padded to rhs_pad columns
            dneg.assign((size_t)m_pad * rp, 0.0);
            for (int a = 0; a < mc; a++) {
                hm::regression_row(gp->mean, &xn[(size_t)a * d], d, f.data());
                for (int l = 0; l < p; l++) a_vec[l] = f[l] - sl[(size_t)a * p + l];
                for (int i = 0; i < p; i++) {  // Rq^T u = A^T (Rq^T lower)
                    double sacc = a_vec[i];
                    for (int l = 0; l < i; l++) sacc -= gp->ft_qr_r[(size_t)l * p + i] * u[l];
                    u[i] = sacc / gp->ft_qr_r[(size_t)i * p + i];
                }
                for (int i = p - 1; i >= 0; i--) {  // Rq D = u (Rq upper)
                    double sacc = u[i];
                    for (int l = i + 1; l < p; l++) sacc -= gp->ft_qr_r[(size_t)i * p + l] * dd[l];
                    dd[i] = sacc / gp->ft_qr_r[(size_t)i * p + i];
                }
                for (int l = 0; l < p; l++) dneg[(size_t)a * rp + l] = -dd[l];
            }
            EGX_HIP_CHECK(hipMemcpyAsync(d_D.p, dneg.data(), sizeof(double) * dneg.size(), hipMemcpyHostToDevice, w.stream));
            // -(Z + E)^T : Wt -= (-R^-1 F) (-D)^T
            EGX_RC(launch_gemm_nt_sub(w.stream, d_Wt.p, m_pad, gp->d_neg_invkf, rp, d_D.p, rp, n_pad, m_pad, rp, 0, 0));
            EGX_RC(launch_xgrad(w.stream, gp->corr, d_xqT.p, m_pad, m_pad, gp->d_xT, n_pad, n, d, gp->d_fit_coef,
                                gp->fit_hcols, d_Wt.p, m_pad, 0, nsplit, d_out.p));
            EGX_HIP_CHECK(hipMemcpyAsync(part.data(), d_out.p, sizeof(double) * out_sz, hipMemcpyDeviceToHost, w.stream));
            EGX_HIP_CHECK(hipStreamSynchronize(w.stream));
            for (int a = 0; a < mc; a++) {
                for (int l = 0; l < p; l++) dd[l] = -dneg[(size_t)a * rp + l];
                hm::regression_jac_dot(gp->mean, &xn[(size_t)a * d], d, dd.data(), df.data());
                for (int k = 0; k < d; k++)
                    gv[(size_t)(m0 + a) * d + k] = 2.0 * gp->sigma2 * (df[k] + reduce_out(a, k)) / gp->x_std[k];
            }
        }
    }
    return EGX_SUCCESS;
}

}  // namespace egx

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int32_t egx_abi_version(void) { return EGX_GP_ABI_VERSION; }
const char *egx_last_error(void) { return g_last_error.c_str(); }

int32_t egx_device_count(void) {
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
}

void egx_gp_config_default(egx_gp_config *cfg) {
    if (!cfg) return;
    cfg->corr = EGX_CORR_SQUARED_EXPONENTIAL;  // Kriging = ConstantMean + SquaredExponentialCorr, algorithm.rs:200-207
    cfg->mean = EGX_MEAN_CONSTANT;
    cfg->nugget = 100.0 * std::numeric_limits<double>::epsilon();  // parameters.rs:118
    cfg->device = -1;
    cfg->n_workspaces = 1;
    cfg->w_star = nullptr;
    cfg->kpls_dim = 0;
}

int32_t egx_normalize(const double *x, int64_t n, int64_t d, double *xnorm, double *mean, double *std_) {
    if (!x || !xnorm || !mean || !std_ || n < 2 || d < 1) {
        set_error("egx_normalize: need n >= 2, d >= 1 and non-null arrays");
        return EGX_ERR_INVALID_VALUE;
    }
    hm::normalize(x, n, d, xnorm, mean, std_);
    return EGX_SUCCESS;
}

int64_t egx_regression_ncols(int32_t mean, int64_t d) { return hm::regression_ncols(mean, d); }

int32_t egx_regression_basis(int32_t mean, const double *x, int64_t n, int64_t d, double *f) {
    const int64_t p = hm::regression_ncols(mean, d);
    if (p < 0 || !x || !f || n < 0 || d < 1) {
        set_error("egx_regression_basis: bad arguments");
        return EGX_ERR_INVALID_VALUE;
    }
    for (int64_t i = 0; i < n; i++) hm::regression_row(mean, x + i * d, d, f + i * p);
    return EGX_SUCCESS;
}

int32_t egx_gp_create(const egx_gp_config *cfg_in, const double *x, const double *y, int64_t n, int64_t d,
                      egx_gp **out) {
    if (!out) {
        set_error("out handle pointer is NULL");
        return EGX_ERR_INVALID_VALUE;
    }
    *out = nullptr;
    egx_gp_config cfg;
    if (cfg_in) cfg = *cfg_in; else egx_gp_config_default(&cfg);
    if (!x || !y) {
        set_error("x / y is NULL");
        return EGX_ERR_INVALID_VALUE;
    }
    if (n < 2) {
        set_error("need at least 2 training points (sample standard deviation, utils.rs:48), got " + std::to_string(n));
        return EGX_ERR_INVALID_VALUE;
    }
    if (d < 1 || d > kMaxDim) {
        set_error("input dimension must be in [1, 64], got " + std::to_string(d));
        return EGX_ERR_INVALID_VALUE;
    }
    if (cfg.corr < 0 || cfg.corr > 3 || cfg.mean < 0 || cfg.mean > 2) {
        set_error("unknown correlation / regression model");
        return EGX_ERR_INVALID_VALUE;
    }
    if (!std::isfinite(cfg.nugget)) {  // the reference does not constrain the nugget's sign
        set_error("nugget must be finite");
        return EGX_ERR_INVALID_VALUE;
    }
    if (cfg.w_star) {
        if (cfg.kpls_dim == 0) {  // parameters.rs:290-294
            set_error("`kpls_dim` canot be 0!");
            return EGX_ERR_INVALID_VALUE;
        }
        if (cfg.kpls_dim < 0 || cfg.kpls_dim > d) {  // algorithm.rs:798-807
            set_error("Dimension reduction " + std::to_string(cfg.kpls_dim) +
                      " should be smaller than actual training input dimensions " + std::to_string(d));
            return EGX_ERR_INVALID_VALUE;
        }
    }
    for (int64_t i = 0; i < n * d; i++)
        if (!std::isfinite(x[i])) {
            set_error("x contains non-finite values");
            return EGX_ERR_INVALID_VALUE;
        }
    for (int64_t i = 0; i < n; i++)
        if (!std::isfinite(y[i])) {
            set_error("y contains non-finite values");
            return EGX_ERR_INVALID_VALUE;
        }
    const int64_t p = hm::regression_ncols(cfg.mean, d);
    if (p >= n) {
        // the reference would produce a rank-deficient GLS; refuse explicitly
        set_error("need more training points than regression basis columns");
        return EGX_ERR_INVALID_VALUE;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device: libegx_gp_hip has no CPU fallback (needs an MI355X / gfx950 GPU)");
        return EGX_ERR_NO_DEVICE;
    }
    int dev = cfg.device;
    if (dev < 0) {
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    }
    if (dev >= ndev) {
        set_error("device ordinal out of range");
        return EGX_ERR_INVALID_VALUE;
    }
    egx_gp *gp = new egx_gp();
    gp->device = dev;
    gp->n = (int)n;
    gp->d = (int)d;
    gp->p = (int)p;
    gp->corr = cfg.corr;
    gp->mean = cfg.mean;
    gp->nugget = cfg.nugget;
    gp->has_w = cfg.w_star != nullptr;
    gp->h = gp->has_w ? (int)cfg.kpls_dim : (int)d;
    if (gp->has_w) gp->w_star.assign(cfg.w_star, cfg.w_star + d * cfg.kpls_dim);
    gp->q = gp->p + 1;
    // 256-column granularity lets every trailing update of a large fit use the 128x256 tile (N % 256 == 0)
    gp->n_pad = (int)round_up(n, n >= 4096 ? kNB : kTile);
    gp->rhs_pad = (int)round_up(gp->q, kRhsPad);
    gp->m_tot = gp->n_pad + gp->rhs_pad;
    gp->ld = gp->n_pad;
    gp->x_raw.assign(x, x + n * d);
    gp->y_raw.assign(y, y + n);
    gp->xnorm.resize(n * d);
    gp->x_mean.resize(d);
    gp->x_std.resize(d);
    gp->ynorm.resize(n);
    hm::normalize(x, n, d, gp->xnorm.data(), gp->x_mean.data(), gp->x_std.data());  // algorithm.rs:840
    hm::normalize(y, n, 1, gp->ynorm.data(), &gp->y_mean, &gp->y_std);               // algorithm.rs:841
    gp->F.resize(n * p);
    for (int64_t i = 0; i < n; i++) hm::regression_row(gp->mean, &gp->xnorm[i * d], d, &gp->F[i * p]);  // :866

    auto fail = [&](int rc) {
        egx_gp_destroy(gp);
        return rc;
    };
    if (hipSetDevice(dev) != hipSuccess) {
        set_error("hipSetDevice failed");
        return fail(EGX_ERR_HIP);
    }
    int rc = chol_init();
    if (rc) return fail(rc);
    // k-major, zero padded copies
    std::vector<double> xT((size_t)d * gp->n_pad, 0.0), rhsT((size_t)gp->q * gp->n_pad, 0.0);
    for (int64_t i = 0; i < n; i++)
        for (int64_t j = 0; j < d; j++) xT[(size_t)j * gp->n_pad + i] = gp->xnorm[i * d + j];
    for (int64_t l = 0; l < p; l++)
        for (int64_t i = 0; i < n; i++) rhsT[(size_t)l * gp->n_pad + i] = gp->F[i * p + l];
    for (int64_t i = 0; i < n; i++) rhsT[(size_t)p * gp->n_pad + i] = gp->ynorm[i];
#define EGX_HIPF(expr)                                                            \
    do {                                                                          \
        hipError_t _e = (expr);                                                   \
        if (_e != hipSuccess) {                                                   \
            set_error(std::string(#expr) + ": " + hipGetErrorString(_e));         \
            return fail(EGX_ERR_HIP);                                             \
        }                                                                         \
    } while (0)
    EGX_HIPF(hipMalloc(&gp->d_xT, sizeof(double) * xT.size()));
    EGX_HIPF(hipMalloc(&gp->d_rhsT, sizeof(double) * rhsT.size()));
    EGX_HIPF(hipMalloc(&gp->d_gamma, sizeof(double) * gp->n_pad));
    EGX_HIPF(hipMalloc(&gp->d_fit_coef, sizeof(double) * (size_t)d * (gp->has_w ? gp->h : 1)));
    EGX_HIPF(hipMemcpy(gp->d_xT, xT.data(), sizeof(double) * xT.size(), hipMemcpyHostToDevice));
    EGX_HIPF(hipMemcpy(gp->d_rhsT, rhsT.data(), sizeof(double) * rhsT.size(), hipMemcpyHostToDevice));
    int nws = cfg.n_workspaces < 1 ? 1 : cfg.n_workspaces;
    gp->ws.resize(nws);
    for (int i = 0; i < nws; i++) {
        rc = alloc_workspace(gp, gp->ws[i]);
        if (rc) return fail(rc);
    }
    *out = gp;
    return EGX_SUCCESS;
}

void egx_gp_destroy(egx_gp *gp) {
    if (!gp) return;
    hipSetDevice(gp->device);
    for (auto &w : gp->ws) free_workspace(w);
    if (gp->d_xT) hipFree(gp->d_xT);
    if (gp->d_rhsT) hipFree(gp->d_rhsT);
    if (gp->d_gamma) hipFree(gp->d_gamma);
    if (gp->d_fit_coef) hipFree(gp->d_fit_coef);
    if (gp->d_W) hipFree(gp->d_W);
    if (gp->d_neg_invkf) hipFree(gp->d_neg_invkf);
    for (double *q : {gp->sp_R, gp->sp_P, gp->sp_y, gp->sp_z, gp->sp_wt, gp->sp_out, gp->sp_xq})
        if (q) hipFree(q);
    if (gp->d_Rinv) hipFree(gp->d_Rinv);
    if (gp->d_gout) hipFree(gp->d_gout);
    if (gp->d_theta) hipFree(gp->d_theta);
    delete gp;
}

int32_t egx_gp_dims(const egx_gp *gp, int64_t *n, int64_t *d, int64_t *p, int64_t *h) {
    if (!gp) {
        set_error("NULL handle");
        return EGX_ERR_INVALID_VALUE;
    }
    if (n) *n = gp->n;
    if (d) *d = gp->d;
    if (p) *p = gp->p;
    if (h) *h = gp->h;
    return EGX_SUCCESS;
}

int32_t egx_gp_likelihood(egx_gp *gp, const double *theta, int64_t theta_len, double *lkh, int32_t *status) {
    if (!gp || !theta || !lkh || !status) {
        set_error("NULL argument");
        return EGX_ERR_INVALID_VALUE;
    }
    std::shared_lock<std::shared_mutex> lock(gp->mu);
    EGX_RC(set_device(gp));
    // take a free workspace; workspace 0 (where a fitted factor lives) is the last choice
    int wi = -1;
    {
        std::unique_lock<std::mutex> pl(gp->pool_mu);
        if (gp->ws_busy.size() != gp->ws.size()) gp->ws_busy.assign(gp->ws.size(), 0);
        for (;;) {
            for (int i = (int)gp->ws.size() - 1; i >= 0 && wi < 0; i--)
                if (!gp->ws_busy[i]) wi = i;
            if (wi >= 0) break;
            gp->pool_cv.wait(pl);
        }
        gp->ws_busy[wi] = 1;
    }
    EvalResult res;
    const int rc = eval_one(gp, wi, theta, theta_len, res, false);
    {
        std::lock_guard<std::mutex> pl(gp->pool_mu);
        gp->ws_busy[wi] = 0;
    }
    gp->pool_cv.notify_one();
    if (rc) return rc;
    *lkh = res.lkh;
    *status = res.status;
    return EGX_SUCCESS;
}

int32_t egx_gp_likelihood_batch(egx_gp *gp, const double *thetas, int64_t k, int64_t theta_len, double *lkh,
                                int32_t *status) {
    if (!gp || (k > 0 && (!thetas || !lkh || !status)) || k < 0) {
        set_error("NULL argument");
        return EGX_ERR_INVALID_VALUE;
    }
    std::unique_lock<std::shared_mutex> lock(gp->mu);
    EGX_RC(set_device(gp));
    const int nws = (int)gp->ws.size();
    // Software pipeline over the workspaces: every workspace has its own stream pair, so the serial panel
    // factorisations of one candidate overlap the trailing updates of the others; as soon as a candidate has
    // been read back its workspace is re-used for candidate c + nws.
    std::vector<int> launched(nws, 0);
    auto enqueue = [&](int64_t c) -> int {
        const int wi = (int)(c % nws);
        const double *th = thetas + c * theta_len;
        std::vector<double> coef;
        int hcols = 1;
        EGX_RC(make_coef(gp, th, theta_len, coef, hcols, nullptr));
        launched[wi] = 0;
        if (has_nan(th, theta_len)) {
            lkh[c] = -std::numeric_limits<double>::infinity();
            status[c] = EGX_STATUS_NAN_THETA;
            return EGX_SUCCESS;
        }
        if (wi == 0) gp->fitted = false;
        EGX_RC(enqueue_eval(gp, gp->ws[wi], coef, hcols));
        launched[wi] = 1;
        return EGX_SUCCESS;
    };
    for (int64_t c = 0; c < k && c < nws; c++) EGX_RC(enqueue(c));
    for (int64_t c = 0; c < k; c++) {
        const int wi = (int)(c % nws);
        if (launched[wi]) {
            EvalResult res;
            EGX_RC(finish_eval(gp, gp->ws[wi], res, false));
            lkh[c] = res.lkh;
            status[c] = res.status;
            if (wi == 0) record_timings(gp, gp->ws[0], 0.0, 0.0);
        }
        if (c + nws < k) EGX_RC(enqueue(c + nws));
    }
    return EGX_SUCCESS;
}

int32_t egx_gp_finalize(egx_gp *gp, const double *theta, int64_t theta_len) {
    if (!gp || !theta) {
        set_error("NULL argument");
        return EGX_ERR_INVALID_VALUE;
    }
    std::unique_lock<std::shared_mutex> lock(gp->mu);
    EGX_RC(set_device(gp));
    return do_finalize(gp, theta, theta_len);
}

// Multistart derivative-free fit over the ACTIVE theta components (all of them for ThetaTuning::Full; a subset for
// ThetaTuning::Partial, algorithm.rs:822-826, 873-960: the inactive components stay at theta_base).
static int32_t fit_nm_core(egx_gp *gp, const double *theta_base /*h*/, const std::vector<int> &active,
                           const double *theta0s /*n_starts x k*/, int64_t n_starts, const double *lo,
                           const double *hi, int64_t bounds_len, int64_t max_eval, int64_t *n_evals_out) {
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
    double best_f = std::numeric_limits<double>::infinity();
    std::vector<double> best_x(h, 0.0);
    int64_t evals = 0;
    for (int64_t s = 0; s < n_starts * h; s++)
        if (!(theta0s[s] > 0.0)) {
            set_error("theta start points must be > 0");
            return EGX_ERR_INVALID_VALUE;
        }
    // The starts are independent optimisations (rayon par_iter over theta_inits rows, algorithm.rs:928-945):
    // one host thread per workspace, start s runs on workspace s % n_threads.
    const int nthreads = (int)std::min<int64_t>((int64_t)gp->ws.size(), n_starts);
    std::vector<NmResult> results((size_t)n_starts);
    std::vector<int> rcs((size_t)nthreads, EGX_SUCCESS);
    std::vector<std::string> errs((size_t)nthreads);
    gp->fitted = false;
    auto worker = [&](int t) {
        if (hipSetDevice(gp->device) != hipSuccess) {
            rcs[t] = EGX_ERR_HIP;
            errs[t] = "hipSetDevice failed in optimiser thread";
            return;
        }
        auto objective = [&](const std::vector<double> &x) -> double {
            std::vector<double> th(theta_base, theta_base + hfull);
            for (int i = 0; i < h; i++) th[active[i]] = std::pow(10.0, x[i]);
            EvalResult res;
            int rc = eval_one(gp, t, th.data(), hfull, res, false);
            if (rc) {
                if (!rcs[t]) {
                    rcs[t] = rc;
                    errs[t] = g_last_error;
                }
                return std::numeric_limits<double>::infinity();
            }
            if (res.status != EGX_STATUS_OK || std::isnan(res.lkh)) return std::numeric_limits<double>::infinity();
            return -res.lkh;
        };
        for (int64_t s = t; s < n_starts; s += nthreads) {
            std::vector<double> x0(h);
            for (int i = 0; i < h; i++) x0[i] = std::log10(theta0s[s * h + i]);
            results[(size_t)s] = nelder_mead(objective, x0, blo, bhi, per_start);
        }
    };
    if (nthreads <= 1) {
        worker(0);
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nthreads; t++) pool.emplace_back(worker, t);
        for (auto &th : pool) th.join();
    }
    for (int t = 0; t < nthreads; t++)
        if (rcs[t]) {
            set_error(errs[t]);
            return rcs[t];
        }
    for (int64_t s = 0; s < n_starts; s++) {
        evals += results[(size_t)s].evals;
        if (results[(size_t)s].f < best_f) {  // algorithm.rs:942-945 reduce to min (first wins ties)
            best_f = results[(size_t)s].f;
            best_x = results[(size_t)s].x;
        }
    }
    if (n_evals_out) *n_evals_out = evals;
    std::vector<double> th(theta_base, theta_base + hfull);
    if (std::isfinite(best_f))
        for (int i = 0; i < h; i++) th[active[i]] = std::pow(10.0, best_x[i]);
    else  // every start failed: the reference falls through with ones (algorithm.rs:943) -> 10^1... keep start 0
        for (int i = 0; i < h; i++) th[active[i]] = theta0s[i];
    return do_finalize(gp, th.data(), hfull);
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

int32_t egx_gp_predict(egx_gp *gp, const double *xq, int64_t m, double *y) {
    if (!gp || (m > 0 && !y)) {
        set_error("NULL argument");
        return EGX_ERR_INVALID_VALUE;
    }
    std::unique_lock<std::shared_mutex> lock(gp->mu);
    return predict_impl(gp, xq, m, y, nullptr);
}
int32_t egx_gp_predict_var(egx_gp *gp, const double *xq, int64_t m, double *var) {
    if (!gp || (m > 0 && !var)) {
        set_error("NULL argument");
        return EGX_ERR_INVALID_VALUE;
    }
    std::unique_lock<std::shared_mutex> lock(gp->mu);
    return predict_impl(gp, xq, m, nullptr, var);
}
int32_t egx_gp_predict_valvar(egx_gp *gp, const double *xq, int64_t m, double *y, double *var) {
    if (!gp || (m > 0 && (!y || !var))) {
        set_error("NULL argument");
        return EGX_ERR_INVALID_VALUE;
    }
    std::unique_lock<std::shared_mutex> lock(gp->mu);
    return predict_impl(gp, xq, m, y, var);
}

int32_t egx_gp_predict_gradients(egx_gp *gp, const double *x, int64_t m, double *grad) {
    if (!gp || (m > 0 && !grad)) {
        set_error("NULL argument");
        return EGX_ERR_INVALID_VALUE;
    }
    std::unique_lock<std::shared_mutex> lock(gp->mu);
    return xgrad_impl(gp, x, m, grad, nullptr);
}

int32_t egx_gp_predict_var_gradients(egx_gp *gp, const double *x, int64_t m, double *grad) {
    if (!gp || (m > 0 && !grad)) {
        set_error("NULL argument");
        return EGX_ERR_INVALID_VALUE;
    }
    std::unique_lock<std::shared_mutex> lock(gp->mu);
    return xgrad_impl(gp, x, m, nullptr, grad);
}

int32_t egx_gp_predict_valvar_gradients(egx_gp *gp, const double *x, int64_t m, double *grad_y, double *grad_var) {
    if (!gp || (m > 0 && (!grad_y || !grad_var))) {
        set_error("NULL argument");
        return EGX_ERR_INVALID_VALUE;
    }
    std::unique_lock<std::shared_mutex> lock(gp->mu);
    return xgrad_impl(gp, x, m, grad_y, grad_var);
}

int32_t egx_gp_get_inner(egx_gp *gp, const egx_gp_inner_view *v) {
    if (!gp || !v) {
        set_error("NULL argument");
        return EGX_ERR_INVALID_VALUE;
    }
    std::unique_lock<std::shared_mutex> lock(gp->mu);
    if (!gp->fitted) {
        set_error("model is not fitted");
        return EGX_ERR_NOT_FITTED;
    }
    EGX_RC(set_device(gp));
    const int n = gp->n, p = gp->p, d = gp->d;
    if (v->theta) std::memcpy(v->theta, gp->theta.data(), sizeof(double) * gp->h);
    if (v->likelihood) *v->likelihood = gp->likelihood;
    if (v->sigma2) *v->sigma2 = gp->sigma2;
    if (v->beta) std::memcpy(v->beta, gp->beta.data(), sizeof(double) * p);
    if (v->gamma) std::memcpy(v->gamma, gp->gamma.data(), sizeof(double) * n);
    if (v->ft) std::memcpy(v->ft, gp->ft.data(), sizeof(double) * (size_t)n * p);
    if (v->ft_qr_r) std::memcpy(v->ft_qr_r, gp->ft_qr_r.data(), sizeof(double) * (size_t)p * p);
    if (v->x_mean) std::memcpy(v->x_mean, gp->x_mean.data(), sizeof(double) * d);
    if (v->x_std) std::memcpy(v->x_std, gp->x_std.data(), sizeof(double) * d);
    if (v->y_mean) *v->y_mean = gp->y_mean;
    if (v->y_std) *v->y_std = gp->y_std;
    if (v->xt_norm) std::memcpy(v->xt_norm, gp->xnorm.data(), sizeof(double) * (size_t)n * d);
    if (v->yt_norm) std::memcpy(v->yt_norm, gp->ynorm.data(), sizeof(double) * n);
    if (v->r_chol) {
        Workspace &w = gp->ws[0];
        EGX_HIP_CHECK(hipMemcpy2DAsync(v->r_chol, sizeof(double) * n, w.M, sizeof(double) * gp->ld, sizeof(double) * n,
                                       n, hipMemcpyDeviceToHost, w.stream));
        EGX_HIP_CHECK(hipStreamSynchronize(w.stream));
        for (int i = 0; i < n; i++)
            for (int j = i + 1; j < n; j++) v->r_chol[(size_t)i * n + j] = 0.0;
    }
    return EGX_SUCCESS;
}

int32_t egx_gp_set_inner(egx_gp *gp, const egx_gp_inner_view *v) {
    if (!gp || !v || !v->theta || !v->likelihood || !v->sigma2 || !v->beta || !v->gamma || !v->r_chol || !v->ft ||
        !v->ft_qr_r) {
        set_error("egx_gp_set_inner: theta, likelihood, sigma2, beta, gamma, r_chol, ft and ft_qr_r are required");
        return EGX_ERR_INVALID_VALUE;
    }
    std::unique_lock<std::shared_mutex> lock(gp->mu);
    EGX_RC(set_device(gp));
    const int n = gp->n, p = gp->p, n_pad = gp->n_pad;
    for (int i = 0; i < n; i++)
        if (!(v->r_chol[(size_t)i * n + i] > 0.0)) {
            set_error("egx_gp_set_inner: r_chol must have a positive diagonal");
            return EGX_ERR_INVALID_VALUE;
        }
    std::vector<double> coef, thfull;
    int hcols = 1;
    EGX_RC(make_coef(gp, v->theta, gp->h, coef, hcols, &thfull));
    gp->fitted = false;
    Workspace &w = gp->ws[0];
    // factor (identity padded, lower) + ft^T rows below it, exactly the layout a factorisation leaves behind
    std::vector<double> hm((size_t)gp->m_tot * gp->ld, 0.0);
    for (int i = 0; i < n; i++)
        for (int j = 0; j <= i; j++) hm[(size_t)i * gp->ld + j] = v->r_chol[(size_t)i * n + j];
    for (int i = n; i < n_pad; i++) hm[(size_t)i * gp->ld + i] = 1.0;
    for (int l = 0; l < p; l++)
        for (int i = 0; i < n; i++) hm[(size_t)(n_pad + l) * gp->ld + i] = v->ft[(size_t)i * p + l];
    EGX_HIP_CHECK(hipMemcpyAsync(w.M, hm.data(), sizeof(double) * hm.size(), hipMemcpyHostToDevice, w.stream));
    EGX_RC(launch_diag_tile_inverses(w.stream, w.M, gp->ld, n_pad, w.dinv));
    std::memset(w.h_vec, 0, sizeof(double) * n_pad);
    std::memcpy(w.h_vec, v->gamma, sizeof(double) * n);
    EGX_HIP_CHECK(hipMemcpyAsync(gp->d_gamma, w.h_vec, sizeof(double) * n_pad, hipMemcpyHostToDevice, w.stream));
    std::memcpy(w.h_coef, coef.data(), sizeof(double) * coef.size());
    EGX_HIP_CHECK(hipMemcpyAsync(gp->d_fit_coef, w.h_coef, sizeof(double) * coef.size(), hipMemcpyHostToDevice, w.stream));
    EGX_HIP_CHECK(hipStreamSynchronize(w.stream));
    gp->theta = thfull;
    gp->likelihood = *v->likelihood;
    gp->sigma2 = *v->sigma2;
    gp->beta.assign(v->beta, v->beta + p);
    gp->gamma.assign(v->gamma, v->gamma + n);
    gp->ft.assign(v->ft, v->ft + (size_t)n * p);
    gp->ft_qr_r.assign(v->ft_qr_r, v->ft_qr_r + (size_t)p * p);
    gp->fit_coef = coef;
    gp->fit_hcols = hcols;
    gp->fitted = true;
    gp->fit_epoch++;
    gp->small_var_calls = 0;
    return EGX_SUCCESS;
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
        for (int k = 0; k < d; k++) grad[k] = 0.0;
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
        for (int k = 0; k < d; k++) grad[k] = 0.0;
        return EGX_SUCCESS;
    }
    const size_t sq = (size_t)n_pad * n_pad;
    if (!gp->d_W) EGX_HIP_CHECK(hipMalloc(&gp->d_W, sizeof(double) * sq));
    if (!gp->d_Rinv) EGX_HIP_CHECK(hipMalloc(&gp->d_Rinv, sizeof(double) * sq));
    if (!gp->d_gout) EGX_HIP_CHECK(hipMalloc(&gp->d_gout, sizeof(double) * 2 * kMaxDim));
    if (!gp->d_theta) EGX_HIP_CHECK(hipMalloc(&gp->d_theta, sizeof(double) * kMaxDim));
    // gamma = C^-T rho
    std::memset(w.h_vec, 0, sizeof(double) * n_pad);
    std::memcpy(w.h_vec, res.rho.data(), sizeof(double) * n);
    EGX_HIP_CHECK(hipMemcpyAsync(w.d_rhs, w.h_vec, sizeof(double) * n_pad, hipMemcpyHostToDevice, w.stream));
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
    EGX_HIP_CHECK(hipMemcpyAsync(gp->d_theta, thfull.data(), sizeof(double) * d, hipMemcpyHostToDevice, w.stream));
    EGX_HIP_CHECK(hipStreamSynchronize(w.stream));
    EGX_RC(launch_grad_accum(w.stream, gp->corr, gp->d_xT, n_pad, n, d, gp->d_theta, gp->d_Rinv, n_pad, w.d_vec,
                             gp->d_gout));
    std::vector<double> gout(2 * d);
    EGX_HIP_CHECK(hipMemcpyAsync(gout.data(), gp->d_gout, sizeof(double) * 2 * d, hipMemcpyDeviceToHost, w.stream));
    EGX_HIP_CHECK(hipStreamSynchronize(w.stream));
    // dL/dtheta_k = (1/ln10) [ gamma^T dR_k gamma / sigma2 - tr(R^-1 dR_k) ] ; d_Rinv holds -R^-1
    const double ln10 = std::log(10.0);
    for (int k = 0; k < d; k++) grad[k] = (gout[d + k] / res.sigma2n + gout[k]) / ln10;
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
    if (gp->has_w) {
        set_error("likelihood gradient with KPLS weights is not implemented");
        return EGX_ERR_UNSUPPORTED;
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
    if (gp->has_w) {
        set_error("likelihood gradient with KPLS weights is not implemented");
        return EGX_ERR_UNSUPPORTED;
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

namespace egx {
static int upload_kmajor(const double *x, int64_t n, int64_t d, int n_pad, DevBuf &buf) {
    std::vector<double> xT((size_t)d * n_pad, 0.0);
    for (int64_t i = 0; i < n; i++)
        for (int64_t j = 0; j < d; j++) xT[(size_t)j * n_pad + i] = x[i * d + j];
    EGX_RC(buf.alloc(xT.size()));
    EGX_HIP_CHECK(hipMemcpy(buf.p, xT.data(), sizeof(double) * xT.size(), hipMemcpyHostToDevice));
    return EGX_SUCCESS;
}
static int require_device() {
    if (egx_device_count() <= 0) {
        set_error("no HIP device: libegx_gp_hip has no CPU fallback (needs an MI355X / gfx950 GPU)");
        return EGX_ERR_NO_DEVICE;
    }
    return EGX_SUCCESS;
}
}  // namespace egx

extern "C" {

int32_t egx_corr_matrix(int32_t corr, const double *xnorm, int64_t n, int64_t d, const double *theta, double nugget,
                        double *r) {
    if (!xnorm || !theta || !r || n < 1 || d < 1 || d > kMaxDim || corr < 0 || corr > 3) {
        set_error("egx_corr_matrix: bad arguments");
        return EGX_ERR_INVALID_VALUE;
    }
    EGX_RC(require_device());
    const int n_pad = (int)round_up(n, kTile);
    DevBuf d_xT, d_coef, d_M;
    EGX_RC(upload_kmajor(xnorm, n, d, n_pad, d_xT));
    EGX_RC(d_coef.alloc(d));
    EGX_HIP_CHECK(hipMemcpy(d_coef.p, theta, sizeof(double) * d, hipMemcpyHostToDevice));
    EGX_RC(d_M.alloc((size_t)n_pad * n_pad));
    EGX_RC(launch_corr_sym(0, corr, d_xT.p, n_pad, (int)n, (int)d, d_coef.p, 1, nugget, d_M.p, n_pad, n_pad));
    EGX_HIP_CHECK(hipMemcpy2D(r, sizeof(double) * n, d_M.p, sizeof(double) * n_pad, sizeof(double) * n, n,
                              hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n; i++)  // mirror the lower triangle (the scatter loop writes both, algorithm.rs:999-1000)
        for (int64_t j = i + 1; j < n; j++) r[i * n + j] = r[j * n + i];
    return EGX_SUCCESS;
}

int32_t egx_cross_corr(int32_t corr, const double *xq_norm, int64_t m, const double *xt_norm, int64_t n, int64_t d,
                       const double *theta, double *r) {
    if (!xq_norm || !xt_norm || !theta || !r || n < 1 || m < 1 || d < 1 || d > kMaxDim || corr < 0 || corr > 3) {
        set_error("egx_cross_corr: bad arguments");
        return EGX_ERR_INVALID_VALUE;
    }
    EGX_RC(require_device());
    const int n_pad = (int)round_up(n, kTile), m_pad = (int)round_up(m, kTile);
    DevBuf d_xT, d_qT, d_coef, d_R;
    EGX_RC(upload_kmajor(xt_norm, n, d, n_pad, d_xT));
    EGX_RC(upload_kmajor(xq_norm, m, d, m_pad, d_qT));
    EGX_RC(d_coef.alloc(d));
    EGX_HIP_CHECK(hipMemcpy(d_coef.p, theta, sizeof(double) * d, hipMemcpyHostToDevice));
    EGX_RC(d_R.alloc((size_t)m_pad * n_pad));
    EGX_RC(launch_cross_corr(0, corr, d_qT.p, m_pad, m_pad, d_xT.p, n_pad, n_pad, (int)d, d_coef.p, 1, d_R.p, n_pad));
    EGX_HIP_CHECK(hipMemcpy2D(r, sizeof(double) * n, d_R.p, sizeof(double) * n_pad, sizeof(double) * n, m,
                              hipMemcpyDeviceToHost));
    return EGX_SUCCESS;
}

int32_t egx_potrf(double *a, int64_t n, int32_t *info) {
    if (!a || !info || n < 1) {
        set_error("egx_potrf: bad arguments");
        return EGX_ERR_INVALID_VALUE;
    }
    EGX_RC(require_device());
    const int n_pad = (int)round_up(n, kTile);
    std::vector<double> hp((size_t)n_pad * n_pad, 0.0);
    for (int64_t i = 0; i < n; i++) std::memcpy(&hp[(size_t)i * n_pad], a + i * n, sizeof(double) * n);
    for (int64_t i = n; i < n_pad; i++) hp[(size_t)i * n_pad + i] = 1.0;
    DevBuf d_M, d_dinv, d_info;  // d_info: one int stored in a double-sized slot
    EGX_RC(d_M.alloc(hp.size()));
    EGX_RC(d_dinv.alloc((size_t)(n_pad / 64) * 4096));
    EGX_RC(d_info.alloc(1));
    EGX_HIP_CHECK(hipMemset(d_info.p, 0, sizeof(double)));
    EGX_HIP_CHECK(hipMemcpy(d_M.p, hp.data(), sizeof(double) * hp.size(), hipMemcpyHostToDevice));
    EGX_RC(launch_potrf(0, d_M.p, n_pad, n_pad, n_pad, d_dinv.p, reinterpret_cast<int *>(d_info.p)));
    EGX_HIP_CHECK(hipMemcpy(hp.data(), d_M.p, sizeof(double) * hp.size(), hipMemcpyDeviceToHost));
    EGX_HIP_CHECK(hipMemcpy(info, d_info.p, sizeof(int), hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n; i++)
        for (int64_t j = 0; j < n; j++) a[i * n + j] = (j <= i) ? hp[(size_t)i * n_pad + j] : 0.0;
    return EGX_SUCCESS;
}

int32_t egx_gp_last_timings(const egx_gp *gp, egx_timings *t) {
    if (!gp || !t) {
        set_error("NULL argument");
        return EGX_ERR_INVALID_VALUE;
    }
    *t = gp->timings;
    return EGX_SUCCESS;
}

int32_t egx_mfma_probe(double *max_abs_err) {
    if (!max_abs_err) return EGX_ERR_INVALID_VALUE;
    if (egx_device_count() <= 0) {
        set_error("no HIP device");
        return EGX_ERR_NO_DEVICE;
    }
    return mfma_probe(max_abs_err);
}

}  // extern "C"
