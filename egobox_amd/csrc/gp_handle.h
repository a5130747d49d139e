// Internal header shared by the dense-GP host translation units (gp_host.hip: handle, evaluation, state;
// gp_predict.hip: predictions and their x-gradients; gp_fit.hip: optimiser drivers and the theta-gradient).
#pragma once
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <list>
#include <condition_variable>
#include <mutex>
#include <shared_mutex>
#include <thread>
#include <algorithm>
#include <vector>

#include "egx_internal.h"
#include "host_math.h"
#include "cobyla.h"

#define EGX_RC(call)              \
    do {                          \
        int _rc = (call);         \
        if (_rc) return _rc;      \
    } while (0)

namespace egx {

// Scoped device allocation for the temporaries of one call (freed on every exit path).
struct DevBuf {
    double *p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    size_t cap = 0;  // doubles
    // grow-only: a buffer declared outside a chunk loop is allocated once (a hipMalloc / hipFree pair of a 1 GiB block per
    // 16 384 query points used to sit inside predict_var's loop)
    int alloc(size_t n_doubles) {
        if (n_doubles == 0) n_doubles = 1;
        if (p && cap >= n_doubles) return EGX_SUCCESS;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        EGX_HIP_CHECK(dev_malloc(&p, sizeof(double) * n_doubles));
        cap = n_doubles;
        return EGX_SUCCESS;
    }
};

struct Workspace {
    hipStream_t stream = nullptr;
    hipStream_t eval_stream = nullptr;  // the stream the evaluation in flight was enqueued on: the workspace's own, or
                                        // the leader's when it rides in a lock-step group (finish_eval waits on it)
    PotrfLookahead lk;  // look-ahead streams + events (lk.s2 == nullptr: look-ahead off)
    // the stream + events on which C^-T rides along a factorisation (theta-gradient; PotrfInverse, egx_internal.h)
    hipStream_t inv_stream = nullptr;
    hipEvent_t ev_inv_grp = nullptr, ev_inv_done = nullptr;
    // M, dinv and d_info are VIEWS into the handle's slabs (egx_gp::slab_*): consecutive workspaces sit at fixed
    // strides, which is what lets a group of them be factored in lock-step by one launch sequence
    double *M = nullptr;       // (m_tot x ld): correlation matrix / factor + appended RHS rows
    double *dinv = nullptr;    // (n_pad/64) x 64 x 64 inverses of the diagonal tiles
    double *dW = nullptr;      // (n_pad/256) x 256 x 256 transposed inverses of the diagonal blocks (lazy)
    double *d_coef = nullptr;  // d x hcols
    double *d_xs = nullptr;    // d x n_pad: the inputs times this candidate's coefficients (K1's scalar-row form)
    double *d_diag = nullptr;  // n
    double *d_vec = nullptr;   // n_pad (gamma)
    double *d_rhs = nullptr;   // n_pad (rho, destroyed by the back-substitution)
    int *d_info = nullptr;
    const int *sync_lead = nullptr;  // hand-off words of the lead of the lock-step group this evaluation ran in (diagnostics)
    // what finish_eval needs to enqueue this evaluation once more, alone and by separate launches, when a chain launch ran into
    // its wait bound: coefficient columns and count (the coefficients themselves are still in h_coef), the C^-T buffer of the
    // theta-gradient's rider, and whether this already IS the second attempt
    int retry_hcols = 1, retry_ncoef = 0;
    double *retry_W = nullptr;
    bool retried = false;
    // device-side GLS (p > 1 trend columns): Gram matrix of [ft | yt] and its factor, all (rhs_pad x rhs_pad)
    double *d_gneg = nullptr, *d_gram = nullptr, *d_gdinv = nullptr, *d_gramP = nullptr, *d_beta = nullptr,
           *d_part = nullptr;
    int *d_ginfo = nullptr;
    double *h_gram = nullptr, *h_part = nullptr, *h_beta = nullptr;  // pinned
    int *h_ginfo = nullptr;                                          // pinned
    bool gls_enqueued = false;  // this evaluation took the device GLS route (set by enqueue_eval)
    double *h_coef = nullptr;  // pinned
    double *h_rows = nullptr;  // pinned: q x n_pad solved RHS rows (ft^T, yt^T)
    double *h_diag = nullptr;  // pinned: n
    double *h_vec = nullptr;   // pinned: n_pad
    bool block_inv_ready = false;  // dW holds the inverse blocks of the factor now in M (launched ahead of the host's GLS: finalize)
    int *h_info = nullptr;     // pinned: [0] the factorisation's info, [1..8] the abort word + diagnostics of its chain launches
    // theta-gradient scratch (lazy, gp_fit.hip): the workgroups' partial sums, the reduced sums, their pinned copy
    double *d_gpart = nullptr, *d_gout = nullptr, *h_gout = nullptr;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    GemmTrace trace;
};

// outcome of one start of a multistart optimisation: objective (-likelihood), its minimiser in log10 theta, evaluations
struct StartResult {
    double f;
    std::vector<double> x;
    int64_t evals;
};

struct EvalResult {
    double lkh = -std::numeric_limits<double>::infinity();
    int status = EGX_STATUS_OK;
    double sigma2n = 0.0;         // rho^2 / n in normalised units
    std::vector<double> beta;     // p
    bool rho_on_device = false;   // device GLS: rho already sits (zero padded) in the workspace's d_rhs
    std::vector<double> rho;      // n (host GLS)
    std::vector<double> ft;       // n x p row-major
    std::vector<double> ft_qr_r;  // p x p row-major
};

}  // namespace egx

namespace egx {
// Slabs shared by the members of a GROUP of models (egx_gp_create_group): k models of one shape -- the experts of a mixture
// (crates/moe/src/algorithm.rs:167-177), an optimiser's objective and constraint surrogates -- whose matrices sit at the
// fixed strides a lock-step launch needs, although every member has its own training set.  Freed with the last member.
struct GroupSlabs {
    int device = 0, k = 0;
    double *M = nullptr, *D = nullptr;
    int *I = nullptr;
    size_t bytes_M = 0, bytes_D = 0, bytes_I = 0;
    ~GroupSlabs();  // gp_host.hip: the slabs go to the resource pool (the next group of this shape adopts them), or are freed
};
}  // namespace egx

struct egx_gp {
    int device = 0;
    int n = 0, d = 0, p = 0, h = 0, corr = 0, mean = 0;
    double nugget = 0.0;
    int n_pad = 0, rhs_pad = 0, m_tot = 0, q = 0;
    int64_t ld = 0;
    bool has_w = false;
    bool gls_device = false;  // p >= 2: GLS through the Gram matrix on the device (gp_host.hip finish_eval)
    std::vector<double> w_star;  // d x h
    std::vector<double> x_raw, y_raw, xnorm, x_mean, x_std, ynorm, F;
    double y_mean = 0.0, y_std = 1.0;
    double *d_xT = nullptr;    // d x n_pad (k-major normalised inputs, zero padded), then the same times the fit's coefficients
    double *d_rhsT = nullptr;  // q x n_pad: columns of F then y (normalised), as rows
    std::vector<egx::Workspace> ws;
    // one allocation each for all workspaces' matrices, tile inverses and failure flags (strides in elements)
    double *slab_M = nullptr, *slab_D = nullptr;
    int *slab_I = nullptr;  // one failure flag per workspace, then (from sync_off on) stride_S hand-off words per workspace
                            // for the pipelined chain kernel (kernels_pipe.hip; zeroed by every factorisation)
    int64_t stride_M = 0, stride_D = 0, stride_S = 0, sync_off = 0;
    std::shared_ptr<egx::GroupSlabs> group;  // member of a group: slab_M / slab_D / slab_I are VIEWS of slot `group_slot`
    int group_slot = -1;
    int lockstep = 1;  // candidates of a likelihood batch factored in lock-step (consecutive workspaces), <= ws.size()
    egx::PotrfSchedule sched;  // how this handle factors: decided at create / egx_gp_set_lockstep (schedule_for), kept by shrink
    // exclusive for everything that touches the fitted state or all workspaces; SHARED for egx_gp_likelihood, whose
    // concurrent callers (the reference's rayon multistart closures, algorithm.rs:928-945) each take a workspace
    // from the pool below
    std::shared_mutex mu;
    std::mutex pool_mu;
    std::condition_variable pool_cv;
    std::vector<char> ws_busy;
    // fitted state (lives in ws[0])
    std::atomic<bool> fitted{false};  // read by concurrent egx_gp_likelihood callers while another one clears it
    std::vector<double> theta;  // h
    double likelihood = 0.0, sigma2 = 0.0;
    std::vector<double> beta, gamma, ft, ft_qr_r;
    std::vector<double> fit_coef;
    int fit_hcols = 1;
    double *d_gamma = nullptr;  // n_pad
    double *d_fit_coef = nullptr;  // d x hcols coefficients of the fit, then x_mean (d) | x_std (d): dev_xnorm()
    // theta-gradient scratch (allocated on first use, gp_fit.hip): C^-T of `slab_W_count` candidates at a fixed stride of
    // n_pad^2 doubles (lock-step), |w_star| on the device (KPLS + Matern)
    double *slab_W = nullptr, *d_wabs = nullptr;
    int slab_W_count = 0;
    // x-gradient state (lazy, per fitted factor): d_W = C^-T of the FITTED factor and
    // -R^-1 F = -C^-T ft as an (n_pad x rhs_pad) matrix
    double *d_W = nullptr;
    double *d_neg_invkf = nullptr;
    std::vector<double> h_neg_invkf;  // host copy (n x p) for the single-point path
    // device scratch of the single-point path, allocated once (a hipMalloc per call would cost more than the kernels)
    double *sp_R = nullptr, *sp_P = nullptr, *sp_y = nullptr, *sp_z = nullptr, *sp_wt = nullptr, *sp_out = nullptr,
           *sp_xq = nullptr;
    int sp_nsplit = 0;
    int small_var_calls = 0;  // single-point predict_var calls since the fit: the third one builds W = C^-T
    uint64_t fit_epoch = 0, winv_epoch = ~(uint64_t)0;
    uint64_t winv_fail_epoch = ~(uint64_t)0;  // fit for which the C^-T cache could not be built (batched path serves it)
    egx_timings timings{};
};

namespace egx {
// the two halves of the multistart COBYLA fit (gp_fit.hip): the starts s = rank, rank + world, ... of this rank, then the
// reduction over ALL starts (results of the other ranks filled in by the caller) and the final factorisation
int fit_run_starts(egx_gp *gp, const double *theta_base, const std::vector<int> &active, const double *theta0s,
                   int64_t n_starts, const double *lo, const double *hi, int64_t bounds_len, int64_t max_eval, int rank,
                   int world, std::vector<StartResult> &results);
int fit_reduce_finalize(egx_gp *gp, const double *theta_base, const std::vector<int> &active, const double *theta0s,
                        const std::vector<StartResult> &results, int64_t *n_evals_out);
}  // namespace egx

// x_mean (d) | x_std (d) on the device, behind the coefficients of the fit in the same allocation
// the training inputs times the coefficients of the fit (d x n_pad, k-major; valid when fit_hcols == 1), behind d_xT in
// the same allocation: what the scalar-row prediction kernel reads (kernels_corr.hip k_predict_mean_srow)
// ints of slab_I for `nws` workspaces, and workspace i's hand-off words in it
inline size_t slab_I_ints(const egx_gp *gp, int nws) { return (size_t)egx::round_up(nws, 64) + (size_t)gp->stride_S * (size_t)nws; }
inline int *dev_sync(const egx_gp *gp, int i) { return gp->slab_I + gp->sync_off + (int64_t)i * gp->stride_S; }
inline double *dev_xs_fit(const egx_gp *gp) { return gp->d_xT + (size_t)gp->d * gp->n_pad; }
inline double *dev_xnorm(const egx_gp *gp) { return gp->d_fit_coef + (size_t)gp->d * (gp->has_w ? gp->h : 1); }

namespace egx {

const std::string &last_error_string();  // this thread's message (egx_last_error)
int set_device(const egx_gp *gp);
int make_coef(const egx_gp *gp, const double *theta, int64_t theta_len, std::vector<double> &coef, int &hcols,
              std::vector<double> *theta_full);
int enqueue_eval(egx_gp *gp, Workspace &w, const std::vector<double> &coef, int hcols);
// `count` evaluations on the consecutive workspaces w0 .. w0 + count - 1, factored in lock-step on the streams of w0
// W0 != nullptr: the `count` buffers W0 + j n_pad^2 receive C^-T of the candidates, riding along the factorisation
int enqueue_eval_group(egx_gp *gp, int w0, int count, const std::vector<double> *coefs, int hcols, double *W0 = nullptr);
// the same for `count` MODELS of one group (consecutive slots): member j evaluates on its own workspace 0, its own training set
int enqueue_eval_members(egx_gp *const *gps, int count, const std::vector<double> *coefs, int hcols);
int finish_eval(egx_gp *gp, Workspace &w, EvalResult &out, int keep);  // 0 scalars, 1 fitted state, 2 rho only
void record_timings(egx_gp *gp, Workspace &w, double host_ms, double solve_ms);
bool has_nan(const double *theta, int64_t len);
int eval_one(egx_gp *gp, int widx, const double *theta, int64_t theta_len, EvalResult &res, bool keep);
int backward_solve(egx_gp *gp, Workspace &w);
int do_finalize(egx_gp *gp, const double *theta, int64_t theta_len);
// Where a batch takes its candidates from: the sequence 0 .. k-1 (nullptr), a rank's static shard or the node-wide
// counter of a dynamic sweep (sweep.hip).  pull() hands out up to `want` candidate indices, 0 = exhausted.
struct CandidateSource {
    virtual int pull(int want, int64_t *out) = 0;
    virtual ~CandidateSource() = default;
};
// Evaluates the candidates `src` hands out (rows of thetas, k x theta_len) pipelined over the handle's workspaces;
// results go to lkh[c] / status[c] of the candidate's own index, evaluated[c] (optional, k chars) is set for them.
int likelihood_batch_core(egx_gp *gp, const double *thetas, int64_t k, int64_t theta_len, double *lkh, int32_t *status,
                          CandidateSource *src = nullptr, char *evaluated = nullptr);
// likelihood + dL/dtheta of k candidates, pipelined over the workspaces in lock-step slots (gp_fit.hip)
int likelihood_grad_batch_core(egx_gp *gp, const double *thetas, int64_t k, int64_t theta_len, double *lkh, double *grad,
                               int32_t *status);
// gp_predict.hip
int predict_impl(egx_gp *gp, const double *xq, int64_t m, double *yout, double *vout);
int xgrad_impl(egx_gp *gp, const double *xq, int64_t m, double *gy, double *gv);

}  // namespace egx
