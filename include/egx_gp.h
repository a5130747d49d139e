/*
 * egx_gp.h -- C ABI of libegx_gp_hip.so: MI355X-native (gfx950) kriging hot path.
 *
 * This is the drop-in boundary for egobox-gp's fit / reduced-likelihood /
 * predict / predict_var path.  The reference (relf/egobox 0.34.0) has no FFI
 * today: its seams are Rust traits and one cargo-feature `cfg`.  Every entry
 * point below names the reference interface it replaces (paths relative to the
 * reference checkout).  A Rust `extern "C"` shim implementing those traits on
 * top of this header is shown in INTEGRATION.md.
 *
 * Conventions
 *   - all arrays are caller-owned HOST memory, f64, C-contiguous row-major
 *     (ndarray standard layout); the library owns device memory behind the
 *     opaque handle; no callbacks cross the boundary.
 *   - every function returns an egx_rc (0 = success); on failure
 *     egx_last_error() returns a thread-local message.
 *   - numerical failures of ONE likelihood evaluation are VALUES, not errors:
 *     they come back in the per-candidate `status` (egx_status), exactly as the
 *     reference swallows them into +inf during optimisation
 *     (crates/gp/src/algorithm.rs:885-896).  Only egx_gp_finalize / egx_gp_fit
 *     turn a bad status into an error return, as the reference's final
 *     `reduced_likelihood(...)?` does (algorithm.rs:968).
 *   - a handle is safe to call from several host threads (calls are
 *     serialised per handle); distinct handles are independent.
 *   - there is NO CPU fallback: without a gfx950 device egx_gp_create fails
 *     with EGX_ERR_NO_DEVICE.
 */
#ifndef EGX_GP_H
#define EGX_GP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EGX_GP_ABI_VERSION 2

typedef struct egx_gp egx_gp; /* opaque: one training set resident on one GPU */

/* return codes */
typedef enum {
    EGX_SUCCESS = 0,
    EGX_ERR_INVALID_VALUE = 1, /* GpError::InvalidValueError, crates/gp/src/errors.rs:38 (and the
                                  panics of algorithm.rs:835-837, 907-911 turned into errors) */
    EGX_ERR_NO_DEVICE = 2,     /* no gfx950 device / HIP runtime unusable */
    EGX_ERR_HIP = 3,           /* a HIP call failed (message has the HIP error string) */
    EGX_ERR_NOT_FITTED = 4,    /* predict* / get_inner before finalize/fit */
    EGX_ERR_LINALG = 5,        /* GpError::LinalgError (not positive definite), errors.rs:19 */
    EGX_ERR_LIKELIHOOD = 6,    /* GpError::LikelihoodComputationError, errors.rs:12 */
    EGX_ERR_UNSUPPORTED = 7,
    EGX_ERR_PEER = 8           /* collective calls: another rank failed, or did not answer in time */
} egx_rc;

/* per-evaluation status (the value channel) */
typedef enum {
    EGX_STATUS_OK = 0,
    EGX_STATUS_NOT_POSITIVE_DEFINITE = 1, /* cholesky()? failed, algorithm.rs:1004 */
    EGX_STATUS_ILL_CONDITIONED_FT = 2,    /* algorithm.rs:1022-1026 */
    EGX_STATUS_ILL_CONDITIONED_F = 3,     /* algorithm.rs:1015-1020 */
    EGX_STATUS_NAN_THETA = 4,             /* algorithm.rs:885-891 */
    EGX_STATUS_RANK_FAILED = 5            /* sweep only: the rank that held this candidate failed (the call returns
                                             EGX_ERR_PEER on every other rank, the failing rank returns its own error) */
} egx_status;

/* crates/gp/src/correlation_models.rs: the four CorrelationModel impls */
typedef enum {
    EGX_CORR_SQUARED_EXPONENTIAL = 0, /* :91-104  */
    EGX_CORR_ABSOLUTE_EXPONENTIAL = 1, /* :185-196 */
    EGX_CORR_MATERN32 = 2,            /* :277-286, :326-353 */
    EGX_CORR_MATERN52 = 3             /* :446-455, :497-523 */
} egx_corr;

/* crates/gp/src/mean_models.rs: the three RegressionModel impls */
typedef enum {
    EGX_MEAN_CONSTANT = 0, /* :42-44   */
    EGX_MEAN_LINEAR = 1,   /* :68-71   */
    EGX_MEAN_QUADRATIC = 2 /* :97-104  */
} egx_mean;

/* Builder state that reaches the hot path: GpValidParams, crates/gp/src/parameters.rs:93-121 */
typedef struct {
    int32_t corr;         /* egx_corr */
    int32_t mean;         /* egx_mean */
    double nugget;        /* default 100*f64::EPSILON, parameters.rs:118 */
    int32_t device;       /* HIP device ordinal; -1 = current device */
    int32_t n_workspaces; /* correlation-matrix workspaces (concurrent likelihood evaluations), >= 1 */
    const double *w_star; /* optional KPLS rotations (d x kpls_dim), algorithm.rs:843-855; NULL = identity */
    int64_t kpls_dim;     /* columns of w_star (ignored when w_star == NULL) */
} egx_gp_config;

/* ---- library level ------------------------------------------------------ */
int32_t egx_abi_version(void);
const char *egx_last_error(void); /* thread-local, never NULL */
int32_t egx_device_count(void);   /* number of visible HIP devices (0 when none) */
void egx_gp_config_default(egx_gp_config *cfg);
/* Destroyed handles leave their device resources (the correlation-matrix workspaces with their streams and pinned
 * buffers, the training-set buffers) in a per-process pool keyed by the handle's shape (device, padded n, d, trend
 * columns, workspaces): the reference creates a new model per fit (GpValidParams::fit is one-shot, algorithm.rs:785-794;
 * egobox-moe trains one per expert, crates/moe/src/algorithm.rs:167-177), and the next egx_gp_create of that shape
 * starts warm instead of paying a multi-GiB allocation and a first factorisation on untouched memory.  Bounded by
 * EGX_POOL_MAX_GB (environment, default 48, 0 = no pool).  egx_trim frees everything cached and returns the bytes.
 * The same holds for GROUPS (egx_gp_create_group): the group's slabs are pooled by their three sizes, every member's workspace
 * and training-set buffers under the member's shape -- the next group of that shape adopts them (hits: k members + 1).
 * Streams carry no shape: idle workspaces (pooled or freed) leave theirs in a per-device free list and a handle of ANY shape
 * takes them from there (the runtime needs ~3 ms to create a stream, a workspace has four); egx_trim destroys the idle ones. */
int64_t egx_trim(void);
void egx_pool_stats(int64_t *cached_bytes, int64_t *hits, int64_t *misses);
/* Chain launches (the serial chain of a factorisation as one persistent launch) bound every device-side
 * wait by EGX_PIPE_TIMEOUT_MS.  A wait that runs out -- waves pre-empted while several processes share the GPU, a debugger --
 * is not a numerical event and the reference's cholesky() (crates/gp/src/algorithm.rs:1004) knows no such failure (the
 * objective only ever sees numerical errors, :893-896): the evaluation is run ONCE more, alone, by separate launches of the
 * same handle, and only a second failure is reported as EGX_ERR_HIP ("pipe_retry" = 0 / EGX_PIPE_RETRY=0: report at once).
 * Process-wide counters since start-up: *aborted = evaluations whose chain launch ran into the bound, *retried = those that
 * were run again.  Either pointer may be NULL. */
void egx_chain_stats(int64_t *aborted, int64_t *retried);
/* The factorisation's scheduling settings at run time (the EGX_* environment variables of DESIGN.md section 4, read
 * once at start-up; here by name without the prefix, lower case):
 *   "potrf_group"      panels per trailing update (0 = by size)          "stream_min"  tiles from which the stream kernel is used
 *   "gemm_small"       tiles below which the 64 x 64 kernel is used      "look_min"    trailing columns down to which the chain looks ahead
 *   "potrf_left"       left-looking updates 0 never / 1 per handle / 2   "lur_side"    look-ahead update beside (1) or in front of (0) the trailing one
 *   "pipe"             chain launches 0 never / 1 per handle / 2 per group of panels only
 *   "pipe_timeout_ms"  bound of every device-side wait of a chain launch (default 2000; exceeded = one retry by separate
 *                      launches, see egx_chain_stats -- never a hang)
 *   "pipe_retry"       1 (default): that retry; 0: EGX_ERR_HIP at once
 *   "trsv_fused"       1 (default): the back-substitution gamma = C^-T rho (crates/gp/src/algorithm.rs:1034) is ONE launch whose
 *                      workgroups hand the blocks' solutions to each other on the device (waits bounded by "pipe_timeout_ms": after a
 *                      wait that ran out the launch-per-block form is run once, counted by egx_chain_stats; EGX_ERR_HIP only if that
 *                      fails too or with "pipe_retry" = 0 -- never a wrong gamma); 0: one launch per 256-column block (rounds 1-5).
 *                      The same sums in the same order either way: the same bits
 * (the test hook "pipe_stall" exists only in the test build of the library, -DEGX_TEST_HOOKS).  They move launches between streams and kernels between tile shapes, never the
 * arithmetic inside a kernel; "potrf_group" / "stream_min" / "gemm_small" / "potrf_left" / "pipe" change which kernel
 * updates a block, or the order in which updates are summed, hence the rounding -- and "potrf_left" / "pipe" /
 * "potrf_group" enter a handle's schedule (egx_gp_get_schedule) when it is CREATED or its lock-step width is set, not
 * later.  Process-wide, atomic; meant for A/B runs, not for use while evaluations are in flight.  *previous (optional)
 * receives the old value.  Used by bench.py's roofline leg ("lur_side" = 0: clean per-launch durations) and the tests. */
int32_t egx_set_tuning(const char *knob, int32_t value, int32_t *previous);

/* ---- host-side helpers (no device needed) -------------------------------- */
/* utils.rs:45-54 normalize(): column mean, sample std (ddof=1), zero std -> 1. */
int32_t egx_normalize(const double *x, int64_t n, int64_t d, double *xnorm /*n*d*/,
                      double *mean /*d*/, double *std /*d*/);
/* mean_models.rs value(): number of basis columns p for input dimension d. */
int64_t egx_regression_ncols(int32_t mean, int64_t d);
/* mean_models.rs value(): F (n x p) from (normalised) x (n x d). */
int32_t egx_regression_basis(int32_t mean, const double *x, int64_t n, int64_t d, double *f /*n*p*/);

/* ---- handle lifetime ------------------------------------------------------
 * Replaces the theta-independent part of GpValidParams::fit
 * (crates/gp/src/algorithm.rs:795-866): copies x (n x d) and y (n), normalises
 * both, builds the regression basis, uploads, allocates workspaces. */
int32_t egx_gp_create(const egx_gp_config *cfg, const double *x, const double *y, int64_t n,
                      int64_t d, egx_gp **out);
void egx_gp_destroy(egx_gp *gp);
/* dims: n, d (dims().0, algorithm.rs:437-439), p = basis columns, h = theta length */
int32_t egx_gp_dims(const egx_gp *gp, int64_t *n, int64_t *d, int64_t *p, int64_t *h);
/* the handle's own copy of the raw training set (GaussianProcess::training_data, algorithm.rs:969-978: the fitted model
 * owns copies of x and y); x_out (n*d), y_out (n), either may be NULL */
int32_t egx_gp_get_training_data(const egx_gp *gp, double *x_out, double *y_out);

/* ---- likelihood (the unit COBYLA multiplies) ------------------------------
 * One evaluation of `reduced_likelihood(fx, corr.value(d, theta, w), ...)`
 * (algorithm.rs:892-896, 988-1056).  theta_len is h or 1 (broadcast,
 * algorithm.rs:829-838).  *lkh is the reduced likelihood (NOT negated);
 * on status != 0 it is -inf (the reference's objective is then +inf).
 * Thread-safe: concurrent callers each take a workspace from the handle's pool (n_workspaces).  The fitted factor
 * lives in workspace 0: with n_workspaces > 1 a fitted model is never disturbed (callers wait for another workspace);
 * with a single workspace an evaluation overwrites it and the handle must be finalized again before predicting. */
int32_t egx_gp_likelihood(egx_gp *gp, const double *theta, int64_t theta_len, double *lkh,
                          int32_t *status);
/* k candidates, thetas is (k x theta_len); the multistart / theta-sweep unit of
 * algorithm.rs:928-945 (rayon par_iter over starts).  Pipelines over the handle's workspaces; a fitted model keeps
 * workspace 0 (and stays fitted) when n_workspaces > 1. */
int32_t egx_gp_likelihood_batch(egx_gp *gp, const double *thetas, int64_t k, int64_t theta_len,
                                double *lkh /*k*/, int32_t *status /*k*/);
/* Width of the LOCK-STEP groups of egx_gp_likelihood_batch / egx_sweep_likelihood: that many candidates (consecutive
 * workspaces) are factored by ONE launch sequence -- the candidates of a sweep have the same n, hence the same
 * schedule; the serial chain of the factorisation then costs its latency once per group and every launch has `width`
 * times the tiles.  1 = every candidate on its own stream set (round 2's pipeline); 0 = the default
 * (min(n_workspaces, 12) below a padded n of 14336 -- one slot: the serial chain is what such an evaluation waits for --,
 * from there on 4, and 8 from 16 workspaces on); at most n_workspaces.  A candidate's result does
 * not depend on its companions or on how many of them share its launch (same kernels, same arithmetic: bit-identical);
 * it does not depend on the width either, with ONE exception: a handle with a padded n >= 14336 and a width >= 8 factors
 * LEFT-looking over its panel groups (two long-K updates per tile of the factor instead of one per earlier group), a
 * different order of the same sums than the right-looking schedule every other handle uses (EGX_POTRF_LEFT=0 switches
 * it off, 2 forces it everywhere).  Whatever the schedule, it is a property of the HANDLE's shape and setting, so all
 * evaluations of a handle -- and of its replicas on other ranks -- agree bit for bit. */
int32_t egx_gp_set_lockstep(egx_gp *gp, int32_t width);
int32_t egx_gp_get_lockstep(const egx_gp *gp);
/* Which schedule the handle's factorisations follow -- decided when the handle is created or its lock-step width is set,
 * from its padded size, lock-step width and number of workspaces, kept by egx_gp_shrink -- so that a caller can tell two
 * handles that will not give the same bits apart (a sweep on a 16-workspace handle and a re-evaluation of its winner on a
 * 1-workspace handle at n = 16384 differ by 1e-10 relative).  out[0..5] = { left-looking group updates, left-looking
 * C^-T rider, pipelined chain launches (the chain of a panel group -- diagonal blocks, panel solves, in-group updates: the
 * panel step of `cholesky()`, crates/gp/src/algorithm.rs:1004 -- as one persistent launch), whole factorisation as one such
 * launch, panels per group, lock-step width }; out_len >= 6.  out[6] (when out_len >= 7): the whole factorisation as ONE FLOW
 * launch -- critical stage lists + bulk-class rounds per column, csrc/pipe_flow.h; round 6: a lone one-workspace handle that
 * is not a member of a group, padded size 5376 .. 14080 (value 1); from 14336 on (value 2) such a handle factors right-looking by
 * separate launches and hands its LAST 6144 .. 7167 columns, fully updated, to one flow launch.  (An evaluation WITH the
 * theta-gradient factors by separate launches on such a handle too -- the rider C^-T runs beside them, behind a flow launch it
 * would wait: the likelihood egx_gp_likelihood_grad returns there is egx_gp_likelihood's to rounding, 1e-12 relative, not bit for
 * bit.)  Further slots are set to 0. */
int32_t egx_gp_get_schedule(const egx_gp *gp, int32_t *out, int32_t out_len);
/* Give back what only an optimisation needed: the handle keeps its first n_keep (>= 1) workspaces -- workspace 0 holds the
 * fitted factor, which survives -- and frees the others together with the theta-gradient's scratch.  A tuned fit runs its
 * multistart on up to 12 workspaces (2 GiB each at n = 16384); the fitted model that stays resident (an EGO objective, an
 * expert of a mixture) needs one.  The schedule of the handle (egx_gp_get_schedule) does not change -- the model keeps giving
 * the bits it gave before --; the lock-step width is capped by the workspaces that are left. */
int32_t egx_gp_shrink(egx_gp *gp, int32_t n_keep);
/* NEW capability (the reference has no theta-gradient, algorithm.rs:880: the objective closure ignores `_gradient`):
 * the reduced likelihood of algorithm.rs:988-1056 AND dL/dtheta (length h) at one theta; validated against the oracle's
 * closed form and by finite differences of the parity-checked likelihood.  Needs the factor, C^-T (n^2 doubles of
 * scratch per candidate in flight, allocated on first use) and R^-1 (written over the factor).  A fitted model with
 * n_workspaces > 1 keeps workspace 0 and stays fitted; with a single workspace the call un-fits it. */
int32_t egx_gp_likelihood_grad(egx_gp *gp, const double *theta, int64_t theta_len, double *lkh,
                               double *dlkh_dtheta /*h*/, int32_t *status);
/* k candidates at once (thetas is k x theta_len, grads k x h row-major): the gradient counterpart of
 * egx_gp_likelihood_batch -- the objective + gradient evaluations of a gradient-based multistart (the rayon tasks of
 * algorithm.rs:928-945).  Candidates are pipelined over the workspaces; the candidates of a lock-step slot
 * (egx_gp_set_lockstep, at most 16) run EVERY stage -- factorisation, C^-T, R^-1, the trace kernel -- as one launch
 * sequence.  A candidate's (likelihood, gradient) are bit for bit what egx_gp_likelihood_grad returns for it alone. */
int32_t egx_gp_likelihood_grad_batch(egx_gp *gp, const double *thetas, int64_t k, int64_t theta_len,
                                     double *lkh /*k*/, double *dlkh_dtheta /*k x h*/, int32_t *status /*k*/);

/* ---- fit ------------------------------------------------------------------
 * ThetaTuning::Fixed fit (algorithm.rs:869-872, 966-978; python n_start=-1,
 * python/src/gp_mix.rs:202-208): one likelihood evaluation whose factor and
 * inner params stay resident.  Error return when the evaluation fails. */
int32_t egx_gp_finalize(egx_gp *gp, const double *theta, int64_t theta_len);
/* Lock-step across MODELS.  The reference's callers multiply independent models, not only candidates of one model: the
 * expert loop of egobox-moe (crates/moe/src/algorithm.rs:167-177: one GP per cluster, fitted one after the other) and
 * EGO's objective + constraint surrogates (crates/ego/src/solver/solver_impl.rs:370-391).
 *   egx_gp_create_group   k models of ONE shape (same n, d, trend, kernel, nugget: `cfg`; training sets x (k x n x d) and
 *                         y (k x n), one after the other) whose matrices live in one slab at the strides a lock-step launch
 *                         needs.  out[0..k) are ordinary handles with one workspace each -- every entry point works on them
 *                         and each is destroyed on its own (the slab goes with the last one).
 *   egx_gp_finalize_multi `fit` at fixed theta (thetas: k x theta_len, row j for gps[j]) of k DISTINCT handles: members of
 *                         one group in consecutive slots are factored by ONE launch sequence (up to 12 per sequence), any
 *                         other handle on its own -- the fitted models are bit for bit what egx_gp_finalize gives each of them
 *                         (the same kernels on the same values: a matrix never sees its companions).  The first error of
 *                         any model is returned after all of them have been finished.
 *   egx_gp_likelihood_multi  the same for the reduced likelihood alone (algorithm.rs:988-1056); status as egx_gp_likelihood.
 *                         NOTE: both calls evaluate on every member's workspace 0, where a fitted model's factor lives: a fitted
 *                         member is UN-FITTED by egx_gp_likelihood_multi too (egx_gp_likelihood keeps a fitted handle with spare
 *                         workspaces fitted; call egx_gp_finalize[_multi] again before predicting). */
int32_t egx_gp_create_group(const egx_gp_config *cfg, const double *x, const double *y, int64_t n, int64_t d, int32_t k,
                            egx_gp **out /*k*/);
int32_t egx_gp_finalize_multi(egx_gp *const *gps, int32_t k, const double *thetas /*k x theta_len*/, int64_t theta_len);
int32_t egx_gp_likelihood_multi(egx_gp *const *gps, int32_t k, const double *thetas /*k x theta_len*/, int64_t theta_len,
                                double *lkh /*k*/, int32_t *status /*k*/);
/* ThetaTuning::Full across MODELS (round 6): the tuned fit the expert loop runs per cluster (crates/moe/src/algorithm.rs:167-177
 * -> find_best_expert :209-262 -> GpValidParams::fit -> the multistart COBYLA of crates/gp/src/algorithm.rs:921-945), and EGO's
 * per-output surrogates (crates/ego/src/solver/solver_impl.rs:370-391), for k DISTINCT handles with the same number of
 * hyperparameters -- the members of a group (egx_gp_create_group) in consecutive slots are evaluated in lock-step: per round
 * of the k x n_starts COBYLA machines and per start index, the trial points of all models are ONE launch sequence.
 * theta0s is (k x n_starts x h) in linear theta units (model-major), lo / hi / bounds_len / max_eval as egx_gp_fit,
 * n_evals_out (k, optional) the evaluations per model.  Every member ends fitted exactly as egx_gp_fit on a one-workspace
 * handle of its training set would leave it (a machine only sees its own model's values, an evaluation never its companions):
 * theta*, likelihood and predictions are the same bits. */
int32_t egx_gp_fit_multi(egx_gp *const *gps, int32_t k, const double *theta0s /*k x n_starts x h*/, int64_t n_starts,
                         const double *lo, const double *hi, int64_t bounds_len, int64_t max_eval, int64_t *n_evals_out /*k*/);
/* ThetaTuning::Full fit (algorithm.rs:874-948): multistart derivative-free
 * maximisation of the likelihood over log10(theta) in [lo,hi], then finalize.
 * theta0s is (n_starts x h) in LINEAR theta units: the caller supplies the
 * starts (prepare_multistart, optimization.rs:26-71, stays on the caller's side
 * because its LHS stream is the caller's RNG).  bounds_len is h or 1. */
int32_t egx_gp_fit(egx_gp *gp, const double *theta0s, int64_t n_starts, const double *lo,
                   const double *hi, int64_t bounds_len, int64_t max_eval, int64_t *n_evals_out);

/* ThetaTuning::Partial (parameters.rs:24-32, algorithm.rs:822-826, 873-960): only the components listed in
 * active_idx (strictly increasing, n_active of them) are optimised, the others stay at theta_init (h values).
 * theta0s is (n_starts x n_active) in theta space, lo/hi have 1 or n_active entries;
 * maxeval per start = clamp(10 n_active, 25, max_eval). */
int32_t egx_gp_fit_partial(egx_gp *gp, const double *theta_init, const int64_t *active_idx, int64_t n_active,
                           const double *theta0s, int64_t n_starts, const double *lo, const double *hi,
                           int64_t bounds_len, int64_t max_eval, int64_t *n_evals_out);

/* Gradient-based variant (NEW: consumes the theta-gradient; same contract as egx_gp_fit otherwise):
 * projected L-BFGS on log10(theta) per start, ALL starts advanced in lock-step (a round's trial points are one
 * egx_gp_likelihood_grad_batch), best start wins, then finalize.  max_iter bounds the iterations per start;
 * *n_evals_out counts likelihood+gradient evaluations. */
int32_t egx_gp_fit_lbfgs(egx_gp *gp, const double *theta0s, int64_t n_starts, const double *lo,
                         const double *hi, int64_t bounds_len, int64_t max_iter, int64_t *n_evals_out);

/* ---- prediction (GaussianProcess::{predict, predict_var, predict_valvar},
 * algorithm.rs:253-307).  xq is (m x d) in ORIGINAL units. */
int32_t egx_gp_predict(egx_gp *gp, const double *xq, int64_t m, double *y /*m*/);
int32_t egx_gp_predict_var(egx_gp *gp, const double *xq, int64_t m, double *var /*m*/);
int32_t egx_gp_predict_valvar(egx_gp *gp, const double *xq, int64_t m, double *y, double *var);

/* ---- x-gradients of the predictions (what EGO's infill optimiser asks per point):
 * GaussianProcess::predict_gradients algorithm.rs:510-549, predict_var_gradients(_single) :555-617, 702-709,
 * predict_valvar_gradients :711-727; kernels' jacobian correlation_models.rs:106-123, 198-214, 355-413, 524-586,
 * trend jacobian mean_models.rs:50-52, 76-81, 110-128.  xq is (m x d) in ORIGINAL units, grad is (m x d)
 * row-major, d out / d xq_k in original units.  Batched over m (the reference loops point by point and redoes
 * R^-1 F and chol(F^T R^-1 F) for every point); the first variance-gradient call after a fit caches C^-T. */
int32_t egx_gp_predict_gradients(egx_gp *gp, const double *xq, int64_t m, double *grad /*m*d*/);
int32_t egx_gp_predict_var_gradients(egx_gp *gp, const double *xq, int64_t m, double *grad /*m*d*/);
int32_t egx_gp_predict_valvar_gradients(egx_gp *gp, const double *xq, int64_t m, double *grad_y /*m*d*/,
                                        double *grad_var /*m*d*/);

/* ---- fitted state download (GpInnerParams algorithm.rs:47-60 + accessors
 * theta()/variance()/likelihood() :413-431), for serde parity.  Any pointer may
 * be NULL (skipped).  r_chol is (n x n) lower with zero upper triangle;
 * ft_qr_r has a positive diagonal. */
typedef struct {
    double *theta;      /* h */
    double *likelihood; /* 1 */
    double *sigma2;     /* 1 : variance() = sigma2 * y_std^2 */
    double *beta;       /* p */
    double *gamma;      /* n */
    double *r_chol;     /* n*n */
    double *ft;         /* n*p */
    double *ft_qr_r;    /* p*p */
    double *x_mean;     /* d */
    double *x_std;      /* d */
    double *y_mean;     /* 1 */
    double *y_std;      /* 1 */
    double *xt_norm;    /* n*d */
    double *yt_norm;    /* n */
} egx_gp_inner_view;
int32_t egx_gp_get_inner(egx_gp *gp, const egx_gp_inner_view *view);
/* Inverse of egx_gp_get_inner: install a fitted state produced elsewhere (e.g. deserialised from the
 * reference's serde JSON, crates/moe/src/surrogates.rs:426-441 `load`) WITHOUT re-factoring: theta (h),
 * likelihood, sigma2, beta (p), gamma (n), r_chol (n*n, lower), ft (n*p), ft_qr_r (p*p) are read;
 * normalisation fields of the view are ignored (they are recomputed from the training data given to
 * egx_gp_create and must agree).  All eight pointers are required. */
int32_t egx_gp_set_inner(egx_gp *gp, const egx_gp_inner_view *view);

/* ---- kernel-level entry points ---------------------------------------------
 * The CorrelationModel::value seam (correlation_models.rs:19-58) fused with
 * DiffMatrix (utils.rs:80-104) and the scatter loop (algorithm.rs:997-1001):
 * r (n x n, full symmetric, diagonal 1+nugget) straight from normalised x.
 * theta is the per-input-dimension scale vector (length d, w = I). */
int32_t egx_corr_matrix(int32_t corr, const double *xnorm, int64_t n, int64_t d,
                        const double *theta, double nugget, double *r /*n*n*/);
/* _compute_correlation (algorithm.rs:372-380): r (m x n) between normalised
 * query points and normalised training points. */
int32_t egx_cross_corr(int32_t corr, const double *xq_norm, int64_t m, const double *xt_norm,
                       int64_t n, int64_t d, const double *theta, double *r /*m*n*/);
/* cholesky() (algorithm.rs:1004; LAPACK dpotrf with the blas feature :1077):
 * a (n x n, symmetric, lower triangle read) -> lower factor in place (upper
 * zeroed).  *info = 0, or 1-based index of the first non-positive pivot. */
int32_t egx_potrf(double *a, int64_t n, int32_t *info);

/* ==== multi-GPU theta sweep (BASELINE config 4) ================================
 * Replaces the rayon multistart of crates/gp/src/algorithm.rs:928-945 (independent likelihood evaluations, one
 * per start, arg-min reduce :942-945) and the serial expert loop of crates/moe/src/algorithm.rs:167-177 at node
 * scale: ONE PROCESS PER GPU, the training set replicated on every rank (4 MiB at n = 16384, d = 32), candidate
 * k evaluated by rank k mod world through egx_gp_likelihood_batch, and one RCCL all-gather (ncclAllGather over xGMI)
 * of {likelihood f64, status} -- 16 B per candidate -- so that every rank returns the full (k) result.
 * RCCL (librccl.so.1) is bound at run time on the first egx_sweep_* call; a single-GPU user never loads it.
 *
 * Rendezvous: rank 0 calls egx_sweep_unique_id and hands the EGX_SWEEP_ID_BYTES bytes to the other ranks by any
 * out-of-band channel (MPI_Bcast, a file, torch.distributed's store ...); all ranks then call egx_sweep_create
 * collectively.  world == 1 with nccl_id == NULL needs no RCCL at all; world == 1 WITH an id builds a one-rank
 * communicator (the same code path as world > 1). */
#define EGX_SWEEP_ID_BYTES 128 /* sizeof(ncclUniqueId) */
typedef struct egx_sweep egx_sweep;
int32_t egx_sweep_unique_id(void *id_out /*EGX_SWEEP_ID_BYTES*/);
/* cfg / x / y / n / d as egx_gp_create (cfg->device selects this rank's GPU; n_workspaces is raised to 2). */
int32_t egx_sweep_create(const egx_gp_config *cfg, const double *x, const double *y, int64_t n, int64_t d,
                         const void *nccl_id /*EGX_SWEEP_ID_BYTES or NULL*/, int32_t rank, int32_t world,
                         egx_sweep **out);
void egx_sweep_destroy(egx_sweep *sw);
/* this rank's GP handle (owned by the sweep): finalize / predict with the winning theta on it */
egx_gp *egx_sweep_handle(egx_sweep *sw);
/* rccl_ranks = ranks of the RCCL communicator (0 when none was built), rccl_version = ncclGetVersion code */
int32_t egx_sweep_info(const egx_sweep *sw, int32_t *rank, int32_t *world, int32_t *rccl_ranks, int32_t *rccl_version,
                       int64_t *n_allgathers);
/* COLLECTIVE: every rank passes the same thetas (k x theta_len); lkh / status (k) are complete on every rank.
 * Semantics per candidate as egx_gp_likelihood_batch (status is the value channel, failures are not errors). */
int32_t egx_sweep_likelihood(egx_sweep *sw, const double *thetas, int64_t k, int64_t theta_len, double *lkh /*k*/,
                             int32_t *status /*k*/);
/* Assignment of candidates to ranks for the following egx_sweep_likelihood calls: 0 = static (candidate c -> rank
 * c mod world; the default), 1 = dynamic (ranks pull the next candidate from a node-wide counter in POSIX shared
 * memory as their workspaces free up: candidates that are not positive definite return ~10x sooner than the others,
 * SURVEY 8e).  The results are identical either way (every evaluation is deterministic and independent of the rank
 * that runs it).  Every rank must select the same mode.  EGX_ERR_UNSUPPORTED when the shared memory is unavailable. */
int32_t egx_sweep_set_assignment(egx_sweep *sw, int32_t dynamic);
/* Balance of the last egx_sweep_likelihood: candidates evaluated by every rank (world entries, NULL = skip) and the
 * wall time this rank spent in its own evaluations (seconds, NULL = skip). */
int32_t egx_sweep_last_balance(const egx_sweep *sw, int64_t *per_rank /*world*/, double *local_eval_s);
/* FAILURE SAFETY of the collectives: a rank whose local work fails (HIP error, out of memory, ...) still takes part
 * in the all-gather with a poisoned payload and returns its own error afterwards; every other rank returns
 * EGX_ERR_PEER (the candidates of the failed rank come back with status EGX_STATUS_RANK_FAILED, all others are
 * valid).  A peer that never arrives is given EGX_SWEEP_TIMEOUT_S seconds (environment, default 1800), then the
 * communicator is aborted and the call returns EGX_ERR_PEER instead of blocking for ever. */
/* COLLECTIVE helper for the mixture-of-experts path (expert e on rank e mod world): recv (world x count) <- the
 * concatenation over ranks of send (count), e.g. per-expert mean / variance vectors before the recombination of
 * crates/moe/src/algorithm.rs:670-685. */
int32_t egx_sweep_allgather(egx_sweep *sw, const double *send, int64_t count, double *recv /*world*count*/);

/* COLLECTIVE tuned fit: egx_gp_fit with the starts sharded over the ranks (start s on rank s mod world; the reference runs
 * them as rayon tasks on one host, crates/gp/src/algorithm.rs:928-945).  Every rank passes the same arguments; one
 * all-gather carries each start's (objective, evaluations, minimiser); every rank reduces in start order (:942-945) and
 * finalizes ITS replica at the winner, so afterwards egx_sweep_handle(sw) is the same fitted model on every rank -- bit
 * for bit the model egx_gp_fit returns on one GPU (an evaluation gives the same bits wherever it runs).  n_evals_out: the
 * evaluations of all starts.  Failure semantics as egx_sweep_likelihood. */
int32_t egx_sweep_fit(egx_sweep *sw, const double *theta0s /*n_starts x h*/, int64_t n_starts, const double *lo,
                      const double *hi, int64_t bounds_len, int64_t max_eval, int64_t *n_evals_out);

/* ---- mixture-of-experts recombination (BASELINE config 5) ----------------------------------
 * GpMixture::predict_smooth / predict_var_smooth (crates/moe/src/algorithm.rs:411-423, 670-685):
 *     val = sum_e p_e y_e,  var = sum_e p_e^2 v_e        over all m points, every expert sees every point;
 * GpMixture::predict_hard / predict_var_hard (:879-935): a point is answered by the expert of its cluster,
 * argmax_e p_e (first maximum) -- routed once, ONE batched call per expert (the reference calls the expert per row).
 * `experts` are this process' fitted handles, expert_ids[e] in [0, n_experts) their columns in `probas`
 * (m x n_experts row-major: the responsibilities GaussianMixture::predict_probas returns, gaussian_mixture.rs:114-121).
 * xq is (m x d) in ORIGINAL units.  val / var (m): either may be NULL.
 * sw == NULL: single process, every expert is local.  sw != NULL: COLLECTIVE -- every rank passes the same probas /
 * xq and ITS OWN experts (expert e on rank e mod world in config 5; a rank may own none); the partial sums are exchanged
 * by one all-gather on the sweep's communicator and added in rank order, so every rank returns the same bits.  Failure
 * safety as for egx_sweep_likelihood (a failing rank still arrives, the others return EGX_ERR_PEER). */
int32_t egx_moe_predict_valvar(egx_sweep *sw, egx_gp *const *experts, const int32_t *expert_ids, int64_t n_local,
                               int64_t n_experts, const double *probas /*m*n_experts*/, const double *xq /*m*d*/,
                               int64_t m, int64_t d, int32_t smooth, double *val /*m*/, double *var /*m*/);

/* Responsibilities of a fitted Gaussian mixture at m points: GaussianMixture::predict_probas (crates/moe/src/
 * gaussian_mixture.rs:114-121, with compute_log_prob_resp :231-251 and compute_log_gaussian_prob :253-283) -- the `probas`
 * egx_moe_predict_valvar consumes.  weights (k), means (k x d), precisions_chol (k x d x d row-major: (chol(cov_c)^-1)^T as
 * the reference stores it, :182-205; egx_gmx_precisions_chol builds it from covariances (k x d x d), EGX_ERR_LINALG for a
 * covariance that is not positive definite), heaviside_factor > 0 (:105-110: scales the precision factors by its power
 * -1/2; 1 = none).  xq (m x d) in original units, probas (m x k).  One cluster: all ones (:115-116).  Runs on `device`
 * (< 0: the calling thread's current device); the d x d factorisations of egx_gmx_precisions_chol are host arithmetic. */
int32_t egx_gmx_precisions_chol(const double *covariances, int64_t k, int64_t d, double *precisions_chol);
int32_t egx_gmx_predict_probas(int32_t device, const double *weights, const double *means, const double *precisions_chol,
                               int64_t k, int64_t d, double heaviside_factor, const double *xq, int64_t m, double *probas);

/* d p_c(x) / d x of those responsibilities: GaussianMixture::predict_probas_derivatives (gaussian_mixture.rs:127-170),
 * dprobas is (m x k x d) row-major; on the device, one lane per point (needs 3 d + k <= 320). */
int32_t egx_gmx_predict_probas_derivatives(int32_t device, const double *weights, const double *means,
                                           const double *precisions_chol, int64_t k, int64_t d, double heaviside_factor,
                                           const double *xq, int64_t m, double *dprobas /*m*k*d*/);
/* x-gradients of the mixture's mean and variance: GpMixture::predict_gradients_smooth / predict_var_gradients_smooth
 * (crates/moe/src/algorithm.rs:691-783:  sum_e p_e grad y_e + p'_e y_e ,  sum_e p_e^2 grad v_e + 2 p_e p'_e v_e) and
 * predict_gradients_hard / predict_var_gradients_hard (:942-1010: the expert of argmax_e p_e) -- what EGO's infill
 * optimiser asks of a mixture surrogate.  Arguments and sharding as egx_moe_predict_valvar (a rank passes its own
 * experts; one all-gather of the partial (m x d) sums, added in rank order); dprobas (m x n_experts x d, from
 * egx_gmx_predict_probas_derivatives) is read in smooth mode only and may be NULL for a single expert.  grad_val /
 * grad_var are (m x d) row-major in original units; either may be NULL.  Every expert gets ONE batched call per
 * quantity (the reference calls it once per row). */
int32_t egx_moe_predict_valvar_gradients(egx_sweep *sw, egx_gp *const *experts, const int32_t *expert_ids, int64_t n_local,
                                         int64_t n_experts, const double *probas /*m*n_experts*/,
                                         const double *dprobas /*m*n_experts*d*/, const double *xq /*m*d*/, int64_t m,
                                         int64_t d, int32_t smooth, double *grad_val /*m*d*/, double *grad_var /*m*d*/);

/* ---- measurement ------------------------------------------------------------
 * HIP-event durations (ms) of the stages of the most recent likelihood /
 * finalize call on workspace 0, measured on the stream the kernels ran on. */
typedef struct {
    double corr_build_ms; /* K1 */
    double potrf_ms;      /* K3 incl. fused forward solves (K4) */
    double potrf_syrk_ms; /* sum of the HIP-event durations of the big-tile trailing-update launches */
    double solve_ms;      /* gamma back-substitution */
    double host_ms;       /* GLS / QR / reductions on the host */
    double total_ms;
    int64_t potrf_flops;  /* n^3/3 algorithmic */
    int64_t corr_bytes;   /* 8*n*d + 8*n(n+1)/2 algorithmic */
    int64_t syrk_launches; /* number of big-tile (128x128, 512-thread) trailing-update launches timed */
    int64_t syrk_flops;    /* their algorithmic flops: sum of 2*K*ncols*(ncols+1)/2 (lower triangle) */
} egx_timings;
int32_t egx_gp_last_timings(const egx_gp *gp, egx_timings *t);

/* self-test of the FP64 MFMA fragment layout (16x16x4): returns max abs error of
 * a 16x16x16 product computed with v_mfma_f64_16x16x4_f64 vs the host. */
int32_t egx_mfma_probe(double *max_abs_err);


/* ==== sparse Gaussian process (FITC / VFE), SURVEY 8f rank 4 =================================
 * SparseGaussianProcess / SgpValidParams, crates/gp/src/sparse_algorithm.rs:145-168, 422-831 and
 * sparse_parameters.rs.  Works on RAW x / y (the reference does not normalise here) with a zero trend;
 * parameters are theta (d, or 1 broadcast), the process variance sigma2 and the homoscedastic noise variance.
 * z are the nz inducing points (Inducings::Located; a Randomized(n) draw is the caller's shuffle of x). */
typedef struct egx_sgp egx_sgp;
typedef enum { EGX_SGP_FITC = 0, EGX_SGP_VFE = 1 } egx_sgp_method; /* SparseMethod, sparse_parameters.rs:59-67 */
typedef struct {
    int32_t corr;   /* egx_corr */
    int32_t method; /* egx_sgp_method */
    double nugget;  /* added to diag(Kmm); default 100 * f64::EPSILON */
    int32_t device; /* HIP ordinal, -1 = current */
} egx_sgp_config;
void egx_sgp_config_default(egx_sgp_config *cfg);
int32_t egx_sgp_create(const egx_sgp_config *cfg, const double *x /*n*d*/, const double *y /*n*/, int64_t n, int64_t d,
                       const double *z /*nz*d*/, int64_t nz, egx_sgp **out);
void egx_sgp_destroy(egx_sgp *sgp);
int32_t egx_sgp_dims(const egx_sgp *sgp, int64_t *n, int64_t *d, int64_t *nz);
/* reduced_likelihood (fitc :695-766 / vfe :769-831) at given parameters; status as for egx_gp_likelihood
 * (1: Kmm or I + V diag(beta) V^T not positive definite -- the reference unwraps; 4: NaN / non-positive parameters) */
int32_t egx_sgp_likelihood(egx_sgp *sgp, const double *theta, int64_t theta_len, double sigma2, double noise,
                           double *lkh, int32_t *status);
/* keep the state of one evaluation resident = a fit at fixed parameters */
int32_t egx_sgp_finalize(egx_sgp *sgp, const double *theta, int64_t theta_len, double sigma2, double noise);
/* SgpValidParams::fit :488-650: multistart derivative-free maximisation over log10 [theta.., sigma2, (noise)];
 * params0s (n_starts x np), lo / hi (np), np = d + 1 + (estimate_noise != 0); maxeval = clamp(10 d, 25, max_eval) */
int32_t egx_sgp_fit(egx_sgp *sgp, const double *params0s, int64_t n_starts, const double *lo, const double *hi,
                    int32_t estimate_noise, double noise_fixed, int64_t max_eval, int64_t *n_evals_out);
/* SparseGaussianProcess::predict :237-241, predict_var :245-257 (clamp at 1e-15, + noise) */
int32_t egx_sgp_predict(egx_sgp *sgp, const double *xq, int64_t m, double *y /*m*/);
int32_t egx_sgp_predict_var(egx_sgp *sgp, const double *xq, int64_t m, double *var /*m*/);
/* fitted state: theta (d), sigma2, noise, likelihood, WoodburyData vec (nz) and inv (nz*nz) :32-36; NULL = skip */
int32_t egx_sgp_get_state(egx_sgp *sgp, double *theta, double *sigma2, double *noise, double *likelihood,
                          double *w_vec, double *w_inv);

#ifdef __cplusplus
}
#endif
#endif /* EGX_GP_H */
