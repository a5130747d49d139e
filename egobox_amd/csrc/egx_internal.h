// Internal declarations shared by the HIP translation units of libegx_gp_hip.so.
// gfx950 (MI355X / CDNA4) only: wave64, FP64 MFMA 16x16x4, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/egx_gp.h"
#include "schedule.h"

namespace egx {

// ---- error plumbing --------------------------------------------------------
void set_error(const std::string &msg);
#define EGX_HIP_CHECK(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            ::egx::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));         \
            (void)hipGetLastError(); /* the runtime's last error is sticky: reset it */ \
            return EGX_ERR_HIP;                                                          \
        }                                                                                \
    } while (0)

// hipMalloc that gives the pool of destroyed handles' resources on the current device back before it reports
// out-of-memory (gp_host.hip): every device allocation of the library goes through it
hipError_t dev_malloc_bytes(void **p, size_t bytes);
template <typename T>
inline hipError_t dev_malloc(T **p, size_t bytes) {
    return dev_malloc_bytes(reinterpret_cast<void **>(p), bytes);
}

// ---- geometry ---------------------------------------------------------------
constexpr int kTile = 128;      // row/col granularity of the padded correlation matrix
constexpr int kNB = 256;        // outer Cholesky block (K of the trailing update)
constexpr int kDiagTile = 64;   // diagonal tile factored in registers by one wave
constexpr int kRhsPad = 128;    // rows appended below R for the fused forward solves
// (round 4: no cap on the input dimension d any more -- the correlation kernels stage the dimensions in chunks of 64;
//  the x-gradient kernels need d * (hcols + 5) <= 20480: coefficients and one training point in LDS)

inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// ---- kernels_corr.hip -------------------------------------------------------
// coef is (d x hcols) row-major: per-dimension scale (w = I: hcols = 1, coef[j] = theta_j;
// KPLS: sq-exp/abs-exp collapse to hcols = 1, Matern keeps theta_l*|w_jl|).
// xT is k-major (d x ldx): xT[k*ldx + i] = xnorm[i][k]; rows >= n are padding.
// xs_scratch (optional, d x ldx doubles): with it and hcols == 1 the scalar-row kernel runs on prescaled inputs
int launch_corr_sym(hipStream_t s, int corr, const double *xT, int64_t ldx, int n, int d,
                    const double *coef, int hcols, double nugget, double *M, int64_t ld, int n_pad,
                    double *xs_scratch = nullptr);
// full (m_pad x n_pad) cross correlation block, row-major into R (ld), no diagonal handling
int launch_cross_corr(hipStream_t s, int corr, const double *xqT, int64_t ldq, int m_pad,
                      const double *xT, int64_t ldx, int n_pad, int d, const double *coef,
                      int hcols, double *R, int64_t ld);
// racc[split * m_pad + q] = partial sum_i k(xq, x_i) * gamma[i] over the split's training range (gamma zero padded to
// n_pad); nsplit > 1 spreads a few queries over the chip, the caller adds the partial sums (racc: nsplit * m_pad)
// raw row-major queries -> normalised k-major (d x ldq, zero padded to m_pad); par = x_mean (d) | x_std (d) on the device
int launch_normalize_queries(hipStream_t s, const double *xq, int m, int d, const double *par, double *xqT, int64_t ldq,
                             int m_pad);
int launch_predict_mean(hipStream_t s, int corr, const double *xqT, int64_t ldq, int m_pad,
                        const double *xT, int64_t ldx, int n_pad, int d, const double *coef,
                        int hcols, const double *gamma, double *racc, int nsplit = 1,
                        const double *xs_prescaled = nullptr);  // (d x ldx) training inputs times coef (hcols == 1): scalar-row form
// xs[k][i] = coef[k] * xT[k][i] over a (d x ldx) k-major array
int launch_scale_rows(hipStream_t s, const double *xT, int64_t ldx, int d, const double *coef, double *xs);
// x-gradient contraction out[split][a][k] = sum_j w(j, a) d r(x_a, x_j) / d x_ak over the split's training range;
// Wt = gamma (vec != 0, ldw ignored) or the transposed (n x m_pad) weight matrix; m_pad multiple of 128
int launch_xgrad(hipStream_t s, int corr, const double *xqT, int64_t ldq, int m_pad, const double *xT, int64_t ldx,
                 int n, int d, const double *coef, int hcols, const double *Wt, int64_t ldw, int vec, int nsplit,
                 double *out);
// few-query form of launch_xgrad with a weight VECTOR: out (ceil(n / 256) x m x d) partial sums, lanes over training points
int launch_xgrad_point(hipStream_t s, int corr, const double *xqT, int64_t ldq, int m, const double *xT, int64_t ldx, int n,
                       int d, const double *coef, int hcols, const double *wvec, double *out);
// single right-hand side through the cached W = C^-T: y <- W^T r (= C^-1 r), z <- W y (= R^-1 r); P scratch 32 * n_pad
int launch_uptri_solve_pair(hipStream_t s, const double *W, int64_t ld, int n, int n_pad, const double *r, double *P,
                            double *y, double *z);
// dst[(r0 + l) * ld + i] = src[l * lds + i] for l < nrows, i < ncols; rest of [r0, r0+rows_pad) x [0, ld) zeroed
int launch_fill_rows(hipStream_t s, double *M, int64_t ld, int r0, int rows_pad, const double *src,
                     int64_t lds, int nrows, int ncols);
int launch_gather_diag(hipStream_t s, const double *M, int64_t ld, int n, double *out);
// per-candidate pointers of a lock-step batch, passed by value to the batched front-end / tail kernels (kernels_corr.hip)
struct EvalBatchPtrs {
    static constexpr int kMax = 16;  // the widest lock-step batch (egx_gp_set_lockstep)
    const double *xT[kMax], *coef[kMax], *rhsT[kMax];
    double *xs[kMax], *M[kMax];
    double *h_diag[kMax], *h_rows[kMax];  // pinned host memory, written by the device
    int *h_info[kMax];
    const int *d_info[kMax];
};
int launch_eval_front_batch(hipStream_t s, int corr, const EvalBatchPtrs &b, int count, int64_t ldx, int n, int d, int hcols,
                            double nugget, int64_t ld, int n_pad, int rhs_pad, int q);
int launch_eval_tail(hipStream_t s, const EvalBatchPtrs &b, int count, int64_t ld, int n, int n_pad, int q, int rows, const int *sync);
// per-row reductions over the first n columns of rows [0, m): s0[q] = sum rt^2, sl[q*p + l] = sum rt*ft_l
int launch_row_reduce(hipStream_t s, const double *RT, int64_t ld, int m, int n, const double *ftT,
                      int64_t ldf, int p, double *s0, double *sl);
// device-side GLS helpers: G <- -Gneg (lower, leading q x q) + identity padding; rho <- yt - ft beta with block sums of rho^2
int launch_gram_finish(hipStream_t s, const double *Gneg, double *G, int64_t ldg, int rows, int q);
int launch_gls_residual(hipStream_t s, const double *ftT, int64_t ld, const double *yt, const double *beta, int p, int n,
                        int n_pad, double *rho, double *part);
// zero the strict upper triangle of the leading n x n block
int launch_zero_upper(hipStream_t s, double *M, int64_t ld, int n);
// likelihood-gradient accumulation (new capability; kernels_corr.hip K7): per candidate z of the lock-step batch and every
// output o < nout
//   out[z][o] = sum_{i > j} 2 R_ij (gamma_i gamma_j inv_s2 + rneg_ij) d log R_ij / d c_o
// hcols == 1: output o = input dimension o (nout = d; w = I: coef = theta);
// hcols  > 1 (KPLS + Matern): output o = theta_o, d log R / d theta_o = sum_j wabs[j][o] (d log m / dt)(coef[j][o] a_j) a_j
// Deterministic (fixed reduction order): a candidate's result does not depend on the batch it runs in.
constexpr int kGradMaxBatch = 16;
struct GradBatch {
    int count = 1;
    const double *xs[kGradMaxBatch];     // d x ldx: the inputs times the candidate's coefficients (prescaled form only)
    const double *coef[kGradMaxBatch];   // d x hcols
    const double *gamma[kGradMaxBatch];  // n_pad
    const double *rneg[kGradMaxBatch];   // -R^-1, lower triangle, leading dimension ld
    double inv_s2[kGradMaxBatch];        // 1 / sigma2 (normalised units)
    double *part[kGradMaxBatch];         // grad_partial_doubles(nout) doubles of scratch
    double *out[kGradMaxBatch];          // nout
};
int grad_partial_doubles(int nout);
int launch_grad_accum(hipStream_t s, int corr, const double *xT, int64_t ldx, int n, int d, int hcols, const double *wabs,
                      int nout, int64_t ld, const GradBatch &batch, int prescaled = 0);
// z (n) <- W y, W upper triangular row-major
int launch_uptri_gemv(hipStream_t s, const double *W, int64_t ld, int n, const double *y, double *z);

// ---- kernels_chol.hip -------------------------------------------------------
// In-place blocked right-looking Cholesky of the leading n_pad x n_pad block (lower), applied to
// all m_tot >= n_pad rows (rows >= n_pad are right-hand sides: on return they hold (C^-1 B)^T).
// dinv receives the inverses of the 64x64 diagonal tiles ((n_pad/64) * 4096 doubles) followed by one flag per tile
// (1.0: ill-conditioned tile, the solves refine once; see k_panel_trsm): dinv_doubles(n_pad) doubles in all.
// info (device int): 0 or 1-based index of the first non-positive pivot.
// ev_syrk: optional accumulation of per-launch timings is done by the caller via events.
// Optional per-launch timing of the big-tile trailing-update kernel (the dominant kernel): event pairs recorded on
// the stream the kernel is launched on.  Filled by launch_potrf, read back by the caller after a stream sync.
inline size_t dinv_doubles(int n_pad) { return (size_t)(n_pad / 64) * 4096 + (size_t)(n_pad / 64); }
struct GemmTrace {
    static constexpr int kMax = 192;
    hipEvent_t e0[kMax], e1[kMax];
    double flops[kMax];
    int used = 0;
    bool ready = false;
};
// Look-ahead resources: s2 = chain stream (diagonal blocks, panel solves), s3 = side stream (the off-critical-path rest
// of the chain's updates), both high priority; events: ev_lu (next diagonal block updated), ev_lur (rest of the next
// group's columns updated), ev_panel (chain of the next group done), ev_a / ev_b (hand-offs between s2 and s3).
struct PotrfLookahead {
    hipStream_t s2 = nullptr, s3 = nullptr;
    hipEvent_t ev_lu = nullptr, ev_lur = nullptr, ev_panel = nullptr, ev_a = nullptr, ev_b = nullptr;
};
// The inverse of the factor riding along the factorisation (theta-gradient): W (n_pad x n_pad, ldw; holds the rows of the
// identity on entry, launch_identity_rows) becomes C^-T.  As soon as a group of panels is final, the forward block
// substitution of W's rows through THAT group -- panel solves with the 16 x 16 inverses the diagonal-block kernel left
// behind, the in-group updates and the update of all later columns -- is enqueued on the stream `sw`, beside the
// factorisation's own trailing update: the substitution's chip-filling GEMMs hide the factorisation's serial chain (most
// of all in its last ~5000 columns, where the trailing matrix is small and W's rows are many), and its small launches
// hide behind the factorisation's big ones.  Same flops as launch_trsm_rows(tri_rows) after the factorisation (n^3/3).
struct PotrfInverse {
    double *W = nullptr;
    int64_t ldw = 0, sW = 0;  // sW: doubles between consecutive matrices' W (lock-step batch)
    hipStream_t sw = nullptr;
    hipEvent_t ev_grp = nullptr, ev_done = nullptr;
};
// Lock-step batches: `count` matrices of one shape, matrix z at (pointer of matrix 0) + z * stride (elements).  Every
// kernel of the factorisation takes the batch as grid.z; a matrix gets the same arithmetic alone and in a batch.
struct GemmBatch {
    int count = 1;
    int64_t sC = 0, sA = 0, sB = 0;  // doubles between consecutive matrices' C / A / B
    int sInfo = 0;                   // ints between their failure flags
};
struct PotrfBatch {
    int count = 1;
    int64_t sM = 0, sD = 0;  // doubles between consecutive matrices / their dinv blocks
    int sI = 0;              // ints between their info words
    int left = 0;            // left-looking group updates (launch_potrf); chosen per handle: potrf_left_for()
    int w_left = 0;          // ... and the C^-T rider's (PotrfInverse) left-looking update: w_left_for()
    int *sync = nullptr;     // hand-off words of the pipelined chain kernel (kernels_pipe.hip): pipe_sync_ints() ints per matrix,
    int64_t sS = 0;          // sS ints apart; nullptr: the chain runs as separate launches (k_potf2_reg, k_panel_trsm16, updates)
    int pipe = 0;            // ... the chain of every group of panels is one chain launch (schedule.h: per handle)
    int whole = 0;           // ... the WHOLE factorisation is one chain launch (every update inside it)
    int flow = 0;            // ... the whole factorisation is one FLOW launch (pipe_flow.h; one matrix per launch)
    int flow_tail = 0;       // ... the LAST flow_tail columns of a right-looking factorisation by separate launches are one flow launch
    int group_panels = 0;    // panels per trailing update, from the handle's schedule (0: by size, potrf_group_panels): launch_potrf
                             // never reads the process-wide knobs for a handle that carries its schedule
    int seqs = 1;            // launch sequences of this handle that may be in flight at once (workspaces / lock-step width):
                             // beyond two, a look-ahead chain launch waits for its columns on the STREAM, not on the device
};
int potrf_left_for(int n_pad, int lockstep);
int w_left_for(int n_pad, int lockstep);
// the schedule of a handle's factorisations, decided once per handle shape: schedule.h (table, rationale, host test)
PotrfSchedule schedule_for(int n_pad, int lockstep, int n_workspaces);
int potrf_group_panels(int n_pad);
// the solves after a factorisation in lock-step: `count` factors (sM apart, tile inverses sD apart) and as many
// right-hand-side buffers (sR apart)
struct TrsmBatch {
    int count = 1;
    int64_t sM = 0, sD = 0, sR = 0;
};
// lk == nullptr runs everything in order on s.
int launch_potrf(hipStream_t s, double *M, int64_t ld, int n_pad, int m_tot, double *dinv, int *info,
                 const PotrfLookahead *lk = nullptr, GemmTrace *trace = nullptr, const PotrfBatch *batch = nullptr,
                 const PotrfInverse *inv = nullptr);
// rows [0, m) of RT (ld) are right-hand sides: RT <- RT * C^-T  (C = lower factor in M, n_pad cols)
// tri_rows != 0: the rows are those of the identity (solution upper triangular): zero blocks are skipped
int launch_trsm_rows(hipStream_t s, const double *M, int64_t ldm, int n_pad, const double *dinv,
                     double *RT, int64_t ldr, int m, int tri_rows = 0, const TrsmBatch *batch = nullptr);
// W (n_pad x n_pad, ld; `count` of them `stride` apart) <- the identity's rows, as far as the two calls below touch them
int launch_identity_rows(hipStream_t s, double *W, int64_t ld, int n_pad, int count = 1, int64_t stride = 0);
// C (lower tiles; n x n) <- -(W W^T), W upper triangular (C^-T): the theta-gradient's -R^-1; C is not read
int launch_syrk_uptri_neg(hipStream_t s, double *C, int64_t ldc, const double *W, int64_t ldw, int n,
                          const GemmBatch *batch = nullptr);
// dinv <- inverses of the 64x64 diagonal tiles of a given lower factor (model load path)
int launch_diag_tile_inverses(hipStream_t s, const double *M, int64_t ld, int n_pad, double *dinv);
// Wall ((n_pad/256) x 256 x 256) <- transposed inverses of the 256x256 diagonal blocks of the factor
// per-model pointers of a lock-step back-substitution (egx_gp_finalize_multi), by value in the kernel arguments
struct SolveBatchPtrs {
    static constexpr int kMax = 16;
    const double *M[kMax], *dinv[kMax];
    double *dW[kMax], *rhs[kMax], *vec[kMax];
};
// the buffer of the 256 x 256 inverse blocks (launch_block_inverse: Wall / dW) carries, behind the blocks, the start ticket and
// the exchange buffer of the one-launch back-substitution (k_trsv_t_fused): allocate block_inverse_doubles(n_pad) doubles
inline size_t trsv_tail_doubles(int n_pad) { return 4 + (size_t)n_pad; }
inline size_t block_inverse_doubles(int n_pad) { return (size_t)((n_pad + kNB - 1) / kNB) * 65536 + trsv_tail_doubles(n_pad); }
long long pipe_timeout_ticks();  // EGX_PIPE_TIMEOUT_MS in ticks of the 100-MHz wall clock (kernels_pipe.hip)
int launch_block_inverse_batch(hipStream_t s, const SolveBatchPtrs &b, int count, int64_t ld, int n_pad);
int launch_trsv_t_batch(hipStream_t s, const SolveBatchPtrs &b, int count, int64_t ld, int n_pad);
int launch_block_inverse(hipStream_t s, const double *M, int64_t ld, int n_pad, const double *dinv, double *Wall);
// xout (n_pad) <- C^-T v   (needs launch_block_inverse first).  One launch (k_trsv_t_fused; v is left alone) unless
// per_block or "trsv_fused" = 0: then one launch per 256-column block, and v is destroyed
int launch_trsv_t(hipStream_t s, const double *M, int64_t ld, int n_pad, const double *Wall, double *v,
                  double *xout, bool per_block = false);
// C (M x N, ldc) -= A (M x K, lda) * B (N x K, ldb)^T ; lower != 0 skips tiles strictly above the diagonal
// ktri != 0 (with lower): A and B are upper triangular, the K loop of tile (bx, by) starts at row bx*tile
// info != nullptr: device flag of the enclosing factorisation; the kernel returns at once when it is non-zero
int launch_gemm_nt_sub(hipStream_t s, double *C, int64_t ldc, const double *A, int64_t lda,
                       const double *B, int64_t ldb, int M, int N, int K, int lower, int ktri = 0,
                       bool *used_big_tile = nullptr, const int *info = nullptr, const GemmBatch *batch = nullptr,
                       int tag = 0);  // tag: kernel symbol of the stream kernel (profiling only; k_gemm_stream's TAG)
// Gneg (rows x rows, ldg; lower 64x64 tiles) <- -(W W^T) for a small output and a long contraction (split-K, partial
// tiles in the scratch P of gram_scratch_doubles(rows, K) doubles, summed in a fixed order)
size_t gram_scratch_doubles(int rows, int K);
int launch_gram_lower(hipStream_t s, const double *W, int64_t ldw, int rows, int K, double *Gneg, int64_t ldg, double *P);
int mfma_probe(double *max_abs_err);

// ---- kernels_pipe.hip -------------------------------------------------------
// The serial chain of a GROUP of panels -- diagonal blocks, panel solves, the updates of the group's own later columns -- as
// ONE persistent launch with device-side hand-offs (k_potrf_pipe): the group [g0, g0 + gw) of `pb.count` matrices in lock-step,
// whose columns have received every update of the earlier groups.  `pb.sync` = pipe_sync_ints(n_pad, m_tot) ints per matrix,
// ZEROED once per factorisation (launch_potrf does it) -- all hand-off words count upwards over the launches of one
// factorisation.  After the factorisation word 0 of the FIRST matrix' block is non-zero iff a bounded wait inside a launch
// ran out (EGX_PIPE_TIMEOUT_MS): the factors are then unusable and the caller reports EGX_ERR_HIP.
size_t pipe_sync_ints(int n_pad, int m_tot);
// ext_need != 0: the solves of the group's first panel wait, on the device, until pipe_signal(.., ext_need) has run (the rest
// of the group's columns is being updated by a launch beside this one); shared_chip: other launches of the factorisation run
// beside this one (look-ahead): the grid leaves them compute units.
int launch_potrf_pipe(hipStream_t s, double *M, int64_t ld, int n_pad, int m_tot, double *dinv, int *info,
                      const PotrfBatch &pb, int g0, int gw, int ext_need = 0, int shared_chip = 0);
int pipe_signal(hipStream_t s, const PotrfBatch &pb, int value);
// The WHOLE factorisation of ONE large matrix as a flow launch (k_potrf_flow; pipe_flow.h: critical stage lists + bulk-class
// rounds per column, round 6); pb.sync as for the chain launch (pipe_sync_ints covers both).  flow_fits: 256-column panels and
// a workgroup per diagonal block + workers on the current device.
// col_base != 0: (M, n_pad, m_tot, dinv) describe the trailing part of a larger matrix from that column on, which has received
// every update of the columns before it (launch_potrf's flow tail); only the reported pivot index uses it.
int launch_potrf_flow(hipStream_t s, double *M, int64_t ld, int n_pad, int m_tot, double *dinv, int *info, const PotrfBatch &pb,
                      int col_base = 0);
bool flow_fits(int n_pad);
int pipe_enabled();    // EGX_PIPE (default on); 0 as well when the launch cannot be set up on this device
bool pipe_fits(int nz, int np);  // one workgroup per diagonal block (np panels of nz matrices) + a worker fit the CURRENT device
int pipe_prepare(int n_pad, int m_tot, const PotrfSchedule &sched);  // build + upload the task lists a handle of this shape launches
size_t pipe_release_plans();     // egx_trim: free every task list (synchronises the devices that hold one); bytes freed
int pipe_set_knob(const char *name, int value);  // "pipe", "pipe_timeout_ms" (+ "pipe_stall" in test builds); INT_MIN = unknown
#ifdef EGX_TEST_HOOKS  // egobox_amd/lib/_dev/libegx_gp_hip_testhooks.so and tools/pipe_check only
void pipe_set_trace(long long *device_buf);  // profiling: 8 words per ticket of the next chain launches (nullptr: off)
void pipe_test_set_workgroups(int wgs);      // grid of the next chain launches (0 = the product's)
void pipe_set_trace_cap(int slots);          // slots of the trace buffer (flow launches take them from a counter)
void flow_test_set_leads(int lead_short, int lead_long);  // FlowArgs::lead_* of the next flow launches (-1: the product's)
#endif
int chol_init();  // one-time function attribute setup (dynamic LDS sizes)
int set_knob(const char *name, int value);  // kernels_chol.hip tuning knobs by name; INT_MIN = unknown

}  // namespace egx
