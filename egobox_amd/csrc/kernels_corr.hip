// Correlation kernels for gfx950: the device counterpart of
//   DiffMatrix::_cross_diff        crates/gp/src/utils.rs:80-104
//   pairwise_differences           crates/gp/src/utils.rs:110-131
//   CorrelationModel::value        crates/gp/src/correlation_models.rs:91-104 (sq-exp), :185-196
//                                  (abs-exp), :326-353 (Matern 3/2), :497-523 (Matern 5/2)
//   the R scatter loop             crates/gp/src/algorithm.rs:997-1001
//   _compute_correlation + predict crates/gp/src/algorithm.rs:253-263, 372-380
// fused so that the (n(n-1)/2, d) difference table (34 GB at n = 16384, d = 32) and the (m*n, d)
// prediction table never exist: every workgroup stages two 64-point slabs of the (k-major)
// normalised inputs in LDS and produces a 64x64 tile of correlations with DIRECT differences
// (no |a|^2+|b|^2-2ab trick: cancellation would break 1e-8 parity for near-duplicate points).
// HBM traffic is the algorithmic minimum: X read once per tile row/column from L2, R written once.
#include "egx_internal.h"

#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <string>

namespace egx {

constexpr double kSqrt3 = 1.7320508075688772;
constexpr double kSqrt5 = 2.23606797749979;
constexpr double k5over3 = 5.0 / 3.0;

// exp(x) for x <= 0 (every correlation is exp of a non-positive sum): round-to-nearest range reduction x = k ln2 + r,
// |r| <= ln2 / 2, degree-13 Taylor polynomial in Horner form (truncation 4e-18, all FMA), one v_ldexp_f64: ~20 VALU
// instructions and no branches or selects, against ~30 of the library exp with its special-case handling.  K1 is bound
// by FP64 VALU issue (profiles/r03_*_pmc_corr_*), so instructions per pair are what counts.  Error <= 2 ulp; underflow
// goes through ldexp (gradual, then 0); NaN propagates.
__device__ __forceinline__ double exp_nonpos(double x) {
    // exp(x) == 0 below -745.2: the clamp keeps -inf (a huge theta: the sum overflows) from turning into NaN in the
    // reduction (fma(-inf, -c, -inf)), keeps the cast of kf defined, and lets v_ldexp flush the result to 0 as the reference's
    // exp does.  A select, not fmax: a NaN argument must stay NaN (v_max_f64 would return the other operand).
    x = (x < -800.0) ? -800.0 : x;
    const double kf = __builtin_rint(x * 1.4426950408889634074);
    double r = __builtin_fma(kf, -6.93147180369123816490e-01, x);
    r = __builtin_fma(kf, -1.90821492927058770002e-10, r);
    double p = 1.6059043836821614599e-10;                  // 1 / 13!
    p = __builtin_fma(p, r, 2.0876756987868098979e-09);    // 1 / 12!
    p = __builtin_fma(p, r, 2.5052108385441718775e-08);    // 1 / 11!
    p = __builtin_fma(p, r, 2.7557319223985890653e-07);    // 1 / 10!
    p = __builtin_fma(p, r, 2.7557319223985892511e-06);    // 1 / 9!
    p = __builtin_fma(p, r, 2.4801587301587301566e-05);    // 1 / 8!
    p = __builtin_fma(p, r, 1.9841269841269841253e-04);    // 1 / 7!
    p = __builtin_fma(p, r, 1.3888888888888889419e-03);    // 1 / 6!
    p = __builtin_fma(p, r, 8.3333333333333332177e-03);    // 1 / 5!
    p = __builtin_fma(p, r, 4.1666666666666664354e-02);    // 1 / 4!
    p = __builtin_fma(p, r, 1.6666666666666665741e-01);    // 1 / 3!
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)kf);
}

// Accumulator for one (i, j) pair, updated one input dimension at a time.
template <int CORR>
struct PairAcc {
    double s;  // sum inside the exponential
    double a;  // polynomial product (Matern)
    __device__ __forceinline__ PairAcc() : s(0.0), a(1.0) {}
    __device__ __forceinline__ void add(double diff, const double *__restrict__ coef, int hcols) {
        const double ad = fabs(diff);
        if (CORR == EGX_CORR_SQUARED_EXPONENTIAL) {
            const double t = coef[0] * diff;
            s = __builtin_fma(t, t, s);
        } else if (CORR == EGX_CORR_ABSOLUTE_EXPONENTIAL) {
            s = __builtin_fma(coef[0], ad, s);
        } else if (CORR == EGX_CORR_MATERN32) {
            for (int l = 0; l < hcols; l++) {
                const double t = coef[l] * ad;
                s += t;
                a *= __builtin_fma(kSqrt3, t, 1.0);
            }
        } else {
            for (int l = 0; l < hcols; l++) {
                const double t = coef[l] * ad;
                s += t;
                a *= 1.0 + kSqrt5 * t + k5over3 * (t * t);
            }
        }
    }
    // the same with the coefficient already applied to both coordinates (hcols == 1: diff = c x_i - c x_j, so that
    // c |x_i - x_j| costs no instruction per pair; the rounding of the two products perturbs the exponent by
    // <= 2 eps sum_k |t_k| max |c x|, below 1e-13 wherever r is not negligible): 2 VALU instructions per pair and dimension
    // for the squared exponential instead of 3, 5 instead of 8 for Matern-5/2
    __device__ __forceinline__ void add_scaled(double diff) {
        if (CORR == EGX_CORR_SQUARED_EXPONENTIAL) {
            s = __builtin_fma(diff, diff, s);
        } else if (CORR == EGX_CORR_ABSOLUTE_EXPONENTIAL) {
            s += fabs(diff);
        } else if (CORR == EGX_CORR_MATERN32) {
            const double t = fabs(diff);
            s += t;
            a *= __builtin_fma(kSqrt3, t, 1.0);
        } else {
            const double t = fabs(diff);
            s += t;
            a *= __builtin_fma(__builtin_fma(k5over3, t, kSqrt5), t, 1.0);
        }
    }
    __device__ __forceinline__ double value() const {
        if (CORR == EGX_CORR_SQUARED_EXPONENTIAL) return exp_nonpos(-0.5 * s);
        if (CORR == EGX_CORR_ABSOLUTE_EXPONENTIAL) return exp_nonpos(-s);
        if (CORR == EGX_CORR_MATERN32) return a * exp_nonpos(-kSqrt3 * s);
        return a * exp_nonpos(-kSqrt5 * s);
    }
};

// Stage a 64-point slab (all d dimensions, k-major) into LDS: dst[k*64 + i]; scale != nullptr: times scale[k]
// (PairAcc::add_scaled, hcols == 1).
__device__ __forceinline__ void stage_slab(double *dst, const double *__restrict__ xT, int64_t ldx, int i0,
                                           int d, int tid, const double *__restrict__ scale = nullptr, int nthreads = 256) {
    for (int e = tid; e < d * 64; e += nthreads) {
        const int k = e >> 6, i = e & 63;
        const double v = xT[(int64_t)k * ldx + i0 + i];
        dst[e] = scale ? scale[k] * v : v;
    }
}

// Input dimensions are staged in LDS in chunks of at most kCorrDC (round 4: any d; one chunk for d <= 64, where nothing
// changes): the pair accumulators (exponent sum, Matern product) simply run on across the chunks.
constexpr int kCorrDC = 64;

// 4x4 micro-tile of pair accumulators from two staged slabs: `dn` more dimensions (coef already points at the first one)
template <int CORR, bool PRE = false, int RI = 4>  // PRE: the slabs were staged with the coefficients applied (hcols == 1);
                                                    // RI x 4 pairs per lane (rows RI ty + a, columns 4 tx + b)
__device__ __forceinline__ void tile_pairs_acc(const double *xi, const double *xj, const double *__restrict__ coef,
                                               int hcols, int dn, int ty, int tx, PairAcc<CORR> (&acc)[RI][4]) {
    for (int k = 0; k < dn; k++) {
        double vi[RI], vj[4];
#pragma unroll
        for (int a = 0; a < RI; a++) vi[a] = xi[k * 64 + ty * RI + a];
#pragma unroll
        for (int b = 0; b < 4; b++) vj[b] = xj[k * 64 + tx * 4 + b];
        const double *ck = coef + k * hcols;
#pragma unroll
        for (int a = 0; a < RI; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) {
                if (PRE) acc[a][b].add_scaled(vi[a] - vj[b]);
                else acc[a][b].add(vi[a] - vj[b], ck, hcols);
            }
    }
}
template <int CORR, bool PRE = false, int RI = 4>
__device__ __forceinline__ void tile_pairs(const double *xi, const double *xj, const double *__restrict__ coef,
                                           int hcols, int d, int ty, int tx, double (&r)[RI][4]) {
    PairAcc<CORR> acc[RI][4];
    tile_pairs_acc<CORR, PRE, RI>(xi, xj, coef, hcols, d, ty, tx, acc);
#pragma unroll
    for (int a = 0; a < RI; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) r[a][b] = acc[a][b].value();
}

// K1: symmetric training correlation matrix, 64x64 tiles, 128x128-granular lower triangle.
template <int CORR, bool PRE, int RI>  // 64 x 64 tile per workgroup of 1024 / RI threads, RI x 4 pairs per lane
__global__ __launch_bounds__(1024 / RI) void k_corr_sym(const double *__restrict__ xT, int64_t ldx, int n, int d,
                                                        const double *__restrict__ coef, int hcols, double diag,
                                                        double *__restrict__ M, int64_t ld) {
    // 1-D grid over the 128x128 tiles of the LOWER triangle only (4 workgroups per tile): a 2-D grid with an early exit
    // for the strictly-upper tiles launches twice the workgroups and leaves the real ones unevenly spread over the CUs
    constexpr int NT = 1024 / RI;
    const int t = blockIdx.x >> 2, sub = blockIdx.x & 3;
    int I = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while (I * (I + 1) / 2 > t) I--;
    while ((I + 1) * (I + 2) / 2 <= t) I++;
    const int J = t - I * (I + 1) / 2;
    const int bi = 2 * I + (sub >> 1), bj = 2 * J + (sub & 1);
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int dc = d < kCorrDC ? d : kCorrDC;
    double *xi = sm, *xj = sm + dc * 64;
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    PairAcc<CORR> acc[RI][4];
    for (int c0 = 0; c0 < d; c0 += dc) {
        const int dn = (d - c0 < dc) ? (d - c0) : dc;
        if (c0) __syncthreads();
        stage_slab(xi, xT + (int64_t)c0 * ldx, ldx, bi * 64, dn, tid, PRE ? coef + c0 : nullptr, NT);
        stage_slab(xj, xT + (int64_t)c0 * ldx, ldx, bj * 64, dn, tid, PRE ? coef + c0 : nullptr, NT);
        __syncthreads();
        tile_pairs_acc<CORR, PRE, RI>(xi, xj, coef + (int64_t)c0 * hcols, hcols, dn, ty, tx, acc);
    }
    double r[RI][4];
#pragma unroll
    for (int a = 0; a < RI; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) r[a][b] = acc[a][b].value();
#pragma unroll
    for (int a = 0; a < RI; a++) {
        const int i = bi * 64 + ty * RI + a;
        double out[4];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int j = bj * 64 + tx * 4 + b;
            double v = r[a][b];
            if (i >= n || j >= n) v = (i == j) ? 1.0 : 0.0;  // identity padding
            else if (i == j) v = diag;                       // 1 + nugget, algorithm.rs:997
            out[b] = v;
        }
        double *p = M + (int64_t)i * ld + bj * 64 + tx * 4;
        *reinterpret_cast<double2 *>(p) = make_double2(out[0], out[1]);
        *reinterpret_cast<double2 *>(p + 2) = make_double2(out[2], out[3]);
    }
}
// K1, scalar-row form (round 3).  The 4x4-pairs-per-lane kernel above reads 8 doubles from LDS per 16 pairs and dimension:
// 4 ds_read_b128 per 32 FP64 instructions per wave keep the CU's one LDS pipe as busy as its four SIMDs, and the kernel
// sits at 0.78 of its FP64 VALU-issue bound.  Here a wave owns 16 ROWS x 64 columns (lane = column): the row coordinates
// are WAVE-UNIFORM, so they are fetched by the scalar unit (s_load) into SGPRs and enter the FP64 instructions as their
// scalar operand; only the lane's own column coordinate comes from LDS -- one ds_read_b64 per 32 instructions.  The inputs
// are read PRESCALED (xs[k][i] = c_k x[k][i], one tiny launch per candidate) so that nothing but the subtraction and the
// accumulation is left per pair and dimension.  Same 64x64 tile per workgroup, same grid, same stores (a row of 64 results
// per wave instruction: 512 contiguous bytes).
template <int CORR>
__device__ __forceinline__ void corr_sym_srow_body(const double *__restrict__ xs, int64_t ldx, int n, int d, double diag,
                                                   double *__restrict__ M, int64_t ld) {
    const int t = blockIdx.x >> 2, sub = blockIdx.x & 3;
    int I = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while (I * (I + 1) / 2 > t) I--;
    while ((I + 1) * (I + 2) / 2 <= t) I++;
    const int J = t - I * (I + 1) / 2;
    const int bi = 2 * I + (sub >> 1), bj = 2 * J + (sub & 1);
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *xj = sm;  // the 64 column points, k-major
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    stage_slab(xj, xs, ldx, bj * 64, d, tid);
    __syncthreads();
    const int i0 = bi * 64 + wave * 16;
    const double *xr = xs + i0;  // wave-uniform: rows i0 .. i0 + 15 of dimension k at xr[k * ldx + a]
    PairAcc<CORR> acc[16];
    for (int k = 0; k < d; k++) {
        const double v = xj[k * 64 + lane];
        const double4 r0 = *reinterpret_cast<const double4 *>(xr + (int64_t)k * ldx);
        const double4 r1 = *reinterpret_cast<const double4 *>(xr + (int64_t)k * ldx + 4);
        const double4 r2 = *reinterpret_cast<const double4 *>(xr + (int64_t)k * ldx + 8);
        const double4 r3 = *reinterpret_cast<const double4 *>(xr + (int64_t)k * ldx + 12);
        acc[0].add_scaled(r0.x - v);
        acc[1].add_scaled(r0.y - v);
        acc[2].add_scaled(r0.z - v);
        acc[3].add_scaled(r0.w - v);
        acc[4].add_scaled(r1.x - v);
        acc[5].add_scaled(r1.y - v);
        acc[6].add_scaled(r1.z - v);
        acc[7].add_scaled(r1.w - v);
        acc[8].add_scaled(r2.x - v);
        acc[9].add_scaled(r2.y - v);
        acc[10].add_scaled(r2.z - v);
        acc[11].add_scaled(r2.w - v);
        acc[12].add_scaled(r3.x - v);
        acc[13].add_scaled(r3.y - v);
        acc[14].add_scaled(r3.z - v);
        acc[15].add_scaled(r3.w - v);
    }
    const int j = bj * 64 + lane;
#pragma unroll
    for (int a = 0; a < 16; a++) {
        const int i = i0 + a;
        double vv = acc[a].value();
        if (i >= n || j >= n) vv = (i == j) ? 1.0 : 0.0;  // identity padding
        else if (i == j) vv = diag;                       // 1 + nugget, algorithm.rs:997
        M[(int64_t)i * ld + j] = vv;
    }
}
template <int CORR>
__global__ __launch_bounds__(256) void k_corr_sym_srow(const double *__restrict__ xs, int64_t ldx, int n, int d, double diag,
                                                       double *__restrict__ M, int64_t ld) {
    corr_sym_srow_body<CORR>(xs, ldx, n, d, diag, M, ld);
}
// ... for the candidates of a lock-step batch in ONE launch (blockIdx.y = candidate): the same workgroups doing the same
// arithmetic, a twelfth of the launches (n = 4096 in lock-step 12: 12 x 3 front-end launches of ~20 + 5 + 5 us, serial on the
// lead's stream, were 4 % of a batch -- profiles/r05_n4096_lockstep12_timeline.txt)
template <int CORR>
__global__ __launch_bounds__(256) void k_corr_sym_srow_batch(EvalBatchPtrs b, int64_t ldx, int n, int d, double diag, int64_t ld) {
    corr_sym_srow_body<CORR>(b.xs[blockIdx.y], ldx, n, d, diag, b.M[blockIdx.y], ld);
}

__global__ void k_scale_rows_batch(EvalBatchPtrs b, int64_t ldx, int d) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ldx) return;
    const double *xT = b.xT[blockIdx.y], *coef = b.coef[blockIdx.y];
    double *xs = b.xs[blockIdx.y];
    for (int k = 0; k < d; k++) xs[(int64_t)k * ldx + i] = coef[k] * xT[(int64_t)k * ldx + i];
}

// xs[k][i] = coef[k] * xT[k][i] over the (d x ldx) k-major array (hcols == 1)
__global__ void k_scale_rows(const double *__restrict__ xT, int64_t ldx, int d, const double *__restrict__ coef,
                             double *__restrict__ xs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ldx) return;
    for (int k = 0; k < d; k++) xs[(int64_t)k * ldx + i] = coef[k] * xT[(int64_t)k * ldx + i];
}

// Queries as the caller holds them (m x d row-major, raw) -> normalised and k-major (d x ldq, zero padded): what
// GaussianProcess::predict does first ((x - x_mean) / x_std, algorithm.rs:254).  The same IEEE subtraction and division
// as the host loop this replaces, so the bits are the same; a 64-query slab goes through LDS so that both the reads and
// the writes are contiguous.  par = x_mean (d) then x_std (d).
__global__ __launch_bounds__(256) void k_normalize_queries(const double *__restrict__ xq, int m, int d,
                                                           const double *__restrict__ par, double *__restrict__ xqT,
                                                           int64_t ldq) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int q0 = blockIdx.x * 64, tid = threadIdx.x;
    const int rows = (m - q0 < 64) ? (m - q0) : 64;
    // grid.y: chunks of kCorrDC dimensions (one for d <= 64)
    const int c0 = blockIdx.y * kCorrDC;
    const int dn = (d - c0 < kCorrDC) ? (d - c0) : kCorrDC;
    const int ds = dn | 1;  // odd row stride: the transposed reads below spread over the banks
    for (int e = tid; e < rows * dn; e += 256) {
        const int i = e / dn, k = e - i * dn;
        sm[i * ds + k] = xq[(int64_t)(q0 + i) * d + c0 + k];
    }
    __syncthreads();
    for (int e = tid; e < dn * 64; e += 256) {
        const int k = e >> 6, i = e & 63;
        xqT[(int64_t)(c0 + k) * ldq + q0 + i] = (i < rows) ? (sm[i * ds + k] - par[c0 + k]) / par[d + c0 + k] : 0.0;
    }
}

template <int CORR, bool PRE>
__global__ __launch_bounds__(256) void k_cross_corr(const double *__restrict__ xqT, int64_t ldq,
                                                    const double *__restrict__ xT, int64_t ldx, int d,
                                                    const double *__restrict__ coef, int hcols,
                                                    double *__restrict__ R, int64_t ld) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int dc = d < kCorrDC ? d : kCorrDC;
    double *xi = sm, *xj = sm + dc * 64;
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    PairAcc<CORR> acc[4][4];
    for (int c0 = 0; c0 < d; c0 += dc) {
        const int dn = (d - c0 < dc) ? (d - c0) : dc;
        if (c0) __syncthreads();
        stage_slab(xi, xqT + (int64_t)c0 * ldq, ldq, blockIdx.x * 64, dn, tid, PRE ? coef + c0 : nullptr);
        stage_slab(xj, xT + (int64_t)c0 * ldx, ldx, blockIdx.y * 64, dn, tid, PRE ? coef + c0 : nullptr);
        __syncthreads();
        tile_pairs_acc<CORR, PRE>(xi, xj, coef + (int64_t)c0 * hcols, hcols, dn, ty, tx, acc);
    }
    double r[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) r[a][b] = acc[a][b].value();
#pragma unroll
    for (int a = 0; a < 4; a++) {
        double *p = R + (int64_t)(blockIdx.x * 64 + ty * 4 + a) * ld + blockIdx.y * 64 + tx * 4;
        *reinterpret_cast<double2 *>(p) = make_double2(r[a][0], r[a][1]);
        *reinterpret_cast<double2 *>(p + 2) = make_double2(r[a][2], r[a][3]);
    }
}

// K2+K6 fused: racc[q] = sum_i k(xq, x_i) gamma_i ; one workgroup per 64 queries, loop over slabs.
template <int CORR, bool PRE>
__global__ __launch_bounds__(256) void k_predict_mean(const double *__restrict__ xqT, int64_t ldq,
                                                      const double *__restrict__ xT, int64_t ldx, int n_pad,
                                                      int d, const double *__restrict__ coef, int hcols,
                                                      const double *__restrict__ gamma,
                                                      double *__restrict__ racc, int slabs_per_split, int m_pad) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int dc = d < kCorrDC ? d : kCorrDC;
    double *xi = sm, *xj = sm + dc * 64, *gs = sm + 2 * dc * 64;
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    if (d <= dc) stage_slab(xi, xqT, ldq, blockIdx.x * 64, d, tid, PRE ? coef : nullptr);  // the queries: once
    double sum[4] = {0.0, 0.0, 0.0, 0.0};
    // grid.y splits the training range (a few queries must still fill the chip); partial sums per split
    const int j_lo = blockIdx.y * slabs_per_split * 64;
    int j_hi = j_lo + slabs_per_split * 64;
    if (j_hi > n_pad) j_hi = n_pad;
    for (int j0 = j_lo; j0 < j_hi; j0 += 64) {
        PairAcc<CORR> acc[4][4];
        for (int c0 = 0; c0 < d; c0 += dc) {
            const int dn = (d - c0 < dc) ? (d - c0) : dc;
            __syncthreads();
            if (d > dc) stage_slab(xi, xqT + (int64_t)c0 * ldq, ldq, blockIdx.x * 64, dn, tid, PRE ? coef + c0 : nullptr);
            stage_slab(xj, xT + (int64_t)c0 * ldx, ldx, j0, dn, tid, PRE ? coef + c0 : nullptr);
            if (c0 == 0 && tid < 64) gs[tid] = gamma[j0 + tid];
            __syncthreads();
            tile_pairs_acc<CORR, PRE>(xi, xj, coef + (int64_t)c0 * hcols, hcols, dn, ty, tx, acc);
        }
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) sum[a] = __builtin_fma(acc[a][b].value(), gs[tx * 4 + b], sum[a]);
    }
    // reduce over the 16 tx lanes that share the same queries (consecutive lanes of one wave)
#pragma unroll
    for (int a = 0; a < 4; a++) {
        double v = sum[a];
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 8);
        if (tx == 0) racc[(int64_t)blockIdx.y * m_pad + blockIdx.x * 64 + ty * 4 + a] = v;
    }
}

// K2+K6, scalar-row form (round 3; see k_corr_sym_srow): a lane owns ONE query of the workgroup's 64 (its prescaled
// coordinates in LDS, one ds_read_b64 per 32 FP64 instructions), the training points are the wave-uniform rows -- 16 at a
// time, fetched from the PRESCALED training inputs of the fit (xs = c_k x, k-major) by the scalar unit, gamma with them --
// and r . gamma accumulates in the lane's own register: no cross-lane reduction, no barrier inside the loop.  The four waves
// of a workgroup take every fourth 16-row block of the split's training range; their partial sums meet in LDS at the end.
template <int CORR>
__global__ __launch_bounds__(256) void k_predict_mean_srow(const double *__restrict__ xqT, int64_t ldq,
                                                           const double *__restrict__ xs, int64_t ldx, int n_pad, int d,
                                                           const double *__restrict__ coef,
                                                           const double *__restrict__ gamma, double *__restrict__ racc,
                                                           int slabs_per_split, int m_pad) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *xq = sm, *part = sm + d * 64;  // 64 queries k-major, prescaled; 4 x 64 partial sums
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    stage_slab(xq, xqT, ldq, blockIdx.x * 64, d, tid, coef);
    __syncthreads();
    const int j_lo = blockIdx.y * slabs_per_split * 64;
    int j_hi = j_lo + slabs_per_split * 64;
    if (j_hi > n_pad) j_hi = n_pad;
    double sum = 0.0;
    for (int j0 = j_lo + wave * 16; j0 < j_hi; j0 += 64) {
        const double *xr = xs + j0;  // wave-uniform
        PairAcc<CORR> acc[16];
        for (int k = 0; k < d; k++) {
            const double v = xq[k * 64 + lane];
            const double4 r0 = *reinterpret_cast<const double4 *>(xr + (int64_t)k * ldx);
            const double4 r1 = *reinterpret_cast<const double4 *>(xr + (int64_t)k * ldx + 4);
            const double4 r2 = *reinterpret_cast<const double4 *>(xr + (int64_t)k * ldx + 8);
            const double4 r3 = *reinterpret_cast<const double4 *>(xr + (int64_t)k * ldx + 12);
            acc[0].add_scaled(r0.x - v);
            acc[1].add_scaled(r0.y - v);
            acc[2].add_scaled(r0.z - v);
            acc[3].add_scaled(r0.w - v);
            acc[4].add_scaled(r1.x - v);
            acc[5].add_scaled(r1.y - v);
            acc[6].add_scaled(r1.z - v);
            acc[7].add_scaled(r1.w - v);
            acc[8].add_scaled(r2.x - v);
            acc[9].add_scaled(r2.y - v);
            acc[10].add_scaled(r2.z - v);
            acc[11].add_scaled(r2.w - v);
            acc[12].add_scaled(r3.x - v);
            acc[13].add_scaled(r3.y - v);
            acc[14].add_scaled(r3.z - v);
            acc[15].add_scaled(r3.w - v);
        }
        const double4 g0 = *reinterpret_cast<const double4 *>(gamma + j0);
        const double4 g1 = *reinterpret_cast<const double4 *>(gamma + j0 + 4);
        const double4 g2 = *reinterpret_cast<const double4 *>(gamma + j0 + 8);
        const double4 g3 = *reinterpret_cast<const double4 *>(gamma + j0 + 12);
        const double g[16] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w, g2.x, g2.y, g2.z, g2.w, g3.x, g3.y, g3.z, g3.w};
#pragma unroll
        for (int a = 0; a < 16; a++) sum = __builtin_fma(acc[a].value(), g[a], sum);
    }
    part[wave * 64 + lane] = sum;
    __syncthreads();
    if (tid < 64)
        racc[(int64_t)blockIdx.y * m_pad + blockIdx.x * 64 + tid] = (part[tid] + part[64 + tid]) + (part[128 + tid] + part[192 + tid]);
}

// ---------------------------------------------------------------------------------------------
// x-gradients of the predictions (CorrelationModel::jacobian, correlation_models.rs:106-123, 198-214,
// 355-413, 524-586, contracted as predict_jacobian / predict_var_gradients_single do, algorithm.rs:523-617):
//     out[a][k] = sum_j w(j, a) * d r(x_a, x_j) / d x_ak
// with w = gamma_j (mean, VEC) or a per-(j, a) weight matrix held TRANSPOSED (n x m: lanes <-> queries, coalesced).
// The (n, d) jacobian of one query never exists: one lane per query, training slabs broadcast from LDS, the d
// partial sums of a lane live in registers (DK at a time).  grid.y splits the training range for small batches
// (EGO asks for one point at a time); the partial sums of the splits are added on the host.
// d r / d x_k = r * g_k(x_ak - x_jk):
//   sq-exp   g = -c_k^2 diff                      abs-exp  g = -c_k sign(diff)
//   Matern   g = sign(diff) sum_l c_kl (f'(t_l) / f(t_l) - q),  t_l = c_kl |diff|,  f = 1 + q t [+ 5/3 t^2]
// (the closed form of the reference's product-over-all-but-one loops; sign(+0) = +1 like f64::signum).
// ---------------------------------------------------------------------------------------------
template <int CORR>
__device__ __forceinline__ double xgrad_factor(double diff, const double *__restrict__ c, int hcols) {
    if (CORR == EGX_CORR_SQUARED_EXPONENTIAL) return -(c[0] * c[0]) * diff;
    if (CORR == EGX_CORR_ABSOLUTE_EXPONENTIAL) return -copysign(c[0], diff);
    const double ad = fabs(diff);
    double sacc = 0.0;
    for (int l = 0; l < hcols; l++) {
        const double t = c[l] * ad;
        if (CORR == EGX_CORR_MATERN32)
            sacc += c[l] * (kSqrt3 / __builtin_fma(kSqrt3, t, 1.0) - kSqrt3);
        else
            sacc += c[l] * ((kSqrt5 + 2.0 * k5over3 * t) / (1.0 + kSqrt5 * t + k5over3 * (t * t)) - kSqrt5);
    }
    return copysign(1.0, diff) * sacc;
}

constexpr int kXgThreads = 128;
constexpr size_t kLdsBytes = 160 * 1024;  // LDS of one CU (gfx950)

// XAG (d > 64): the lane's own query coordinates come from global memory (L1 / L2) instead of a [d][kXgThreads] LDS
// image, which together with the training slab would not fit any more; the slab is [d][2^js_shift] training points,
// 64 wide up to d = 256 and narrower beyond (launch_xgrad picks the widest that fits the 160 KB of LDS)
template <int CORR, int DK, bool VEC, bool XAG = false>
__global__ __launch_bounds__(kXgThreads) void k_xgrad(const double *__restrict__ xqT, int64_t ldq,
                                                      const double *__restrict__ xT, int64_t ldx, int n, int d,
                                                      const double *__restrict__ coef, int hcols,
                                                      const double *__restrict__ Wt, int64_t ldw, int slabs_per_split,
                                                      double *__restrict__ out, int m_pad, int js_shift) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int jsh = XAG ? js_shift : 6, JS = 1 << jsh;
    double *xa = sm;                                  // [d][kXgThreads]  (not with XAG)
    double *xj = sm + (XAG ? 0 : d * kXgThreads);     // [d][JS]
    double *cs = xj + d * JS;                         // [d * hcols]
    const int tid = threadIdx.x;
    const int a = blockIdx.x * kXgThreads + tid;
    const double *xag = xqT + a;
#define EGX_XA(k_) (XAG ? xag[(int64_t)(k_) * ldq] : xa[(k_) * kXgThreads + tid])
    if (!XAG)
        for (int k = 0; k < d; k++) xa[k * kXgThreads + tid] = xqT[(int64_t)k * ldq + a];
    for (int e = tid; e < d * hcols; e += kXgThreads) cs[e] = coef[e];
    const int j_lo = blockIdx.y * slabs_per_split * 64;
    int j_hi = j_lo + slabs_per_split * 64;
    if (j_hi > n) j_hi = n;
    double *o = out + ((int64_t)blockIdx.y * m_pad + a) * d;
    for (int k0 = 0; k0 < d; k0 += DK) {
        double acc[DK], xr[DK];
#pragma unroll
        for (int kk = 0; kk < DK; kk++) {
            acc[kk] = 0.0;
            xr[kk] = (k0 + kk < d) ? EGX_XA(k0 + kk) : 0.0;
        }
        for (int j0 = j_lo; j0 < j_hi; j0 += JS) {
            __syncthreads();
            for (int e = tid; e < d * JS; e += kXgThreads) xj[e] = xT[(int64_t)(e >> jsh) * ldx + j0 + (e & (JS - 1))];
            __syncthreads();
            const int jn = (j_hi - j0 < JS) ? (j_hi - j0) : JS;
            for (int jj = 0; jj < jn; jj++) {
                PairAcc<CORR> pa;
                for (int k = 0; k < d; k++) pa.add(EGX_XA(k) - xj[k * JS + jj], cs + k * hcols, hcols);
                const double wv = VEC ? Wt[j0 + jj] : Wt[(int64_t)(j0 + jj) * ldw + a];
                const double rw = pa.value() * wv;
#pragma unroll
                for (int kk = 0; kk < DK; kk++)
                    if (k0 + kk < d)
                        acc[kk] = __builtin_fma(rw, xgrad_factor<CORR>(xr[kk] - xj[(k0 + kk) * JS + jj],
                                                                       cs + (k0 + kk) * hcols, hcols), acc[kk]);
            }
        }
#pragma unroll
        for (int kk = 0; kk < DK; kk++)
            if (k0 + kk < d) o[k0 + kk] = acc[kk];
    }
#undef EGX_XA
}

// Few-query form of the x-gradient contraction (single-point calls): the lanes run over TRAINING points instead of queries
// (one pair per lane, no sequential pair loop), the d partial sums are reduced across the workgroup.
//   out[(blockIdx.y * m + a) * d + k] = sum over the block's 256 training points of w_j d r(x_a, x_j) / d x_ak
template <int CORR>
__global__ __launch_bounds__(256) void k_xgrad_point(const double *__restrict__ xqT, int64_t ldq, const double *__restrict__ xT,
                                                     int64_t ldx, int n, int d, const double *__restrict__ coef, int hcols,
                                                     const double *__restrict__ wvec, double *__restrict__ out, int m) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *xa = sm;            // [d]
    double *cs = sm + d;        // [d * hcols]
    double *red = cs + d * hcols;  // [4][d]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int a = blockIdx.x, j = blockIdx.y * 256 + tid;
    for (int k = tid; k < d; k += 256) xa[k] = xqT[(int64_t)k * ldq + a];
    for (int e = tid; e < d * hcols; e += 256) cs[e] = coef[e];
    __syncthreads();
    const bool live = j < n;
    double rw = 0.0;
    if (live) {
        PairAcc<CORR> pa;
        for (int k = 0; k < d; k++) pa.add(xa[k] - xT[(int64_t)k * ldx + j], cs + k * hcols, hcols);
        rw = pa.value() * wvec[j];
    }
    for (int k = 0; k < d; k++) {
        double v = live ? rw * xgrad_factor<CORR>(xa[k] - xT[(int64_t)k * ldx + j], cs + k * hcols, hcols) : 0.0;
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wave * d + k] = v;
    }
    __syncthreads();
    for (int k = tid; k < d; k += 256)
        out[((int64_t)blockIdx.y * m + a) * d + k] = ((red[k] + red[d + k]) + red[2 * d + k]) + red[3 * d + k];
}

// ---------------------------------------------------------------------------------------------
// Single-point path of the variance (and its x-gradient): with W = C^-T cached (upper triangular, row-major)
//   y = C^-1 r = W^T r      k_uptri_gemv_t  (rows split over grid.y, partial sums to P, summed in a fixed order)
//   z = C^-T y = W y        k_uptri_gemv    (one wave per row, coalesced)
// Two memory-bound passes over half of W instead of 2 x 64 dependent block solves: EGO asks for ONE point at a time.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_uptri_gemv_t(const double *__restrict__ W, int64_t ld, int n,
                                                      const double *__restrict__ r, int rows_per_split,
                                                      double *__restrict__ P, int n_pad) {
    __shared__ double red[4][64];
    const int tid = threadIdx.x, g = tid >> 6, jl = tid & 63;
    const int j = blockIdx.x * 64 + jl;
    const int i_lo = blockIdx.y * rows_per_split;
    int i_hi = i_lo + rows_per_split;
    const int jmax = blockIdx.x * 64 + 63;  // rows beyond the last column of this block are zero (upper triangular)
    if (i_hi > jmax + 1) i_hi = jmax + 1;
    if (i_hi > n) i_hi = n;
    double part = 0.0;
    for (int i0 = i_lo + g; i0 < i_hi; i0 += 32) {  // 8 independent loads in flight per lane
        double mv[8], rv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = i0 + 4 * u;
            const bool ok = i < i_hi;
            mv[u] = ok ? W[(int64_t)i * ld + j] : 0.0;
            rv[u] = ok ? r[i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) part = __builtin_fma(mv[u], rv[u], part);
    }
    red[g][jl] = part;
    __syncthreads();
    if (g == 0) P[(int64_t)blockIdx.y * n_pad + j] = ((red[0][jl] + red[1][jl]) + red[2][jl]) + red[3][jl];
}

__global__ void k_sum_partials(const double *__restrict__ P, int nsplit, int n_pad, double *__restrict__ y) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_pad) return;
    double sacc = 0.0;
    for (int sp = 0; sp < nsplit; sp++) sacc += P[(int64_t)sp * n_pad + j];
    y[j] = sacc;
}

__global__ __launch_bounds__(256) void k_uptri_gemv(const double *__restrict__ W, int64_t ld, int n,
                                                    const double *__restrict__ y, double *__restrict__ z) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    const double *row = W + (int64_t)i * ld;
    double acc = 0.0;
    for (int k = (i & ~63) + lane; k < n; k += 64) acc = __builtin_fma((k >= i) ? row[k] : 0.0, y[k], acc);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) z[i] = acc;
}

__global__ void k_fill_rows(double *__restrict__ M, int64_t ld, int r0, int rows_pad, const double *__restrict__ src,
                            int64_t lds, int nrows, int ncols) {
    const int64_t total = (int64_t)rows_pad * ld;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int l = (int)(e / ld);
        const int i = (int)(e % ld);
        M[(int64_t)(r0 + l) * ld + i] = (l < nrows && i < ncols) ? src[(int64_t)l * lds + i] : 0.0;
    }
}

__global__ void k_gather_diag(const double *__restrict__ M, int64_t ld, int n, double *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = M[(int64_t)i * ld + i];
}

__global__ void k_fill_rows_batch(EvalBatchPtrs b, int64_t ld, int r0, int rows_pad, int64_t lds, int nrows, int ncols) {
    double *M = b.M[blockIdx.y];
    const double *src = b.rhsT[blockIdx.y];
    const int64_t total = (int64_t)rows_pad * ld;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int l = (int)(e / ld);
        const int i = (int)(e % ld);
        M[(int64_t)(r0 + l) * ld + i] = (l < nrows && i < ncols) ? src[(int64_t)l * lds + i] : 0.0;
    }
}

// What the host needs of `count` finished evaluations, written STRAIGHT into each candidate's pinned host buffers (device-
// visible, fine grained: complete when the launch is) by one launch: blockIdx.z = candidate, blockIdx.y = 0: the factor's
// diagonal (n), 1 .. q: the solved right-hand-side rows (q x n_pad, only with `rows`), plus `info` and the eight hand-off
// diagnostics words of the chain launches.  Replaces a gather kernel and four device-to-host copies PER CANDIDATE (~38 us
// each, serial on the lead's stream: 0.46 ms of the 8.7 ms of twelve n = 4096 candidates).
__global__ __launch_bounds__(256) void k_eval_tail(EvalBatchPtrs b, int64_t ld, int n, int n_pad, int rows, const int *__restrict__ sync) {
    const int z = blockIdx.z, y = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const double *M = b.M[z];
    if (y == 0) {
        if (i < n) b.h_diag[z][i] = M[(int64_t)i * ld + i];
        if (i < 9) b.h_info[z][i] = (i == 0) ? *b.d_info[z] : (sync ? sync[i - 1] : 0);
    } else if (rows && i < n_pad) {
        b.h_rows[z][(int64_t)(y - 1) * n_pad + i] = M[(int64_t)(n_pad + y - 1) * ld + i];
    }
}

// Device-side GLS (algorithm.rs:1007-1032 for p > 1 trend columns), helpers around the Gram matrix of [ft | yt]:
// G (rows x rows, ldg) <- -Gneg on the lower triangle of the leading q x q block, identity on the padding diagonal
__global__ void k_gram_finish(const double *__restrict__ Gneg, double *__restrict__ G, int64_t ldg, int rows, int q) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= rows) return;
    double v = 0.0;
    if (i < q && j <= i) v = -Gneg[(int64_t)i * ldg + j];
    else if (i >= q && j == i) v = 1.0;
    G[(int64_t)i * ldg + j] = v;
}

// rho (n_pad, zero beyond n) <- yt - sum_l beta_l ft_l  (ft_l = row l of ftT);  part[block] <- sum of rho^2 over the
// block's 256 points, reduced in a fixed order (the host adds the n_pad / 256 partial sums in order: deterministic)
__global__ __launch_bounds__(256) void k_gls_residual(const double *__restrict__ ftT, int64_t ld, const double *__restrict__ yt,
                                                      const double *__restrict__ beta, int p, int n, int n_pad,
                                                      double *__restrict__ rho, double *__restrict__ part) {
    __shared__ double bs[256];
    __shared__ double red[256];
    const int tid = threadIdx.x, i = blockIdx.x * 256 + tid;
    double acc = (i < n) ? yt[i] : 0.0;
    for (int l0 = 0; l0 < p; l0 += 256) {
        const int cnt = (p - l0 < 256) ? (p - l0) : 256;
        __syncthreads();
        if (tid < cnt) bs[tid] = beta[l0 + tid];
        __syncthreads();
        if (i < n)
            for (int l = 0; l < cnt; l++) acc = __builtin_fma(-bs[l], ftT[(int64_t)(l0 + l) * ld + i], acc);
    }
    if (i < n_pad) rho[i] = acc;
    red[tid] = acc * acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    if (tid == 0) part[blockIdx.x] = red[0];
}

__global__ void k_zero_upper(double *__restrict__ M, int64_t ld, int n) {
    const int64_t total = (int64_t)n * n;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / n), j = (int)(e % n);
        if (j > i) M[(int64_t)i * ld + j] = 0.0;
    }
}

// one wave per query row: s0 = sum_i rt_i^2 ; sl[l] = sum_i rt_i * ft_l[i]
__global__ __launch_bounds__(256) void k_row_reduce(const double *__restrict__ RT, int64_t ld, int m, int n,
                                                    const double *__restrict__ ftT, int64_t ldf, int p,
                                                    double *__restrict__ s0, double *__restrict__ sl) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (q >= m) return;
    const double *row = RT + (int64_t)q * ld;
    double acc = 0.0;
    for (int i = lane; i < n; i += 64) acc = __builtin_fma(row[i], row[i], acc);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) s0[q] = acc;
    for (int l = 0; l < p; l++) {
        const double *f = ftT + (int64_t)l * ldf;
        double a2 = 0.0;
        for (int i = lane; i < n; i += 64) a2 = __builtin_fma(row[i], f[i], a2);
        for (int o = 32; o > 0; o >>= 1) a2 += __shfl_xor(a2, o);
        if (lane == 0) sl[(int64_t)q * p + l] = a2;
    }
}

// ---------------------------------------------------------------------------------------------
// K7: likelihood-gradient accumulation (new capability).  With Rneg = -R^-1 (lower triangle) and
// gamma = C^-T rho, for every output o < nout
//     out[o] = sum_{i > j} 2 R_ij (gamma_i gamma_j / sigma2 + Rneg_ij) * d log R_ij / d c_o
//            = gamma^T dR_o gamma / sigma2 - tr(R^-1 dR_o)
// hcols == 1 (HC1): output o = input dimension o, c_o its coefficient (w = I: theta_o; KPLS sq-exp / abs-exp: the collapsed
// coefficient, the chain rule to theta runs on the host);  hcols > 1 (KPLS + Matern): output o = theta_o,
//     d log R / d theta_o = sum_j |w_jo| (d log m / d t)(theta_o |w_jo| a_j) a_j.
// d log r / d c for one dimension (a = |x_i - x_j|, t = c a):
//     sq-exp -c a^2 ; abs-exp -a ; Matern-3/2 -3 a t / (1 + s3 t) ; Matern-5/2 -(5/3) a t (1 + s5 t) / (1 + s5 t + 5/3 t^2)
// (the Matern forms are m'(t) / m(t) - sqrt(nu') with the constants cancelled: one reciprocal, no subtraction of nearly
// equal terms).
// Round 4: DETERMINISTIC (round 3 added per-tile sums into `out` with atomics, whose order varies from run to run): a
// workgroup walks the 64 x 64 tiles t = blockIdx.x, + gridDim.x, ... of the lower triangle, keeps its DK partial sums
// per lane in registers over all of them, reduces them once (shuffles, then the four waves in a fixed order) and writes
// ONE partial vector; k_grad_reduce adds the workgroups' vectors in a fixed order.  A candidate's gradient is therefore
// the same bits alone and in a lock-step batch (grid.z), whatever else runs.  Two passes per tile: (1) all d dimensions
// -> R_ij and the pair's weight, (2) the DK outputs of this workgroup's chunk (grid.y) -> DK accumulations per pair.
// Dimensions are staged in LDS in chunks of at most 64 (any d).  The division of round 3 (~30 FP64 instructions per pair
// and dimension) is v_rcp_f64 + two Newton steps (m >= 1: no scaling needed).
// ---------------------------------------------------------------------------------------------
constexpr int kGradDC = 64;      // input dimensions staged per pass (2 x 64 x 64 doubles = 64 KB of LDS)
constexpr int kGradMaxWG = 2048;  // workgroups (partial vectors) per candidate and output chunk

__device__ __forceinline__ double rcp_ge1(double m) {  // 1 / m for m >= 1 (finite), <= 1 ulp-ish
    double y = __builtin_amdgcn_rcp(m);
    double e = __builtin_fma(-m, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-m, y, 1.0);
    return __builtin_fma(y, e, y);
}
template <int CORR>
__device__ __forceinline__ double dlog_dc(double c, double a) {
    if (CORR == EGX_CORR_SQUARED_EXPONENTIAL) return -c * a * a;
    if (CORR == EGX_CORR_ABSOLUTE_EXPONENTIAL) return -a;
    const double t = c * a;
    if (CORR == EGX_CORR_MATERN32) return -3.0 * a * t * rcp_ge1(__builtin_fma(kSqrt3, t, 1.0));
    const double u = __builtin_fma(kSqrt5, t, 1.0);
    return -k5over3 * (a * t) * u * rcp_ge1(__builtin_fma(k5over3 * t, t, u));
}

// The same with the coefficient already inside the distance (t = c a; HC1 with every c > 0, inputs prescaled as for K1's
// scalar-row form): d log r / d c = F(c) G(t) with G below and F = -1/c (sq-exp, abs-exp), -3/c (Matern-3/2),
// -(5/3)/c (Matern-5/2) applied once per output when the partial sum is written.  One Newton step on v_rcp_f64 (2^-26
// or better, squared: the last bit of a double; the gradient is asserted at 1e-6).
__device__ __forceinline__ double rcp_ge1_fast(double m) {
    const double y = __builtin_amdgcn_rcp(m);
    return __builtin_fma(__builtin_fma(-m, y, 1.0), y, y);
}
template <int CORR>
__device__ __forceinline__ double dlog_g_scaled(double t) {  // t >= 0
    if (CORR == EGX_CORR_SQUARED_EXPONENTIAL) return t * t;
    if (CORR == EGX_CORR_ABSOLUTE_EXPONENTIAL) return t;
    const double t2 = t * t;
    if (CORR == EGX_CORR_MATERN32) return t2 * rcp_ge1_fast(__builtin_fma(kSqrt3, t, 1.0));
    const double u = __builtin_fma(kSqrt5, t, 1.0);
    return (t2 * u) * rcp_ge1_fast(__builtin_fma(k5over3, t2, u));
}
template <int CORR>
__device__ __forceinline__ double dlog_f_scaled(double c) {
    if (CORR == EGX_CORR_MATERN32) return -3.0 / c;
    if (CORR == EGX_CORR_MATERN52) return -k5over3 / c;
    return -1.0 / c;
}

// MODE 0: hcols > 1 (KPLS + Matern);  1: hcols == 1, raw inputs;  2: hcols == 1, inputs prescaled per candidate (gb.xs)
template <int CORR, int MODE, int DK>
__global__ __launch_bounds__(256) void k_grad_accum(const double *__restrict__ xT, int64_t ldx, int n, int d, int hcols,
                                                    const double *__restrict__ wabs, int nout, int64_t ld, int ntiles,
                                                    int nout_pad, GradBatch gb) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int z = blockIdx.z;
    constexpr bool HC1 = MODE != 0, PRE = MODE == 2;
    if (PRE) xT = gb.xs[z];  // this candidate's c_k x_k (k-major like xT)
    const double *__restrict__ coef = gb.coef[z];
    const double *__restrict__ gamma = gb.gamma[z];
    const double *__restrict__ Rneg = gb.rneg[z];
    const double inv_s2 = gb.inv_s2[z];
    const int dc = d < kGradDC ? d : kGradDC;
    double *xi = sm, *xj = sm + dc * 64;
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const int o0 = blockIdx.y * DK;
    double acc[DK];
#pragma unroll
    for (int oo = 0; oo < DK; oo++) acc[oo] = 0.0;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        int bi = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
        while (bi * (bi + 1) / 2 > t) bi--;
        while ((bi + 1) * (bi + 2) / 2 <= t) bi++;
        const int bj = t - bi * (bi + 1) / 2;
        // ---- pass 1: the correlations of the tile's 16 pairs per lane (all d dimensions, staged dc at a time)
        PairAcc<CORR> pa[4][4];
        for (int c0 = 0; c0 < d; c0 += dc) {
            const int dn = (d - c0 < dc) ? (d - c0) : dc;
            __syncthreads();  // the previous readers of the slabs are done
            stage_slab(xi, xT + (int64_t)c0 * ldx, ldx, bi * 64, dn, tid);
            stage_slab(xj, xT + (int64_t)c0 * ldx, ldx, bj * 64, dn, tid);
            __syncthreads();
            for (int k = 0; k < dn; k++) {
                double vi[4], vj[4];
#pragma unroll
                for (int a = 0; a < 4; a++) vi[a] = xi[k * 64 + ty * 4 + a];
#pragma unroll
                for (int b = 0; b < 4; b++) vj[b] = xj[k * 64 + tx * 4 + b];
                const double *ck = coef + (int64_t)(c0 + k) * hcols;
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        if (PRE) pa[a][b].add_scaled(vi[a] - vj[b]);
                        else pa[a][b].add(vi[a] - vj[b], ck, hcols);
                    }
            }
        }
        double wp[4][4];  // 2 R_ij (gamma_i gamma_j / sigma2 - Rinv_ij) for i > j inside the matrix, else 0
        {
            double gi[4], gj[4];
#pragma unroll
            for (int a = 0; a < 4; a++) gi[a] = gamma[bi * 64 + ty * 4 + a] * inv_s2;
#pragma unroll
            for (int b = 0; b < 4; b++) gj[b] = gamma[bj * 64 + tx * 4 + b];
#pragma unroll
            for (int a = 0; a < 4; a++) {
                const int i = bi * 64 + ty * 4 + a;
                const double *rrow = Rneg + (int64_t)i * ld + bj * 64 + tx * 4;
                const double2 r01 = *reinterpret_cast<const double2 *>(rrow);
                const double2 r23 = *reinterpret_cast<const double2 *>(rrow + 2);
                const double rn[4] = {r01.x, r01.y, r23.x, r23.y};
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const int j = bj * 64 + tx * 4 + b;
                    const bool on = (i > j) && (i < n);
                    wp[a][b] = on ? 2.0 * pa[a][b].value() * __builtin_fma(gi[a], gj[b], rn[b]) : 0.0;
                }
            }
        }
        // ---- pass 2: this workgroup's outputs o0 .. o0 + DK - 1
        if (HC1) {
            // output o <-> input dimension o.  d > dc: the chunk's dimensions are staged again, from slab row 0
            int base = 0;
            if (d > dc) {
                const int dn = (nout - o0 < DK) ? (nout - o0) : DK;
                __syncthreads();
                stage_slab(xi, xT + (int64_t)o0 * ldx, ldx, bi * 64, dn, tid);
                stage_slab(xj, xT + (int64_t)o0 * ldx, ldx, bj * 64, dn, tid);
                __syncthreads();
                base = o0;
            }
#pragma unroll
            for (int oo = 0; oo < DK; oo++) {
                const int o = o0 + oo;
                if (o < nout) {
                    const double c = coef[o];
                    double vi[4], vj[4];
#pragma unroll
                    for (int a = 0; a < 4; a++) vi[a] = xi[(o - base) * 64 + ty * 4 + a];
#pragma unroll
                    for (int b = 0; b < 4; b++) vj[b] = xj[(o - base) * 64 + tx * 4 + b];
                    double sacc = acc[oo];
#pragma unroll
                    for (int a = 0; a < 4; a++)
#pragma unroll
                        for (int b = 0; b < 4; b++) {
                            const double ad = fabs(vi[a] - vj[b]);
                            sacc = __builtin_fma(wp[a][b], PRE ? dlog_g_scaled<CORR>(ad) : dlog_dc<CORR>(c, ad), sacc);
                        }
                    acc[oo] = sacc;
                }
            }
        } else {
            // KPLS + Matern: output o = theta_o gets a term from every input dimension j
            for (int c0 = 0; c0 < d; c0 += dc) {
                const int dn = (d - c0 < dc) ? (d - c0) : dc;
                if (d > dc) {
                    __syncthreads();
                    stage_slab(xi, xT + (int64_t)c0 * ldx, ldx, bi * 64, dn, tid);
                    stage_slab(xj, xT + (int64_t)c0 * ldx, ldx, bj * 64, dn, tid);
                    __syncthreads();
                }
                for (int k = 0; k < dn; k++) {
                    double ad[4][4];
                    {
                        double vi[4], vj[4];
#pragma unroll
                        for (int a = 0; a < 4; a++) vi[a] = xi[k * 64 + ty * 4 + a];
#pragma unroll
                        for (int b = 0; b < 4; b++) vj[b] = xj[k * 64 + tx * 4 + b];
#pragma unroll
                        for (int a = 0; a < 4; a++)
#pragma unroll
                            for (int b = 0; b < 4; b++) ad[a][b] = fabs(vi[a] - vj[b]);
                    }
#pragma unroll
                    for (int oo = 0; oo < DK; oo++) {
                        const int o = o0 + oo;
                        if (o < nout) {
                            const double c = coef[(int64_t)(c0 + k) * hcols + o], wa = wabs[(int64_t)(c0 + k) * hcols + o];
                            if (wa != 0.0) {  // (wave-uniform)
                                double sacc = 0.0;
#pragma unroll
                                for (int a = 0; a < 4; a++)
#pragma unroll
                                    for (int b = 0; b < 4; b++) sacc = __builtin_fma(wp[a][b], dlog_dc<CORR>(c, ad[a][b]), sacc);
                                acc[oo] = __builtin_fma(wa, sacc, acc[oo]);
                            }
                        }
                    }
                }
            }
        }
    }
    // ---- one reduction per workgroup: lanes of a wave (xor shuffles), then the four waves in a fixed order
    __syncthreads();
    double *red = sm;  // [4][DK]
#pragma unroll
    for (int oo = 0; oo < DK; oo++) {
        double v = acc[oo];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((tid & 63) == 0) red[(tid >> 6) * DK + oo] = v;
    }
    __syncthreads();
    if (tid < DK && o0 + tid < nout) {
        double v = ((red[tid] + red[DK + tid]) + red[2 * DK + tid]) + red[3 * DK + tid];
        if (PRE) v *= dlog_f_scaled<CORR>(coef[o0 + tid]);
        gb.part[z][(int64_t)blockIdx.x * nout_pad + o0 + tid] = v;
    }
}

// out[z][o] = sum over the nwg partial vectors, in a fixed order: 256 lanes take every 256th, then a tree in LDS
__global__ __launch_bounds__(256) void k_grad_reduce(int nwg, int nout, int nout_pad, GradBatch gb) {
    __shared__ double red[256];
    const int o = blockIdx.x, z = blockIdx.y, tid = threadIdx.x;
    const double *part = gb.part[z];
    double v = 0.0;
    for (int b = tid; b < nwg; b += 256) v += part[(int64_t)b * nout_pad + o];
    red[tid] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0) gb.out[z][o] = red[0];
}

// =============================================================================================
#define EGX_DISPATCH_CORR(corr, CALL)                                                        \
    switch (corr) {                                                                          \
        case EGX_CORR_SQUARED_EXPONENTIAL: { constexpr int C_ = EGX_CORR_SQUARED_EXPONENTIAL; CALL; } break;   \
        case EGX_CORR_ABSOLUTE_EXPONENTIAL: { constexpr int C_ = EGX_CORR_ABSOLUTE_EXPONENTIAL; CALL; } break; \
        case EGX_CORR_MATERN32: { constexpr int C_ = EGX_CORR_MATERN32; CALL; } break;       \
        case EGX_CORR_MATERN52: { constexpr int C_ = EGX_CORR_MATERN52; CALL; } break;       \
        default: set_error("unknown correlation kind"); return EGX_ERR_INVALID_VALUE;        \
    }

int launch_corr_sym(hipStream_t s, int corr, const double *xT, int64_t ldx, int n, int d, const double *coef,
                    int hcols, double nugget, double *M, int64_t ld, int n_pad, double *xs_scratch) {
    const int nt2 = n_pad / 128;  // n_pad is a multiple of 128
    dim3 grid((unsigned)(4 * (nt2 * (nt2 + 1) / 2)));
    const int dc = d < kCorrDC ? d : kCorrDC;  // dimensions staged at a time
    const size_t lds = (size_t)2 * dc * 64 * sizeof(double);
    // (the LDS-only form was measured against it: 0.51 vs 0.42 ms sq-exp, 1.10 vs 0.76 at d = 64: profiles/r03_run11_k1_scalar_rows_ab.txt)
    if (hcols == 1 && xs_scratch != nullptr && d <= kCorrDC) {
        // scalar-row form on prescaled inputs (xs_scratch: d x ldx doubles, owned by the caller's workspace); d > 64: the
        // LDS-only form below, which stages the dimensions in chunks
        hipLaunchKernelGGL(k_scale_rows, dim3((unsigned)((ldx + 255) / 256)), dim3(256), 0, s, xT, ldx, d, coef, xs_scratch);
        EGX_DISPATCH_CORR(corr, hipLaunchKernelGGL((k_corr_sym_srow<C_>), grid, dim3(256), lds / 2, s, (const double *)xs_scratch, ldx,
                                                   n, d, 1.0 + nugget, M, ld));
        EGX_HIP_CHECK(hipGetLastError());
        return EGX_SUCCESS;
    }
    // (forms with more pairs per lane -- 8 x 4 in a 128x128 tile or in the same tile with 128 threads -- were measured SLOWER:
    //  profiles/r03_run5_k1_tile128_discarded.jsonl, profiles/r03_run10_k1_rows8_discarded.txt)
    if (hcols == 1) {
        EGX_DISPATCH_CORR(corr, hipLaunchKernelGGL((k_corr_sym<C_, true, 4>), grid, dim3(256), lds, s, xT, ldx, n, d, coef, hcols,
                                                   1.0 + nugget, M, ld));
    } else {
        EGX_DISPATCH_CORR(corr, hipLaunchKernelGGL((k_corr_sym<C_, false, 4>), grid, dim3(256), lds, s, xT, ldx, n, d, coef, hcols,
                                                   1.0 + nugget, M, ld));
    }
    EGX_HIP_CHECK(hipGetLastError());
    return EGX_SUCCESS;
}

int launch_cross_corr(hipStream_t s, int corr, const double *xqT, int64_t ldq, int m_pad, const double *xT,
                      int64_t ldx, int n_pad, int d, const double *coef, int hcols, double *R, int64_t ld) {
    dim3 grid(m_pad / 64, n_pad / 64);
    const size_t lds = (size_t)2 * (d < kCorrDC ? d : kCorrDC) * 64 * sizeof(double);
    if (hcols == 1) {
        EGX_DISPATCH_CORR(corr, hipLaunchKernelGGL((k_cross_corr<C_, true>), grid, dim3(256), lds, s, xqT, ldq, xT, ldx, d, coef,
                                                   hcols, R, ld));
    } else {
        EGX_DISPATCH_CORR(corr, hipLaunchKernelGGL((k_cross_corr<C_, false>), grid, dim3(256), lds, s, xqT, ldq, xT, ldx, d, coef,
                                                   hcols, R, ld));
    }
    EGX_HIP_CHECK(hipGetLastError());
    return EGX_SUCCESS;
}

int launch_normalize_queries(hipStream_t s, const double *xq, int m, int d, const double *par, double *xqT, int64_t ldq,
                             int m_pad) {
    const size_t lds = (size_t)64 * ((d < kCorrDC ? d : kCorrDC) | 1) * sizeof(double);
    hipLaunchKernelGGL(k_normalize_queries, dim3(m_pad / 64, (unsigned)((d + kCorrDC - 1) / kCorrDC)), dim3(256), lds, s, xq, m, d,
                       par, xqT, ldq);
    EGX_HIP_CHECK(hipGetLastError());
    return EGX_SUCCESS;
}

int launch_scale_rows(hipStream_t s, const double *xT, int64_t ldx, int d, const double *coef, double *xs) {
    hipLaunchKernelGGL(k_scale_rows, dim3((unsigned)((ldx + 255) / 256)), dim3(256), 0, s, xT, ldx, d, coef, xs);
    EGX_HIP_CHECK(hipGetLastError());
    return EGX_SUCCESS;
}

int launch_predict_mean(hipStream_t s, int corr, const double *xqT, int64_t ldq, int m_pad, const double *xT,
                        int64_t ldx, int n_pad, int d, const double *coef, int hcols, const double *gamma,
                        double *racc, int nsplit, const double *xs_prescaled) {
    const size_t lds = (size_t)(2 * (d < kCorrDC ? d : kCorrDC) * 64 + 64) * sizeof(double);
    const int slabs = n_pad / 64;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > slabs) nsplit = slabs;
    const int per = (slabs + nsplit - 1) / nsplit;
    nsplit = (slabs + per - 1) / per;
    if (hcols == 1 && xs_prescaled != nullptr && d <= kCorrDC) {
        const size_t lds_s = (size_t)(d * 64 + 256) * sizeof(double);
        EGX_DISPATCH_CORR(corr, hipLaunchKernelGGL((k_predict_mean_srow<C_>), dim3(m_pad / 64, nsplit), dim3(256), lds_s, s, xqT,
                                                   ldq, xs_prescaled, ldx, n_pad, d, coef, gamma, racc, per, m_pad));
        EGX_HIP_CHECK(hipGetLastError());
        return EGX_SUCCESS;
    }
    if (hcols == 1) {
        EGX_DISPATCH_CORR(corr, hipLaunchKernelGGL((k_predict_mean<C_, true>), dim3(m_pad / 64, nsplit), dim3(256), lds, s, xqT,
                                                   ldq, xT, ldx, n_pad, d, coef, hcols, gamma, racc, per, m_pad));
    } else {
        EGX_DISPATCH_CORR(corr, hipLaunchKernelGGL((k_predict_mean<C_, false>), dim3(m_pad / 64, nsplit), dim3(256), lds, s, xqT,
                                                   ldq, xT, ldx, n_pad, d, coef, hcols, gamma, racc, per, m_pad));
    }
    EGX_HIP_CHECK(hipGetLastError());
    return EGX_SUCCESS;
}

// out: nsplit x m_pad x d partial sums (nsplit returned); Wt = gamma (vec != 0) or the (n x m_pad) weight matrix
int launch_xgrad(hipStream_t s, int corr, const double *xqT, int64_t ldq, int m_pad, const double *xT, int64_t ldx,
                 int n, int d, const double *coef, int hcols, const double *Wt, int64_t ldw, int vec, int nsplit,
                 double *out) {
    const int slabs = (n + 63) / 64;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > slabs) nsplit = slabs;
    const int per = (slabs + nsplit - 1) / nsplit;
    dim3 grid(m_pad / kXgThreads, nsplit);
    // d <= 64: the lane's query coordinates in LDS; beyond: from global memory (XAG), the training slab alone in LDS,
    // as many training points wide (64, 32, ... 1) as fit beside the d x hcols coefficients
    const bool xag = d > kCorrDC;
    int js_shift = 6;
    auto lds_for = [&](int sh) { return (size_t)((xag ? 0 : d * kXgThreads) + ((size_t)d << sh) + (size_t)d * hcols) * sizeof(double); };
    while (js_shift > 0 && lds_for(js_shift) > kLdsBytes) js_shift--;
    const size_t lds = lds_for(js_shift);
    if (lds > kLdsBytes) {
        set_error("x-gradients: " + std::to_string(d) + " inputs x " + std::to_string(hcols) +
                  " coefficient columns do not fit the 160 KB of LDS (d * (hcols + 1) <= 20480)");
        return EGX_ERR_UNSUPPORTED;
    }
#define EGX_XG1(C_, DK_, VEC_, XAG_)                                                                                         \
    {                                                                                                                        \
        if (lds > 65536)                                                                                                     \
            EGX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_xgrad<C_, DK_, VEC_, XAG_>),                 \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                        \
        hipLaunchKernelGGL((k_xgrad<C_, DK_, VEC_, XAG_>), grid, dim3(kXgThreads), lds, s, xqT, ldq, xT, ldx, n, d, coef,    \
                           hcols, Wt, ldw, per, out, m_pad, js_shift);                                                       \
    }
#define EGX_XG(C_, DK_)                                    \
    if (xag) {                                             \
        if (vec) EGX_XG1(C_, 32, true, true)               \
        else EGX_XG1(C_, 32, false, true)                  \
    } else if (vec) EGX_XG1(C_, DK_, true, false)          \
    else EGX_XG1(C_, DK_, false, false)
    if (d <= 8) {
        EGX_DISPATCH_CORR(corr, EGX_XG(C_, 8));
    } else if (d <= 16) {
        EGX_DISPATCH_CORR(corr, EGX_XG(C_, 16));
    } else {
        EGX_DISPATCH_CORR(corr, EGX_XG(C_, 32));
    }
#undef EGX_XG1
#undef EGX_XG
    EGX_HIP_CHECK(hipGetLastError());
    return EGX_SUCCESS;
}

// few queries (m <= 8), weights = a vector over the training points: out is (ceil(n / 256) x m x d) partial sums
int launch_xgrad_point(hipStream_t s, int corr, const double *xqT, int64_t ldq, int m, const double *xT, int64_t ldx, int n,
                       int d, const double *coef, int hcols, const double *wvec, double *out) {
    dim3 grid(m, (n + 255) / 256);
    const size_t lds = (size_t)(d + (size_t)d * hcols + 4 * (size_t)d) * sizeof(double);
    if (lds > kLdsBytes) {
        set_error("x-gradients: " + std::to_string(d) + " inputs x " + std::to_string(hcols) +
                  " coefficient columns do not fit the 160 KB of LDS (d * (hcols + 5) <= 20480)");
        return EGX_ERR_UNSUPPORTED;
    }
#define EGX_XGP(C_)                                                                                                          \
    {                                                                                                                        \
        if (lds > 65536)                                                                                                     \
            EGX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_xgrad_point<C_>),                            \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                        \
        hipLaunchKernelGGL(k_xgrad_point<C_>, grid, dim3(256), lds, s, xqT, ldq, xT, ldx, n, d, coef, hcols, wvec, out, m);  \
    }
    EGX_DISPATCH_CORR(corr, EGX_XGP(C_));
#undef EGX_XGP
    EGX_HIP_CHECK(hipGetLastError());
    return EGX_SUCCESS;
}

// y (n_pad) <- W^T r, z (n_pad) <- W y for the upper-triangular W (n_pad x n_pad, ld); P: scratch of 32 * n_pad doubles
int launch_uptri_solve_pair(hipStream_t s, const double *W, int64_t ld, int n, int n_pad, const double *r, double *P,
                            double *y, double *z) {
    int nsplit = (n + 255) / 256;
    if (nsplit > 32) nsplit = 32;
    const int per = (int)round_up((n + nsplit - 1) / nsplit, 4);
    nsplit = (n + per - 1) / per;
    hipLaunchKernelGGL(k_uptri_gemv_t, dim3(n_pad / 64, nsplit), dim3(256), 0, s, W, ld, n, r, per, P, n_pad);
    hipLaunchKernelGGL(k_sum_partials, dim3((n_pad + 255) / 256), dim3(256), 0, s, (const double *)P, nsplit, n_pad, y);
    hipLaunchKernelGGL(k_uptri_gemv, dim3((n + 3) / 4), dim3(256), 0, s, W, ld, n, (const double *)y, z);
    EGX_HIP_CHECK(hipGetLastError());
    return EGX_SUCCESS;
}

int launch_fill_rows(hipStream_t s, double *M, int64_t ld, int r0, int rows_pad, const double *src, int64_t lds,
                     int nrows, int ncols) {
    hipLaunchKernelGGL(k_fill_rows, dim3(1024), dim3(256), 0, s, M, ld, r0, rows_pad, src, lds, nrows, ncols);
    EGX_HIP_CHECK(hipGetLastError());
    return EGX_SUCCESS;
}

// correlation matrices + right-hand-side rows of `count` candidates (scalar-row form only: hcols == 1, d <= 64; the caller
// falls back to launch_corr_sym / launch_fill_rows per candidate otherwise): returns EGX_ERR_UNSUPPORTED without launching
int launch_eval_front_batch(hipStream_t s, int corr, const EvalBatchPtrs &b, int count, int64_t ldx, int n, int d, int hcols,
                            double nugget, int64_t ld, int n_pad, int rhs_pad, int q) {
    if (hcols != 1 || d > kCorrDC || count < 1 || count > EvalBatchPtrs::kMax) return EGX_ERR_UNSUPPORTED;
    const int nt2 = n_pad / 128;
    const dim3 grid((unsigned)(4 * (nt2 * (nt2 + 1) / 2)), (unsigned)count);
    const size_t lds = (size_t)d * 64 * sizeof(double);
    hipLaunchKernelGGL(k_scale_rows_batch, dim3((unsigned)((ldx + 255) / 256), (unsigned)count), dim3(256), 0, s, b, ldx, d);
    EGX_DISPATCH_CORR(corr, hipLaunchKernelGGL((k_corr_sym_srow_batch<C_>), grid, dim3(256), lds, s, b, ldx, n, d, 1.0 + nugget, ld));
    hipLaunchKernelGGL(k_fill_rows_batch, dim3(256, (unsigned)count), dim3(256), 0, s, b, ld, n_pad, rhs_pad, ldx, q, n_pad);
    EGX_HIP_CHECK(hipGetLastError());
    return EGX_SUCCESS;
}

int launch_eval_tail(hipStream_t s, const EvalBatchPtrs &b, int count, int64_t ld, int n, int n_pad, int q, int rows, const int *sync) {
    hipLaunchKernelGGL(k_eval_tail, dim3((unsigned)((n_pad + 255) / 256), (unsigned)(rows ? 1 + q : 1), (unsigned)count), dim3(256), 0, s, b,
                       ld, n, n_pad, rows, sync);
    EGX_HIP_CHECK(hipGetLastError());
    return EGX_SUCCESS;
}

int launch_gather_diag(hipStream_t s, const double *M, int64_t ld, int n, double *out) {
    hipLaunchKernelGGL(k_gather_diag, dim3((n + 255) / 256), dim3(256), 0, s, M, ld, n, out);
    EGX_HIP_CHECK(hipGetLastError());
    return EGX_SUCCESS;
}

int launch_gram_finish(hipStream_t s, const double *Gneg, double *G, int64_t ldg, int rows, int q) {
    hipLaunchKernelGGL(k_gram_finish, dim3((rows + 255) / 256, rows), dim3(256), 0, s, Gneg, G, ldg, rows, q);
    EGX_HIP_CHECK(hipGetLastError());
    return EGX_SUCCESS;
}

int launch_gls_residual(hipStream_t s, const double *ftT, int64_t ld, const double *yt, const double *beta, int p, int n,
                        int n_pad, double *rho, double *part) {
    hipLaunchKernelGGL(k_gls_residual, dim3((n_pad + 255) / 256), dim3(256), 0, s, ftT, ld, yt, beta, p, n, n_pad, rho,
                       part);
    EGX_HIP_CHECK(hipGetLastError());
    return EGX_SUCCESS;
}

int launch_zero_upper(hipStream_t s, double *M, int64_t ld, int n) {
    hipLaunchKernelGGL(k_zero_upper, dim3(2048), dim3(256), 0, s, M, ld, n);
    EGX_HIP_CHECK(hipGetLastError());
    return EGX_SUCCESS;
}

int launch_row_reduce(hipStream_t s, const double *RT, int64_t ld, int m, int n, const double *ftT, int64_t ldf,
                      int p, double *s0, double *sl) {
    hipLaunchKernelGGL(k_row_reduce, dim3((m + 3) / 4), dim3(256), 0, s, RT, ld, m, n, ftT, ldf, p, s0, sl);
    EGX_HIP_CHECK(hipGetLastError());
    return EGX_SUCCESS;
}

int grad_partial_doubles(int nout) { return kGradMaxWG * (int)round_up(nout, 32); }

// batch.count candidates (grid.z); per candidate: coef (d x hcols), gamma (n_pad), rneg (-R^-1, lower, ld), inv_s2,
// part (grad_partial_doubles(nout) doubles of scratch), out (nout doubles)
// prescaled != 0 (hcols == 1, every coefficient > 0): batch.xs[z] holds candidate z's inputs times its coefficients
int launch_grad_accum(hipStream_t s, int corr, const double *xT, int64_t ldx, int n, int d, int hcols, const double *wabs,
                      int nout, int64_t ld, const GradBatch &batch, int prescaled) {
    if (batch.count < 1 || batch.count > kGradMaxBatch) {
        set_error("grad_accum: batch count out of range");
        return EGX_ERR_INVALID_VALUE;
    }
    const int nt = (n + 63) / 64, ntiles = nt * (nt + 1) / 2;
    const int nwg = ntiles < kGradMaxWG ? ntiles : kGradMaxWG;
    const int nout_pad = (int)round_up(nout, 32);
    const int dc = d < kGradDC ? d : kGradDC;
    const int dk = nout <= 8 ? 8 : (nout <= 16 ? 16 : 32);
    const size_t lds = sizeof(double) * (size_t)std::max(2 * dc * 64, 4 * dk);
    const dim3 grid((unsigned)nwg, (unsigned)((nout + dk - 1) / dk), (unsigned)batch.count);
#define EGX_GA(C_, MODE_, DK_)                                                                                             \
    hipLaunchKernelGGL((k_grad_accum<C_, MODE_, DK_>), grid, dim3(256), lds, s, xT, ldx, n, d, hcols, wabs, nout, ld, ntiles, \
                       nout_pad, batch)
#define EGX_GA_DK(C_, MODE_)                  \
    if (dk == 8) EGX_GA(C_, MODE_, 8);        \
    else if (dk == 16) EGX_GA(C_, MODE_, 16); \
    else EGX_GA(C_, MODE_, 32)
    if (hcols == 1 && prescaled) {
        EGX_DISPATCH_CORR(corr, EGX_GA_DK(C_, 2));
    } else if (hcols == 1) {
        EGX_DISPATCH_CORR(corr, EGX_GA_DK(C_, 1));
    } else {
        EGX_DISPATCH_CORR(corr, EGX_GA_DK(C_, 0));
    }
#undef EGX_GA_DK
#undef EGX_GA
    hipLaunchKernelGGL(k_grad_reduce, dim3((unsigned)nout, (unsigned)batch.count), dim3(256), 0, s, nwg, nout, nout_pad, batch);
    EGX_HIP_CHECK(hipGetLastError());
    return EGX_SUCCESS;
}

// z (n) <- W y for the upper triangular W (row-major): gamma = C^-T rho through the explicit C^-T of the theta-gradient
int launch_uptri_gemv(hipStream_t s, const double *W, int64_t ld, int n, const double *y, double *z) {
    hipLaunchKernelGGL(k_uptri_gemv, dim3((n + 3) / 4), dim3(256), 0, s, W, ld, n, y, z);
    EGX_HIP_CHECK(hipGetLastError());
    return EGX_SUCCESS;
}

}  // namespace egx
