// Sparse Gaussian process (FITC / VFE) on gfx950 -- SURVEY 8f rank 4, the reference's own large-n model:
//   SgpValidParams::reduced_likelihood / fitc / vfe   crates/gp/src/sparse_algorithm.rs:654-831
//   SparseGaussianProcess::predict / predict_var       crates/gp/src/sparse_algorithm.rs:237-257
//   SgpValidParams::fit (objective, parameter layout)  crates/gp/src/sparse_algorithm.rs:422-650
// n training points, nz << n inducing points; everything O(n nz^2) runs on the kernels of the dense path:
//   R_nz                     k_cross_corr (rows = training points, columns = inducing points; raw x, no normalisation)
//   Kmm = sigma2 (R_zz + nugget/sigma2 I) = U U^T, U = sigma C_z      k_corr_sym + launch_potrf
//   V^T = sigma R_nz C_z^-T  (n x nz rows)                            launch_trsm_rows
//   nu_t = sigma2 (1 - |v~_t|^2) + noise                              k_row_reduce
//   G = V diag(beta) V^T, V (beta o y), y^T diag(beta) y              ONE NT GEMM on the transposed, scaled and
//                                                                     y-augmented copy W = [sigma V~ sqrt(beta); sqrt(beta) y]
//   A = I + G = L L^T,  b = L^-1 V (beta o y)                         launch_potrf with the right-hand side riding below A
// The Woodbury matrix of the reference (nz x nz inverses) is never formed for predictions:
//   kx^T Kmm^-1 kx = |U^-1 kx|^2,   kx^T U^-T (L L^T)^-1 U^-1 kx = |L^-1 U^-1 kx|^2   (two launch_trsm_rows + k_row_reduce)
#include <cmath>
#include <cstring>
#include <limits>
#include <mutex>
#include <vector>

#include "egx_internal.h"
#include "cobyla.h"

using namespace egx;

namespace {

// outcome of one start of the multistart optimisation (objective, minimiser in log10 parameters, evaluations)
struct StartResult {
    double f;
    std::vector<double> x;
    int64_t evals;
};

struct Dev {
    double *p = nullptr;
    ~Dev() {
        if (p) (void)hipFree(p);
    }
    int alloc(size_t n) {
        if (p) (void)hipFree(p);
        p = nullptr;
        EGX_HIP_CHECK(dev_malloc(&p, sizeof(double) * (n ? n : 1)));
        return EGX_SUCCESS;
    }
};

// W[i][t] = scale * sb[t] * RT[t][i]  (i < nz, t < n), row z_pad: W[z_pad][t] = sb[t] * y[t]; everything else 0.
// RT is (n_pad x z_pad) row-major, W is (zext x n_pad) row-major; 32x32 LDS transpose tiles.
__global__ __launch_bounds__(256) void k_sgp_transpose_scale(const double *__restrict__ RT, int z_pad, int n, int nz,
                                                             const double *__restrict__ sb, const double *__restrict__ y,
                                                             double scale, double *__restrict__ W, int n_pad) {
    __shared__ double tile[32][33];
    const int t0 = blockIdx.x * 32, i0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    if (i0 < z_pad) {
        for (int r = ty; r < 32; r += 8) {
            const int t = t0 + r, i = i0 + tx;
            tile[r][tx] = (t < n && i < nz) ? RT[(int64_t)t * z_pad + i] * sb[t] * scale : 0.0;
        }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) W[(int64_t)(i0 + r) * n_pad + t0 + tx] = tile[tx][r];
    } else {  // augmented block of 128 rows: only its first row carries data
        for (int r = ty; r < 32; r += 8) {
            const int i = i0 + r, t = t0 + tx;
            W[(int64_t)i * n_pad + t] = (i == z_pad && t < n) ? sb[t] * y[t] : 0.0;
        }
    }
}

// sb[t] = sqrt(beta_t): FITC beta_t = 1 / (sigma2 (1 - s0_t) + noise), VFE beta = 1 / max(noise, nugget)
__global__ void k_sgp_sqrt_beta(const double *__restrict__ s0, int n, double sigma2, double noise, double nugget,
                                int vfe, double *__restrict__ sb) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const double nu = vfe ? fmax(noise, nugget) : sigma2 * (1.0 - s0[t]) + noise;
    sb[t] = (nu > 0.0) ? sqrt(1.0 / nu) : 0.0;  // nu <= 0 is reported by the host from s0
}

// A (zext rows x z_pad cols, ld = z_pad): lower part of I - Gneg for the first z_pad rows, rhs = -Gneg[z_pad][.] in row
// z_pad, zeros below.  Gneg is (zext x zext) with the LOWER tiles valid.
__global__ void k_sgp_form_a(const double *__restrict__ Gneg, int zext, int z_pad, double *__restrict__ A) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= z_pad) return;
    double v = 0.0;
    if (i < z_pad) {
        if (j <= i) v = ((i == j) ? 1.0 : 0.0) - Gneg[(int64_t)i * zext + j];
    } else if (i == z_pad) {
        v = -Gneg[(int64_t)i * zext + j];
    }
    A[(int64_t)i * z_pad + j] = v;
}

struct Eval {
    double lkh = -std::numeric_limits<double>::infinity();
    int status = EGX_STATUS_OK;
};

}  // namespace

struct egx_sgp {
    int device = 0, corr = 0, method = 0;
    double nugget = 0.0;
    int n = 0, d = 0, nz = 0, n_pad = 0, z_pad = 0, zext = 0;
    std::vector<double> y_host;
    double yty = 0.0;
    hipStream_t stream = nullptr;
    Dev xT, zT, y, coef, RT, W, G, P, Kz, A, dinv_z, dinv_a, s0, sb, diag, brow, vec, wall, tmpv;
    int *d_info = nullptr;
    std::mutex mu;
    // fitted state
    bool fitted = false;
    std::vector<double> theta;
    double sigma2 = 0.0, noise = 0.0, likelihood = 0.0;
    std::vector<double> w_vec;
};

namespace {

int sgp_eval(egx_sgp *g, const double *theta, int64_t theta_len, double sigma2, double noise, Eval &out, bool keep) {
    const int n = g->n, d = g->d, nz = g->nz, n_pad = g->n_pad, z_pad = g->z_pad, zext = g->zext;
    out = Eval();
    if (theta_len != 1 && theta_len != d) {
        set_error("theta should be either 1-dim or dim of xtrain, got " + std::to_string(theta_len));
        return EGX_ERR_INVALID_VALUE;
    }
    std::vector<double> th(d);
    bool bad = !(sigma2 > 0.0) || !(noise >= 0.0) || !std::isfinite(sigma2) || !std::isfinite(noise);
    for (int k = 0; k < d; k++) {
        th[k] = theta[theta_len == 1 ? 0 : k];
        if (std::isnan(th[k])) bad = true;
    }
    if (bad) {  // sparse_algorithm.rs:521-527: NaN parameters are the worst value
        out.status = EGX_STATUS_NAN_THETA;
        return EGX_SUCCESS;
    }
    hipStream_t s = g->stream;
    const double sigma = std::sqrt(sigma2);
    g->fitted = false;  // the resident factors are about to be overwritten; set again below when `keep`
    EGX_HIP_CHECK(hipMemcpyAsync(g->coef.p, th.data(), sizeof(double) * d, hipMemcpyHostToDevice, s));
    EGX_HIP_CHECK(hipStreamSynchronize(s));  // th is a local
    EGX_HIP_CHECK(hipMemsetAsync(g->d_info, 0, 2 * sizeof(int), s));
    // Kmm / sigma2 = R_zz + (nugget / sigma2) I  -> C_z
    int rc = launch_corr_sym(s, g->corr, g->zT.p, z_pad, nz, d, g->coef.p, 1, g->nugget / sigma2, g->Kz.p, z_pad, z_pad);
    if (rc) return rc;
    rc = launch_potrf(s, g->Kz.p, z_pad, z_pad, z_pad, g->dinv_z.p, g->d_info);
    if (rc) return rc;
    // R_nz, then V~^T = R_nz C_z^-T
    rc = launch_cross_corr(s, g->corr, g->xT.p, n_pad, n_pad, g->zT.p, z_pad, z_pad, d, g->coef.p, 1, g->RT.p, z_pad);
    if (rc) return rc;
    rc = launch_trsm_rows(s, g->Kz.p, z_pad, z_pad, g->dinv_z.p, g->RT.p, z_pad, n_pad);
    if (rc) return rc;
    rc = launch_row_reduce(s, g->RT.p, z_pad, n_pad, nz, nullptr, 0, 0, g->s0.p, nullptr);
    if (rc) return rc;
    hipLaunchKernelGGL(k_sgp_sqrt_beta, dim3((n + 255) / 256), dim3(256), 0, s, g->s0.p, n, sigma2, noise, g->nugget,
                       g->method, g->sb.p);
    hipLaunchKernelGGL(k_sgp_transpose_scale, dim3(n_pad / 32, zext / 32), dim3(256), 0, s, g->RT.p, z_pad, n, nz, g->sb.p,
                       g->y.p, sigma, g->W.p, n_pad);
    // Gneg = 0 - W W^T (lower tiles): G, V (beta o y) in row z_pad, y^T diag(beta) y in the corner
    rc = launch_gram_lower(s, g->W.p, n_pad, zext, n_pad, g->G.p, zext, g->P.p);
    if (rc) return rc;
    hipLaunchKernelGGL(k_sgp_form_a, dim3((z_pad + 255) / 256, zext), dim3(256), 0, s, g->G.p, zext, z_pad, g->A.p);
    rc = launch_potrf(s, g->A.p, z_pad, z_pad, zext, g->dinv_a.p, g->d_info + 1);
    if (rc) return rc;
    rc = launch_gather_diag(s, g->A.p, z_pad, z_pad, g->diag.p);
    if (rc) return rc;
    std::vector<double> s0(n), diag(z_pad), brow(z_pad);
    double corner = 0.0;
    int info[2] = {0, 0};
    EGX_HIP_CHECK(hipMemcpyAsync(s0.data(), g->s0.p, sizeof(double) * n, hipMemcpyDeviceToHost, s));
    EGX_HIP_CHECK(hipMemcpyAsync(diag.data(), g->diag.p, sizeof(double) * z_pad, hipMemcpyDeviceToHost, s));
    EGX_HIP_CHECK(hipMemcpyAsync(brow.data(), g->A.p + (size_t)z_pad * z_pad, sizeof(double) * z_pad, hipMemcpyDeviceToHost, s));
    EGX_HIP_CHECK(hipMemcpyAsync(&corner, g->G.p + (size_t)z_pad * zext + z_pad, sizeof(double), hipMemcpyDeviceToHost, s));
    EGX_HIP_CHECK(hipMemcpyAsync(info, g->d_info, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    EGX_HIP_CHECK(hipStreamSynchronize(s));
    if (info[0] != 0 || info[1] != 0) {  // the reference unwraps (panics); the objective wrapper turns it into +inf
        out.status = EGX_STATUS_NOT_POSITIVE_DEFINITE;
        return EGX_SUCCESS;
    }
    double term2 = 0.0, bb = 0.0;
    for (int i = 0; i < nz; i++) {
        term2 += 2.0 * std::log(diag[i]);
        bb += brow[i] * brow[i];
    }
    const double term3 = -corner;  // Gneg corner = -(y^T diag(beta) y)
    double lkh;
    if (g->method == 0) {  // FITC  sparse_algorithm.rs:747-754
        double term1 = 0.0;
        for (int t = 0; t < n; t++) {
            const double nu = sigma2 * (1.0 - s0[t]) + noise;
            if (!(nu > 0.0)) {
                out.status = EGX_STATUS_NOT_POSITIVE_DEFINITE;
                return EGX_SUCCESS;
            }
            term1 += std::log(nu);
        }
        lkh = -0.5 * (term1 + term2 + term3 - bb);
    } else {  // VFE  :813-821
        const double beta = 1.0 / std::fmax(noise, g->nugget);
        double tr = 0.0;
        for (int t = 0; t < n; t++) tr += s0[t];
        lkh = -0.5 * (-(double)n * std::log(beta) + term2 + term3 - bb + (double)n * beta * sigma2 - beta * sigma2 * tr);
    }
    if (std::isnan(lkh)) {
        out.status = EGX_STATUS_NAN_THETA;
        return EGX_SUCCESS;
    }
    out.lkh = lkh;
    if (keep) {
        // vec = U^-T L^-T b = (1 / sigma) C_z^-T (L^-T b): two backward substitutions with the resident factors
        std::vector<double> tmp(z_pad, 0.0);
        for (int i = 0; i < nz; i++) tmp[i] = brow[i];
        EGX_HIP_CHECK(hipMemcpyAsync(g->tmpv.p, tmp.data(), sizeof(double) * z_pad, hipMemcpyHostToDevice, s));
        rc = launch_block_inverse(s, g->A.p, z_pad, z_pad, g->dinv_a.p, g->wall.p);
        if (rc) return rc;
        rc = launch_trsv_t(s, g->A.p, z_pad, z_pad, g->wall.p, g->tmpv.p, g->vec.p, true);  // (a few blocks: launch per block)
        if (rc) return rc;
        EGX_HIP_CHECK(hipMemcpyAsync(g->tmpv.p, g->vec.p, sizeof(double) * z_pad, hipMemcpyDeviceToDevice, s));
        rc = launch_block_inverse(s, g->Kz.p, z_pad, z_pad, g->dinv_z.p, g->wall.p);
        if (rc) return rc;
        rc = launch_trsv_t(s, g->Kz.p, z_pad, z_pad, g->wall.p, g->tmpv.p, g->vec.p, true);
        if (rc) return rc;
        g->w_vec.assign(z_pad, 0.0);
        EGX_HIP_CHECK(hipMemcpyAsync(g->w_vec.data(), g->vec.p, sizeof(double) * z_pad, hipMemcpyDeviceToHost, s));
        EGX_HIP_CHECK(hipStreamSynchronize(s));
        for (int i = 0; i < z_pad; i++) g->w_vec[i] = (i < nz) ? g->w_vec[i] / sigma : 0.0;
        // sigma2 * vec is what the mean kernel contracts with R(x, z): predict = sigma2 R . vec
        std::vector<double> sv(z_pad);
        for (int i = 0; i < z_pad; i++) sv[i] = sigma2 * g->w_vec[i];
        EGX_HIP_CHECK(hipMemcpyAsync(g->vec.p, sv.data(), sizeof(double) * z_pad, hipMemcpyHostToDevice, s));
        EGX_HIP_CHECK(hipStreamSynchronize(s));
        g->theta = th;
        g->sigma2 = sigma2;
        g->noise = noise;
        g->likelihood = lkh;
        g->fitted = true;
    }
    return EGX_SUCCESS;
}

int sgp_predict(egx_sgp *g, const double *xq, int64_t m, double *yout, double *vout) {
    if (!g->fitted) {
        set_error("sparse model is not fitted (call egx_sgp_finalize or egx_sgp_fit first)");
        return EGX_ERR_NOT_FITTED;
    }
    if (m < 0 || (m > 0 && !xq)) {
        set_error("bad query array");
        return EGX_ERR_INVALID_VALUE;
    }
    const int d = g->d, nz = g->nz, z_pad = g->z_pad;
    hipStream_t s = g->stream;
    const int64_t cap = 65536;
    for (int64_t m0 = 0; m0 < m; m0 += cap) {
        const int mc = (int)((m - m0 < cap) ? (m - m0) : cap);
        const int m_pad = (int)round_up(mc, kTile);
        std::vector<double> xt((size_t)d * m_pad, 0.0);
        for (int a = 0; a < mc; a++)
            for (int k = 0; k < d; k++) xt[(size_t)k * m_pad + a] = xq[(size_t)(m0 + a) * d + k];
        Dev dq, dr, dRT, dp, dq2;
        int rc = dq.alloc(xt.size());
        if (rc) return rc;
        EGX_HIP_CHECK(hipMemcpyAsync(dq.p, xt.data(), sizeof(double) * xt.size(), hipMemcpyHostToDevice, s));
        EGX_HIP_CHECK(hipStreamSynchronize(s));
        if (yout) {  // Kx . vec  (:237-241), R never materialised
            std::vector<double> r(m_pad);
            rc = dr.alloc(m_pad);
            if (rc) return rc;
            rc = launch_predict_mean(s, g->corr, dq.p, m_pad, m_pad, g->zT.p, z_pad, z_pad, d, g->coef.p, 1, g->vec.p, dr.p);
            if (rc) return rc;
            EGX_HIP_CHECK(hipMemcpyAsync(r.data(), dr.p, sizeof(double) * m_pad, hipMemcpyDeviceToHost, s));
            EGX_HIP_CHECK(hipStreamSynchronize(s));
            for (int a = 0; a < mc; a++) yout[m0 + a] = r[a];
        }
        if (vout) {  // sigma2 - kx^T inv kx, clamped, + noise  (:245-257)
            std::vector<double> p2(m_pad), q2(m_pad);
            rc = dRT.alloc((size_t)m_pad * z_pad);
            if (rc) return rc;
            rc = dp.alloc(m_pad);
            if (rc) return rc;
            rc = dq2.alloc(m_pad);
            if (rc) return rc;
            rc = launch_cross_corr(s, g->corr, dq.p, m_pad, m_pad, g->zT.p, z_pad, z_pad, d, g->coef.p, 1, dRT.p, z_pad);
            if (rc) return rc;
            rc = launch_trsm_rows(s, g->Kz.p, z_pad, z_pad, g->dinv_z.p, dRT.p, z_pad, m_pad);  // C_z^-1 r
            if (rc) return rc;
            rc = launch_row_reduce(s, dRT.p, z_pad, m_pad, nz, nullptr, 0, 0, dp.p, nullptr);
            if (rc) return rc;
            rc = launch_trsm_rows(s, g->A.p, z_pad, z_pad, g->dinv_a.p, dRT.p, z_pad, m_pad);   // L^-1 C_z^-1 r
            if (rc) return rc;
            rc = launch_row_reduce(s, dRT.p, z_pad, m_pad, nz, nullptr, 0, 0, dq2.p, nullptr);
            if (rc) return rc;
            EGX_HIP_CHECK(hipMemcpyAsync(p2.data(), dp.p, sizeof(double) * m_pad, hipMemcpyDeviceToHost, s));
            EGX_HIP_CHECK(hipMemcpyAsync(q2.data(), dq2.p, sizeof(double) * m_pad, hipMemcpyDeviceToHost, s));
            EGX_HIP_CHECK(hipStreamSynchronize(s));
            // p = U^-1 kx = sigma C_z^-1 r  ->  |p|^2 = sigma2 |C_z^-1 r|^2, likewise q
            for (int a = 0; a < mc; a++) {
                const double quad = (g->method == 0) ? g->sigma2 * (p2[a] - q2[a]) : g->sigma2 * (p2[a] + q2[a]);
                double var = g->sigma2 - quad;
                if (var < 1e-15) var = 1e-15;
                vout[m0 + a] = var + g->noise;
            }
        }
    }
    return EGX_SUCCESS;
}

}  // namespace

extern "C" {

void egx_sgp_config_default(egx_sgp_config *cfg) {
    if (!cfg) return;
    cfg->corr = EGX_CORR_SQUARED_EXPONENTIAL;
    cfg->method = EGX_SGP_FITC;
    cfg->nugget = 100.0 * 2.220446049250313e-16;
    cfg->device = -1;
}

void egx_sgp_destroy(egx_sgp *g) {
    if (!g) return;
    (void)hipSetDevice(g->device);
    if (g->d_info) (void)hipFree(g->d_info);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    delete g;
}

int32_t egx_sgp_create(const egx_sgp_config *cfg_in, const double *x, const double *y, int64_t n, int64_t d,
                       const double *z, int64_t nz, egx_sgp **out) {
    if (!out) {
        set_error("out handle pointer is NULL");
        return EGX_ERR_INVALID_VALUE;
    }
    *out = nullptr;
    egx_sgp_config cfg;
    if (cfg_in) cfg = *cfg_in; else egx_sgp_config_default(&cfg);
    if (!x || !y || !z || n < 2 || nz < 1 || nz > n || d < 1) {
        set_error("egx_sgp_create: need x (n x d), y (n), z (nz x d) with 1 <= nz <= n, n >= 2, d >= 1");
        return EGX_ERR_INVALID_VALUE;
    }
    if (cfg.corr < 0 || cfg.corr > 3 || (cfg.method != EGX_SGP_FITC && cfg.method != EGX_SGP_VFE) ||
        !(cfg.nugget >= 0.0) || !std::isfinite(cfg.nugget)) {
        set_error("egx_sgp_create: unknown correlation / method, or negative nugget");
        return EGX_ERR_INVALID_VALUE;
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
    for (int64_t i = 0; i < nz * d; i++)
        if (!std::isfinite(z[i])) {
            set_error("z contains non-finite values");
            return EGX_ERR_INVALID_VALUE;
        }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device: libegx_gp_hip has no CPU fallback (needs an MI355X / gfx950 GPU)");
        return EGX_ERR_NO_DEVICE;
    }
    int dev = cfg.device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (dev >= ndev) {
        set_error("device ordinal out of range");
        return EGX_ERR_INVALID_VALUE;
    }
    egx_sgp *g = new egx_sgp();
    g->device = dev;
    g->corr = cfg.corr;
    g->method = cfg.method;
    g->nugget = cfg.nugget;
    g->n = (int)n;
    g->d = (int)d;
    g->nz = (int)nz;
    g->n_pad = (int)round_up(n, kTile);
    g->z_pad = (int)round_up(nz, kTile);
    g->zext = g->z_pad + kTile;
    g->y_host.assign(y, y + n);
    for (int64_t i = 0; i < n; i++) g->yty += y[i] * y[i];
    auto fail = [&](int rc) {
        egx_sgp_destroy(g);
        return rc;
    };
#define SGP_TRY(expr)                       \
    do {                                    \
        int _rc = (expr);                   \
        if (_rc) return fail(_rc);          \
    } while (0)
#define SGP_HIP(expr)                                                             \
    do {                                                                          \
        hipError_t _e = (expr);                                                   \
        if (_e != hipSuccess) {                                                   \
            set_error(std::string(#expr) + ": " + hipGetErrorString(_e));         \
            (void)hipGetLastError();                                              \
            return fail(EGX_ERR_HIP);                                             \
        }                                                                         \
    } while (0)
    SGP_HIP(hipSetDevice(dev));
    SGP_TRY(chol_init());
    SGP_HIP(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
    const size_t np = g->n_pad, zp = g->z_pad, ze = g->zext;
    std::vector<double> xT((size_t)d * np, 0.0), zT((size_t)d * zp, 0.0), yp(np, 0.0);
    for (int64_t i = 0; i < n; i++)
        for (int64_t k = 0; k < d; k++) xT[(size_t)k * np + i] = x[i * d + k];
    for (int64_t i = 0; i < nz; i++)
        for (int64_t k = 0; k < d; k++) zT[(size_t)k * zp + i] = z[i * d + k];
    for (int64_t i = 0; i < n; i++) yp[i] = y[i];
    SGP_TRY(g->xT.alloc(xT.size()));
    SGP_TRY(g->zT.alloc(zT.size()));
    SGP_TRY(g->y.alloc(np));
    SGP_TRY(g->coef.alloc((size_t)d));
    SGP_TRY(g->RT.alloc(np * zp));
    SGP_TRY(g->W.alloc(ze * np));
    SGP_TRY(g->G.alloc(ze * ze));
    SGP_TRY(g->P.alloc(gram_scratch_doubles((int)ze, (int)np)));
    SGP_TRY(g->Kz.alloc(zp * zp));
    SGP_TRY(g->A.alloc(ze * zp));
    SGP_TRY(g->dinv_z.alloc(dinv_doubles(zp)));
    SGP_TRY(g->dinv_a.alloc(dinv_doubles(zp)));
    SGP_TRY(g->s0.alloc(np));
    SGP_TRY(g->sb.alloc(np));
    SGP_TRY(g->diag.alloc(zp));
    SGP_TRY(g->brow.alloc(zp));
    SGP_TRY(g->vec.alloc(zp));
    SGP_TRY(g->tmpv.alloc(zp));
    SGP_TRY(g->wall.alloc(block_inverse_doubles(zp)));
    SGP_HIP(dev_malloc(&g->d_info, 2 * sizeof(int)));
    SGP_HIP(hipMemcpy(g->xT.p, xT.data(), sizeof(double) * xT.size(), hipMemcpyHostToDevice));
    SGP_HIP(hipMemcpy(g->zT.p, zT.data(), sizeof(double) * zT.size(), hipMemcpyHostToDevice));
    SGP_HIP(hipMemcpy(g->y.p, yp.data(), sizeof(double) * np, hipMemcpyHostToDevice));
#undef SGP_TRY
#undef SGP_HIP
    *out = g;
    return EGX_SUCCESS;
}

int32_t egx_sgp_dims(const egx_sgp *g, int64_t *n, int64_t *d, int64_t *nz) {
    if (!g) {
        set_error("NULL handle");
        return EGX_ERR_INVALID_VALUE;
    }
    if (n) *n = g->n;
    if (d) *d = g->d;
    if (nz) *nz = g->nz;
    return EGX_SUCCESS;
}

int32_t egx_sgp_likelihood(egx_sgp *g, const double *theta, int64_t theta_len, double sigma2, double noise, double *lkh,
                           int32_t *status) {
    if (!g || !theta || !lkh || !status) {
        set_error("NULL argument");
        return EGX_ERR_INVALID_VALUE;
    }
    std::lock_guard<std::mutex> lock(g->mu);
    EGX_HIP_CHECK(hipSetDevice(g->device));
    Eval ev;
    const int rc = sgp_eval(g, theta, theta_len, sigma2, noise, ev, false);
    if (rc) return rc;
    *lkh = ev.lkh;
    *status = ev.status;
    return EGX_SUCCESS;
}

int32_t egx_sgp_finalize(egx_sgp *g, const double *theta, int64_t theta_len, double sigma2, double noise) {
    if (!g || !theta) {
        set_error("NULL argument");
        return EGX_ERR_INVALID_VALUE;
    }
    std::lock_guard<std::mutex> lock(g->mu);
    EGX_HIP_CHECK(hipSetDevice(g->device));
    Eval ev;
    const int rc = sgp_eval(g, theta, theta_len, sigma2, noise, ev, true);
    if (rc) return rc;
    if (ev.status != EGX_STATUS_OK) {
        set_error(ev.status == EGX_STATUS_NAN_THETA ? "sparse GP: NaN / non-positive parameters"
                                                    : "sparse GP: Kmm or I + V diag(beta) V^T is not positive definite");
        return ev.status == EGX_STATUS_NAN_THETA ? EGX_ERR_LIKELIHOOD : EGX_ERR_LINALG;
    }
    return EGX_SUCCESS;
}

// Multistart derivative-free fit over x = log10([theta_1..theta_d, sigma2, (noise)]) (sparse_algorithm.rs:488-650):
// params0s is (n_starts x np) in parameter space, lo/hi (np) its box, np = d + 1 + estimate_noise.
int32_t egx_sgp_fit(egx_sgp *g, const double *params0s, int64_t n_starts, const double *lo, const double *hi,
                    int32_t estimate_noise, double noise_fixed, int64_t max_eval, int64_t *n_evals_out) {
    if (!g || !params0s || !lo || !hi || n_starts < 1) {
        set_error("NULL argument / no start point");
        return EGX_ERR_INVALID_VALUE;
    }
    std::lock_guard<std::mutex> lock(g->mu);
    EGX_HIP_CHECK(hipSetDevice(g->device));
    const int d = g->d, np = d + 1 + (estimate_noise ? 1 : 0);
    std::vector<double> blo(np), bhi(np);
    for (int i = 0; i < np; i++) {
        if (!(lo[i] > 0.0) || !(hi[i] >= lo[i])) {
            set_error("sparse GP parameter bounds must satisfy 0 < lo <= hi");
            return EGX_ERR_INVALID_VALUE;
        }
        blo[i] = std::log10(lo[i]);
        bhi[i] = std::log10(hi[i]);
    }
    for (int64_t i = 0; i < n_starts * np; i++)
        if (!(params0s[i] > 0.0)) {
            set_error("sparse GP start points must be > 0");
            return EGX_ERR_INVALID_VALUE;
        }
    // maxeval = clamp(10 * theta_dim, 25, max_eval)  sparse_algorithm.rs:601-603
    int64_t per_start = 10 * (int64_t)d;
    if (per_start < 25) per_start = 25;
    if (max_eval >= 25 && per_start > max_eval) per_start = max_eval;
    int first_rc = EGX_SUCCESS;
    auto objective = [&](const std::vector<double> &xv) -> double {
        std::vector<double> th(d);
        for (int i = 0; i < d; i++) th[i] = std::pow(10.0, xv[i]);
        const double s2 = std::pow(10.0, xv[d]);
        const double nv = estimate_noise ? std::pow(10.0, xv[d + 1]) : noise_fixed;
        Eval ev;
        const int rc = sgp_eval(g, th.data(), d, s2, nv, ev, false);
        if (rc) {
            if (!first_rc) first_rc = rc;
            return std::numeric_limits<double>::infinity();
        }
        if (ev.status != EGX_STATUS_OK) return std::numeric_limits<double>::infinity();
        return -ev.lkh;
    };
    double best_f = std::numeric_limits<double>::infinity();
    std::vector<double> best_x;
    int64_t evals = 0;
    for (int64_t st = 0; st < n_starts; st++) {
        std::vector<double> x0(np);
        for (int i = 0; i < np; i++) x0[i] = std::log10(params0s[st * np + i]);
        // COBYLA as the reference configures it for the sparse model too (sparse_algorithm.rs:597-611 ->
        // optimization.rs:122-169)
        StartResult r;
        {
            CobylaBox m(x0, blo, bhi, 0.5, 1e-4, per_start);
            std::vector<double> xv;
            while (m.ask(xv)) m.tell(objective(xv));
            double fb = m.best_f();
            if (std::isnan(fb) || fb >= 1e30) fb = std::numeric_limits<double>::infinity();
            r = StartResult{fb, m.best_x(), m.evals()};
        }
        evals += r.evals;
        if (r.f < best_f) {
            best_f = r.f;
            best_x = r.x;
        }
    }
    if (first_rc) return first_rc;
    if (n_evals_out) *n_evals_out = evals;
    if (!std::isfinite(best_f)) {
        set_error("sparse GP: no start point gave a finite likelihood");
        return EGX_ERR_LIKELIHOOD;
    }
    std::vector<double> th(d);
    for (int i = 0; i < d; i++) th[i] = std::pow(10.0, best_x[i]);
    Eval ev;
    const int rc = sgp_eval(g, th.data(), d, std::pow(10.0, best_x[d]),
                            estimate_noise ? std::pow(10.0, best_x[d + 1]) : noise_fixed, ev, true);
    if (rc) return rc;
    if (ev.status != EGX_STATUS_OK) {
        set_error("sparse GP: final evaluation failed");
        return EGX_ERR_LINALG;
    }
    return EGX_SUCCESS;
}

int32_t egx_sgp_predict(egx_sgp *g, const double *xq, int64_t m, double *y) {
    if (!g || (m > 0 && !y)) {
        set_error("NULL argument");
        return EGX_ERR_INVALID_VALUE;
    }
    std::lock_guard<std::mutex> lock(g->mu);
    EGX_HIP_CHECK(hipSetDevice(g->device));
    return sgp_predict(g, xq, m, y, nullptr);
}

int32_t egx_sgp_predict_var(egx_sgp *g, const double *xq, int64_t m, double *var) {
    if (!g || (m > 0 && !var)) {
        set_error("NULL argument");
        return EGX_ERR_INVALID_VALUE;
    }
    std::lock_guard<std::mutex> lock(g->mu);
    EGX_HIP_CHECK(hipSetDevice(g->device));
    return sgp_predict(g, xq, m, nullptr, var);
}

// Fitted state: theta (d), sigma2, noise, likelihood, Woodbury vector (nz) and -- computed on the host from the two
// downloaded nz x nz factors, for serialisation only -- the Woodbury matrix (nz x nz, sparse_algorithm.rs:757-763, 823-829).
int32_t egx_sgp_get_state(egx_sgp *g, double *theta, double *sigma2, double *noise, double *likelihood, double *w_vec,
                          double *w_inv) {
    if (!g) {
        set_error("NULL handle");
        return EGX_ERR_INVALID_VALUE;
    }
    std::lock_guard<std::mutex> lock(g->mu);
    if (!g->fitted) {
        set_error("sparse model is not fitted");
        return EGX_ERR_NOT_FITTED;
    }
    EGX_HIP_CHECK(hipSetDevice(g->device));
    const int nz = g->nz, z_pad = g->z_pad;
    if (theta) std::memcpy(theta, g->theta.data(), sizeof(double) * g->d);
    if (sigma2) *sigma2 = g->sigma2;
    if (noise) *noise = g->noise;
    if (likelihood) *likelihood = g->likelihood;
    if (w_vec) std::memcpy(w_vec, g->w_vec.data(), sizeof(double) * nz);
    if (w_inv) {
        std::vector<double> cz((size_t)z_pad * z_pad), la((size_t)z_pad * z_pad);
        EGX_HIP_CHECK(hipMemcpy(cz.data(), g->Kz.p, sizeof(double) * cz.size(), hipMemcpyDeviceToHost));
        EGX_HIP_CHECK(hipMemcpy(la.data(), g->A.p, sizeof(double) * la.size(), hipMemcpyDeviceToHost));
        const double sigma = std::sqrt(g->sigma2);
        auto tri_inv = [&](const std::vector<double> &Lm, double scale, std::vector<double> &X) {  // X = (scale L)^-1
            X.assign((size_t)nz * nz, 0.0);
            for (int c = 0; c < nz; c++)
                for (int i = c; i < nz; i++) {
                    double sacc = (i == c) ? 1.0 : 0.0;
                    for (int k = c; k < i; k++) sacc -= Lm[(size_t)i * z_pad + k] * scale * X[(size_t)k * nz + c];
                    X[(size_t)i * nz + c] = sacc / (Lm[(size_t)i * z_pad + i] * scale);
                }
        };
        std::vector<double> ui, li, liui((size_t)nz * nz, 0.0);
        tri_inv(cz, sigma, ui);
        tri_inv(la, 1.0, li);
        for (int i = 0; i < nz; i++)
            for (int k = 0; k <= i; k++) {
                const double lik = li[(size_t)i * nz + k];
                if (lik == 0.0) continue;
                for (int j = 0; j <= k; j++) liui[(size_t)i * nz + j] += lik * ui[(size_t)k * nz + j];
            }
        for (int i = 0; i < nz; i++)
            for (int j = 0; j < nz; j++) {
                double uu = 0.0, ll = 0.0;
                for (int k = (i > j ? i : j); k < nz; k++) {
                    uu += ui[(size_t)k * nz + i] * ui[(size_t)k * nz + j];
                    ll += liui[(size_t)k * nz + i] * liui[(size_t)k * nz + j];
                }
                w_inv[(size_t)i * nz + j] = (g->method == 0) ? uu - ll : uu + ll;
            }
    }
    return EGX_SUCCESS;
}

}  // extern "C"
