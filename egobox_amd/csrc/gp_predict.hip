// Predictions of a fitted dense GP and their x-gradients (GaussianProcess::predict / predict_var / predict_valvar,
// crates/gp/src/algorithm.rs:253-380; predict_gradients / predict_var_gradients :510-727): batched paths on the
// factorisation's own kernels, and few-query paths for EGO's one-point-at-a-time calls.
#include "gp_handle.h"

using namespace egx;

namespace egx {


// ---- prediction -------------------------------------------------------------------------------
// Upload a chunk of query points as the caller holds it and normalise it (algorithm.rs:254) into the k-major layout
// (d x m_pad, zero padded) ON THE DEVICE: the host loop this replaces (a division and a strided store per coordinate) cost
// four times the prediction kernel for 100 000 points.  xn (the normalised rows on the host) is filled only when the trend
// needs the coordinates (Linear / Quadratic).  xq outlives the call, so nothing waits for the copy here.
static int upload_queries(egx_gp *gp, const double *xq, int64_t m0, int m, int m_pad, std::vector<double> &xn,
                          DevBuf &d_xraw, DevBuf &d_xqT, hipStream_t s) {
    const int d = gp->d;
    EGX_RC(d_xraw.alloc((size_t)m * d));
    EGX_RC(d_xqT.alloc((size_t)d * m_pad));
    EGX_HIP_CHECK(hipMemcpyAsync(d_xraw.p, xq + (size_t)m0 * d, sizeof(double) * (size_t)m * d, hipMemcpyHostToDevice, s));
    EGX_RC(launch_normalize_queries(s, d_xraw.p, m, d, dev_xnorm(gp), d_xqT.p, m_pad, m_pad));
    if (gp->mean >= 1) {
        xn.resize((size_t)m * d);
        const double *src = xq + (size_t)m0 * d;
        for (int a = 0; a < m; a++)
            for (int j = 0; j < d; j++) xn[(size_t)a * d + j] = (src[(size_t)a * d + j] - gp->x_mean[j]) / gp->x_std[j];
    } else {
        xn.clear();
    }
    return EGX_SUCCESS;
}

static int predict_var_small(egx_gp *gp, const double *xq, int64_t m, double *vout);
static int ensure_winv(egx_gp *gp);
static int small_path_buffers(egx_gp *gp);

int predict_impl(egx_gp *gp, const double *xq, int64_t m, double *yout, double *vout) {
    if (!gp->fitted) {
        set_error("model is not fitted (call egx_gp_finalize or egx_gp_fit first)");
        return EGX_ERR_NOT_FITTED;
    }
    if (m < 0 || (m > 0 && !xq)) {
        set_error("bad query array");
        return EGX_ERR_INVALID_VALUE;
    }
    EGX_RC(set_device(gp));
    if (vout && m > 0 && m <= 8 && gp->winv_fail_epoch != gp->fit_epoch) {
        // a few points at a time: EGO's inner loop (its criteria ask for value AND variance).  The cached C^-T behind
        // this path is an OPTIMISATION (a second n_pad^2 matrix): when it cannot be had -- allocation failure, or less
        // than its size + 25 % free on the device -- the batched path below serves the call, for this fit and all later
        // small calls on it.
        const bool have_w = gp->winv_epoch == gp->fit_epoch && gp->d_W && gp->d_neg_invkf;
        if (have_w || ++gp->small_var_calls >= 3) {
            int rc = EGX_SUCCESS;
            if (!have_w && !gp->d_W) {
                size_t fr = 0, tot = 0;
                const size_t need = sizeof(double) * (size_t)gp->n_pad * gp->n_pad;
                if (hipMemGetInfo(&fr, &tot) != hipSuccess || fr < need + need / 4) rc = EGX_ERR_HIP;
            }
            if (rc == EGX_SUCCESS) rc = ensure_winv(gp);
            if (rc == EGX_SUCCESS) rc = small_path_buffers(gp);
            if (rc == EGX_SUCCESS) {
                if (yout) EGX_RC(predict_impl(gp, xq, m, yout, nullptr));  // split-range mean kernel
                return predict_var_small(gp, xq, m, vout);
            }
            (void)hipGetLastError();
            if (gp->winv_epoch != gp->fit_epoch) {  // give a half-built cache back
                if (gp->d_W) (void)hipFree(gp->d_W);
                if (gp->d_neg_invkf) (void)hipFree(gp->d_neg_invkf);
                gp->d_W = gp->d_neg_invkf = nullptr;
            }
            gp->winv_fail_epoch = gp->fit_epoch;
        }
    }
    Workspace &w = gp->ws[0];
    const int n = gp->n, n_pad = gp->n_pad, d = gp->d, p = gp->p;
    // chunk so that the (m_tile x n_pad) block of predict_var stays <= 1 GiB
    int64_t cap = ((int64_t)1 << 27) / n_pad / kTile * kTile;
    if (cap < kTile) cap = kTile;
    if (cap > 16384) cap = 16384;
    if (!vout) cap = 65536;
    // Two chunks in flight on two streams: while the GPU works on chunk i the host post-processes chunk i - 1 and
    // normalises / uploads chunk i + 1, and the tail of one chunk's solve overlaps the head of the next (the (m x n)
    // block of a chunk is bounded at 1 GiB, so 100 000 query points at n = 8192 are 7 chunks).
    struct Slot {
        DevBuf d_xraw, d_xqT, d_racc, d_RT, d_s0, d_sl;  // sized by the first (largest) chunk, reused by the later ones
        std::vector<double> xn, racc, s0, sl;
        int64_t m0 = 0;
        int mc = 0, m_pad = 0, msplit = 1;
        hipStream_t stream = nullptr;
    } slots[2];
    slots[0].stream = w.stream;
    slots[1].stream = w.lk.s2 ? w.lk.s2 : w.stream;
    auto enqueue = [&](Slot &sl_, int64_t m0) -> int {
        const int mc = (int)((m - m0 < cap) ? (m - m0) : cap);
        const int m_pad = (int)round_up(mc, kTile);
        sl_.m0 = m0;
        sl_.mc = mc;
        sl_.m_pad = m_pad;
        hipStream_t st = sl_.stream;
        EGX_RC(upload_queries(gp, xq, m0, mc, m_pad, sl_.xn, sl_.d_xraw, sl_.d_xqT, st));
        // few queries: split the training range so that ~1024 workgroups exist (partial sums added below)
        int msplit = 1;
        if (m_pad / 64 < 1024) msplit = (1024 + m_pad / 64 - 1) / (m_pad / 64);
        if (msplit > n_pad / 64) msplit = n_pad / 64;
        {
            const int per = (n_pad / 64 + msplit - 1) / msplit;
            msplit = (n_pad / 64 + per - 1) / per;
        }
        sl_.msplit = msplit;
        if (yout) {
            EGX_RC(sl_.d_racc.alloc((size_t)msplit * m_pad));
            EGX_RC(launch_predict_mean(st, gp->corr, sl_.d_xqT.p, m_pad, m_pad, gp->d_xT, n_pad, n_pad, d, gp->d_fit_coef,
                                       gp->fit_hcols, gp->d_gamma, sl_.d_racc.p, msplit,
                                       gp->fit_hcols == 1 ? dev_xs_fit(gp) : nullptr));
        }
        if (vout) {
            EGX_RC(sl_.d_RT.alloc((size_t)m_pad * n_pad));
            EGX_RC(sl_.d_s0.alloc(m_pad));
            EGX_RC(sl_.d_sl.alloc((size_t)m_pad * p));
            // corr (m x n): algorithm.rs:372-380 ; rt = C^-1 corr^T: :337-350 (held transposed, row per query)
            EGX_RC(launch_cross_corr(st, gp->corr, sl_.d_xqT.p, m_pad, m_pad, gp->d_xT, n_pad, n_pad, d, gp->d_fit_coef,
                                     gp->fit_hcols, sl_.d_RT.p, n_pad));
            EGX_RC(launch_trsm_rows(st, w.M, gp->ld, n_pad, w.dinv, sl_.d_RT.p, n_pad, m_pad));
            // sum rt^2 and ft^T rt (:352): ft^T rows live below the factor in the workspace
            EGX_RC(launch_row_reduce(st, sl_.d_RT.p, n_pad, m_pad, n, w.M + (size_t)n_pad * gp->ld, gp->ld, p, sl_.d_s0.p,
                                     sl_.d_sl.p));
        }
        return EGX_SUCCESS;
    };
    std::vector<double> f(p), rhs(p), u(p);
    auto finish = [&](Slot &sl_) -> int {
        const int mc = sl_.mc, m_pad = sl_.m_pad, msplit = sl_.msplit;
        const int64_t m0 = sl_.m0;
        EGX_HIP_CHECK(hipStreamSynchronize(sl_.stream));
        // (the copies go to pageable memory: issued after the wait, they cost their transfer time only)
        if (yout) {
            sl_.racc.resize((size_t)msplit * m_pad);
            EGX_HIP_CHECK(hipMemcpy(sl_.racc.data(), sl_.d_racc.p, sizeof(double) * (size_t)msplit * m_pad, hipMemcpyDeviceToHost));
        }
        if (vout) {
            sl_.s0.resize(m_pad);
            sl_.sl.resize((size_t)m_pad * p);
            EGX_HIP_CHECK(hipMemcpy(sl_.s0.data(), sl_.d_s0.p, sizeof(double) * m_pad, hipMemcpyDeviceToHost));
            EGX_HIP_CHECK(hipMemcpy(sl_.sl.data(), sl_.d_sl.p, sizeof(double) * (size_t)m_pad * p, hipMemcpyDeviceToHost));
        }
        const std::vector<double> &xn = sl_.xn, &racc = sl_.racc, &s0 = sl_.s0, &sl = sl_.sl;
        for (int a = 0; a < mc; a++) {
            hm::regression_row(gp->mean, xn.empty() ? nullptr : &xn[(size_t)a * d], d, f.data());
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
        return EGX_SUCCESS;
    };
    int cur = 0, rc = EGX_SUCCESS;
    bool pending = false;
    for (int64_t m0 = 0; m0 < m && !rc; m0 += cap) {
        rc = enqueue(slots[cur], m0);
        if (!rc && pending) rc = finish(slots[cur ^ 1]);
        pending = !rc;
        cur ^= 1;
    }
    if (!rc && pending) rc = finish(slots[cur ^ 1]);
    if (rc) {  // nothing may still run on the buffers the slots are about to free
        (void)hipStreamSynchronize(slots[0].stream);
        (void)hipStreamSynchronize(slots[1].stream);
        (void)hipGetLastError();
    }
    return rc;
}

// d_W <- C^-T (upper triangular, rows of the identity through the forward block substitution) and
// d_neg_invkf <- -C^-T [ft | yt] for the factor resident in workspace 0; cached per fitted state.
static int ensure_winv(egx_gp *gp) {
    if (gp->winv_epoch == gp->fit_epoch && gp->d_W && gp->d_neg_invkf) return EGX_SUCCESS;
    Workspace &w = gp->ws[0];
    const int n_pad = gp->n_pad;
    const size_t sq = (size_t)n_pad * n_pad;
    if (!gp->d_W) EGX_HIP_CHECK(dev_malloc(&gp->d_W, sizeof(double) * sq));
    if (!gp->d_neg_invkf) EGX_HIP_CHECK(dev_malloc(&gp->d_neg_invkf, sizeof(double) * (size_t)n_pad * gp->rhs_pad));
    // the rows of the identity, written on the device as far as the solve and the readers of W touch them (everything
    // on and above the diagonal 128-tiles): no host round trip, no 2 GiB memset
    EGX_RC(launch_identity_rows(w.stream, gp->d_W, n_pad, n_pad));
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
    EGX_HIP_CHECK(dev_malloc(&gp->sp_R, sizeof(double) * (size_t)kTile * n_pad));
    EGX_HIP_CHECK(dev_malloc(&gp->sp_P, sizeof(double) * (size_t)32 * n_pad));
    EGX_HIP_CHECK(dev_malloc(&gp->sp_y, sizeof(double) * n_pad));
    EGX_HIP_CHECK(dev_malloc(&gp->sp_z, sizeof(double) * n_pad));
    EGX_HIP_CHECK(dev_malloc(&gp->sp_wt, sizeof(double) * n_pad));
    EGX_HIP_CHECK(dev_malloc(&gp->sp_out, sizeof(double) * (size_t)gp->sp_nsplit * kTile * d));
    EGX_HIP_CHECK(dev_malloc(&gp->sp_xq, sizeof(double) * (size_t)d * kTile));
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
int xgrad_impl(egx_gp *gp, const double *xq, int64_t m, double *gy, double *gv) {
    if (!gp->fitted) {
        set_error("model is not fitted (call egx_gp_finalize or egx_gp_fit first)");
        return EGX_ERR_NOT_FITTED;
    }
    if (m < 0 || (m > 0 && !xq)) {
        set_error("bad query array");
        return EGX_ERR_INVALID_VALUE;
    }
    EGX_RC(set_device(gp));
    // (the few-query kernel keeps 5 + hcols doubles per input dimension in LDS, the batched one 1 + hcols: very wide
    //  inputs take the batched form whatever m is)
    if (m > 0 && m <= 8 && (int64_t)gp->d * (gp->fit_hcols + 5) <= 20480) return xgrad_small(gp, xq, m, gy, gv);
    Workspace &w = gp->ws[0];
    const int n = gp->n, n_pad = gp->n_pad, d = gp->d, p = gp->p, rp = gp->rhs_pad;
    if (gv) EGX_RC(ensure_winv(gp));
    int64_t cap = ((int64_t)1 << 27) / n_pad / kTile * kTile;
    if (cap < kTile) cap = kTile;
    if (cap > 16384) cap = 16384;
    if (!gv) cap = 65536;
    std::vector<double> xn, part, sl, f(p), a_vec(p), u(p), dd(p), dneg, df(d);
    DevBuf d_xraw, d_xqT, d_out, d_RT, d_s0, d_sl, d_Wt, d_D;  // sized by the first (largest) chunk, reused by the others
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
        EGX_RC(upload_queries(gp, xq, m0, mc, m_pad, xn, d_xraw, d_xqT, w.stream));
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
                hm::regression_jac_dot(gp->mean, xn.empty() ? nullptr : &xn[(size_t)a * d], d, gp->beta.data(), df.data());
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
                hm::regression_row(gp->mean, xn.empty() ? nullptr : &xn[(size_t)a * d], d, f.data());
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
                hm::regression_jac_dot(gp->mean, xn.empty() ? nullptr : &xn[(size_t)a * d], d, dd.data(), df.data());
                for (int k = 0; k < d; k++)
                    gv[(size_t)(m0 + a) * d + k] = 2.0 * gp->sigma2 * (df[k] + reduce_out(a, k)) / gp->x_std[k];
            }
        }
    }
    return EGX_SUCCESS;
}
}  // namespace egx

extern "C" {

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
}  // extern "C"
