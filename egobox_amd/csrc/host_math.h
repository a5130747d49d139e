// Small dense host-side linear algebra for the p x p / n x p pieces of the generalised least
// squares step of reduced_likelihood (crates/gp/src/algorithm.rs:1006-1034): p is the number of
// regression basis columns (1 for ordinary kriging), so this is O(n p^2) next to the O(n^3)
// factorisation that runs on the GPU.  The reference takes these from linfa-linalg 0.2.1
// (`qr().into_decomp()`, `svd(false,false)`, `solve_triangular`).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace egx {
namespace hm {

// utils.rs:45-54.  Row-major passes with one accumulator per column: every column sees exactly the additions of the
// column-by-column form, in the same order (the same bits), without its d strided sweeps over the (n, d) array.
inline void normalize(const double *x, int64_t n, int64_t d, double *xn, double *mean, double *sd) {
    std::vector<double> acc((size_t)d, 0.0);
    for (int64_t i = 0; i < n; i++) {
        const double *row = x + i * d;
        for (int64_t j = 0; j < d; j++) acc[j] += row[j];
    }
    for (int64_t j = 0; j < d; j++) mean[j] = acc[j] / (double)n;
    std::fill(acc.begin(), acc.end(), 0.0);
    for (int64_t i = 0; i < n; i++) {
        const double *row = x + i * d;
        for (int64_t j = 0; j < d; j++) {
            const double t = row[j] - mean[j];
            acc[j] += t * t;
        }
    }
    for (int64_t j = 0; j < d; j++) {
        double sdev = std::sqrt(acc[j] / (double)(n - 1));
        if (sdev == 0.0) sdev = 1.0;
        sd[j] = sdev;
    }
    for (int64_t i = 0; i < n; i++) {
        const double *row = x + i * d;
        double *out = xn + i * d;
        for (int64_t j = 0; j < d; j++) out[j] = (row[j] - mean[j]) / sd[j];
    }
}

// mean_models.rs:42-44, 68-71, 97-104
inline int64_t regression_ncols(int mean, int64_t d) {
    switch (mean) {
        case 0: return 1;
        case 1: return 1 + d;
        case 2: return 1 + d + d * (d + 1) / 2;
        default: return -1;
    }
}
inline void regression_row(int mean, const double *x, int64_t d, double *f) {
    int64_t c = 0;
    f[c++] = 1.0;
    if (mean >= 1)
        for (int64_t j = 0; j < d; j++) f[c++] = x[j];
    if (mean >= 2)
        for (int64_t k = 0; k < d; k++)
            for (int64_t j = k; j < d; j++) f[c++] = x[j] * x[k];
}

// sum_i log10(d_i) for positive finite d (the log-determinant term of algorithm.rs:1039-1041) without n calls of log10:
// the mantissas are multiplied (renormalised every 256 factors: 256 numbers in [0.5, 1) cannot underflow), the exponents added;
// ONE logarithm at the end.  The product carries a relative error of n eps / 2, i.e. n eps / (2 ln 10) ABSOLUTE in the sum
// (1e-12 at n = 16384 on a sum of order 1e4) -- the same order as the rounding of n separate logarithms, at a twelfth of the
// time (the n = 16384 host half of an evaluation was 0.3 ms of log10 calls).
inline double sum_log10(const double *d, int64_t n) {
    double mant = 1.0;
    int64_t ex = 0;
    for (int64_t i0 = 0; i0 < n; i0 += 256) {
        const int64_t i1 = (i0 + 256 < n) ? i0 + 256 : n;
        for (int64_t i = i0; i < i1; i++) {
            int e;
            mant *= std::frexp(d[i], &e);
            ex += e;
        }
        int e;
        mant = std::frexp(mant, &e);
        ex += e;
    }
    return std::log10(mant) + (double)ex * 0.30102999566398119521;  // log10(2)
}

// out[k] = sum_j v[j] * d f_j(x) / d x_k : RegressionModel::jacobian (mean_models.rs:50-52, 76-81, 110-128)
// contracted with a coefficient vector v (beta for the mean gradient, B^-1 A^T for the variance gradient).
inline void regression_jac_dot(int mean, const double *x, int64_t d, const double *v, double *out) {
    for (int64_t k = 0; k < d; k++) out[k] = 0.0;
    if (mean >= 1)
        for (int64_t k = 0; k < d; k++) out[k] = v[1 + k];
    if (mean >= 2) {
        int64_t c = 1 + d;
        for (int64_t k = 0; k < d; k++)
            for (int64_t j = k; j < d; j++, c++) {  // column x_j * x_k
                out[k] += v[c] * x[j];
                out[j] += v[c] * x[k];
            }
    }
}

// Householder QR of a (n x p, column-major: a[l*n + i]) in place; b (length n) receives Q^T b.
// On return the upper p x p of `a` (a[l*n + i], i <= l) is R.  Rows of R are then sign-flipped so
// that diag(R) > 0 (the convention of the reference's serialized models).
inline void qr_apply(std::vector<double> &a, int64_t n, int64_t p, std::vector<double> *b) {
    std::vector<double> v(n);
    for (int64_t k = 0; k < p && k < n; k++) {
        double *ck = &a[k * n];
        double nrm = 0.0;
        for (int64_t i = k; i < n; i++) nrm += ck[i] * ck[i];
        nrm = std::sqrt(nrm);
        if (nrm == 0.0) continue;
        const double alpha = (ck[k] > 0.0) ? -nrm : nrm;
        for (int64_t i = k; i < n; i++) v[i] = ck[i];
        v[k] -= alpha;
        double vnorm2 = 0.0;
        for (int64_t i = k; i < n; i++) vnorm2 += v[i] * v[i];
        if (vnorm2 == 0.0) continue;
        for (int64_t l = k; l < p; l++) {
            double *cl = &a[l * n];
            double dot = 0.0;
            for (int64_t i = k; i < n; i++) dot += v[i] * cl[i];
            const double f = 2.0 * dot / vnorm2;
            for (int64_t i = k; i < n; i++) cl[i] -= f * v[i];
        }
        if (b) {
            double dot = 0.0;
            for (int64_t i = k; i < n; i++) dot += v[i] * (*b)[i];
            const double f = 2.0 * dot / vnorm2;
            for (int64_t i = k; i < n; i++) (*b)[i] -= f * v[i];
        }
    }
    for (int64_t k = 0; k < p && k < n; k++) {
        if (a[k * n + k] < 0.0) {
            for (int64_t l = k; l < p; l++) a[l * n + k] = -a[l * n + k];
            if (b) (*b)[k] = -(*b)[k];
        }
    }
}

// singular values (descending) of a p x p matrix r (row-major) by one-sided Jacobi
inline std::vector<double> singular_values(const std::vector<double> &r, int64_t p) {
    std::vector<double> u(r);  // columns orthogonalised in place; u[i*p + j]
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0;
        for (int64_t a = 0; a < p - 1; a++)
            for (int64_t b = a + 1; b < p; b++) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int64_t i = 0; i < p; i++) {
                    alpha += u[i * p + a] * u[i * p + a];
                    beta += u[i * p + b] * u[i * p + b];
                    gamma += u[i * p + a] * u[i * p + b];
                }
                if (gamma == 0.0) continue;
                const double lim = std::sqrt(alpha * beta);
                if (lim > 0) off = std::fmax(off, std::fabs(gamma) / lim);
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = ((zeta >= 0) ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
                for (int64_t i = 0; i < p; i++) {
                    const double ua = u[i * p + a], ub = u[i * p + b];
                    u[i * p + a] = c * ua - s * ub;
                    u[i * p + b] = s * ua + c * ub;
                }
            }
        if (off < 1e-15) break;
    }
    std::vector<double> sv(p);
    for (int64_t j = 0; j < p; j++) {
        double s = 0;
        for (int64_t i = 0; i < p; i++) s += u[i * p + j] * u[i * p + j];
        sv[j] = std::sqrt(s);
    }
    for (int64_t i = 0; i < p; i++)
        for (int64_t j = i + 1; j < p; j++)
            if (sv[j] > sv[i]) std::swap(sv[i], sv[j]);
    return sv;
}

}  // namespace hm
}  // namespace egx
