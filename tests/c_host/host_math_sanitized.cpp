// Host-side pieces of the library (egobox_amd/csrc/host_math.h) under AddressSanitizer + UBSan
// (the reference relies on Rust for memory safety; here the host C++ gets sanitizer coverage on the CPU).
// Checks values against independent formulas; exit code = number of failed checks.
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "../../egobox_amd/csrc/host_math.h"

using namespace egx;
static int failures = 0;
#define EXPECT(c)                                                                \
    do {                                                                         \
        if (!(c)) {                                                              \
            std::fprintf(stderr, "%s:%d: EXPECT(%s)\n", __FILE__, __LINE__, #c); \
            failures++;                                                          \
        }                                                                        \
    } while (0)

int main() {
    std::mt19937_64 rng(3);
    std::normal_distribution<double> g(0.0, 1.0);
    // normalize: sample std (ddof = 1), zero std -> 1
    {
        const int n = 7, d = 3;
        std::vector<double> x(n * d), xn(n * d), m(d), s(d);
        for (int i = 0; i < n; i++) {
            x[i * d + 0] = g(rng) * 5 + 2;
            x[i * d + 1] = 4.0;
            x[i * d + 2] = i;
        }
        hm::normalize(x.data(), n, d, xn.data(), m.data(), s.data());
        EXPECT(s[1] == 1.0 && m[1] == 4.0 && std::fabs(m[2] - 3.0) < 1e-15 && std::fabs(s[2] - std::sqrt(28.0 / 6.0)) < 1e-14);
        double chk = 0;
        for (int i = 0; i < n; i++) chk += xn[i * d] * xn[i * d];
        EXPECT(std::fabs(chk - (n - 1)) < 1e-12);
    }
    // sum_log10 (mantissa product, one logarithm) against a long double sum of logarithms: Cholesky pivots from 1e-7 to 1,
    // lengths across the 256-factor renormalisation, and pivots near the ends of the exponent range
    {
        std::uniform_real_distribution<double> u(-7.0, 0.0);
        for (int64_t n : {1, 5, 255, 256, 257, 4096, 16384, 50001}) {
            std::vector<double> dgl(n);
            long double ref = 0.0L;
            for (auto &e : dgl) {
                e = std::pow(10.0, u(rng));
                ref += log10l((long double)e);
            }
            const double got = hm::sum_log10(dgl.data(), n);
            EXPECT(std::fabs(got - (double)ref) <= 4e-16 * (double)n + 1e-15 * std::fabs((double)ref));
        }
        const double edge[6] = {1e-300, 1e300, 3e-308, 1.5e308, 1.0, 0.5};
        long double ref = 0.0L;
        for (double e : edge) ref += log10l((long double)e);
        EXPECT(std::fabs(hm::sum_log10(edge, 6) - (double)ref) <= 1e-12);
    }
    // regression basis and its jacobian contraction vs central differences
    for (int mean = 0; mean < 3; mean++) {
        const int d = 4;
        const int64_t p = hm::regression_ncols(mean, d);
        EXPECT(p == (mean == 0 ? 1 : mean == 1 ? 5 : 15));
        std::vector<double> x(d), v(p), f1(p), f2(p), jd(d);
        for (auto &e : x) e = g(rng);
        for (auto &e : v) e = g(rng);
        hm::regression_jac_dot(mean, x.data(), d, v.data(), jd.data());
        for (int k = 0; k < d; k++) {
            std::vector<double> xp(x), xm(x);
            xp[k] += 1e-6;
            xm[k] -= 1e-6;
            hm::regression_row(mean, xp.data(), d, f1.data());
            hm::regression_row(mean, xm.data(), d, f2.data());
            double fd = 0;
            for (int64_t j = 0; j < p; j++) fd += v[j] * (f1[j] - f2[j]) / 2e-6;
            EXPECT(std::fabs(fd - jd[k]) < 1e-7);
        }
    }
    // Householder QR with a positive diagonal: R^T R = A^T A, Q^T b consistent with least squares
    {
        const int64_t n = 30, p = 4;
        std::vector<double> a(n * p), a0, b(n), b0;
        for (auto &e : a) e = g(rng);
        for (auto &e : b) e = g(rng);
        a0 = a;
        b0 = b;
        hm::qr_apply(a, n, p, &b);
        for (int64_t i = 0; i < p; i++) {
            EXPECT(a[i * n + i] > 0.0);
            for (int64_t j = i; j < p; j++) {
                double rtr = 0, ata = 0;
                for (int64_t k = 0; k <= i; k++) rtr += a[i * n + k] * a[j * n + k];
                for (int64_t k = 0; k < n; k++) ata += a0[i * n + k] * a0[j * n + k];
                EXPECT(std::fabs(rtr - ata) < 1e-10);
            }
        }
        // singular values of R: product = |det R|, max >= min > 0
        std::vector<double> r(p * p, 0.0);
        double det = 1.0;
        for (int64_t i = 0; i < p; i++) {
            det *= a[i * n + i];
            for (int64_t j = i; j < p; j++) r[i * p + j] = a[j * n + i];
        }
        auto sv = hm::singular_values(r, p);
        double prod = 1.0;
        for (double s : sv) prod *= s;
        EXPECT(sv.size() == (size_t)p && sv.front() >= sv.back() && sv.back() > 0.0 && std::fabs(prod - std::fabs(det)) < 1e-9 * prod);
    }
    std::printf("%s (%d)\n", failures ? "FAILED" : "OK", failures);
    return failures;
}
