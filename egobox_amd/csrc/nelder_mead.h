// Host-side derivative-free maximiser shared by the dense and the sparse GP drivers.
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>
#include <utility>
#include <vector>

namespace egx {

// ---- derivative-free maximiser (stand-in for cobyla 0.8.0, optimization.rs:122-169) -------------
// Nelder-Mead on x = log10(theta) inside the box, initial simplex edge rhobeg = 0.5, relative
// tolerance 1e-4 on f, maxeval evaluations.  The reference's optimised theta* is "parity unpinned"
// (it depends on the un-vendored COBYLA trajectory), so only the contract is mirrored: minimise
// -likelihood over log10 theta within bounds, errors count as +inf, return the best point.
struct NmResult {
    double f;
    std::vector<double> x;
    int64_t evals;
};

template <typename F>
inline NmResult nelder_mead(F &&fn, const std::vector<double> &x0, const std::vector<double> &lo,
                            const std::vector<double> &hi, int64_t maxeval) {
    const int h = (int)x0.size();
    auto clip = [&](std::vector<double> &x) {
        for (int i = 0; i < h; i++) x[i] = std::fmin(hi[i], std::fmax(lo[i], x[i]));
    };
    std::vector<std::vector<double>> sx(h + 1, x0);
    std::vector<double> sf(h + 1);
    int64_t evals = 0;
    clip(sx[0]);
    for (int i = 0; i < h; i++) {
        sx[i + 1] = sx[0];
        double step = 0.5;
        if (sx[i + 1][i] + step > hi[i]) step = -step;
        sx[i + 1][i] += step;
        clip(sx[i + 1]);
    }
    for (int i = 0; i <= h && evals < maxeval; i++) {
        sf[i] = fn(sx[i]);
        evals++;
    }
    for (int i = (int)evals; i <= h; i++) sf[i] = std::numeric_limits<double>::infinity();
    std::vector<int> ord(h + 1);
    while (evals < maxeval) {
        for (int i = 0; i <= h; i++) ord[i] = i;
        for (int i = 0; i <= h; i++)
            for (int j = i + 1; j <= h; j++)
                if (sf[ord[j]] < sf[ord[i]]) std::swap(ord[i], ord[j]);
        const int b = ord[0], wv = ord[h], sw = ord[h > 0 ? h - 1 : 0];
        if (std::isfinite(sf[b]) && std::isfinite(sf[wv]) &&
            std::fabs(sf[wv] - sf[b]) <= 1e-4 * std::fabs(sf[b]) + 1e-300)
            break;
        std::vector<double> c(h, 0.0);
        for (int i = 0; i <= h; i++)
            if (i != wv)
                for (int k = 0; k < h; k++) c[k] += sx[i][k] / h;
        auto along = [&](double t) {
            std::vector<double> x(h);
            for (int k = 0; k < h; k++) x[k] = c[k] + t * (sx[wv][k] - c[k]);
            clip(x);
            return x;
        };
        std::vector<double> xr = along(-1.0);
        const double fr = fn(xr);
        evals++;
        if (fr < sf[b]) {
            if (evals < maxeval) {
                std::vector<double> xe = along(-2.0);
                const double fe = fn(xe);
                evals++;
                if (fe < fr) { sx[wv] = xe; sf[wv] = fe; } else { sx[wv] = xr; sf[wv] = fr; }
            } else { sx[wv] = xr; sf[wv] = fr; }
        } else if (fr < sf[sw]) {
            sx[wv] = xr; sf[wv] = fr;
        } else {
            if (evals >= maxeval) {  // budget exhausted: keep the reflection if it improved on the worst vertex
                if (fr < sf[wv]) { sx[wv] = xr; sf[wv] = fr; }
                break;
            }
            std::vector<double> xc = along(fr < sf[wv] ? -0.5 : 0.5);
            const double fc = fn(xc);
            evals++;
            if (fc < std::fmin(fr, sf[wv])) { sx[wv] = xc; sf[wv] = fc; }
            else {
                for (int i = 0; i <= h && evals < maxeval; i++) {
                    if (i == b) continue;
                    for (int k = 0; k < h; k++) sx[i][k] = sx[b][k] + 0.5 * (sx[i][k] - sx[b][k]);
                    sf[i] = fn(sx[i]);
                    evals++;
                }
            }
        }
    }
    int best = 0;
    for (int i = 1; i <= h; i++)
        if (sf[i] < sf[best]) best = i;
    return NmResult{sf[best], sx[best], evals};
}


}  // namespace egx
