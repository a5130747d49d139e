// COBYLA for bound-constrained minimisation, as a resumable (ask / tell) state machine.
//
// The reference tunes theta with the `cobyla` crate 0.8.0 (crates/gp/src/optimization.rs:122-169: `minimize(objfn, x0,
// bounds, cons = [], maxeval, RhoBeg::All(0.5), StopTols { ftol_rel: 1e-4, .. })`), a port of the COBYLA in NLopt, which is
// M. J. D. Powell's "direct search optimization method that models the objective and constraint functions by linear
// interpolation" (1994) plus S. G. Johnson's wrapper: variables rescaled by the initial step, bounds turned into 2n
// linear inequality constraints AND the objective evaluated at the point clamped into the box, initial simplex steps
// flipped to stay inside the box, rho doubled after a step whose actual reduction is within 10 % of the predicted
// one, and the ftol test made where rho is about to be reduced (best value now against the best value at the previous
// reduction).  The crate is not vendored in the reference checkout, so this file restates the PUBLISHED algorithm:
//   * the simplex / merit-function / trust-radius logic of Powell's COBYLB (vertex replacement by the sigma / eta
//     acceptability test, penalty parameter mu raised to 2 * barmu, rho halved when no progress) follows the paper;
//   * the trust-region subproblem -- minimise the greatest violation of the linearised constraints inside the ball, then
//     the linear model of f without increasing it (Powell's TRSTLP) -- is solved in CLOSED FORM: the only constraints
//     here are bounds, whose linear models are exact, so the two stages reduce to a shrink-the-excess step and to
//     d(tau) = clip(-tau g, L, U) with tau fixed by |d| = rho (the point Powell's active-set path ends at for a box).
// Every start of the multistart is one machine; the driver advances all of them in lock-step and evaluates their
// requests as ONE egx_gp_likelihood_batch (gp_fit.hip).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

namespace egx {

class CobylaBox {
public:
    enum Status { RUNNING = 0, MAXEVAL = 1, FTOL = 2, RHOEND = 3, ROUNDOFF = 4 };

    // x0, lo, hi in the caller's units; rhobeg = initial step (all coordinates), ftol_rel as nlopt_stop_ftol
    CobylaBox(const std::vector<double> &x0, const std::vector<double> &lo, const std::vector<double> &hi, double rhobeg,
              double ftol_rel, int64_t maxeval, double rhoend_scaled = 0.0, bool rho_doubling = true, bool clamp_eval = true)
        : n_((int)x0.size()), scale_(rhobeg), ftol_rel_(ftol_rel), maxfun_(maxeval), rhoend_(rhoend_scaled),
          rho_doubling_(rho_doubling), clamp_eval_(clamp_eval) {
        const int n = n_;
        lo_.resize(n);
        hi_.resize(n);
        x_.resize(n);
        for (int i = 0; i < n; i++) {  // rescaled variables: the initial step is 1 in every coordinate
            lo_[i] = lo[i] / scale_;
            hi_[i] = hi[i] / scale_;
            x_[i] = std::fmin(hi_[i], std::fmax(lo_[i], x0[i] / scale_));
        }
        sim_.assign((size_t)n * (n + 1), 0.0);
        simi_.assign((size_t)n * n, 0.0);
        datf_.assign(n + 1, 0.0);
        datr_.assign(n + 1, 0.0);
        vsig_.assign(n, 0.0);
        veta_.assign(n, 0.0);
        sigbar_.assign(n, 0.0);
        dx_.assign(n, 0.0);
        g_.assign(n, 0.0);
        rho_ = 1.0;
        parmu_ = 0.0;
        for (int i = 0; i < n; i++) sim(i, n) = x_[i];
        jdrop_ = n;
        ibrnch_ = 0;
        label_ = L_EVAL;
    }

    // Next point to evaluate (caller's units, inside the box).  Returns false when the run has finished.
    bool ask(std::vector<double> &x_out) {
        if (label_ != L_EVAL) run();
        if (status_ != RUNNING) return false;
        if (nfvals_ >= maxfun_ && nfvals_ > 0) {
            finish(MAXEVAL);
            return false;
        }
        x_out.resize(n_);
        for (int i = 0; i < n_; i++) x_out[i] = (clamp_eval_ ? std::fmin(hi_[i], std::fmax(lo_[i], x_[i])) : x_[i]) * scale_;
        return true;
    }

    // Objective value at the point handed out by the last ask().  Non-finite values (a failed likelihood is +inf in
    // the reference's objective, algorithm.rs:893-896) enter as a large finite barrier so that the simplex algebra
    // stays finite.
    void tell(double f) {
        if (!(f == f) || f > kBarrier) f = kBarrier;
        nfvals_++;
        f_ = f;
        resmax_ = violation(x_);
        label_ = L_AFTER_EVAL;
        run();
    }

    Status status() const { return status_; }
    int64_t evals() const { return nfvals_; }
    double best_f() const { return fbest_; }
    std::vector<double> best_x() const {
        std::vector<double> r(n_);
        for (int i = 0; i < n_; i++) r[i] = std::fmin(hi_[i], std::fmax(lo_[i], xbest_[i])) * scale_;
        return r;
    }

private:
    static constexpr double kBarrier = 1e30;
    enum Label { L_EVAL, L_AFTER_EVAL, L140, L370, L440, L550, L_DONE };
    int n_;
    double scale_, ftol_rel_;
    int64_t maxfun_;
    double rhoend_;
    bool rho_doubling_;  // false = Powell's original radius schedule (used to validate against his Fortran code)
    bool clamp_eval_;    // false = hand out the raw trial point even outside the box (Powell's original; validation only)
    std::vector<double> lo_, hi_, x_, sim_, simi_, datf_, datr_, vsig_, veta_, sigbar_, dx_, g_, xbest_;
    double rho_ = 1.0, parmu_ = 0.0, f_ = 0.0, resmax_ = 0.0, prerec_ = 0.0, prerem_ = 0.0, parsig_ = 0.0;
    double minf_ = std::numeric_limits<double>::infinity(), fbest_ = std::numeric_limits<double>::infinity();
    int jdrop_ = 0, ibrnch_ = 0, iflag_ = 0, ifull_ = 0;
    int64_t nfvals_ = 0;
    Label label_ = L_EVAL;
    Status status_ = RUNNING;

    double &sim(int i, int j) { return sim_[(size_t)i * (n_ + 1) + j]; }
    double &simi(int i, int j) { return simi_[(size_t)i * n_ + j]; }

    // greatest violation of the 2n bound constraints c = x - lo >= 0, hi - x >= 0 (rescaled units)
    double violation(const std::vector<double> &x) const {
        double r = 0.0;
        for (int i = 0; i < n_; i++) r = std::fmax(r, std::fmax(lo_[i] - x[i], x[i] - hi_[i]));
        return r;
    }

    // simi_ <- inverse of the n x n displacement matrix (Gauss-Jordan, partial pivoting); false when singular
    bool invert_simplex() {
        const int n = n_;
        std::vector<double> a((size_t)n * 2 * n, 0.0);
        for (int i = 0; i < n; i++) {
            for (int j = 0; j < n; j++) a[(size_t)i * 2 * n + j] = sim(i, j);
            a[(size_t)i * 2 * n + n + i] = 1.0;
        }
        for (int c = 0; c < n; c++) {
            int piv = c;
            for (int r = c + 1; r < n; r++)
                if (std::fabs(a[(size_t)r * 2 * n + c]) > std::fabs(a[(size_t)piv * 2 * n + c])) piv = r;
            if (a[(size_t)piv * 2 * n + c] == 0.0) return false;
            if (piv != c)
                for (int j = 0; j < 2 * n; j++) std::swap(a[(size_t)piv * 2 * n + j], a[(size_t)c * 2 * n + j]);
            const double inv = 1.0 / a[(size_t)c * 2 * n + c];
            for (int j = 0; j < 2 * n; j++) a[(size_t)c * 2 * n + j] *= inv;
            for (int r = 0; r < n; r++)
                if (r != c) {
                    const double f = a[(size_t)r * 2 * n + c];
                    if (f != 0.0)
                        for (int j = 0; j < 2 * n; j++) a[(size_t)r * 2 * n + j] -= f * a[(size_t)c * 2 * n + j];
                }
        }
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) simi(i, j) = a[(size_t)i * 2 * n + n + j];
        return true;
    }

    void finish(Status s) {
        status_ = s;
        label_ = L_DONE;
        // the optimal vertex sits in the pole position of the simplex
        xbest_.resize(n_);
        for (int i = 0; i < n_; i++) xbest_[i] = sim(i, n_);
        fbest_ = datf_[n_];
        if (s == RHOEND && ifull_ == 1) {  // Powell returns the last trial point itself in this case
            xbest_ = x_;
            fbest_ = f_;
        }
    }

    // Powell's TRSTLP for box constraints in closed form: dx_ <- step from the pole x0, ifull_ <- 1 when |dx| = rho
    void trust_step(const std::vector<double> &x0) {
        const int n = n_;
        const double rho = rho_;
        std::vector<double> vlo(n), vhi(n);
        double res0 = 0.0;
        for (int i = 0; i < n; i++) {
            vlo[i] = lo_[i] - x0[i];
            vhi[i] = x0[i] - hi_[i];
            res0 = std::fmax(res0, std::fmax(vlo[i], vhi[i]));
        }
        double tstar = 0.0;
        if (res0 > 0.0) {  // stage 1: the least greatest violation t reachable inside the ball
            auto need = [&](double t) {
                double s = 0.0;
                for (int i = 0; i < n; i++) {
                    const double a = std::fmax(0.0, vlo[i] - t), b = std::fmax(0.0, vhi[i] - t);
                    s += a * a + b * b;
                }
                return s;
            };
            if (need(0.0) > rho * rho) {
                double a = 0.0, b = res0;
                for (int it = 0; it < 200; it++) {
                    const double mid = 0.5 * (a + b);
                    if (need(mid) > rho * rho) a = mid; else b = mid;
                }
                tstar = b;
            }
        }
        std::vector<double> L(n), U(n);
        for (int i = 0; i < n; i++) {
            L[i] = vlo[i] - tstar;   // d_i >= L_i keeps the lower-bound violation at or below t*
            U[i] = -vhi[i] + tstar;  // d_i <= U_i
        }
        auto clipd = [&](double tau, std::vector<double> &d) {
            double s = 0.0;
            for (int i = 0; i < n; i++) {
                double v = -tau * g_[i];
                if (g_[i] == 0.0) v = 0.0;
                v = std::fmin(U[i], std::fmax(L[i], v));
                d[i] = v;
                s += v * v;
            }
            return s;
        };
        ifull_ = 0;
        if (tstar > 0.0) {  // the ball was used up reducing the violation: no second stage
            clipd(0.0, dx_);
            ifull_ = 1;
            return;
        }
        // stage 2: d(tau) = clip(-tau g, L, U), tau as large as the ball allows
        double gmax = 0.0;
        for (int i = 0; i < n; i++) gmax = std::fmax(gmax, std::fabs(g_[i]));
        if (gmax == 0.0 || !(gmax < std::numeric_limits<double>::infinity())) {
            clipd(0.0, dx_);
            return;
        }
        double tau_inf = 0.0;  // beyond this tau every moving coordinate sits on its bound
        for (int i = 0; i < n; i++)
            if (g_[i] != 0.0) tau_inf = std::fmax(tau_inf, (g_[i] < 0.0 ? U[i] : -L[i]) / std::fabs(g_[i]));
        tau_inf = std::fmax(tau_inf, 0.0);
        if (clipd(tau_inf, dx_) <= rho * rho) return;  // the whole box corner lies inside the ball
        double a = 0.0, b = tau_inf;
        for (int it = 0; it < 200; it++) {
            const double mid = 0.5 * (a + b);
            if (clipd(mid, dx_) > rho * rho) b = mid; else a = mid;
        }
        clipd(a, dx_);
        ifull_ = 1;
    }

    void run() {
        const int n = n_, np = n_;
        const double alpha = 0.25, beta = 2.1, gamma = 0.5, delta = 1.1;
        for (;;) {
            switch (label_) {
            case L_EVAL:
            case L_DONE:
                return;
            case L_AFTER_EVAL: {
                if (ibrnch_ == 1) {
                    label_ = L440;
                    break;
                }
                // a vertex of the initial simplex (or of a geometry step)
                datf_[jdrop_] = f_;
                datr_[jdrop_] = resmax_;
                if (nfvals_ <= np + 1) {
                    if (jdrop_ < n) {
                        if (datf_[np] <= f_) {
                            x_[jdrop_] = sim(jdrop_, np);
                        } else {  // the new point becomes the pole
                            // (the old pole and the vertices made so far all move by -step along this coordinate)
                            const double step = sim(jdrop_, jdrop_);
                            sim(jdrop_, np) = x_[jdrop_];
                            std::swap(datf_[jdrop_], datf_[np]);
                            std::swap(datr_[jdrop_], datr_[np]);
                            for (int k = 0; k <= jdrop_; k++) sim(jdrop_, k) = -step;
                        }
                    }
                    if (nfvals_ <= n) {
                        jdrop_ = (int)nfvals_ - 1;
                        // step of the initial simplex, flipped / shortened so that the vertex stays inside the box
                        double step = rho_;
                        const double xj = x_[jdrop_];
                        if (xj + step > hi_[jdrop_]) {
                            if (xj - step >= lo_[jdrop_]) step = -step;
                            else if (hi_[jdrop_] - xj > xj - lo_[jdrop_]) step = 0.5 * (hi_[jdrop_] - xj);
                            else step = -0.5 * (xj - lo_[jdrop_]);
                        }
                        if (step == 0.0) step = rho_;  // degenerate box (lo == hi): the violation term takes over
                        x_[jdrop_] += step;
                        sim(jdrop_, jdrop_) = step;
                        label_ = L_EVAL;
                        return;
                    }
                    if (!invert_simplex()) {  // initial simplex complete: SIMI = inverse of the displacement matrix
                        finish(ROUNDOFF);
                        return;
                    }
                }
                ibrnch_ = 1;
                label_ = L140;
                break;
            }
            case L140: {
                // the optimal vertex (merit f + mu * violation) goes to the pole position
                double phimin = datf_[np] + parmu_ * datr_[np];
                int nbest = np;
                for (int j = 0; j < n; j++) {
                    const double temp = datf_[j] + parmu_ * datr_[j];
                    if (temp < phimin) {
                        nbest = j;
                        phimin = temp;
                    } else if (temp == phimin && parmu_ == 0.0 && datr_[j] < datr_[nbest]) {
                        nbest = j;
                    }
                }
                if (nbest < n) {
                    std::swap(datf_[np], datf_[nbest]);
                    std::swap(datr_[np], datr_[nbest]);
                    for (int i = 0; i < n; i++) {
                        const double temp = sim(i, nbest);
                        sim(i, nbest) = 0.0;
                        sim(i, np) += temp;
                        double tempa = 0.0;
                        for (int k = 0; k < n; k++) {
                            sim(i, k) -= temp;
                            tempa -= simi(k, i);
                        }
                        simi(nbest, i) = tempa;
                    }
                }
                // SIMI must still be the inverse of SIM
                double error = 0.0;
                for (int i = 0; i < n; i++)
                    for (int j = 0; j < n; j++) {
                        double temp = (i == j) ? -1.0 : 0.0;
                        for (int k = 0; k < n; k++) temp += simi(i, k) * sim(k, j);
                        error = std::fmax(error, std::fabs(temp));
                    }
                if (!(error <= 0.1)) {
                    finish(ROUNDOFF);
                    return;
                }
                // gradient of the linear interpolant of f
                for (int i = 0; i < n; i++) {
                    double temp = 0.0;
                    for (int j = 0; j < n; j++) temp += (datf_[j] - datf_[np]) * simi(j, i);
                    g_[i] = temp;
                }
                // acceptability of the simplex
                iflag_ = 1;
                parsig_ = alpha * rho_;
                const double pareta = beta * rho_;
                for (int j = 0; j < n; j++) {
                    double wsig = 0.0, weta = 0.0;
                    for (int i = 0; i < n; i++) {
                        wsig += simi(j, i) * simi(j, i);
                        weta += sim(i, j) * sim(i, j);
                    }
                    vsig_[j] = 1.0 / std::sqrt(wsig);
                    veta_[j] = std::sqrt(weta);
                    if (vsig_[j] < parsig_ || veta_[j] > pareta) iflag_ = 0;
                }
                if (ibrnch_ == 1 || iflag_ == 1) {
                    label_ = L370;
                    break;
                }
                // geometry step: replace the worst vertex
                int jd = -1;
                double temp = pareta;
                for (int j = 0; j < n; j++)
                    if (veta_[j] > temp) {
                        jd = j;
                        temp = veta_[j];
                    }
                if (jd < 0)
                    for (int j = 0; j < n; j++)
                        if (vsig_[j] < temp) {
                            jd = j;
                            temp = vsig_[j];
                        }
                jdrop_ = jd;
                temp = gamma * rho_ * vsig_[jd];
                for (int i = 0; i < n; i++) dx_[i] = temp * simi(jd, i);
                // sign of the step: the one with the smaller predicted merit (constraints are linear: exact)
                double cvmaxp = 0.0, cvmaxm = 0.0, sum = 0.0;
                for (int i = 0; i < n; i++) {
                    const double clo = sim(i, np) - lo_[i], chi = hi_[i] - sim(i, np);
                    cvmaxp = std::fmax(cvmaxp, std::fmax(-dx_[i] - clo, dx_[i] - chi));
                    cvmaxm = std::fmax(cvmaxm, std::fmax(dx_[i] - clo, -dx_[i] - chi));
                    sum -= g_[i] * dx_[i];
                }
                const double dxsign = (parmu_ * (cvmaxp - cvmaxm) > sum + sum) ? -1.0 : 1.0;
                temp = 0.0;
                for (int i = 0; i < n; i++) {
                    dx_[i] *= dxsign;
                    sim(i, jd) = dx_[i];
                    temp += simi(jd, i) * dx_[i];
                }
                for (int i = 0; i < n; i++) simi(jd, i) /= temp;
                for (int j = 0; j < n; j++) {
                    if (j != jd) {
                        double t2 = 0.0;
                        for (int i = 0; i < n; i++) t2 += simi(j, i) * dx_[i];
                        for (int i = 0; i < n; i++) simi(j, i) -= t2 * simi(jd, i);
                    }
                    x_[j] = sim(j, np) + dx_[j];
                }
                label_ = L_EVAL;
                return;
            }
            case L370: {
                std::vector<double> x0(n);
                for (int i = 0; i < n; i++) x0[i] = sim(i, np);
                trust_step(x0);
                if (ifull_ == 0) {
                    double temp = 0.0;
                    for (int i = 0; i < n; i++) temp += dx_[i] * dx_[i];
                    if (temp < rho_ * 0.25 * rho_) {
                        ibrnch_ = 1;
                        label_ = L550;
                        break;
                    }
                }
                // predicted change of f and of the greatest violation
                std::vector<double> xn(n);
                double sum = 0.0;
                for (int i = 0; i < n; i++) {
                    xn[i] = x0[i] + dx_[i];
                    sum += g_[i] * dx_[i];
                }
                const double resnew = violation(xn);
                double barmu = 0.0;
                prerec_ = datr_[np] - resnew;
                if (prerec_ > 0.0) barmu = sum / prerec_;
                if (parmu_ < barmu * 1.5) {
                    parmu_ = barmu * 2.0;
                    const double phi = datf_[np] + parmu_ * datr_[np];
                    bool again = false;
                    for (int j = 0; j < n && !again; j++) {
                        const double temp = datf_[j] + parmu_ * datr_[j];
                        if (temp < phi) again = true;
                        else if (temp == phi && parmu_ == 0.0 && datr_[j] < datr_[np]) again = true;
                    }
                    if (again) {
                        label_ = L140;
                        break;
                    }
                }
                prerem_ = parmu_ * prerec_ - sum;
                x_ = xn;
                ibrnch_ = 1;
                label_ = L_EVAL;
                return;
            }
            case L440: {
                const double vmold = datf_[np] + parmu_ * datr_[np];
                const double vmnew = f_ + parmu_ * resmax_;
                double trured = vmold - vmnew;
                if (parmu_ == 0.0 && f_ == datf_[np]) {
                    prerem_ = prerec_;
                    trured = datr_[np] - resmax_;
                }
                // which vertex does the new point replace?
                double ratio = (trured <= 0.0) ? 1.0 : 0.0;
                int jd = -1;
                for (int j = 0; j < n; j++) {
                    double temp = 0.0;
                    for (int i = 0; i < n; i++) temp += simi(j, i) * dx_[i];
                    temp = std::fabs(temp);
                    if (temp > ratio) {
                        jd = j;
                        ratio = temp;
                    }
                    sigbar_[j] = temp * vsig_[j];
                }
                double edgmax = delta * rho_;
                int l = -1;
                for (int j = 0; j < n; j++)
                    if (sigbar_[j] >= parsig_ || sigbar_[j] >= vsig_[j]) {
                        double temp = veta_[j];
                        if (trured > 0.0) {
                            temp = 0.0;
                            for (int i = 0; i < n; i++) temp += (dx_[i] - sim(i, j)) * (dx_[i] - sim(i, j));
                            temp = std::sqrt(temp);
                        }
                        if (temp > edgmax) {
                            l = j;
                            edgmax = temp;
                        }
                    }
                if (l >= 0) jd = l;
                if (jd < 0) {
                    label_ = L550;
                    break;
                }
                jdrop_ = jd;
                double temp = 0.0;
                for (int i = 0; i < n; i++) {
                    sim(i, jd) = dx_[i];
                    temp += simi(jd, i) * dx_[i];
                }
                for (int i = 0; i < n; i++) simi(jd, i) /= temp;
                for (int j = 0; j < n; j++)
                    if (j != jd) {
                        double t2 = 0.0;
                        for (int i = 0; i < n; i++) t2 += simi(j, i) * dx_[i];
                        for (int i = 0; i < n; i++) simi(j, i) -= t2 * simi(jd, i);
                    }
                datf_[jd] = f_;
                datr_[jd] = resmax_;
                if (trured > 0.0 && trured >= prerem_ * 0.1) {
                    // (S. G. Johnson's modification) a step that did what the model promised earns a larger radius
                    if (rho_doubling_ && trured >= prerem_ * 0.9 && trured <= prerem_ * 1.1 && iflag_) rho_ *= 2.0;
                    label_ = L140;
                    break;
                }
                label_ = L550;
                break;
            }
            case L550: {
                if (iflag_ == 0) {
                    ibrnch_ = 0;
                    label_ = L140;
                    break;
                }
                // the function-value stopping test lives where rho is about to be reduced
                {
                    const double fb = (ifull_ == 1) ? f_ : datf_[np];
                    if (fb < minf_ && ftol_rel_ > 0.0) {
                        const double d = std::fabs(fb - minf_);
                        if (std::isfinite(minf_) && (d < ftol_rel_ * (std::fabs(fb) + std::fabs(minf_)) * 0.5 || fb == minf_)) {
                            finish(FTOL);
                            return;
                        }
                    }
                    minf_ = fb;
                }
                if (rho_ > rhoend_) {
                    rho_ *= 0.5;
                    if (rho_ <= rhoend_ * 1.5) rho_ = rhoend_;
                    if (parmu_ > 0.0) {
                        // mu is reset from the spread of f over the simplex against the spread of the constraints
                        double denom = 0.0;
                        for (int c = 0; c < 2 * n; c++) {
                            const int i = c >> 1;
                            auto cval = [&](int j) {
                                const double xi = sim(i, np) + (j < n ? sim(i, j) : 0.0);
                                return (c & 1) ? hi_[i] - xi : xi - lo_[i];
                            };
                            double cmin = cval(np), cmax = cmin;
                            for (int j = 0; j < n; j++) {
                                cmin = std::fmin(cmin, cval(j));
                                cmax = std::fmax(cmax, cval(j));
                            }
                            if (cmin < cmax * 0.5) {
                                const double temp = std::fmax(cmax, 0.0) - cmin;
                                denom = (denom <= 0.0) ? temp : std::fmin(denom, temp);
                            }
                        }
                        double fmin = datf_[np], fmax = datf_[np];
                        for (int j = 0; j < n; j++) {
                            fmin = std::fmin(fmin, datf_[j]);
                            fmax = std::fmax(fmax, datf_[j]);
                        }
                        if (denom == 0.0) parmu_ = 0.0;
                        else if (fmax - fmin < parmu_ * denom) parmu_ = (fmax - fmin) / denom;
                    }
                    label_ = L140;
                    break;
                }
                finish(RHOEND);
                return;
            }
            }
        }
    }
};

}  // namespace egx
