// CPU check of the flow launch's scheduling rules (egobox_amd/csrc/pipe_flow.h; the device code that applies them is
// k_potrf_flow / flow_worker_loop in kernels_pipe.hip).  No GPU: a discrete-event simulation of W workgroups that claim tasks
// by exactly the rules of pipe_flow.h -- critical tickets of the one open stage first, then the nearest column's released round,
// rounds of a column in order, gates G1 / G2 / G3 -- and BLOCK inside a claimed task until its producers are finished, with random
// task durations.  Checked, for several shapes, worker counts and seeds:
//   * the simulation ends (no deadlock: at every instant some claimed task can run or some worker can claim one),
//   * every update of the blocked algorithm is applied exactly once and in panel order: tile (I, J) of block column q receives
//     the panels 0 .. q - 1 one after the other (far rounds of two, near rounds of one, the last one as LAST / FINE),
//   * every diagonal block is factored after its ten fine tiles, every row chunk solved after its four fine-tile counts,
//   * the completion counts the gates wait for (flow_need_pre / flow_need_last) are exactly what the rounds deliver,
//   * at most one stage has unclaimed critical tickets at any time.
// Build + run: g++ -O2 -std=c++17 -I egobox_amd/csrc tests/c_host/flow_order_test.cpp -o /tmp/flow_order_test && /tmp/flow_order_test
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "pipe_flow.h"

using namespace egx;

namespace {

struct Task {
    int type = -1;            // PT_*
    int p = 0, a = 0, b = 0;  // as PipeTask; PT_BULK: p = p0, a = I, b = q; p1 below
    int p1 = 0, q = 0, last = 0, bulk_class = 0;
};

struct Sim {
    int n_pad, m_tot, NP, NC, NI, W, lead_short, lead_long;
    FlowShape sh;
    std::mt19937 rng;
    // hand-off state
    int open_upto = 0;
    std::vector<int> cnext, pre_done, last_done, rcur, diag_done;
    std::vector<std::vector<int>> rcnt, cver, trsm_done, fcnt;
    std::vector<std::vector<PipeTask>> stage;
    // workers
    struct Worker {
        int state = 0;  // 0 idle, 1 claimed (blocked), 2 running, 3 waiting for its diagonal block, 4 factoring it
        Task t;
        double end = 0;
        int my_block = -1;
    };
    std::vector<Worker> w;
    long claimed = 0, finished = 0;
    int fail = 0;

    Sim(int n_pad_, int m_tot_, int W_, unsigned seed, int ls, int ll)
        : n_pad(n_pad_), m_tot(m_tot_), NP(n_pad_ / 256), NC(m_tot_ / 64), NI(m_tot_ / 128), W(W_), lead_short(ls), lead_long(ll),
          sh{NP, NC, NI}, rng(seed) {
        cnext.assign(NP, 0), pre_done.assign(NP, 0), last_done.assign(NP, 0), rcur.assign(NP, 0), diag_done.assign(NP, 0);
        rcnt.assign(NP, std::vector<int>(NP / kFlowGP + 3, 0));
        cver.assign(NI, std::vector<int>(2 * NP, 0));
        trsm_done.assign(NP, std::vector<int>(NC, 0));
        fcnt.assign(NP + 1, std::vector<int>(NC, 0));
        for (int s = 0; s < NP; s++) stage.push_back(flow_stage_tasks(n_pad, m_tot, s));
        w.resize(W);
        for (int i = 0; i < W && i < NP; i++) w[i].my_block = i;
    }
    int blocks_done() const {
        int c = 0;
        while (c < NP && diag_done[c]) c++;
        return c;
    }
    void check(bool ok, const char *what, int x = 0, int y = 0, int z = 0) {
        if (!ok) {
            if (fail < 10) std::printf("  VIOLATION %s (%d %d %d)\n", what, x, y, z);
            fail++;
        }
    }
    bool ready(const Task &t) const {
        auto solved = [&](int p, int c) { return trsm_done[p][c] != 0; };
        switch (t.type) {
            case PT_TRSM: return diag_done[t.p] && (t.p == 0 || fcnt[t.p][t.a] >= 4);
            case PT_FINE: {
                const int c_col = 4 * (t.p + 1) + t.b;
                return solved(t.p, t.a) && solved(t.p, c_col) && cver[t.a / 2][2 * (t.p + 1) + t.b / 2] >= t.p;
            }
            case PT_COARSE:
            case PT_COARSE_LAST:
                return solved(t.p, 2 * t.a) && solved(t.p, 2 * t.a + 1) && solved(t.p, 2 * t.b) && solved(t.p, 2 * t.b + 1) && cver[t.a][t.b] >= t.p;
            case PT_BULK: {
                bool ok = solved(t.p1 - 1, 2 * t.a) && solved(t.p1 - 1, 2 * t.a + 1);
                for (int c = 4 * t.q; c < 4 * t.q + 4; c++) ok = ok && solved(t.p1 - 1, c);
                ok = ok && cver[t.a][2 * t.q] >= t.p;
                if (t.a != 2 * t.q) ok = ok && cver[t.a][2 * t.q + 1] >= t.p;
                return ok;
            }
        }
        return false;
    }
    void finish(const Task &t) {
        finished++;
        switch (t.type) {
            case PT_TRSM:
                check(!trsm_done[t.p][t.a], "row chunk solved twice", t.p, t.a);
                trsm_done[t.p][t.a] = 1;
                break;
            case PT_FINE: fcnt[t.p + 1][t.a]++; break;
            case PT_COARSE:
            case PT_COARSE_LAST:
                check(cver[t.a][t.b] == t.p, "near update out of panel order", t.p, t.a, t.b);
                cver[t.a][t.b] = t.p + 1;
                if (t.type == PT_COARSE_LAST) fcnt[t.p + 1][2 * t.a] += 2, fcnt[t.p + 1][2 * t.a + 1] += 2;
                break;
            case PT_BULK:
                check(cver[t.a][2 * t.q] == t.p, "far update out of panel order", t.p, t.a, t.q);
                cver[t.a][2 * t.q] = t.p1;
                if (t.a != 2 * t.q) {
                    check(cver[t.a][2 * t.q + 1] == t.p, "far update out of panel order (right half)", t.p, t.a, t.q);
                    cver[t.a][2 * t.q + 1] = t.p1;
                }
                break;
        }
        if (t.bulk_class) (t.last ? last_done : pre_done)[t.q]++;
    }
    // the claim rules of pipe_flow.h; returns true when the worker got a task
    bool claim(Worker &wk, bool allow_long) {
        // gates
        for (;;) {
            const int S = open_upto;
            if (wk.my_block >= 0 && S >= wk.my_block) return false;  // (its own stage has opened meanwhile: next pass leaves)
            if (cnext[S] < (int)stage[S].size()) {
                const PipeTask &pt = stage[S][cnext[S]++];
                wk.t = Task();
                wk.t.type = pt.type, wk.t.p = pt.p, wk.t.a = pt.a, wk.t.b = pt.b, wk.t.q = pt.p + 1;
                return true;
            }
            if (S + 1 < NP) {
                const int s1 = S + 1;
                const bool g2 = last_done[s1] >= flow_need_last(sh, s1) && pre_done[s1] >= flow_need_pre(sh, s1);
                const bool g3 = s1 + 1 >= NP || pre_done[s1 + 1] >= flow_need_pre(sh, s1 + 1);
                if (g2 && g3) {
                    for (int s = 0; s < s1; s++) check(cnext[s] >= (int)stage[s].size(), "a stage opened while an earlier one has tickets", s1, s);
                    open_upto = s1;
                    continue;
                }
            }
            break;
        }
        if (!allow_long) return false;
        for (int q = 2; q < NP; q++) {
            int &r = rcur[q];
            while (r < flow_nrounds(q) && flow_round_release_stage(q, r) <= open_upto && rcnt[q][r] >= flow_round_size(sh, q, r)) r++;
            if (r >= flow_nrounds(q) || flow_round_release_stage(q, r) > open_upto) continue;
            const int tk = rcnt[q][r]++;
            const FlowBulkTask bt = flow_round_task(sh, q, r, tk);
            wk.t = Task();
            wk.t.type = bt.type, wk.t.p = bt.p0, wk.t.p1 = bt.p1, wk.t.a = bt.I, wk.t.b = bt.type == PT_BULK ? q : bt.J, wk.t.q = q;
            wk.t.last = bt.last, wk.t.bulk_class = 1;
            return true;
        }
        return false;
    }
    double duration(const Task &t) {
        std::uniform_real_distribution<double> u(0.5, 1.5);
        const double base = t.type == PT_BULK ? 146.0 : (t.type == PT_FINE ? 17.0 : (t.type == PT_TRSM ? 20.0 : 40.0));
        return base * u(rng);
    }
    bool run() {
        double now = 0;
        long guard = 0;
        for (;;) {
            bool progress = false;
            // claims + starts
            for (auto &wk : w) {
                if (wk.state == 0) {
                    const int cur = blocks_done();
                    // (leaves `lead_short` blocks ahead of its own -- and, whatever the lead, once ITS stage is open: it must never hold
                    //  a ticket that waits for the block only it can factor)
                    if (wk.my_block >= 0 && (wk.my_block - cur <= lead_short || open_upto >= wk.my_block)) {
                        wk.state = 3;
                        progress = true;
                        continue;
                    }
                    const bool allow_long = wk.my_block < 0 || wk.my_block - cur > lead_long;
                    if (claim(wk, allow_long)) {
                        wk.state = 1, claimed++, progress = true;
                    }
                }
                if (wk.state == 1 && ready(wk.t)) wk.state = 2, wk.end = now + duration(wk.t), progress = true;
                if (wk.state == 3) {
                    const int p = wk.my_block;
                    bool ok = true;
                    for (int i = 0; i < 4 && p > 0; i++) ok = ok && fcnt[p][4 * p + i] >= i + 1;
                    if (ok) wk.state = 4, wk.end = now + 62.0, progress = true;
                }
            }
            // next completion
            double tmin = -1;
            for (auto &wk : w)
                if ((wk.state == 2 || wk.state == 4) && (tmin < 0 || wk.end < tmin)) tmin = wk.end;
            if (tmin < 0) {
                if (progress) continue;
                break;  // nothing runs and nothing can be claimed or started
            }
            now = tmin;
            for (auto &wk : w) {
                if (wk.state == 2 && wk.end <= now) finish(wk.t), wk.state = 0;
                else if (wk.state == 4 && wk.end <= now) {
                    check(!diag_done[wk.my_block], "diagonal block factored twice", wk.my_block);
                    check(wk.my_block == 0 || diag_done[wk.my_block - 1], "diagonal block ahead of its predecessor", wk.my_block);
                    diag_done[wk.my_block] = 1, wk.my_block = -1, wk.state = 0;
                }
            }
            if (++guard > 50000000) {
                check(false, "simulation does not end");
                break;
            }
        }
        // ---- what must hold at the end
        for (const auto &wk : w) check(wk.state == 0, "a workgroup is still blocked at the end (deadlock)", wk.state, wk.t.type, wk.t.p);
        check(claimed == finished, "claimed != finished");
        for (int p = 0; p < NP; p++) {
            check(diag_done[p], "diagonal block not factored", p);
            check(cnext[p] >= (int)stage[p].size(), "critical tickets left", p);
            for (int c = 4 * p + 4; c < NC; c++) check(trsm_done[p][c], "row chunk not solved", p, c);
        }
        for (int q = 2; q < NP; q++) {
            check(rcur[q] == flow_nrounds(q) || rcnt[q][flow_nrounds(q) - 1] >= flow_round_size(sh, q, flow_nrounds(q) - 1), "rounds left", q);
            check(pre_done[q] == flow_need_pre(sh, q), "pre_done != need_pre", q, pre_done[q], flow_need_pre(sh, q));
            check(last_done[q] == flow_need_last(sh, q), "last_done != need_last", q, last_done[q], flow_need_last(sh, q));
        }
        for (int q = 1; q < NP; q++)
            for (int J = 2 * q; J <= 2 * q + 1; J++)
                for (int I = J; I < NI; I++) {
                    const bool in_diag = I < 2 * q + 2;
                    check(cver[I][J] == (in_diag ? q - 1 : q), "a tile did not receive every panel", I, J, cver[I][J]);
                }
        for (int q = 1; q < NP; q++)
            for (int c = 4 * q; c < NC; c++) check(fcnt[q][c] == (c < 4 * q + 4 ? c - 4 * q + 1 : 4), "fine-tile count of a row chunk", q, c, fcnt[q][c]);
        return fail == 0;
    }
};

}  // namespace

int main() {
    int bad = 0;
    long tasks = 0;
    struct Shape { int n_pad, rhs; } shapes[] = {{512, 128}, {768, 128}, {1024, 128}, {2048, 256}, {4096, 128}, {8192, 128}};
    for (const auto &sp : shapes)
        for (int W : {0, 40, 256})
            for (unsigned seed : {1u, 2u, 3u}) {
                const int NP = sp.n_pad / 256;
                const int Wn = W == 0 ? NP + 1 : (W < NP + 1 ? NP + 1 : W);
                for (int leads = 0; leads < 2; leads++) {
                    Sim s(sp.n_pad, sp.n_pad + sp.rhs, Wn, seed, leads ? 0 : 1, leads ? 0 : 3);
                    const bool ok = s.run();
                    tasks += s.finished;
                    if (!ok) {
                        std::printf("FAIL n_pad %d rhs %d workers %d seed %u leads %d: %d violations\n", sp.n_pad, sp.rhs, Wn, seed, leads, s.fail);
                        bad++;
                    }
                }
            }
    // every stage list: tasks of its own panel only, in dependency order (last, solves, next block's near tiles, fine tiles)
    for (int n_pad : {1024, 4096}) {
        const int NP = n_pad / 256;
        for (int s = 0; s < NP; s++) {
            int phase = 0;
            for (const PipeTask &t : flow_stage_tasks(n_pad, n_pad + 128, s)) {
                const int ph = t.type == PT_COARSE_LAST ? 0 : (t.type == PT_TRSM ? 1 : (t.type == PT_COARSE ? 2 : 3));
                if (ph < phase) std::printf("FAIL stage %d: task types out of order\n", s), bad++;
                phase = ph;
            }
        }
    }
    std::printf("%s: %ld simulated tasks\n", bad ? "flow order test FAILED" : "flow order test ok", tasks);
    return bad ? 1 : 0;
}
