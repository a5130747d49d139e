// CPU check of the flow launch's scheduling rules (egobox_amd/csrc/pipe_flow.h; the device code that applies them is
// k_potrf_flow / flow_worker_loop in kernels_pipe.hip).  No GPU: a discrete-event simulation of W workgroups that claim tasks
// by exactly the rules of pipe_flow.h -- LOOK (the next ticket of the open stage, else the two columns next to the chain, else
// the oldest released round: only a ticket whose task would start at once, head solves and fine tiles excepted) then TAKE (the
// counter's next ticket, which is a LATER, unchecked one when several workgroups looked at the same moment), stage s opened
// when every ticket of stage s - 1 is claimed, rounds of a column in order, a workgroup that owns a diagonal block never
// holding a ticket of its own stage -- and BLOCK inside a claimed task until its producers are finished, with random task
// durations.  Checked, for several shapes, worker counts (down to NP + 1) and seeds:
//   * the simulation ends (no deadlock: at every instant some claimed task can run or some worker can claim one),
//   * every update of the blocked algorithm is applied exactly once and in panel order: tile (I, J) of block column q receives
//     the panels 0 .. q - 1 one after the other (far rounds of two, near rounds of one, the last one as LAST / FINE),
//   * every diagonal block is factored after its ten fine tiles, every row chunk solved after its four fine-tile counts,
//   * at most one stage has unclaimed critical tickets at any time, every list is exhausted at the end.
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
    int p1 = 0, q = 0, last = 0, bulk_class = 0, dg = 0;
};

struct Sim {
    int n_pad, m_tot, NP, NC, NI, W, lead_short, lead_long;
    bool pairs = false;

    FlowShape sh;
    std::mt19937 rng;
    // hand-off state
    int open_upto = 0;
    std::vector<int> cnext, lnext, tnext, pre_done, last_done, rcur, diag_done, diag_started;
    int tail_lo = 0;
    std::vector<std::vector<int>> rcnt, cver, trsm_done, trsm_started, fcnt;
    std::vector<std::vector<PipeTask>> stage;
    // workers
    struct Worker {
        int state = 0;  // 0 idle, 1 claimed (blocked), 2 running, 3 waiting for its diagonal block, 4 factoring it
        Task t;
        double end = 0;
        int my_block = -1;
        Task extra;
        bool has_extra = false;
    };
    std::vector<Worker> w;
    long claimed = 0, finished = 0;
    int fail = 0;

    Sim(int n_pad_, int m_tot_, int W_, unsigned seed, int ls, int ll)
        : n_pad(n_pad_), m_tot(m_tot_), NP(n_pad_ / 256), NC(m_tot_ / 64), NI(m_tot_ / 128), W(W_), lead_short(ls), lead_long(ll),
          sh{NP, NC, NI}, rng(seed) {
        cnext.assign(NP, 0), lnext.assign(NP, 0), tnext.assign(NP, 0), pre_done.assign(NP, 0), last_done.assign(NP, 0), rcur.assign(NP, 0);
        diag_done.assign(NP, 0), diag_started.assign(NP, 0);
        trsm_started.assign(NP, std::vector<int>(NC, 0));
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
    // ---- the claim rules of pipe_flow.h, in the two steps the device takes them: LOOK (which list has a next ticket whose task
    //      would start at once: a candidate, no side effect) and TAKE (one atomicAdd on that list's counter: the ticket obtained
    //      may be a LATER one than the ticket looked at when several workgroups looked at the same moment -- it is then unchecked)
    struct Cand {
        int kind = 0;  // 0 none, 1 head (stage s), 2 bulk-class (q, r), 3 tail (stage s, list: 0 solves, 1 last updates)
        int s = 0, q = 0, r = 0, list = 0;
    };
    static Task from(const PipeTask &pt) {
        Task t;
        t.type = pt.type, t.p = pt.p, t.a = pt.a, t.b = pt.b, t.q = pt.p + 1;
        return t;
    }
    Task bulk_task(int q, int r, int idx) const {
        const FlowBulkTask bt = flow_round_task(sh, q, r, idx);
        Task t;
        t.type = bt.type, t.p = bt.p0, t.p1 = bt.p1, t.a = bt.I, t.b = bt.type == PT_BULK ? q : bt.J, t.q = q;
        t.last = bt.last, t.bulk_class = 1, t.dg = bt.dg;
        return t;
    }
    // the look of flow_worker_loop (A): streaming tasks are taken once their producer has started, the rest when it can start
    bool head_look(const Task &t) const {
        // (what a checked claim waits for inside its task is RUNNING, never merely claimed)
        if (t.type == PT_TRSM) return blocks_done() >= t.p && (t.p == 0 || fcnt[t.p][t.a] >= 4);
        if (t.type == PT_FINE) return trsm_started[t.p][t.a] && trsm_started[t.p][4 * (t.p + 1) + t.b] && cver[t.a / 2][2 * (t.p + 1) + t.b / 2] >= t.p;
        return ready(t);
    }
    Cand look(Worker &wk, bool allow_long) {
        Cand c;
        for (;;) {
            const int S = open_upto;
            if (wk.my_block >= 0 && S >= wk.my_block) return c;  // (its own stage is open: next pass leaves)
            if (cnext[S] < (int)stage[S].size()) {
                if (head_look(from(stage[S][cnext[S]]))) {
                    c.kind = 1, c.s = S;
                    return c;
                }
                break;  // (the next ticket cannot be taken yet: nothing in the head)
            }
            if (S + 1 < NP) {
                open_upto = S + 1;  // every ticket of head S is claimed
                continue;
            }
            break;
        }
        // (B) tails of the open stages, oldest first (a window of eight stages from tail_lo)
        while (tail_lo <= open_upto && tail_lo < NP && tnext[tail_lo] >= flow_tt_size(sh, tail_lo) && lnext[tail_lo] >= flow_lt_size(sh, tail_lo)) tail_lo++;
        for (int st = tail_lo; st <= open_upto && st < NP && st < tail_lo + 8; st++) {
            if (tnext[st] < flow_tt_size(sh, st)) {
                const Task t = from(flow_tt_task(st, tnext[st]));
                if ((st == 0 || fcnt[st][t.a] >= 4) && diag_done[st]) {
                    c.kind = 3, c.s = st, c.list = 0;
                    return c;
                }
            }
            if (lnext[st] < flow_lt_size(sh, st) && ready(from(flow_lt_task(st, lnext[st])))) {
                c.kind = 3, c.s = st, c.list = 1;
                return c;
            }
        }
        if (!allow_long) return c;
        int best_key = 1 << 30;
        for (int q = 2; q < NP; q++) {
            int &r = rcur[q];
            while (r < flow_nrounds(q) && flow_round_release_stage(q, r) <= open_upto && rcnt[q][r] >= flow_round_size(sh, q, r)) r++;
            if (r >= flow_nrounds(q) || flow_round_release_stage(q, r) > open_upto) continue;
            if (!ready(bulk_task(q, r, rcnt[q][r]))) continue;
            const int key = q <= open_upto + 2 ? q : 1024 + flow_round_release_stage(q, r) * 256 + q;
            if (key < best_key) best_key = key, c.kind = 2, c.q = q, c.r = r;
        }
        return c;
    }
    bool take(Worker &wk, const Cand &c) {
        if (c.kind == 1) {
            if (cnext[c.s] >= (int)stage[c.s].size()) return false;
            wk.t = from(stage[c.s][cnext[c.s]++]);
            return true;
        }
        if (c.kind == 3) {
            if (c.list == 0) {
                if (tnext[c.s] >= flow_tt_size(sh, c.s)) return false;
                wk.t = from(flow_tt_task(c.s, tnext[c.s]++));
            } else {
                if (lnext[c.s] >= flow_lt_size(sh, c.s)) return false;
                wk.t = from(flow_lt_task(c.s, lnext[c.s]++));
            }
            return true;
        }
        if (c.kind == 2) {
            if (rcnt[c.q][c.r] >= flow_round_size(sh, c.q, c.r)) return false;
            wk.t = bulk_task(c.q, c.r, rcnt[c.q][c.r]++);
            // (a claim of two consecutive far tiles, as the device does from n_pad 8192 on: the second one unchecked)
            if (wk.t.type == PT_BULK && pairs && rcnt[c.q][c.r] < flow_round_size(sh, c.q, c.r)) wk.extra = bulk_task(c.q, c.r, rcnt[c.q][c.r]++), wk.has_extra = true;
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
            // every idle workgroup LOOKS first, then they all TAKE: the ones that looked at the same ticket get later ones, unchecked
            std::vector<Cand> cands(w.size());
            for (size_t i = 0; i < w.size(); i++) {
                Worker &wk = w[i];
                if (wk.state != 0) continue;
                const int cur = blocks_done();
                // (leaves `lead_short` blocks ahead of its own -- and, whatever the lead, once ITS stage is open: it must never hold
                //  a ticket that waits for the block only it can factor)
                if (wk.my_block >= 0 && (wk.my_block - cur <= lead_short || open_upto >= wk.my_block)) {
                    wk.state = 3;
                    progress = true;
                    continue;
                }
                cands[i] = look(wk, wk.my_block < 0 || wk.my_block - cur > lead_long);
            }
            for (size_t i = 0; i < w.size(); i++)
                if (w[i].state == 0 && cands[i].kind && !(w[i].my_block >= 0 && open_upto >= w[i].my_block) && take(w[i], cands[i]))
                    w[i].state = 1, claimed++, progress = true;
            for (auto &wk : w) {
                if (wk.state == 1 && ready(wk.t)) {
                    wk.state = 2, wk.end = now + duration(wk.t), progress = true;
                    if (wk.t.type == PT_TRSM) trsm_started[wk.t.p][wk.t.a] = 1;
                }
                if (wk.state == 3) {
                    const int p = wk.my_block;
                    bool ok = true;
                    for (int i = 0; i < 4 && p > 0; i++) ok = ok && fcnt[p][4 * p + i] >= i + 1;
                    if (ok) wk.state = 4, wk.end = now + 62.0, progress = true, diag_started[p] = 1;
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
                if (wk.state == 2 && wk.end <= now) {
                    finish(wk.t), wk.state = 0;
                    if (wk.has_extra) wk.t = wk.extra, wk.has_extra = false, wk.state = 1, claimed++;  // the second tile of a claim of two
                }
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
        for (const auto &wk : w) {
            check(wk.state == 0, "a workgroup is still blocked at the end (deadlock)", wk.state, wk.t.type, wk.t.p);
            if (wk.state != 0 && std::getenv("FLOW_TEST_VERBOSE"))
                std::printf("    blocked: state %d my_block %d task type %d p %d a %d b %d q %d | open %d tail_lo %d blocks_done %d\n", wk.state, wk.my_block, wk.t.type,
                            wk.t.p, wk.t.a, wk.t.b, wk.t.q, open_upto, tail_lo, blocks_done());
        }
        check(claimed == finished, "claimed != finished");
        for (int p = 0; p < NP; p++) {
            check(diag_done[p], "diagonal block not factored", p);
            check(cnext[p] >= (int)stage[p].size(), "head tickets left", p);
            check(tnext[p] >= flow_tt_size(sh, p) && lnext[p] >= flow_lt_size(sh, p), "tail tickets left", p);
            for (int c = 4 * p + 4; c < NC; c++) check(trsm_done[p][c], "row chunk not solved", p, c);
        }
        for (int q = 2; q < NP; q++) {
            check(rcur[q] == flow_nrounds(q) || rcnt[q][flow_nrounds(q) - 1] >= flow_round_size(sh, q, flow_nrounds(q) - 1), "rounds left", q);
            check(pre_done[q] == flow_need_pre(sh, q), "tasks of the rounds before the last near one", q, pre_done[q], flow_need_pre(sh, q));
            check(last_done[q] == flow_need_last(sh, q), "tasks of the last near round", q, last_done[q], flow_need_last(sh, q));
        }
        for (int q = 1; q < NP; q++)
            for (int J = 2 * q; J <= 2 * q + 1; J++)
                for (int I = J; I < NI; I++) {
                    const bool fine = I < flow_lt_first(q);  // the diagonal block and the window below it: the last panel came as fine tiles (counted in fcnt)
                    check(cver[I][J] == (fine ? q - 1 : q), "a tile did not receive every panel", I, J, cver[I][J]);
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
                    s.pairs = sp.n_pad >= 4096;
                    const bool ok = s.run();
                    tasks += s.finished;
                    if (!ok) {
                        std::printf("FAIL n_pad %d rhs %d workers %d seed %u leads %d: %d violations\n", sp.n_pad, sp.rhs, Wn, seed, leads, s.fail);
                        bad++;
                    }
                }
            }
    // every head list: the window's solves behind the four fine tiles of their rows, the next block's tiles behind the solves of
    // the block's rows; heads + tails solve every chunk below the block exactly once
    for (int n_pad : {1024, 4096}) {
        const int NP = n_pad / 256, NC = (n_pad + 128) / 64, NI = (n_pad + 128) / 128;
        const FlowShape sh{NP, NC, NI};
        for (int s = 0; s < NP; s++) {
            std::vector<int> fines(NC, 0), solved(NC, 0);
            for (const PipeTask &t : flow_stage_tasks(n_pad, n_pad + 128, s)) {
                if (t.type == PT_FINE && t.p == s - 1) fines[t.a]++;
                if (t.type == PT_TRSM) {
                    if (solved[t.a] || (s >= 1 && fines[t.a] != 4)) std::printf("FAIL stage %d: solve of chunk %d out of order\n", s, t.a), bad++;
                    solved[t.a]++;
                }
                if ((t.type == PT_FINE && t.p == s) || t.type == PT_COARSE)
                    for (int c = 4 * s + 4; c < 4 * s + 8; c++)
                        if (!solved[c]) std::printf("FAIL stage %d: the next block's tiles ahead of the solve of chunk %d\n", s, c), bad++;
            }
            for (int t = 0; t < flow_tt_size(sh, s); t++) solved[flow_tt_task(s, t).a]++;
            for (int c = 4 * s + 4; c < NC; c++)
                if (solved[c] != 1) std::printf("FAIL stage %d: chunk %d solved %d times\n", s, c, solved[c]), bad++;
            std::vector<int> lastc(NI, 0);
            for (int t = 0; t < flow_lt_size(sh, s); t++) lastc[flow_lt_task(s, t).a]++;
            for (int I = flow_lt_first(s); I < NI && s >= 1; I++)
                if (lastc[I] != 2) std::printf("FAIL stage %d: row tile %d has %d last updates\n", s, I, lastc[I]), bad++;
        }
    }
    std::printf("%s: %ld simulated tasks\n", bad ? "flow order test FAILED" : "flow order test ok", tasks);
    return bad ? 1 : 0;
}
