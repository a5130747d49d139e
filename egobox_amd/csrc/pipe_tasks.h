// The task list of a chain launch (kernels_pipe.hip) and its ORDER -- host code only, so that the one property the launch's
// deadlock freedom rests on can be checked without a GPU (tests/c_host/pipe_order_test.cpp): every task depends on EARLIER
// entries only.  Tickets are handed out in this order to workgroups that are resident; a task whose producers are all
// finished, running, or held by a resident workgroup that only waits for still earlier tasks can always complete.
#pragma once

#include <vector>

namespace egx {

enum { PT_TRSM = 0, PT_FINE = 1, PT_COARSE = 2, PT_DIAG = 3 };
struct PipeTask {
    int type, p, a, b;  // TRSM: a = first 64-row chunk (absolute); FINE: a = row chunk, b = column tile 0..3; COARSE: a = I, b = J (128-tiles, absolute)
};

// The task list of the group [g0, g0 + 256 np), in STAGES.  Stage s: the FINE tiles of X(s - 1 -> s) BELOW block column s's
// diagonal block (panel s - 1 into block column s: what TRSM(s) waits for, row chunk by row chunk), the TRSM tasks of panel s
// (rows of the next diagonal block first), the update of the NEXT diagonal block (its coarse tiles from panel s - 1, its ten
// fine tiles from panel s: what DIAG(s + 1) waits for), then the rest of the COARSE tiles of the updates X(p -> q), q >= p + 2,
// with max(p + 1, q - la) == s in the order of their panels.  (DIAG(s) is not in the list: the diagonal blocks have workgroups
// of their own.)  la = 0 queues
// every update just in time (left-looking order), a large la right behind its panel (right-looking order); in every such
// order a task depends on EARLIER entries only (X(p -> q) follows TRSM(p) and X(p - 1 -> q)), which is what makes the
// ticket scheme of the kernel deadlock free.
inline std::vector<PipeTask> pipe_tasks(int n_pad, int m_tot, int g0, int np, int rt, int la) {
    std::vector<PipeTask> v;
    auto width = [&](int p) { const int k0 = g0 + 256 * p; return (n_pad - k0 < 256) ? (n_pad - k0) : 256; };
    // the FINE tiles of panel p into block column p + 1 in the 64-row chunks [lo, hi) counted from that column's diagonal block
    auto fine = [&](int p, int lo, int hi) {
        const int c0 = (g0 + 256 * (p + 1)) / 64;
        for (int c = c0 + lo; c < c0 + hi && c < m_tot / 64; c++)
            for (int j = 0; j < width(p + 1) / 64; j++)
                if (c - c0 >= j) v.push_back({PT_FINE, p, c, j});  // (tiles above the diagonal block's diagonal: none)
    };
    for (int s = 0; s < np; s++) {
        // COARSE: X(p -> q), q >= p + 2, of this stage; `next_diag`: only / all but the tiles of the NEXT diagonal block
        auto coarse = [&](bool own_column, int next_diag) {
            for (int p = 0; p + 1 < np; p++)
                for (int q = p + 2; q < np; q++) {
                    const int stage = (p + 1 > q - la) ? p + 1 : q - la;
                    if (stage != s || (q == s) != own_column) continue;
                    const int cq = g0 + 256 * q, j_end = (cq + width(q)) / 128;
                    for (int J = cq / 128; J < j_end; J++)
                        for (int I = J; I < m_tot / 128; I++) {
                            const bool in_next_diag = q == s + 1 && I < j_end;
                            if (next_diag == (in_next_diag ? 0 : 1)) continue;  // (next_diag < 0: everything)
                            v.push_back({PT_COARSE, p, I, J});
                        }
                }
        };
        coarse(true, -1);  // (la == 0 only: block column s itself still has coarse updates to receive, ahead of its fine ones)
        if (s > 0) fine(s - 1, width(s) / 64, m_tot / 64);  // FINE: panel s - 1 into block column s, the rows BELOW its diagonal block
        const int r0 = g0 + 256 * s + width(s);
        for (int c = r0 / 64; c + rt <= m_tot / 64; c += rt) v.push_back({PT_TRSM, s, c, 0});
        // The NEXT diagonal block's own update -- its (up to three) coarse tiles from panel s - 1, then its ten fine tiles from panel
        // s, which run quarter by quarter beside DIAG(s) -- goes IN FRONT of the bulk of this stage's coarse tiles: what the chain waits
        // for after a block's last strip does not queue behind ~40 us x hundreds of tasks (n = 4096: 22-32 us between blocks in the
        // first panels instead of 13-20; profiles/r05_fine_quarters.txt).  Ten workers wait inside those tiles while DIAG(s) runs.
        if (s + 1 < np && la >= 1) {
            coarse(false, 1);
            fine(s, 0, width(s + 1) / 64);
            coarse(false, 0);
        } else {
            coarse(false, -1);  // (behind the chain's own tasks: nothing of this stage waits for them)
        }
    }
    return v;
}

}  // namespace egx
