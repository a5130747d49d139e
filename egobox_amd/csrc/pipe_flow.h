// The WHOLE factorisation of a large matrix as one persistent launch with TWO classes of work (k_potrf_flow,
// kernels_pipe.hip; round 6) -- the shape of every list and the arithmetic of every gate, host + device code, so that the
// scheduling argument can be checked without a GPU (tests/c_host/flow_order_test.cpp).
//
// What it replaces: launch_potrf's separate launches from ~7000 columns on (the panel step of LAPACK dpotrf / linfa-linalg
// `cholesky()` behind crates/gp/src/algorithm.rs:1004), where the chain kernels of the next group of panels each wait ~250 us for
// a compute unit behind the stream kernel's workgroups, and the whole-factorisation chain launch of round 5, whose every update
// was a 128 x 128 tile with K = 256 in ONE ordered ticket list (bulk in front of the chain's next tickets).
//
// PANELS are 256 columns (NP of them, n_pad a multiple of 256); column block q receives its q earlier panels as
//   FAR rounds     groups of GP = 2 panels [GP g, GP g + GP), g = 0 .. q / GP - 2: one 128 x 256 tile per 128 rows, K = 512, on
//                  the LDS-DMA ring (role BULK)                                             -- the bulk of the flops
//   NEAR rounds    the remaining one or two panels p < q - 1, one at a time: 128 x 128 tiles, K = 256 (role COARSE); of the
//                  LAST one (p = q - 2) the three tiles of the next diagonal block are CRITICAL, the rest is a round
//   the LAST panel p = q - 1: the ten 64 x 64 tiles of the diagonal block quarter by quarter beside the diagonal block of
//                  p (role FINE), the rows below as 128 x 128 tiles that also count towards the rows' readiness (COARSE, `last`)
// CRITICAL work belongs to STAGES (stage s = panel s): a HEAD -- the chain and its two-block streaming window, one ordered ticket
// list -- and two TAILS -- the last updates (LT) and the solves (TT) of all rows below the window -- see "critical work of stage
// s" below.  Head s+1 opens (a monotonic word) when every ticket of head s is claimed; tails are open with their head.
// BULK-CLASS work is one ticket counter per (column, round); a round is RELEASED when the head of the stage behind its newest
// source panel is open, and the rounds of a column are claimed in order (round r+1 only when every ticket of round r is taken).
// LOOK BEFORE YOU CLAIM: a workgroup takes a ticket only if its task would start AT ONCE (rows solved, tile up to date: a
// handful of relaxed loads) -- except the head's STREAMING tasks, which are taken once everything they wait for inside is
// RUNNING: the solves of the window (strip by strip behind the diagonal block: taken when their rows' fine tiles are in and the
// blocks before theirs are factored) and the fine tiles (quarter by quarter behind two solves: taken when both have published
// a quarter and the tile is up to date).  Workers take head tickets first, then the oldest open tail, then the two columns next
// to the chain, then the OLDEST released round of any column.
// DEADLOCK FREEDOM: a ticket claimed after a look (a CHECKED claim) never waits inside its task for anything but running
// tasks and the diagonal block's own workgroup.  The counter may hand a workgroup a LATER ticket than the one it looked at (others
// looked at the same moment): that UNCHECKED ticket may wait for a producer nobody has claimed yet -- but nothing checked ever
// queues up behind it, and the workgroup that won the race finishes its task, is free, looks alone (hence checked) and takes what
// the blocked ones wait for: at every instant some claimed task runs or a free workgroup can claim one.  A workgroup that owns a
// diagonal block never takes a head ticket of its own stage or later.  tests/c_host/flow_order_test.cpp simulates exactly
// these rules -- looks at the same instant included -- down to NP + 1 workgroups.
// (First versions of this design: one list per stage with the tail behind the head, a stage opened only when the previous one
//  was fully claimed and its column complete -- gates G1/G2/G3.  Measured: n = 4096 lost 80 us every other panel to a 135-us far
//  tile with one stage of slack; with look-before-claim the column gates went, and the chain's period was then 165 us: the rows
//  the chain needs NEXT went through a non-streaming 128 x 128 last update and a non-streaming solve, and no stage could run
//  ahead of the slowest row of the previous one.  profiles/r06_flow_*.txt)
#pragma once

#include <vector>

#include "pipe_tasks.h"

#ifdef __HIPCC__
#define EGX_FLOW_HD __host__ __device__ __forceinline__
#else
#define EGX_FLOW_HD inline
#endif

namespace egx {

enum { PT_COARSE_LAST = 4, PT_BULK = 5 };  // (PT_TRSM .. PT_DIAG: pipe_tasks.h)
constexpr int kFlowGP = 2;                 // panels per far round (K = 512)

struct FlowShape {
    int NP, NC, NI;  // panels, 64-row chunks, 128-row tiles (m_tot / 128)
};

// ---- column q: its rounds ---------------------------------------------------------------------
// (far rounds end TWO to three panels before the column's last near one: the newest far tile of the next diagonal block -- 135 us,
//  behind a non-streaming solve -- then has two stages of slack; with one, n = 4096 stalled 50 us every other panel.  A near round
//  costs 76 us per 128 x 256 of a panel against the far round's 65: moving one panel per column over is < 1 % of the work.)
EGX_FLOW_HD int flow_nbulk(int q) { return (q - 1) / kFlowGP - 1 > 0 ? (q - 1) / kFlowGP - 1 : 0; }
EGX_FLOW_HD int flow_first_single(int q) { return kFlowGP * flow_nbulk(q); }                     // B(q)
EGX_FLOW_HD int flow_nsingles(int q) { return q >= 2 ? q - 1 - flow_first_single(q) : 0; }       // panels [B(q), q - 1)
EGX_FLOW_HD int flow_nrounds(int q) { return flow_nbulk(q) + flow_nsingles(q); }
EGX_FLOW_HD bool flow_round_is_bulk(int q, int r) { return r < flow_nbulk(q); }
// the newest source panel of round r, + 1: the round is released once that STAGE is open
EGX_FLOW_HD int flow_round_release_stage(int q, int r) {
    return flow_round_is_bulk(q, r) ? kFlowGP * r + kFlowGP : flow_first_single(q) + (r - flow_nbulk(q)) + 1;
}
EGX_FLOW_HD bool flow_round_is_last_single(int q, int r) { return !flow_round_is_bulk(q, r) && r == flow_nrounds(q) - 1; }
// tickets of round r: far = one per 128-row tile from the column's diagonal block down; near = the lower 128-tiles of both
// 128-column halves, without the three tiles of the diagonal block when it is the last near round
EGX_FLOW_HD int flow_round_size(const FlowShape &sh, int q, int r) {
    const int nrt = sh.NI - 2 * q;
    if (flow_round_is_bulk(q, r)) return nrt;
    return flow_round_is_last_single(q, r) ? 2 * (nrt - 2) : 2 * nrt - 1;
}
struct FlowBulkTask {
    int type;    // PT_BULK or PT_COARSE
    int p0, p1;  // source panels [p0, p1)
    int I, J;    // PT_BULK: 128-row tile I, column block q (J unused = q); PT_COARSE: 128-tiles (I, J)
    int last;    // counts towards last_done (the last near round) instead of pre_done
    int dg;      // a tile of the column's DIAGONAL block (rows 2 q, 2 q + 1): counts towards dpre_done as well (gate G3)
};
// (the tiles of the diagonal block are the FIRST tickets of every round: the chain needs them first)
EGX_FLOW_HD FlowBulkTask flow_round_task(const FlowShape &sh, int q, int r, int t) {
    FlowBulkTask k;
    k.last = 0, k.dg = 0;
    if (flow_round_is_bulk(q, r)) {
        k.type = PT_BULK, k.p0 = kFlowGP * r, k.p1 = kFlowGP * r + kFlowGP, k.I = 2 * q + t, k.J = q;
        k.dg = t < 2;
        return k;
    }
    const int p = flow_first_single(q) + (r - flow_nbulk(q));
    k.type = PT_COARSE, k.p0 = p, k.p1 = p + 1;
    const int n0 = sh.NI - 2 * q - 2;  // 128-row tiles below the diagonal block
    if (flow_round_is_last_single(q, r)) {
        k.last = 1;
    } else if (t < 3) {
        k.dg = 1;
        k.I = 2 * q + (t > 0), k.J = 2 * q + (t > 1);
        return k;
    } else {
        t -= 3;
    }
    k.J = t < n0 ? 2 * q : 2 * q + 1;
    k.I = 2 * q + 2 + (t < n0 ? t : t - n0);
    return k;
}
// completions the gates wait for
EGX_FLOW_HD int flow_need_pre(const FlowShape &sh, int q) {
    int n = 0;
    for (int r = 0; r < flow_nrounds(q); r++)
        if (!flow_round_is_last_single(q, r)) n += flow_round_size(sh, q, r);
    return n;
}
EGX_FLOW_HD int flow_need_last(const FlowShape &sh, int q) { return q >= 2 ? 2 * (sh.NI - 2 * q - 2) : 0; }

// ---- hand-off words of the flow launch (ints, behind the chain launch's: PipeLayout) -------------
struct FlowLayout {
    int RMAX;  // round counters per column
    int off_open, off_cnext, off_lnext, off_tnext, off_rcur, off_rcnt, off_trace, total;
};
EGX_FLOW_HD FlowLayout flow_layout(int NP, int base) {
    FlowLayout l;
    l.RMAX = NP / kFlowGP + 3;
    l.off_open = base;            // [0] highest stage whose HEAD is open, [1] columns' scan hint (stage << 8 | column), [2] oldest stage with tail tickets left
    l.off_cnext = base + 4;       // [NP] head tickets taken per stage
    l.off_lnext = l.off_cnext + NP;    // [NP] LT tickets taken per stage
    l.off_tnext = l.off_lnext + NP;    // [NP] TT tickets taken per stage
    l.off_rcur = l.off_tnext + NP;     // [NP] first round of the column that is not fully claimed
    l.off_rcnt = l.off_rcur + NP;      // [NP][RMAX] tickets taken per (column, round)
    l.off_trace = l.off_rcnt + NP * l.RMAX;  // [1] next trace slot (profiling builds)
    l.total = l.off_trace + 1 - base;
    l.total = (l.total + 63) / 64 * 64;
    return l;
}

// ---- critical work of stage s ----------------------------------------------------------------------
// HEAD (host-built list, uploaded once per shape): the rows of the next THREE diagonal blocks -- the chain's window -- are
// handled streaming: their last update (s-1 -> s) as fine tiles quarter by quarter behind the solves of panel s-1, their
// solves strip by strip behind diagonal block s.  In ticket order:
//   the window's first four row chunks (block s+1):   16 fine tiles (s-1 -> s), 4 solves (s)
//   the next diagonal block itself:                    3 near tiles (s-1 -> s+1), 10 fine tiles (s -> s+1)
//   the window's other row chunks (blocks s+2, s+3):   16 fine tiles (s-1 -> s), 4 solves (s) per block
// so that a row block is solved for panel s right behind block s's last strip for two stages before the chain needs it, and
// the chain's period is the diagonal block + one strip of solve + one quarter of fine tile (~90 us), not two non-streaming
// hops (165 us measured with a one-block window and 128 x 128 last updates).
// TAIL (arithmetic, no list): every row below the window -- LT(s): the last update (s-1 -> s) as 128 x 128 tiles (I >= flow_lt_first(s),
// both halves of the block column), TT(s): the solves (chunks >= 4 s + 4 + kFlowWindow) -- each with a ticket counter of its own.  Tails of
// several stages may be open at once (oldest first); the head of stage s+1 opens when every ticket of head s is claimed.
constexpr int kFlowWindow = 12;  // row chunks (64 rows) of the streaming window below a diagonal block: three blocks
inline std::vector<PipeTask> flow_stage_tasks(int n_pad, int m_tot, int s) {
    std::vector<PipeTask> v;
    const int NP = n_pad / 256, NC = m_tot / 64;
    auto window_rows = [&](int c_lo) {
        for (int c = c_lo; c < c_lo + 4 && c < NC && s >= 1; c++)
            for (int j = 0; j < 4; j++) v.push_back({PT_FINE, s - 1, c, j});
        for (int c = c_lo; c < c_lo + 4 && c < NC; c++) v.push_back({PT_TRSM, s, c, 0});
    };
    window_rows(4 * s + 4);
    if (s + 1 < NP) {
        if (s >= 1) {  // the next diagonal block's three near tiles
            v.push_back({PT_COARSE, s - 1, 2 * s + 2, 2 * s + 2});
            v.push_back({PT_COARSE, s - 1, 2 * s + 3, 2 * s + 2});
            v.push_back({PT_COARSE, s - 1, 2 * s + 3, 2 * s + 3});
        }
        const int c0 = 4 * (s + 1);
        for (int c = c0; c < c0 + 4; c++)
            for (int j = 0; j <= c - c0; j++) v.push_back({PT_FINE, s, c, j});
    }
    for (int c = 4 * s + 8; c < 4 * s + 4 + kFlowWindow; c += 4) window_rows(c);
    return v;
}
// tails: tickets of LT(s) / TT(s) and what ticket t is
EGX_FLOW_HD int flow_lt_first(int s) { return 2 * s + 2 + kFlowWindow / 2; }  // first 128-row tile below the window
EGX_FLOW_HD int flow_lt_size(const FlowShape &sh, int s) { return s >= 1 && sh.NI > flow_lt_first(s) ? 2 * (sh.NI - flow_lt_first(s)) : 0; }
EGX_FLOW_HD int flow_tt_size(const FlowShape &sh, int s) { return sh.NC > 4 * s + 4 + kFlowWindow ? sh.NC - (4 * s + 4 + kFlowWindow) : 0; }
EGX_FLOW_HD PipeTask flow_lt_task(int s, int t) { return PipeTask{PT_COARSE_LAST, s - 1, flow_lt_first(s) + (t >> 1), 2 * s + (t & 1)}; }
EGX_FLOW_HD PipeTask flow_tt_task(int s, int t) { return PipeTask{PT_TRSM, s, 4 * s + 4 + kFlowWindow + t, 0}; }

}  // namespace egx
