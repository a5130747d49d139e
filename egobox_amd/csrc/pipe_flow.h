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
// CRITICAL work is one ticket list per STAGE s (= panel s): [last(s-1 -> s) below the block, TRSM(s), the next block's three
// near tiles (s-1 -> s+1), its ten fine tiles (s -> s+1)], claimed with one atomicAdd on the stage's counter; at most ONE
// stage has unclaimed tickets, because stage s is OPENED (a monotonic word) only when
//   G1  every ticket of stage s-1 has been claimed,
//   G2  column s has received all its rounds (so that no critical task ever waits for bulk-class data), and
//   G3  column s+1 has received every round but its last near one (whose three critical tiles belong to stage s).
// BULK-CLASS work is one ticket counter per (column, round); a round is RELEASED when the stage behind its newest source panel
// is open (then that panel's solves are all claimed), and the rounds of a column are claimed in order (round r+1 only when
// every ticket of round r is taken).  Workers take critical tickets first, then the nearest column's released round.
// DEADLOCK FREEDOM: a worker only ever blocks INSIDE a task, on words whose producers are (a) tasks already claimed by
// resident workgroups -- earlier tickets of the same stage / an earlier stage (G1), the previous round on the same tile (claim
// order), the solves of a released round's panels -- or (b) the diagonal block's own workgroup, or (c) bulk-class tasks the
// gates G2 / G3 have already seen finished.  Induction over a topological order of the tasks: the earliest unfinished task is
// claimed (or claimable by any free workgroup, and no workgroup blocks outside a task) and all its producers are finished.
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
EGX_FLOW_HD int flow_nbulk(int q) { return q / kFlowGP - 1 > 0 ? q / kFlowGP - 1 : 0; }
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
};
EGX_FLOW_HD FlowBulkTask flow_round_task(const FlowShape &sh, int q, int r, int t) {
    FlowBulkTask k;
    k.last = 0;
    if (flow_round_is_bulk(q, r)) {
        k.type = PT_BULK, k.p0 = kFlowGP * r, k.p1 = kFlowGP * r + kFlowGP, k.I = 2 * q + t, k.J = q;
        return k;
    }
    const int p = flow_first_single(q) + (r - flow_nbulk(q));
    k.type = PT_COARSE, k.p0 = p, k.p1 = p + 1;
    const int nrt = sh.NI - 2 * q;
    if (flow_round_is_last_single(q, r)) {
        k.last = 1;
        const int n0 = nrt - 2;
        k.J = t < n0 ? 2 * q : 2 * q + 1;
        k.I = 2 * q + 2 + (t < n0 ? t : t - n0);
    } else {
        k.J = t < nrt ? 2 * q : 2 * q + 1;
        k.I = t < nrt ? 2 * q + t : 2 * q + 1 + (t - nrt);
    }
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
    int off_open, off_cnext, off_rcur, off_rcnt, off_pre, off_last, off_trace, total;
};
EGX_FLOW_HD FlowLayout flow_layout(int NP, int base) {
    FlowLayout l;
    l.RMAX = NP / kFlowGP + 3;
    l.off_open = base;            // [0] highest open stage, [1] columns' scan hint (stage << 8 | column), [2] workers that have left
    l.off_cnext = base + 4;       // [NP] tickets taken per stage
    l.off_rcur = l.off_cnext + NP;     // [NP] first round of the column that is not fully claimed
    l.off_rcnt = l.off_rcur + NP;      // [NP][RMAX] tickets taken per (column, round)
    l.off_pre = l.off_rcnt + NP * l.RMAX;  // [NP] finished tasks of the rounds before the last near one
    l.off_last = l.off_pre + NP;           // [NP] finished tasks of the last near round
    l.off_trace = l.off_last + NP;         // [1] next trace slot (profiling builds)
    l.total = l.off_trace + 1 - base;
    l.total = (l.total + 63) / 64 * 64;
    return l;
}

// ---- the critical list of stage s (host: uploaded once per shape) ------------------------------
inline std::vector<PipeTask> flow_stage_tasks(int n_pad, int m_tot, int s) {
    std::vector<PipeTask> v;
    const int NP = n_pad / 256, NC = m_tot / 64, NI = m_tot / 128;
    if (s >= 1)  // last(s-1 -> s): the rows below column s's diagonal block
        for (int I = 2 * s + 2; I < NI; I++)
            for (int J = 2 * s; J <= 2 * s + 1; J++) v.push_back({PT_COARSE_LAST, s - 1, I, J});
    for (int c = 4 * s + 4; c < NC; c++) v.push_back({PT_TRSM, s, c, 0});
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
    return v;
}

}  // namespace egx
