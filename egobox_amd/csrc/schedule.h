// How a handle factors: the ONE place where the schedule of launch_potrf is decided (host-testable: no HIP in here;
// tests/c_host/schedule_test.cpp walks the table).  Decided once per handle shape -- egx_gp_create, egx_gp_set_lockstep --
// and stored in the handle (egx_gp::sched), never per launch, never by the number of matrices in a launch, and not by
// egx_gp_shrink: all evaluations of a handle agree bit for bit (egx_gp_get_schedule shows the decision).
//
//   n_pad (padded size)   lock-step width   workspaces x panels     left   w_left   pipe   whole   panels per group
//   <= 7168               any               <= 32                   0      0        1      1       (one launch)
//   <= 4096               any               >  32                   0      0        1      0       2
//   4097 .. 14335         any               >  32 or n_pad > 7168   0      0        0      0       2
//   >= 14336              < 4               -                       0      0        0      0       4
//   >= 14336              4 .. 7            -                       0      1        0      0       4
//   >= 14336              >= 8              -                       1      1        0      0       4
//   5376 .. 14080         1                 1 workspace, no group   the whole factorisation is ONE FLOW launch (flow = 1; pipe_flow.h)
//   >= 14336              1                 1 workspace, no group   right-looking by separate launches, the last 6144 .. 7167 columns as a flow launch (flow_tail)
//
// left / w_left  left-looking group updates of the factorisation / of the theta-gradient's C^-T rider (EGX_POTRF_LEFT: 0 never,
//                1 by the table, 2 always).  Measured: n = 16384 in lock-step groups of eight +2.3 % on the sweep, a lone
//                matrix 2.4x slower, n = 8192 -14 % (profiles/r04_run4_*, r04_run5_*, r04_run12_*)
// pipe           the serial chain of a group of panels -- diagonal blocks, panel solves, in-group updates -- is ONE persistent
//                launch with device-side hand-offs (kernels_pipe.hip; EGX_PIPE=0: separate launches everywhere).  Up to
//                kPipeMaxCols (4096) columns: beyond, a group's panel solves and updates are chip-filling launches of their
//                own right and the chain launch loses against them (n = 8192: 6.4 against 6.0 ms, profiles/r05_pipe_*)
// whole          ... and the whole factorisation is one such launch (every update inside it; EGX_PIPE=2: never); every
//                diagonal block of a chain launch has a workgroup of its own from the start, hence the bound of
//                kPipeDiagBlocks (32) on workspaces x panels.  Up to kPipeWholeMaxCols (7168) columns: one matrix at
//                n = 4608 / 5120 / 6144 / 7168 / 7680 / 8192: 1.63 / 1.98 / 2.90 / 4.24 / 5.32 / 6.5 ms against 2.32 / 2.72 / 3.52 /
//                4.64 / 5.35 / 5.9 ms of separate launches (the in-launch 128 x 128 update tasks are slower than the stream
//                kernel, and from ~8000 columns on the updates are what a factorisation is: profiles/r05_whole_launch_size_ab.txt)
// group_panels   panels per trailing update (EGX_POTRF_GROUP; profiles/r02_run13_*: four pay from n ~ 14000 on)
// (Round 5 also measured (a) look-ahead further down the matrix for lock-step widths >= 4 -- 1024 instead of 3072 trailing
//  columns: n = 4096 in lock-step 12 unchanged, 7.9 ms of launches per batch either way, because a chain launch that shares the
//  chip runs 3x longer than alone, profiles/r05_look_by_width_discarded.txt -- and (b) a chain launch for the LAST columns of a large right-looking factorisation behind one update of the
//  whole trailing matrix: n = 16384 30.2 against 28.9 ms, n = 8192 6.8 against 5.9 -- profiles/r05_pipe_check_tail_*.txt; gone.)
#pragma once

namespace egx {

constexpr int kFlowMinCols = 5120;       // a LONE handle (one workspace, not a member of a group) beyond this padded size ...
constexpr int kFlowMaxCols = 14336;      // ... and below this one factors as ONE flow launch (pipe_flow.h, round 6)
constexpr int kFlowTailCols = 6144;      // from kFlowMaxCols on: (at least) this many last columns of the right-looking factorisation as a flow launch
constexpr int kPipeMaxCols = 4096;       // padded size up to which the chain of a group of panels is one launch
constexpr int kPipeWholeMaxCols = 7168;  // ... up to which the WHOLE factorisation is one launch,
constexpr int kPipeDiagBlocks = 32;      // while workspaces x panels stays within this

struct PotrfSchedule {
    int left = 0, w_left = 0, pipe = 0, whole = 0, group_panels = 2;
    int flow = 0;  // the whole factorisation of ONE matrix per launch as a flow launch (pipe_flow.h, round 6)
    int flow_tail = 0;  // columns at the END of a larger lone matrix' right-looking factorisation that are one flow launch (0: none)
};
struct ScheduleKnobs {  // egx_set_tuning / environment: "potrf_left", "pipe", "potrf_group"
    int potrf_left = 1, pipe = 1, potrf_group = 0;
};
#ifdef EGX_DEV_KNOBS  // (scratch builds for A/B measurements only: tools/dev_build.sh; never in the product library)
}  // namespace egx
#include <cstdlib>
namespace egx {
inline int dev_env(const char *name, int dflt) {
    const char *e = std::getenv(name);
    return e ? std::atoi(e) : dflt;
}
#define EGX_PIPE_MAX_COLS dev_env("EGX_DEV_PIPE_MAX", kPipeMaxCols)
#define EGX_PIPE_WHOLE_MAX_COLS dev_env("EGX_DEV_PIPE_WHOLE_MAX", kPipeWholeMaxCols)
#else
#define EGX_PIPE_MAX_COLS kPipeMaxCols
#define EGX_PIPE_WHOLE_MAX_COLS kPipeWholeMaxCols
#endif
inline PotrfSchedule schedule_table(int n_pad, int lockstep, int n_workspaces, const ScheduleKnobs &k) {
    PotrfSchedule s;
    s.left = k.potrf_left >= 2 || (k.potrf_left == 1 && n_pad >= 14336 && lockstep >= 8);
    s.w_left = n_pad % 256 == 0 && (k.potrf_left >= 2 || (k.potrf_left == 1 && n_pad >= 14336 && lockstep >= 4));
    s.whole = k.pipe == 1 && n_pad <= EGX_PIPE_WHOLE_MAX_COLS && (long long)n_workspaces * ((n_pad + 255) / 256) <= kPipeDiagBlocks;
    s.pipe = k.pipe != 0 && (n_pad <= EGX_PIPE_MAX_COLS || s.whole);
    s.group_panels = k.potrf_group ? k.potrf_group : (n_pad >= 14336 ? 4 : 2);
    // flow: measured on one matrix (tools/pipe_check flow, profiles/r06_flow_sizes.txt; ms, separate launches / whole chain launch /
    // flow): n = 4096 1.95 / 1.37 / 1.47, 5120 2.89 / 1.94 / 1.92, 6144 3.97 / 2.77 / 2.62, 7168 5.21 / 4.16 / 3.57,
    // 8192 5.85 / 6.45 / 4.93, 10240 9.55 / - / 8.57, 12288 14.67 / - / 13.68, 14336 20.72 / - / 21.02, 16384 28.92 / - / 30.38.
    // One matrix per launch: handles with several workspaces and the members of a group (egx_gp_create_group clears the bit) keep
    // the rows above -- `whole` / `pipe` stay set as what such a handle falls back to (a retry, a device the launch does not fit).
    s.flow = k.pipe == 1 && n_workspaces == 1 && lockstep <= 1 && n_pad % 256 == 0 && n_pad > kFlowMinCols && n_pad < kFlowMaxCols;
    // ... and from kFlowMaxCols on such a handle factors its first columns right-looking by separate launches -- where the trailing
    // updates are chip-filling 128 x 256 stream launches with K = 1024 at 0.83 of peak -- and its LAST kFlowTailCols columns, where
    // the chain is what the separate launches wait for, as one flow launch on the fully updated trailing matrix
    // (profiles/r06_flow_tail_ab.txt)
    // (n = 16384, ms: separate launches 28.89; the last 4096 / 6144 / 8192 / 10240 / 12288 columns as a flow launch 28.49 / 27.71 / 27.88 /
    //  27.66 / 28.18; n = 20480: 52.23 against 50.86 / 51.30 with 8192 / 12288.  The switch sits on a boundary of the four-panel
    //  groups, the tail is the 6144 .. 7167 columns that leaves.)
    s.flow_tail = (k.pipe == 1 && n_workspaces == 1 && lockstep <= 1 && n_pad % 256 == 0 && n_pad >= kFlowMaxCols && !s.left)
                      ? n_pad - 1024 * ((n_pad - kFlowTailCols) / 1024)
                      : 0;
    return s;
}

}  // namespace egx
