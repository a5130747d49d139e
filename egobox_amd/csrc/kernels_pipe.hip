// The serial chain of a group of Cholesky panels as ONE persistent launch with device-side hand-offs (gfx950 / MI355X).
//
// What it replaces: launch_potrf's chain -- per 256-column panel a diagonal-block launch (k_potf2_reg, one workgroup,
// ~57 us), a panel solve launch (k_panel_trsm16, ~20 us) and one or two update launches of the group's own later columns --
// i.e. the panel step of LAPACK dpotrf / linfa-linalg `cholesky()` behind crates/gp/src/algorithm.rs:1004, which is what a
// LONE factorisation waits for (n = 4096: 16 x (57 + 20 + 15..80) us of launches in a row for 0.3 ms worth of flops).
// Here the panel solve consumes the diagonal block STRIP BY STRIP while it is being factored, the update of the next
// panel's columns starts row chunk by row chunk as the solve finishes them, and the next diagonal block starts as soon as
// its own ten tiles are up to date: no launch boundary, no cross-stream event on the chain.
//
// ROLES.  Every workgroup has 1024 threads.  It first takes a START ticket (one device-scope atomicAdd per launch):
//   the first np * nz      DIAG workgroups, one per (diagonal block of the group, matrix of the lock-step batch): each factors
//   workgroups to start    ITS block with rb_factor_block<16, PIPE> (potf2_blocks.h, inlined at the top of the kernel: no
//                          call, no spill) as soon as the block's ten tiles are up to date, publishing every finished
//                          16-column strip -- by order of START, not by blockIdx: whatever the dispatcher places first is
//                          resident, so launches that share the chip cannot starve each other's diagonal blocks
//   everybody (the DIAG    workers: TASK ticket -> (task, matrix) from a host-built list in an order in which every task
//   workgroups afterwards) depends on EARLIER tasks only; a worker that finishes takes the next ticket
//     TRSM(p, rows)        64 * RT rows of the panel below diagonal block p: the register-resident 16-row-tile solve of
//                          k_panel_trsm16, driven by the strips as they are published (two workgroup barriers per strip,
//                          the fragments of the next strip prefetched whenever that strip is already there)
//     FINE(p, i, j)        64 x 64 tile of block column p + 1 (the NEXT panel: on the chain) -= P_p P_p^T, QUARTER BY QUARTER of the
//                          panel's 256 columns as the TRSM tasks of its two row chunks publish them: sixteen waves x one
//                          16 x 16 sub-tile, the quarter's operands through LDS, ascending k, no reduction across waves
//     COARSE(p, I, J)      128 x 128 tile of the group's columns beyond the next panel -= P_p P_p^T, K = 256
// Because tickets are taken in order by workgroups that are RESIDENT, every task's producers are finished, running, or
// held by a resident workgroup that only waits for still earlier tasks: no dispatch-order or placement assumption, no
// deadlock.  Every wait is bounded (EGX_PIPE_TIMEOUT_MS): a waiter whose time runs out raises the launch's abort word
// and returns, everybody else sees the word in its own wait or at its next ticket, the launch drains and the host reports
// EGX_ERR_HIP (tests/test_gpu_pipe.py forces this path).
//
// HAND-OFFS (MI355X guide, "agent-scope release/acquire"; all words zeroed once per factorisation, counting upwards):
//   strips[z]            DIAG -> TRSM: 16-column strips published (absolute); payload = the strip's column of the factor,
//                        its diagonal tile, the tile's inverse + refinement flag: write-through (sc1) stores, drained by
//                        every storing wave in front of a workgroup barrier, then one relaxed agent-scope flag store;
//                        the consumer reads them with agent-scope loads (they bypass its L1)
//   rowT[P][c]           TRSM -> FINE / COARSE: how many QUARTERS (four strips = 64 columns) of panel P are solved for the 64-row
//                        chunk c (sc1 stores + drain + flag per quarter); COARSE waits for all of them
//   fcnt[Q][c]           FINE -> TRSM / DIAG: fine tiles of block column Q, row chunk c, that are up to date
//   cver[I][J]           COARSE -> COARSE / FINE: updates of this launch applied to the 128 x 128 tile (I, J)
// A task first waits (ONE lane polls, relaxed loads + s_sleep), then ONE agent-scope acquire drops the stale lines of its
// CU's L1, a workgroup barrier, then plain loads.
//
// ARITHMETIC.  DIAG and TRSM do exactly what k_potf2_reg / k_panel_trsm16 do (same instructions on the same values: a
// group of ONE panel gives the same bits as the separate launches, tools/pipe_check.hip); FINE / COARSE add their K = 256
// products in an order of their own (fixed: a matrix gets the same bits alone and in a lock-step batch, on any grid size).
#include "egx_internal.h"
#include "mfma_gemm_core.h"
#include "potf2_blocks.h"
#include "pipe_tasks.h"
#include "pipe_flow.h"

#define EGX_RC_PIPE(call)     \
    do {                      \
        int _rc = (call);     \
        if (_rc) return _rc;  \
    } while (0)

#include <atomic>
#include <climits>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace egx {


// sync words of one matrix (ints): [0] abort (the launch's: matrix 0's), [1] strips published, [2] scratch (stalled strips of
// the timeout test), [4..7] diagnostics of the first wait that ran out, [8 + P] task ticket and [8 + NP + P] start ticket of the
// launch whose group starts at panel P (matrix 0's), then rowT, fcnt, cver
constexpr int kPipeHdr = 8;
struct PipeLayout {
    int NP, NC, NJ;  // panels, 64-row chunks, 128-column tiles
    int off_rowT, off_fcnt, off_cver, total;
};
static PipeLayout pipe_layout(int n_pad, int m_tot) {
    PipeLayout l;
    l.NP = (n_pad + 255) / 256;
    l.NC = m_tot / 64;
    l.NJ = n_pad / 128;
    l.off_rowT = kPipeHdr + 2 * l.NP;  // (behind the launches' task tickets and start tickets)
    l.off_fcnt = l.off_rowT + l.NP * l.NC;
    l.off_cver = l.off_fcnt + (l.NP + 1) * l.NC;
    l.total = l.off_cver + (m_tot / 128) * l.NJ;
    l.total = (l.total + 63) / 64 * 64;
    return l;
}
size_t pipe_sync_ints(int n_pad, int m_tot) {
    const PipeLayout l = pipe_layout(n_pad, m_tot);
    return (size_t)l.total + (size_t)flow_layout(l.NP, l.total).total;  // + the flow launch's words (pipe_flow.h)
}

struct PipeArgs {
    double *M;
    int64_t ld;
    int n_pad, m_tot;
    double *dinv;
    int *info;
    int *sync;
    int64_t sM, sD, sS;
    int sI, nz;
    int g0, np;
    const PipeTask *tasks;
    int ntasks;
    int NP, NC, NJ, off_rowT, off_fcnt, off_cver;
    int rt;             // 64-row chunks per TRSM task (1 or 2)
    int ext_need;       // != 0: the solve tasks of the launch's FIRST panel wait until word 3 (matrix 0's) reaches this value: the
                        // rest of the group's columns has been updated by a launch that runs beside this one (pipe_signal)
    int stall;          // test hook: 1 + the strip (absolute) from which the DIAG role publishes into the scratch word
    long long timeout;  // wall_clock64 ticks (100 MHz)
    long long *trace;   // profiling (tools/pipe_check): 8 words per ticket -- task, times taken / ready / done (100 MHz), XCD
};

// The roles are separate (non-inlined) functions -- each gets a register allocation of its own under the kernel's 128-VGPR
// budget (inlined into one body the diagonal role alone filled it and everything spilled) -- that read the launch arguments
// from the kernel-argument segment (scalar loads through the pointer the kernel hands them) and get the LDS base as an LDS pointer.
typedef const __attribute__((address_space(4))) PipeArgs *pipe_kargs_t;
typedef __attribute__((address_space(3))) double *pipe_lds_t;
// Function arguments arrive in VECTOR registers: whatever is computed from them counts as divergent and lives in vector
// registers too (the diagonal role lost ~20 of its 128 that way and kept two accumulator tiles in scratch).  The roles'
// arguments are wave-uniform by construction, so they are moved to scalar registers on entry.
__device__ __forceinline__ int pipe_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T>
__device__ __forceinline__ T *pipe_uniform(T *p) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return (T *)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double *pipe_uniform_lds(pipe_lds_t p) {  // (an LDS address is 32 bits)
    const unsigned long long u = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)p);
    return (double *)(pipe_lds_t)u;
}
__device__ __forceinline__ PipeArgs pipe_kargs(pipe_kargs_t kv) {  // (kv: the kernel's own kernarg pointer -- inside a
    PipeArgs a;                                                   //  called function the intrinsic yields a null pointer)
    const unsigned long long u = (unsigned long long)kv;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    pipe_kargs_t k = (pipe_kargs_t)(((unsigned long long)hi << 32) | lo);
    a.M = k->M, a.ld = k->ld, a.n_pad = k->n_pad, a.m_tot = k->m_tot, a.dinv = k->dinv, a.info = k->info, a.sync = k->sync;
    a.sM = k->sM, a.sD = k->sD, a.sS = k->sS, a.sI = k->sI, a.nz = k->nz, a.g0 = k->g0, a.np = k->np, a.tasks = k->tasks;
    a.ntasks = k->ntasks, a.NP = k->NP, a.NC = k->NC, a.NJ = k->NJ, a.off_rowT = k->off_rowT, a.off_fcnt = k->off_fcnt, a.off_cver = k->off_cver;
    a.rt = k->rt, a.ext_need = k->ext_need, a.stall = k->stall, a.timeout = k->timeout, a.trace = k->trace;
    return a;
}

constexpr int kFineStage2 = 2 * GemmShape<64, 64, 32, 32, 256>::STAGE;  // (a quarter of the launch's LDS: the unit the roles' areas are sized in)
constexpr int kPipeLdsDoubles = 4 * kFineStage2;                         // 147 456 B: the COARSE role's two stages
constexpr int kPipeLdsBytes = kPipeLdsDoubles * 8 + 64;                  // + the control words
static_assert(kPipeLdsDoubles * 8 >= RB_LDS_BYTES && kPipeLdsDoubles >= 2 * GemmShape<128, 128, 32, 32, 1024>::STAGE,
              "LDS of the other roles");

// ONE lane waits until *flag >= need.  0: there; 1: this matrix lost a pivot (nothing left to compute for it); 2: the launch
// is aborted (somebody's wait ran out, maybe this one's)
__device__ __forceinline__ int pipe_wait_ge(const int *flag, int need, int *abortp, const int *infop, long long limit,
                                            int who = 0) {
    if (load_flag(flag) >= need) return 0;
    const long long t0 = wall_clock64();
    for (unsigned spins = 1;; spins++) {
        __builtin_amdgcn_s_sleep(2);
        if (load_flag(flag) >= need) return 0;
        if ((spins & 7) == 0) {
            if (load_flag(abortp) != 0) return 2;
            if (load_flag(infop) != 0) return 1;
            if (wall_clock64() - t0 > limit) {
                // (diagnostics behind the abort word: who gave up, on which word, what it wanted and what it saw)
                if (__hip_atomic_fetch_add(abortp, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                    store_flag(abortp + 4, who);
                    store_flag(abortp + 5, (int)(flag - abortp));
                    store_flag(abortp + 6, need);
                    store_flag(abortp + 7, load_flag(flag));
                }
                return 2;
            }
        }
    }
}

// the outcome of thread 0's waits, for the whole workgroup (two barriers; on success thread 0 has executed the agent-scope
// acquire that lets the workgroup read, with plain loads, what the producers it waited for have published)
// (ACQUIRE = false: the caller reads what it waited for with agent-scope loads only)
template <bool ACQUIRE = true, typename F>
__device__ __forceinline__ int pipe_wg_wait(int *s_ctl, F &&waits) {
    if (threadIdx.x == 0) {
        const int r = waits();
        if (ACQUIRE && r == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        s_ctl[0] = r;
    }
    __syncthreads();
    const int r = s_ctl[0];
    __syncthreads();
    return r;
}

// ---------------------------------------------------------------------------------------------
// TRSM: 64 RT rows of the panel below diagonal block p.  Wave w: row-tile group w / 4 (RT tiles of 16 rows), strips
// C = w % 4 (mod 4) of those tiles as FP64-MFMA accumulators (k_panel_trsm16's layout and arithmetic).  What a strip
// needs from the DIAG role -- its column of the factor below and including the diagonal tile, the tile's inverse and
// refinement flag -- is STAGED through LDS by the whole workgroup (one agent-scope 32-byte piece per thread; two buffers:
// strip k + 1 is fetched, whenever it is already published, while strip k is being applied).
// ---------------------------------------------------------------------------------------------
constexpr int kTrsmXDoubles = 2 * 4 * 2 * 16 * RB_LD;  // X_k of 4 row-tile groups x (up to) 2 tiles, two buffers
constexpr int kTrsmLBuf = 256 * RB_LD + 16 * RB_LD + 8;  // strip column (row r of the block at r * RB_LD), Linv_k, flag
static_assert(kTrsmXDoubles + 2 * kTrsmLBuf <= kFineStage2 * 4, "LDS of the TRSM role");

template <int RT>
__device__ __noinline__ int pipe_role_trsm(pipe_kargs_t ka, pipe_lds_t sm3, long long *tr, int z, int p, int c0) {
    const PipeArgs a = pipe_kargs(ka);
    z = pipe_uniform(z), p = pipe_uniform(p), c0 = pipe_uniform(c0);
#ifdef EGX_PIPE_TRACE
    tr = pipe_uniform(tr);
#else
    tr = nullptr;
#endif
    double *sm = pipe_uniform_lds(sm3);
    int *s_ctl = reinterpret_cast<int *>(sm + kPipeLdsDoubles);
    const int k0 = a.g0 + 256 * p, P = k0 >> 8, base = k0 >> 4;
    const int nbk = (a.n_pad - k0 < 256) ? (a.n_pad - k0) : 256, nb16 = nbk >> 4;
    double *Mz = a.M + (int64_t)z * a.sM;
    const double *lin = a.dinv + (int64_t)z * a.sD + (int64_t)(k0 / 64) * 4096;
    const int *info = a.info + (int64_t)z * a.sI;
    int *S = a.sync + (int64_t)z * a.sS;
    const int tid = threadIdx.x, lane = tid & 63, frow = lane & 15, fk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, sw = wave & 3;
    const int prow = ((frow & 3) << 2) | (frow >> 2);  // pi(frow), see rb_factor_block
    // the rows' columns of this panel have received the previous panel (fine tiles), and strip 0 is there
    if (tid == 0) {
        int r = load_flag(info) != 0 ? 1 : 0;
        if (p == 0 && a.ext_need && r == 0) r = pipe_wait_ge(a.sync + 3, a.ext_need, a.sync, info, a.timeout, (9 << 28) | (c0 & 0xfffff));
        if (p > 0)
            for (int c = 0; c < RT && r == 0; c++)
                r = pipe_wait_ge(S + a.off_fcnt + P * a.NC + c0 + c, nbk / 64, a.sync, info, a.timeout, (1 << 28) | ((p) << 20) | ((c0) & 0xfffff));
        if (r == 0) r = pipe_wait_ge(S + 1, base + 1, a.sync, info, a.timeout, (2 << 28) | ((p) << 20) | ((c0) & 0xfffff));
        if (r == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        s_ctl[0] = r;
        s_ctl[1] = load_flag(S + 1);
    }
    __syncthreads();
    {
        const int r = s_ctl[0];
        if (r) {
            __syncthreads();
            return r;
        }
    }
    int avail = s_ctl[1];  // strips published when thread 0 last looked (>= base + 1)
    if (tr && threadIdx.x == 0) tr[3] = wall_clock64();
    double *Pw = Mz + (int64_t)(64 * c0 + grp * RT * 16 + frow) * a.ld + k0 + 4 * fk;  // + 16 r ld + 16 C: row tile r, strip C
    double acc[RT][16];  // row tile r, slot t <-> strip C = 4 t + sw
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int C = 4 * t + sw, Cc = C < nb16 ? C : 0;
#pragma unroll
        for (int r = 0; r < RT; r++) {
            const double *src = Pw + (int64_t)(16 * r) * a.ld + 16 * Cc;
            const d2_t v0 = *reinterpret_cast<const d2_t *>(src), v1 = *reinterpret_cast<const d2_t *>(src + 2);
            acc[r][4 * t] = v0[0];
            acc[r][4 * t + 1] = v0[1];
            acc[r][4 * t + 2] = v1[0];
            acc[r][4 * t + 3] = v1[1];
        }
    }
    double *Xs = sm, *Ls = sm + kTrsmXDoubles;
    // staging of strip k: thread -> row tid / 4 of the diagonal block, doubles [4 (tid % 4), + 4) of the strip's 16 columns
    // (rows above the strip's diagonal tile are not needed); threads 0..63 also carry the tile's inverse, thread 64 the flag
    const int srow = tid >> 2, spart = (tid & 3) * 4;
    const double *Lcol = Mz + (int64_t)(k0 + srow) * a.ld + k0 + spart;  // + 16 k
    d2_t sv0, sv1, si0, si1;
    double sflag = 0.0;
    auto stage_load = [&](int k) {
        if (srow >= 16 * k && srow < nbk) {
            sv0 = load_d2_sc1(Lcol + 16 * k);
            sv1 = load_d2_sc1(Lcol + 16 * k + 2);
        }
        const double *lk = lin + (int64_t)(k >> 2) * 4096 + (k & 3) * 256;
        if (tid < 64) {
            si0 = load_d2_sc1(lk + tid * 4);
            si1 = load_d2_sc1(lk + tid * 4 + 2);
        }
        if (tid == 64) sflag = load_sc1(lin + (int64_t)(k >> 2) * 4096 + 1024 + (k & 3));
    };
    auto stage_store = [&](int k) {
        double *Lb = Ls + (k & 1) * kTrsmLBuf;
        if (srow >= 16 * k && srow < nbk) {
            *reinterpret_cast<d2_t *>(Lb + srow * RB_LD + spart) = sv0;
            *reinterpret_cast<d2_t *>(Lb + srow * RB_LD + spart + 2) = sv1;
        }
        if (tid < 64) {
            double *li = Lb + 256 * RB_LD + (tid >> 2) * RB_LD + (tid & 3) * 4;
            *reinterpret_cast<d2_t *>(li) = si0;
            *reinterpret_cast<d2_t *>(li + 2) = si1;
        }
        if (tid == 64) Lb[256 * RB_LD + 16 * RB_LD] = sflag;
    };
    auto xbuf = [&](int k, int r) { return Xs + ((size_t)((k & 1) * 4 + grp) * 2 + r) * (16 * RB_LD); };
    // solve strip k (slot k / 4 of the waves with sw == k % 4): X_k = T_k Linv_k^T (+ one refinement step) -> LDS, global
    auto solve = [&](int k) {
        const int t = k >> 2;
        const double *Lb = Ls + (k & 1) * kTrsmLBuf;
        const double *li = Lb + 256 * RB_LD + prow * RB_LD + 4 * fk;
        const d2_t n01 = *reinterpret_cast<const d2_t *>(li), n23 = *reinterpret_cast<const d2_t *>(li + 2);
        const bool refine = Lb[256 * RB_LD + 16 * RB_LD] != 0.0;
        d2_t l01 = d2_t{0.0, 0.0}, l23 = d2_t{0.0, 0.0};
        if (refine) {  // rows pi(frow) of the raw diagonal tile, zero right of the diagonal
            const double *lr = Lb + (16 * k + prow) * RB_LD + 4 * fk;
            const d2_t r01 = *reinterpret_cast<const d2_t *>(lr), r23 = *reinterpret_cast<const d2_t *>(lr + 2);
            l01 = d2_t{(4 * fk <= prow) ? r01[0] : 0.0, (4 * fk + 1 <= prow) ? r01[1] : 0.0};
            l23 = d2_t{(4 * fk + 2 <= prow) ? r23[0] : 0.0, (4 * fk + 3 <= prow) ? r23[1] : 0.0};
        }
#pragma unroll
        for (int r = 0; r < RT; r++) {
            double4_t x = double4_t{0.0, 0.0, 0.0, 0.0};
            RB_MFMA4(x, 0, n01, n23, acc[r][4 * t], acc[r][4 * t + 1], acc[r][4 * t + 2], acc[r][4 * t + 3]);
            if (refine) {
                double4_t rs = double4_t{acc[r][4 * t], acc[r][4 * t + 1], acc[r][4 * t + 2], acc[r][4 * t + 3]};
                RB_MFMA4(rs, 1, l01, l23, x[0], x[1], x[2], x[3]);
                RB_MFMA4(x, 0, n01, n23, rs[0], rs[1], rs[2], rs[3]);
            }
            double *xs = xbuf(k, r) + frow * RB_LD + 4 * fk;
            *reinterpret_cast<d2_t *>(xs) = d2_t{x[0], x[1]};
            *reinterpret_cast<d2_t *>(xs + 2) = d2_t{x[2], x[3]};
            double *dst = Pw + (int64_t)(16 * r) * a.ld + 16 * k;
            store_d2_sc1(dst, d2_t{x[0], x[1]});
            store_d2_sc1(dst + 2, d2_t{x[2], x[3]});
        }
    };
    stage_load(0);
    stage_store(0);
    __syncthreads();  // strip 0 is staged
    bool have = false;  // strip k + 1's pieces are in this thread's registers
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if (k < nb16) {
            have = (k + 1 < nb16) && (avail >= base + k + 2);
            // a QUARTER of the panel's columns (four strips, one per wave residue) is complete with this strip: its stores are
            // drained in front of the barrier and the rows' progress word is raised behind it -- FINE tasks contract quarter by
            // quarter, so the update of the next diagonal block is three quarters done when the last strip arrives
            const bool qend = (k & 3) == 3 || k + 1 == nb16;
            if (have && !qend) stage_load(k + 1);
            if (sw == (k & 3)) solve(k);
            if (qend) {
                drain_stores();  // (in front of the next strip's fetch: the wait is for the stores only)
                if (have) stage_load(k + 1);
            }
            __syncthreads();  // X_k is in LDS
            if (qend && tid == 0)
                for (int c = 0; c < RT; c++) store_flag(S + a.off_rowT + P * a.NC + c0 + c, (k >> 2) + 1);
            {
                const double *Lb = Ls + (k & 1) * kTrsmLBuf;
                d2_t b01[RT], b23[RT];
#pragma unroll
                for (int r = 0; r < RT; r++) {
                    const double *xs = xbuf(k, r) + frow * RB_LD + 4 * fk;
                    b01[r] = *reinterpret_cast<const d2_t *>(xs);
                    b23[r] = *reinterpret_cast<const d2_t *>(xs + 2);
                }
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int C = 4 * t + sw;
                    if (C > k && C < nb16) {  // T_C -= X_k L(C, k)^T
                        const double *lf = Lb + (16 * C + prow) * RB_LD + 4 * fk;
                        const d2_t a01 = *reinterpret_cast<const d2_t *>(lf), a23 = *reinterpret_cast<const d2_t *>(lf + 2);
#pragma unroll
                        for (int r = 0; r < RT; r++) {
                            double4_t c4 = double4_t{acc[r][4 * t], acc[r][4 * t + 1], acc[r][4 * t + 2], acc[r][4 * t + 3]};
                            RB_MFMA4(c4, 1, a01, a23, b01[r][0], b01[r][1], b23[r][0], b23[r][1]);
                            acc[r][4 * t] = c4[0];
                            acc[r][4 * t + 1] = c4[1];
                            acc[r][4 * t + 2] = c4[2];
                            acc[r][4 * t + 3] = c4[3];
                        }
                    }
                }
            }
            if (k + 1 < nb16) {
                if (!have) {  // strip k + 1 was not there yet: thread 0 waits for it, then everybody fetches
                    if (tid == 0) {
                        const int r = pipe_wait_ge(S + 1, base + k + 2, a.sync, info, a.timeout, (3 << 28) | ((p) << 20) | ((c0) & 0xfffff));
                        s_ctl[1] = r ? -r : load_flag(S + 1);
                    }
                    __syncthreads();
                    avail = s_ctl[1];
                    if (avail < 0) return -avail;
                    stage_load(k + 1);
                } else if (tid == 0) {
                    s_ctl[1] = load_flag(S + 1);  // (a fresh look for the next strip's prefetch decision)
                }
                stage_store(k + 1);
                __syncthreads();  // strip k + 1 is staged (and s_ctl[1] is everybody's)
                avail = s_ctl[1];
            }
        }
    }
    return 0;  // (the last strip closed the last quarter: drained and published above)
}

// ---------------------------------------------------------------------------------------------
// FINE: C (64 x 64 at rows 64 c_row, columns k0 + 256 + 64 j) -= A B^T over panel p's 256 columns, QUARTER BY QUARTER as
// the TRSM tasks of the two row chunks publish them (rowT counts finished quarters): a quarter's 64 x 64 pieces of A and B
// come through LDS (one agent-scope 32-byte piece of each per thread: every byte crosses the fabric once), wave w owns the
// 16 x 16 sub-tile (w / 4, w % 4) and contracts the quarter with sixteen MFMAs, in ascending k -- no cross-wave reduction.
// When the last strip of the panel arrives, the quarters before it are done: what is left on the chain is one quarter and
// the tile's read-modify-write.  (Operands straight from L2 into the MFMAs, without LDS, were 4x the fabric traffic -- every
// row slice is used by four waves -- and 64 us per task instead of 17: profiles/r05_fine_quarters.txt)
// ---------------------------------------------------------------------------------------------
constexpr int kFineLd = 68;                    // doubles per staged row (64 + 4: conflict-free 32-byte fragment reads)
constexpr int kFineBuf = 2 * 64 * kFineLd;     // A and B pieces of one quarter
static_assert(2 * kFineBuf <= kPipeLdsDoubles, "LDS of the FINE role");
__device__ __noinline__ int pipe_role_fine(pipe_kargs_t ka, pipe_lds_t sm3, long long *tr, int z, int p, int c_row, int j) {
    const PipeArgs a = pipe_kargs(ka);
    z = pipe_uniform(z), p = pipe_uniform(p), c_row = pipe_uniform(c_row), j = pipe_uniform(j);
#ifdef EGX_PIPE_TRACE
    tr = pipe_uniform(tr);
#else
    tr = nullptr;
#endif
    double *sm = pipe_uniform_lds(sm3);
    int *s_ctl = reinterpret_cast<int *>(sm + kPipeLdsDoubles);
    const int k0 = a.g0 + 256 * p, P = k0 >> 8;
    const int nq = ((a.n_pad - k0 < 256) ? (a.n_pad - k0) : 256) >> 6;  // quarters of panel p
    const int R0 = 64 * c_row, C0 = k0 + 256 + 64 * j, c_col = C0 >> 6;
    double *Mz = a.M + (int64_t)z * a.sM;
    const int *info = a.info + (int64_t)z * a.sI;
    int *S = a.sync + (int64_t)z * a.sS;
    const int tid = threadIdx.x, lane = tid & 63, frow = lane & 15, fk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int prow = ((frow & 3) << 2) | (frow >> 2);  // pi(frow), see rb_factor_block
    // staging: thread -> row tid / 16 of the 64, doubles [4 (tid % 16), + 4) of the quarter's 64 columns
    const int srow = tid >> 4, spart = (tid & 15) * 4;
    const double *Ag = Mz + (int64_t)(R0 + srow) * a.ld + k0 + spart;
    const double *Bg = Mz + (int64_t)(C0 + srow) * a.ld + k0 + spart;
    // accumulator layout as everywhere here: lane (frow, fk), register i <-> C[frow][4 fk + i] of the sub-tile; the operand
    // whose rows index the tile's COLUMNS is read at row pi(frow), the one whose rows index the tile's ROWS at row frow
    const int a_off = (16 * (wave >> 2) + frow) * kFineLd + 4 * fk, b_off = 64 * kFineLd + (16 * (wave & 3) + prow) * kFineLd + 4 * fk;
    double *Cp = Mz + (int64_t)(R0 + 16 * (wave >> 2) + frow) * a.ld + C0 + 16 * (wave & 3) + 4 * fk;
    double4_t acc = double4_t{0.0, 0.0, 0.0, 0.0};
    d2_t c01 = d2_t{0.0, 0.0}, c23 = d2_t{0.0, 0.0};
    int done = 0;
    while (done < nq) {
        if (tid == 0) {  // the next quarter of both row chunks -- and how many more are there already
            int rr = (done == 0 && load_flag(info) != 0) ? 1 : 0;
            const int *fr = S + a.off_rowT + P * a.NC + c_row, *fc = S + a.off_rowT + P * a.NC + c_col;
            if (rr == 0) rr = pipe_wait_ge(fr, done + 1, a.sync, info, a.timeout, (4 << 28) | ((p) << 20) | ((c_row) & 0xfffff));
            if (rr == 0) rr = pipe_wait_ge(fc, done + 1, a.sync, info, a.timeout, (5 << 28) | ((p) << 20) | ((c_row) & 0xfffff));
            int av = load_flag(fr);
            const int av2 = load_flag(fc);
            av = av < av2 ? av : av2;
            av = av < nq ? av : nq;
            if (rr == 0 && av == nq && p > 0)  // with the last quarter the tile itself: every earlier panel's update has been applied
                rr = pipe_wait_ge(S + a.off_cver + (R0 >> 7) * a.NJ + (C0 >> 7), p, a.sync, info, a.timeout, (6 << 28) | ((p) << 20) | ((c_row) & 0xfffff));
            s_ctl[0] = rr;
            s_ctl[4] = av;
        }
        __syncthreads();
        const int r = s_ctl[0], avail = s_ctl[4];
        __syncthreads();
        if (r) return r;
        if (tr && done == 0 && threadIdx.x == 0) tr[3] = wall_clock64();
        if (avail == nq) {  // (operands and the C tile are read with agent-scope loads: no fence; in flight with the staging loads)
            c01 = load_d2_sc1(Cp);
            c23 = load_d2_sc1(Cp + 2);
        }
        for (int q = done; q < avail; q++) {
            double *buf = sm + (q & 1) * kFineBuf;
            const d2_t a0 = load_d2_sc1(Ag + 64 * q), a1 = load_d2_sc1(Ag + 64 * q + 2);
            const d2_t b0 = load_d2_sc1(Bg + 64 * q), b1 = load_d2_sc1(Bg + 64 * q + 2);
            *reinterpret_cast<d2_t *>(buf + srow * kFineLd + spart) = a0;
            *reinterpret_cast<d2_t *>(buf + srow * kFineLd + spart + 2) = a1;
            *reinterpret_cast<d2_t *>(buf + 64 * kFineLd + srow * kFineLd + spart) = b0;
            *reinterpret_cast<d2_t *>(buf + 64 * kFineLd + srow * kFineLd + spart + 2) = b1;
            __syncthreads();  // (two buffers: the next quarter's stores cannot overtake this one's reads)
#pragma unroll
            for (int kb = 0; kb < 4; kb++) {
                const d2_t av0 = *reinterpret_cast<const d2_t *>(buf + b_off + 16 * kb), av1 = *reinterpret_cast<const d2_t *>(buf + b_off + 16 * kb + 2);
                const d2_t bv0 = *reinterpret_cast<const d2_t *>(buf + a_off + 16 * kb), bv1 = *reinterpret_cast<const d2_t *>(buf + a_off + 16 * kb + 2);
                RB_MFMA4(acc, 0, av0, av1, bv0[0], bv0[1], bv1[0], bv1[1]);
            }
        }
        done = avail;
    }
    store_d2_sc1(Cp, d2_t{c01[0] - acc[0], c01[1] - acc[1]});
    store_d2_sc1(Cp + 2, d2_t{c23[0] - acc[2], c23[1] - acc[3]});
    drain_stores();
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(S + a.off_fcnt + (P + 1) * a.NC + c_row, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// COARSE: C (128 x 128 tile (I, J), absolute) -= A B^T over panel p's 256 columns; sixteen waves x 32 x 32
// ---------------------------------------------------------------------------------------------
// last != 0 (flow launch, pipe_flow.h): this is the LAST update of the tile's column block (panel p into block column p + 1, rows
// below its diagonal block): the two 64-row chunks of the tile also count two fine tiles each towards fcnt, what the solve of
// those rows waits for
__device__ __noinline__ int pipe_role_coarse(pipe_kargs_t ka, pipe_lds_t sm3, long long *tr, int z, int p, int I, int J, int last = 0) {
    const PipeArgs a = pipe_kargs(ka);
    z = pipe_uniform(z), p = pipe_uniform(p), I = pipe_uniform(I), J = pipe_uniform(J), last = pipe_uniform(last);
#ifdef EGX_PIPE_TRACE
    tr = pipe_uniform(tr);
#else
    tr = nullptr;
#endif
    double *sm = pipe_uniform_lds(sm3);
    int *s_ctl = reinterpret_cast<int *>(sm + kPipeLdsDoubles);
    const int k0 = a.g0 + 256 * p, P = k0 >> 8;
    const int R0 = 128 * I, C0 = 128 * J;
    double *Mz = a.M + (int64_t)z * a.sM;
    const int *info = a.info + (int64_t)z * a.sI;
    int *S = a.sync + (int64_t)z * a.sS;
    int *ver = S + a.off_cver + I * a.NJ + J;
    const int nqp = ((a.n_pad - k0 < 256) ? (a.n_pad - k0) : 256) >> 6;  // rowT counts the finished QUARTERS of a row chunk
    const int r = pipe_wg_wait(s_ctl, [&]() {
        int rr = load_flag(info) != 0 ? 1 : 0;
        const int *rowT = S + a.off_rowT + P * a.NC;
        {   // (all five words in one round of loads: the flow launch claims these tasks when they can start)
            const int f0 = load_flag(rowT + 2 * I), f1 = load_flag(rowT + 2 * I + 1), f2 = load_flag(rowT + 2 * J), f3 = load_flag(rowT + 2 * J + 1);
            const int v = p > 0 ? load_flag(ver) : 0;
            if ((rr == 0) & (f0 >= nqp) & (f1 >= nqp) & (f2 >= nqp) & (f3 >= nqp) & (v >= p)) return 0;
        }
        if (rr == 0) rr = pipe_wait_ge(rowT + 2 * I, nqp, a.sync, info, a.timeout, (7 << 28) | ((p) << 20) | ((I * 1024 + J) & 0xfffff));
        if (rr == 0) rr = pipe_wait_ge(rowT + 2 * I + 1, nqp, a.sync, info, a.timeout, (7 << 28) | ((p) << 20) | ((I * 1024 + J) & 0xfffff));
        if (rr == 0) rr = pipe_wait_ge(rowT + 2 * J, nqp, a.sync, info, a.timeout, (7 << 28) | ((p) << 20) | ((I * 1024 + J) & 0xfffff));
        if (rr == 0) rr = pipe_wait_ge(rowT + 2 * J + 1, nqp, a.sync, info, a.timeout, (7 << 28) | ((p) << 20) | ((I * 1024 + J) & 0xfffff));
        if (rr == 0 && p > 0) rr = pipe_wait_ge(ver, p, a.sync, info, a.timeout, (7 << 28) | ((p) << 20) | ((I * 1024 + J) & 0xfffff));
        return rr;
    });
    if (r) return r;
    if (tr && threadIdx.x == 0) tr[3] = wall_clock64();
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double4_t acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
        for (int ni = 0; ni < 2; ni++) acc[mi][ni] = double4_t{0.0, 0.0, 0.0, 0.0};
    gemm_core<128, 128, 32, 32, 1024>(Mz + (int64_t)R0 * a.ld + k0, a.ld, Mz + (int64_t)C0 * a.ld + k0, a.ld, 256, acc, sm, tid);
    double *Ct = Mz + (int64_t)(R0 + (wave >> 2) * 32 + (lane >> 4)) * a.ld + C0 + (wave & 3) * 32 + (lane & 15);
    double cv[2][2][4];
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
        for (int ni = 0; ni < 2; ni++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) cv[mi][ni][rr] = Ct[(int64_t)(16 * mi + 4 * rr) * a.ld + 16 * ni];
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
        for (int ni = 0; ni < 2; ni++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++)
                __hip_atomic_store(Ct + (int64_t)(16 * mi + 4 * rr) * a.ld + 16 * ni, cv[mi][ni][rr] - acc[mi][ni][rr], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
    drain_stores();
    __syncthreads();
    if (tid == 0) {
        store_flag(ver, p + 1);
        if (last) {
            __hip_atomic_fetch_add(S + a.off_fcnt + (P + 1) * a.NC + 2 * I, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(S + a.off_fcnt + (P + 1) * a.NC + 2 * I + 1, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    return 0;
}

// the workers: ticket -> (task, matrix), until the list is exhausted or the launch is aborted
__device__ __noinline__ void pipe_worker_loop(pipe_kargs_t ka, pipe_lds_t sm3) {
    const PipeArgs a = pipe_kargs(ka);
    double *sm = pipe_uniform_lds(sm3);
    int *s_ctl = reinterpret_cast<int *>(sm + kPipeLdsDoubles);
    int *ticket = a.sync + kPipeHdr + (a.g0 >> 8);
    const int total = a.ntasks * a.nz;
    for (;;) {
        if (threadIdx.x == 0) s_ctl[2] = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int t = s_ctl[2];
        __syncthreads();
        if (t >= total) return;
        const int z = t % a.nz;
        const PipeTask task = a.tasks[t / a.nz];
#ifdef EGX_PIPE_TRACE
        long long *tr = a.trace ? a.trace + 8 * (int64_t)t : nullptr;
#else
        long long *tr = nullptr;  // (task timelines: tools/pipe_check builds with -DEGX_PIPE_TRACE)
#endif
        if (tr && threadIdx.x == 0) {
            tr[0] = task.type | (task.p << 8) | (z << 16);
            tr[1] = (unsigned)task.a | ((long long)task.b << 32);
            tr[2] = wall_clock64();
            tr[5] = __builtin_amdgcn_s_getreg((3 << 11) | 20) | ((long long)blockIdx.x << 8);  // XCC_ID, workgroup
        }
        int r;
        if (task.type == PT_TRSM) r = pipe_role_trsm<1>(ka, (pipe_lds_t)sm, tr, z, task.p, task.a);
        else if (task.type == PT_FINE) r = pipe_role_fine(ka, (pipe_lds_t)sm, tr, z, task.p, task.a, task.b);
        else r = pipe_role_coarse(ka, (pipe_lds_t)sm, tr, z, task.p, task.a, task.b);
        if (tr && threadIdx.x == 0) tr[4] = wall_clock64();
        if (r == 2) return;  // aborted: the launch drains
    }
}

// The first np * nz workgroups TO START (a start ticket, not the block index: with several chain launches in flight a block's
// compute-unit slot may free up long after its neighbours', and workers spinning on a diagonal block that is still queued
// behind them would never let it in) are the DIAG workgroups: start ticket t factors diagonal block t / nz of matrix t % nz,
// INLINE in the kernel body -- in kernel context the register-resident block fits its 128 vector registers without a spill
// (as a called function it ran at 75-95 us per block instead of 60-68) -- and is a worker afterwards.  They are resident
// before any worker holds a task, and wait for their block's fine tiles.
__global__ __launch_bounds__(1024) void k_potrf_pipe(PipeArgs a) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    int *s_ctl = reinterpret_cast<int *>(sm + kPipeLdsDoubles);
    if (threadIdx.x == 0)
        s_ctl[3] = __hip_atomic_fetch_add(a.sync + kPipeHdr + a.NP + (a.g0 >> 8), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int start = s_ctl[3];
    if (start < a.np * a.nz) {
        const int z = start % a.nz, p = start / a.nz;
        const int k0 = a.g0 + 256 * p, P = k0 >> 8;
        const int nbk = (a.n_pad - k0 < 256) ? (a.n_pad - k0) : 256;
        int *info = a.info + (int64_t)z * a.sI;
        int *S = a.sync + (int64_t)z * a.sS;
#ifdef EGX_PIPE_TRACE
        long long *tr = a.trace ? a.trace + 8 * (int64_t)(a.ntasks * a.nz + start) : nullptr;
        if (tr && threadIdx.x == 0) {
            tr[0] = PT_DIAG | (p << 8) | (z << 16);
            tr[1] = 0;
            tr[2] = wall_clock64();
            tr[5] = __builtin_amdgcn_s_getreg((3 << 11) | 20) | ((long long)blockIdx.x << 8);
        }
#endif
        // the block's fine tiles (i, j <= i) have received the previous panel (first block of a launch: nothing to wait for)
        const int r = pipe_wg_wait(s_ctl, [&]() {
            int rr = load_flag(info) != 0 ? 1 : 0;
            const int *fc = S + a.off_fcnt + P * a.NC + (k0 >> 6);
            for (int i = 0; i < nbk / 64 && rr == 0 && p > 0; i++) rr = pipe_wait_ge(fc + i, i + 1, a.sync, info, a.timeout, (8 << 28) | ((p) << 20) | ((z) & 0xfffff));
            return rr;
        });
        if (r == 2) return;
        if (r == 0) {
            RbPublish pub;
            pub.base = k0 >> 4;
            pub.strips = S + 1;
            if (a.stall && pub.base + (nbk >> 4) >= a.stall) pub.strips = S + 2;  // (test hook: this block's strips are never seen)
#ifdef EGX_PIPE_TRACE
            pub.trace = tr;
            if (tr && threadIdx.x == 0) tr[3] = wall_clock64();
#endif
            (void)rb_factor_block<16, true>(a.M + (int64_t)z * a.sM + (int64_t)k0 * a.ld + k0, a.ld, nbk,
                                            a.dinv + (int64_t)z * a.sD + (int64_t)(k0 / 64) * 4096, info, k0, a.n_pad, sm, pub);
#ifdef EGX_PIPE_TRACE
            if (tr && threadIdx.x == 0) tr[4] = wall_clock64();
#endif
        }
        __syncthreads();  // (the waves of a block that lost a pivot leave rb_factor_block at different points)
    }
    pipe_worker_loop((pipe_kargs_t)__builtin_amdgcn_kernarg_segment_ptr(), (pipe_lds_t)sm);
}

// =============================================================================================
// FLOW: the whole factorisation of a large matrix as ONE launch with critical and bulk-class work (pipe_flow.h has the
// shape of the lists, the gates and the deadlock argument; round 6)
// =============================================================================================
struct FlowArgs {
    PipeArgs p;          // (first: the roles read the launch arguments through a pointer to it)
    const int *coff;     // [NP + 1] first HEAD task of every stage in p.tasks
    int NI;              // 128-row tiles of the matrix (m_tot / 128)
    int RMAX, off_open, off_cnext, off_lnext, off_tnext, off_rcur, off_rcnt, off_trace;
    int bulk_pairs;      // a workgroup may take two consecutive far tiles with one claim (one look of the scheduler per two tiles)
    int col_base;        // column of the enclosing matrix the launch's matrix starts at (a flow launch for the LAST columns of a
                         // larger factorisation): only the reported pivot index needs it
    int lead_short, lead_long;  // a workgroup with a diagonal block ahead of it stops taking short / long tasks this many blocks before its own
    int trace_cap;       // trace slots (profiling builds)
};
typedef const __attribute__((address_space(4))) FlowArgs *flow_kargs_t;
__device__ __forceinline__ FlowArgs flow_kargs(flow_kargs_t kv) {
    const unsigned long long u = (unsigned long long)kv;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    flow_kargs_t k = (flow_kargs_t)(((unsigned long long)hi << 32) | lo);
    FlowArgs a;
    a.p = pipe_kargs((pipe_kargs_t)k);
    a.coff = k->coff, a.NI = k->NI, a.RMAX = k->RMAX, a.off_open = k->off_open, a.off_cnext = k->off_cnext, a.off_rcur = k->off_rcur;
    a.off_rcnt = k->off_rcnt, a.off_lnext = k->off_lnext, a.off_tnext = k->off_tnext, a.off_trace = k->off_trace, a.bulk_pairs = k->bulk_pairs;
    a.lead_short = k->lead_short, a.lead_long = k->lead_long, a.trace_cap = k->trace_cap;
    return a;
}

// ---------------------------------------------------------------------------------------------
// BULK: C (128 rows x 256 columns: row tile I of block column q) -= A B^T over the panels [p0, p1) (K = 512): the stream
// kernel's LDS-DMA ring (k_gemm_stream, kernels_chol.hip: unpadded [384][16] chunk image, XOR-swizzled 16-byte slots, three
// 48 KB stages, prefetch distance two, one counted wait + one barrier per chunk in the MIDDLE of the chunk's MFMAs) re-cut for
// the launch's sixteen waves: a wave owns 32 x 64 of the tile (64 accumulator registers of its 128) and three of a chunk's 48
// LDS-DMA pieces.  Accumulators start as -C; the tile is stored write-through and its two 128-column halves are versioned.
// ---------------------------------------------------------------------------------------------
constexpr int kBulkRows = 384, kBulkStage = kBulkRows * KC, kBulkStages = 3;
static_assert(kBulkStages * kBulkStage <= kPipeLdsDoubles, "LDS of the BULK role");
typedef __attribute__((address_space(3))) void *pipe_lds_void_t;
typedef const __attribute__((address_space(1))) void *pipe_gbl_void_t;

__device__ __noinline__ int pipe_role_bulk(pipe_kargs_t ka, pipe_lds_t sm3, long long *tr, int z, int q, int I, int p0, int p1) {
    const PipeArgs a = pipe_kargs(ka);
    z = pipe_uniform(z), q = pipe_uniform(q), I = pipe_uniform(I), p0 = pipe_uniform(p0), p1 = pipe_uniform(p1);
#ifdef EGX_PIPE_TRACE
    tr = pipe_uniform(tr);
#else
    tr = nullptr;
#endif
    double *smem = pipe_uniform_lds(sm3);
    int *s_ctl = reinterpret_cast<int *>(smem + kPipeLdsDoubles);
    double *Mz = a.M + (int64_t)z * a.sM;
    const int *info = a.info + (int64_t)z * a.sI;
    int *S = a.sync + (int64_t)z * a.sS;
    const bool top = I == 2 * q;  // the tile's right half lies above the diagonal: computed, never read, not versioned
    int *ver0 = S + a.off_cver + I * a.NJ + 2 * q, *ver1 = ver0 + 1;
    {
        const int r = pipe_wg_wait(s_ctl, [&]() {
            int rr = load_flag(info) != 0 ? 1 : 0;
            const int *rowT = S + a.off_rowT + (p1 - 1) * a.NC;  // (a row chunk solved for panel p1 - 1 is solved for every earlier panel)
            const int who = (10 << 28) | (p0 << 20) | ((I * 1024 + q) & 0xfffff);
            {   // the usual case -- the scheduler looked before it claimed -- is one round of parallel loads, not eight one after the other
                const int f0 = load_flag(rowT + 2 * I), f1 = load_flag(rowT + 2 * I + 1), c0 = load_flag(rowT + 4 * q), c1 = load_flag(rowT + 4 * q + 1),
                          c2 = load_flag(rowT + 4 * q + 2), c3 = load_flag(rowT + 4 * q + 3), v0 = load_flag(ver0), v1 = top ? p0 : load_flag(ver1);
                if ((rr == 0) & (f0 >= 4) & (f1 >= 4) & (c0 >= 4) & (c1 >= 4) & (c2 >= 4) & (c3 >= 4) & (v0 >= p0) & (v1 >= p0)) return 0;
            }
            if (rr == 0) rr = pipe_wait_ge(rowT + 2 * I, 4, a.sync, info, a.timeout, who);
            if (rr == 0) rr = pipe_wait_ge(rowT + 2 * I + 1, 4, a.sync, info, a.timeout, who);
            for (int c = 4 * q; c < 4 * q + 4 && rr == 0; c++) rr = pipe_wait_ge(rowT + c, 4, a.sync, info, a.timeout, who);
            if (rr == 0) rr = pipe_wait_ge(ver0, p0, a.sync, info, a.timeout, who);
            if (rr == 0 && !top) rr = pipe_wait_ge(ver1, p0, a.sync, info, a.timeout, who);
            return rr;
        });
        if (r) return r;
    }
    if (tr && threadIdx.x == 0) tr[3] = wall_clock64();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t ld = a.ld;
    const double *A = Mz + (int64_t)(128 * I) * ld + 256 * p0;
    const double *B = Mz + (int64_t)(256 * q) * ld + 256 * p0;
    const int nch = (256 * (p1 - p0)) / KC;
    // ---- load side: wave w fills the 8-row groups w of A and w, w + 16 of B of every stage
    const int lrow = lane >> 3;
    const int sw_ld = (4 * wave + (lane >> 4)) & 7;  // ((row >> 1) & 7) of this lane's row, the same for the three groups
    const unsigned lpart = (unsigned)(((lane & 7) ^ sw_ld) * 2);
    const unsigned offA = (unsigned)((8 * wave + lrow) * (int)ld) + lpart;
    unsigned offB[2];
#pragma unroll
    for (int i = 0; i < 2; i++) offB[i] = (unsigned)((8 * (wave + 16 * i) + lrow) * (int)ld) + lpart;
    auto issue = [&](int stage, int ch) {
        double *dst = smem + stage * kBulkStage + wave * 128;  // group g starts at g * 8 rows * 16 doubles
        __builtin_amdgcn_global_load_lds((pipe_gbl_void_t)(A + ch * KC + offA), (pipe_lds_void_t)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((pipe_gbl_void_t)(B + ch * KC + offB[0]), (pipe_lds_void_t)(dst + 2048), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((pipe_gbl_void_t)(B + ch * KC + offB[1]), (pipe_lds_void_t)(dst + 2048 + 2048), 16, 0, 0);
    };
    issue(0, 0);
    if (nch > 1) issue(1, 1);
    // ---- compute side: wave (w / 4, w % 4) owns rows [32 (w / 4), + 32) x columns [64 (w % 4), + 64)
    const int wm0 = (wave >> 2) * 32, wn0 = (wave & 3) * 64;
    const int frow = lane & 15, fk = lane >> 4;
    int foff[2];
    foff[0] = frow * KC + ((fk ^ ((frow >> 1) & 7)) << 1);
    foff[1] = foff[0] ^ 8;
    const int aoff = wm0 * KC, boff = (128 + wn0) * KC;
    double *Ct = Mz + (int64_t)(128 * I + wm0) * ld + 256 * q + wn0;
    unsigned coff[4];
#pragma unroll
    for (int r = 0; r < 4; r++) coff[r] = (unsigned)(((lane >> 4) + 4 * r) * (int)ld + (lane & 15));
    double4_t acc[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
        for (int ni = 0; ni < 4; ni++)
#pragma unroll
            for (int r = 0; r < 4; r++) acc[mi][ni][r] = -Ct[(int64_t)mi * 16 * ld + ni * 16 + coff[r]];
    auto read_half = [&](int st, int kb, d2_t (&av)[2], d2_t (&bv)[4]) {
        const double *As = smem + st * kBulkStage + aoff, *Bs = smem + st * kBulkStage + boff;
#pragma unroll
        for (int mi = 0; mi < 2; mi++) av[mi] = *reinterpret_cast<const d2_t *>(As + mi * 16 * KC + foff[kb]);
#pragma unroll
        for (int ni = 0; ni < 4; ni++) bv[ni] = *reinterpret_cast<const d2_t *>(Bs + ni * 16 * KC + foff[kb]);
    };
    auto mma_quarter = [&](const d2_t (&av)[2], const d2_t (&bv)[4], int h) {
#pragma unroll
        for (int mi = 0; mi < 2; mi++)
#pragma unroll
            for (int ni = 0; ni < 4; ni++) acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[mi][h], bv[ni][h], acc[mi][ni], 0, 0, 0);
    };
    d2_t a0[2], b0[4];
    if (nch > 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_half(0, 0, a0, b0);
    int stage = 0;
    for (int ch = 0; ch < nch; ch++) {
        d2_t a1[2], b1[4];
        read_half(stage, 1, a1, b1);
        mma_quarter(a0, b0, 0);
        mma_quarter(a0, b0, 1);
        __builtin_amdgcn_sched_barrier(0);
        const bool next = ch + 1 < nch, more = ch + 2 < nch;
        const int st1 = (stage == kBulkStages - 1) ? 0 : stage + 1;
        if (next) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // own pieces of chunk ch + 1 (the only ones in flight)
            __builtin_amdgcn_s_barrier();
            if (more) issue(stage == 0 ? 2 : stage - 1, ch + 2);  // chunk ch + 2 -> stage of chunk ch - 1
        }
        __builtin_amdgcn_sched_barrier(0);
        mma_quarter(a1, b1, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (next) read_half(st1, 0, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        mma_quarter(a1, b1, 1);
        stage = st1;
    }
#pragma unroll
    for (int r = 0; r < 4; r++) asm volatile("" : "+v"(coff[r]));
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
        for (int ni = 0; ni < 4; ni++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                __hip_atomic_store(Ct + (int64_t)mi * 16 * ld + ni * 16 + coff[r], -acc[mi][ni][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    drain_stores();
    __syncthreads();
    if (tid == 0) {
        store_flag(ver0, p1);
        if (!top) store_flag(ver1, p1);
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// The scheduler of a flow launch: every workgroup runs it between tasks.  Wave 0 looks -- all in one pass of parallel
// loads -- at the open stage's ticket counter and at the first unfinished round of (up to) 64 columns from the scan hint on;
// a critical ticket wins, then the nearest column's released round.  `my_block` >= 0: this workgroup owns a diagonal block
// and leaves (return 3) once that block is `lead` blocks away, taking long (bulk) tasks only while it is `lead_long` away.
// Returns 0 = the factorisation's lists are exhausted, 2 = the launch is aborted.
// ---------------------------------------------------------------------------------------------
__device__ __noinline__ int flow_worker_loop(flow_kargs_t kf, pipe_lds_t sm3, int z, int my_block) {
    const FlowArgs f = flow_kargs(kf);
    const PipeArgs &a = f.p;
    const pipe_kargs_t ka = (pipe_kargs_t)kf;
    z = pipe_uniform(z), my_block = pipe_uniform(my_block);
    double *sm = pipe_uniform_lds(sm3);
    int *s_ctl = reinterpret_cast<int *>(sm + kPipeLdsDoubles);
    int *S = a.sync + (int64_t)z * a.sS;
    const int *info = a.info + (int64_t)z * a.sI;
    const FlowShape sh{a.NP, a.NC, f.NI};
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int stage_guess = 0;       // the stage whose head this workgroup last saw open (a wrong guess only costs the slow path)
    long long idle_since = 0;  // wall clock of the first pass without work since the last task (0: not idle)
    // Would this near-class task (COARSE, also as LAST) start at once?  (rows solved for its panel, the tile up to date)
    // (all loads first, one decision at the end: `&&` would make every load wait for the one before it -- a pass of this
    //  scheduler is a handful of L2 round trips of ~1 us each, and 256 workgroups make one per task)
    auto near_ready = [&](int p, int I, int J) {
        const int *rowT = S + a.off_rowT + p * a.NC;
        const int r0 = load_flag(rowT + 2 * I), r1 = load_flag(rowT + 2 * I + 1), r2 = load_flag(rowT + 2 * J), r3 = load_flag(rowT + 2 * J + 1);
        const int v = load_flag(S + a.off_cver + I * a.NJ + J);
        return (r0 >= 4) & (r1 >= 4) & (r2 >= 4) & (r3 >= 4) & (v >= p);
    };
    for (;;) {
        // ---- decide (wave 0), publish in s_ctl[8 ..]: [8] kind (0 none / retry, 1 critical, 2 bulk-class, 3 leave for the
        //      diagonal block, 4 exhausted, 5 aborted); critical: [9] stage, [10] ticket, [11] list (0 head, 1 solves, 2 last
        //      updates of the tail); bulk-class: [9] column, [10] round, [11] first ticket, [12] tickets (1 or 2)
        if (wave == 0) {
            int kind = 0, w0 = 0, w1 = 0, w2 = 0, w3 = 1;
            const int strips = load_flag(S + 1);
            const int cur = strips >> 4;               // diagonal blocks finished (16 strips each)
            const bool failed = load_flag(info) != 0;  // (a matrix that lost a pivot: every task returns at once; nothing is "ready")
            const int aborted = load_flag(a.sync);
            bool allow_short = true, allow_long = true;
            if (my_block >= 0) {
                allow_short = my_block - cur > f.lead_short;
                allow_long = my_block - cur > f.lead_long;
            }
            // (a workgroup that owns block p never holds a ticket that could wait for block p: no head ticket of a stage >= p -- the
            //  tails and the bulk-class rounds only ever wait for blocks that are finished when the ticket is taken)
            if (aborted != 0) kind = 5;
            else if (!allow_short || (my_block >= 0 && stage_guess >= my_block)) kind = 3;
            else {
                // ---- (A) the head of the stage last seen open: its next ticket
                int Sh = stage_guess;
                int t = -1;
                bool exhausted = false;
                if (lane == 0) {
                    const int cnt = f.coff[Sh + 1] - f.coff[Sh];
                    const int nxt = load_flag(S + f.off_cnext + Sh);
                    exhausted = nxt >= cnt;
                    if (!exhausted) {
                        const PipeTask tk = a.tasks[f.coff[Sh] + nxt];
                        bool rdy = true;
                        if (!failed) {
                            // (what a CHECKED claim may wait for inside its task is RUNNING, never merely claimed: a ticket obtained
                            //  unchecked -- after losing the race for the one looked at -- may sit on an unclaimed producer, and nothing
                            //  that was claimed with a look must queue up behind it)
                            if (tk.type == PT_TRSM) {
                                // its rows' fine tiles are in, the blocks before its own are factored: its block starts within ~25 us
                                const int fc = tk.p == 0 ? 4 : load_flag(S + a.off_fcnt + tk.p * a.NC + tk.a);
                                rdy = (strips >= 16 * tk.p) & (fc >= 4);
                            } else if (tk.type == PT_FINE) {
                                // both solves it streams behind have published a quarter, its tile is up to date
                                const int *rowT = S + a.off_rowT + tk.p * a.NC;
                                const int r0 = load_flag(rowT + tk.a), r1 = load_flag(rowT + 4 * (tk.p + 1) + tk.b);
                                const int v = load_flag(S + a.off_cver + (tk.a >> 1) * a.NJ + 2 * (tk.p + 1) + (tk.b >> 1));
                                rdy = (r0 >= 1) & (r1 >= 1) & (v >= tk.p);
                            } else rdy = near_ready(tk.p, tk.a, tk.b);
                        }
                        if (rdy) {
                            t = __hip_atomic_fetch_add(S + f.off_cnext + Sh, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (t >= cnt) t = -1, exhausted = true;
                        }
                    }
                }
                t = __builtin_amdgcn_readfirstlane(t);
                exhausted = __builtin_amdgcn_readfirstlane((int)exhausted) != 0;
                if (t >= 0) kind = 1, w0 = Sh, w1 = t, w2 = 0;
                else if (exhausted) {
                    // every ticket of this head is taken: the next head opens (or has been opened by somebody else meanwhile)
                    const int So = load_flag(S + f.off_open);
                    if (So != stage_guess) stage_guess = So, kind = 0, w0 = 1;  // (w0 = 1: look again at once, not idle)
                    else if (So + 1 < a.NP) {
                        if (lane == 0) __hip_atomic_fetch_max(S + f.off_open, So + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        stage_guess = So + 1, kind = 0, w0 = 1;
                    }
                    Sh = stage_guess;
                }
                if (kind == 0 && my_block >= 0 && stage_guess >= my_block) kind = 3, w0 = 0;
                // ---- (B) the tails of the open stages, oldest first: lane 2 k + w looks at list w (0 solves, 1 last updates) of stage tlo + k
                if (kind == 0 && w0 == 0) {
                    const int tlo = load_flag(S + f.off_open + 2);
                    const int st = tlo + (lane >> 1), which = lane & 1;
                    int nxt = 0, size = 0, key = 0x7fffffff;
                    bool done = false;  // this list has no ticket left
                    if (lane < 16 && st <= Sh && st < a.NP) {
                        size = which ? flow_lt_size(sh, st) : flow_tt_size(sh, st);
                        nxt = load_flag(S + (which ? f.off_lnext : f.off_tnext) + st);
                        done = nxt >= size;
                        if (!done) {
                            bool rdy = true;
                            if (!failed) {
                                if (which) {
                                    const PipeTask tk = flow_lt_task(st, nxt);
                                    rdy = near_ready(tk.p, tk.a, tk.b);
                                } else {
                                    const PipeTask tk = flow_tt_task(st, nxt);
                                    const int fc = st == 0 ? 4 : load_flag(S + a.off_fcnt + st * a.NC + tk.a);
                                    rdy = (fc >= 4) & (strips >= 16 * (st + 1));
                                }
                            }
                            if (rdy) key = 2 * st + which;
                        }
                    }
                    // the oldest stage's lists both exhausted (and its head open): it leaves the window
                    const unsigned long long dmask = __builtin_amdgcn_ballot_w64(done);
                    if ((dmask & 3) == 3 && tlo <= Sh && lane == 0) __hip_atomic_fetch_max(S + f.off_open + 2, tlo + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    unsigned long long left = __builtin_amdgcn_ballot_w64(key != 0x7fffffff);
                    for (int tries = 0; tries < 3 && left && kind == 0; tries++) {
                        int best = -1, best_key = 0x7fffffff;
                        for (unsigned long long m = left; m; m &= m - 1) {
                            const int l = __builtin_ctzll(m);
                            const int k = __builtin_amdgcn_readlane(key, l);
                            if (k < best_key) best_key = k, best = l;
                        }
                        left &= ~(1ull << best);
                        int tk = -1;
                        if (lane == best) {
                            tk = __hip_atomic_fetch_add(S + (which ? f.off_lnext : f.off_tnext) + st, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (tk >= size) tk = -1;
                        }
                        tk = __builtin_amdgcn_readlane(tk, best);
                        if (tk >= 0) kind = 1, w0 = tlo + (best >> 1), w1 = tk, w2 = 1 + (best & 1);
                    }
                }
                // ---- (C) bulk-class: lane l looks at column q0 + l: the first round with tickets left, released, and whose NEXT ticket
                //      would start at once.  A workgroup that takes a tile whose rows are not solved yet, or whose previous round still
                //      runs, sits inside the task and is lost to everybody (n = 16384 without this look: 320 us of waiting per 130-us
                //      tile, 83 ms per factorisation).  Rows are solved top-down and tickets run top-down, so "not yet" holds for the
                //      tickets behind it too.
                if (kind == 0 && w0 == 0 && allow_long) {
                    // (every column from 2 on, 64 per look: heads run ahead of the tails, so columns well before the open head may
                    //  still be receiving rounds)
                    bool unfinished = false;
                    for (int qb = 2; qb < a.NP && kind == 0; qb += 64) {
                        const int q = qb + lane;
                        int r = -1, tsize = 0, key = 0x7fffffff, nxt = 0;
                        bool more = false;   // the column still has rounds to hand out, now or later
                        bool two = false;    // ... and the ticket behind the next one is a far tile that would start at once as well
                        if (q < a.NP) {
                            const int nr = flow_nrounds(q);
                            r = load_flag(S + f.off_rcur + q);
                            const int r_was = r;
                            for (int it = 0; it < 3 && r < nr; it++) {
                                if (flow_round_release_stage(q, r) > Sh) break;
                                tsize = flow_round_size(sh, q, r);
                                nxt = load_flag(S + f.off_rcnt + q * f.RMAX + r);
                                if (nxt < tsize) break;
                                r++;  // every ticket of round r is taken: the next round may be claimed
                                tsize = 0;
                            }
                            more = r < nr;
                            if (r > r_was) __hip_atomic_fetch_max(S + f.off_rcur + q, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (tsize > 0 && !failed) {
                                const FlowBulkTask bt = flow_round_task(sh, q, r, nxt);
                                bool rdy;
                                if (bt.type == PT_BULK) {
                                    const int *rowT = S + a.off_rowT + (bt.p1 - 1) * a.NC;
                                    const int *cv = S + a.off_cver + bt.I * a.NJ;
                                    const int r0 = load_flag(rowT + 2 * bt.I), r1 = load_flag(rowT + 2 * bt.I + 1);
                                    const int c0 = load_flag(rowT + 4 * q), c1 = load_flag(rowT + 4 * q + 1), c2 = load_flag(rowT + 4 * q + 2),
                                              c3 = load_flag(rowT + 4 * q + 3);
                                    const int v0 = load_flag(cv + 2 * q), v1 = load_flag(cv + 2 * q + 1);
                                    // the row tile below it, for a claim of two (its rows and its tile; the column's rows are the same)
                                    const bool pair = nxt + 1 < tsize && nxt >= 2 + kFlowWindow / 2;  // (never the diagonal block's or the window's tiles: the chain waits for those)
                                    const int r2 = pair ? load_flag(rowT + 2 * bt.I + 2) : 0, r3 = pair ? load_flag(rowT + 2 * bt.I + 3) : 0;
                                    const int v2 = pair ? load_flag(cv + a.NJ + 2 * q) : 0, v3 = pair ? load_flag(cv + a.NJ + 2 * q + 1) : 0;
                                    rdy = (r0 >= 4) & (r1 >= 4) & (c0 >= 4) & (c1 >= 4) & (c2 >= 4) & (c3 >= 4) & (v0 >= bt.p0) & ((bt.I == 2 * q) | (v1 >= bt.p0));
                                    two = rdy & pair & (r2 >= 4) & (r3 >= 4) & (v2 >= bt.p0) & (v3 >= bt.p0) & (f.bulk_pairs != 0);
                                } else {
                                    rdy = near_ready(bt.p0, bt.I, bt.J);
                                }
                                if (!rdy) tsize = 0;
                            }
                            // the two columns next to the chain first, then the OLDEST round (a column that falls rounds behind
                            // becomes a chain of its own at the end: its rounds run one after the other on every tile)
                            if (tsize > 0) key = q <= Sh + 2 ? q : 1024 + flow_round_release_stage(q, r) * 256 + q;
                        }
                        unsigned long long left = __builtin_amdgcn_ballot_w64(tsize > 0);
                        unfinished = unfinished || __builtin_amdgcn_ballot_w64(more) != 0;
                        for (int tries = 0; tries < 4 && left && kind == 0; tries++) {
                            int best = -1, best_key = 0x7fffffff;
                            for (unsigned long long m = left; m; m &= m - 1) {
                                const int l = __builtin_ctzll(m);
                                const int k = __builtin_amdgcn_readlane(key, l);
                                if (k < best_key) best_key = k, best = l;
                            }
                            left &= ~(1ull << best);
                            int tk = -1, got = 1;
                            if (lane == best) {
                                got = two ? 2 : 1;
                                tk = __hip_atomic_fetch_add(S + f.off_rcnt + q * f.RMAX + r, got, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (tk >= tsize) tk = -1;
                                else if (tk + got > tsize) got = tsize - tk;
                            }
                            tk = __builtin_amdgcn_readlane(tk, best);
                            if (tk >= 0) kind = 2, w0 = qb + best, w1 = __builtin_amdgcn_readlane(r, best), w2 = tk, w3 = __builtin_amdgcn_readlane(got, best);
                        }
                    }
                    // exhausted: the last head is open and fully claimed, every tail is exhausted, no column has a round left
                    if (kind == 0 && !unfinished && Sh == a.NP - 1 && exhausted && load_flag(S + f.off_open + 2) >= a.NP) {
                        bool any = false;
                        for (int qb = 2; qb < a.NP; qb += 64) {
                            const int q = qb + lane;
                            const bool m = q < a.NP && load_flag(S + f.off_rcur + q) < flow_nrounds(q);
                            any = any || __builtin_amdgcn_ballot_w64(m) != 0;
                        }
                        if (!any) kind = 4;
                    }
                }
            }
            if (kind == 0 && w0 == 0) {  // idle: nothing to take right now
                const long long now = wall_clock64();
                if (idle_since == 0) idle_since = now;
                else if (now - idle_since > a.timeout) {
                    if (lane == 0 && __hip_atomic_fetch_add(a.sync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                        store_flag(a.sync + 4, (11 << 28) | (stage_guess << 20) | (load_flag(S + f.off_open + 2) & 0xfffff));
                        store_flag(a.sync + 5, f.off_open);
                        store_flag(a.sync + 6, a.NP - 1);
                        store_flag(a.sync + 7, load_flag(S + f.off_open));
                    }
                    kind = 5;
                }
                if (kind == 0) __builtin_amdgcn_s_sleep(32);
            } else {
                idle_since = 0;
            }
            if (lane == 0) s_ctl[8] = kind, s_ctl[9] = w0, s_ctl[10] = w1, s_ctl[11] = w2, s_ctl[13] = w3;
        }
        __syncthreads();
        const int kind = s_ctl[8], w0 = s_ctl[9], w1 = s_ctl[10], w2 = s_ctl[11], w3 = s_ctl[13];
        __syncthreads();
        if (kind == 0) continue;
        if (kind == 3) return 3;
        if (kind == 4) return 0;
        if (kind == 5) return 2;
        for (int rep = 0; rep < (kind == 2 ? w3 : 1); rep++) {
            long long *tr = nullptr;
#ifdef EGX_PIPE_TRACE
            if (a.trace) {
                if (tid == 0) s_ctl[12] = __hip_atomic_fetch_add(a.sync + f.off_trace, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __syncthreads();
                const int slot = s_ctl[12];
                __syncthreads();
                if (slot < f.trace_cap) tr = a.trace + 8 * (int64_t)slot;
            }
#endif
            int r;
            if (kind == 1) {
                const PipeTask task = w2 == 0 ? a.tasks[f.coff[w0] + w1] : (w2 == 1 ? flow_tt_task(w0, w1) : flow_lt_task(w0, w1));
                if (tr && tid == 0) {
                    tr[0] = task.type | (task.p << 8) | (z << 16) | ((long long)w2 << 32);
                    tr[1] = (unsigned)task.a | ((long long)task.b << 32);
                    tr[2] = wall_clock64();
                    tr[5] = __builtin_amdgcn_s_getreg((3 << 11) | 20) | ((long long)blockIdx.x << 8);
                }
                if (task.type == PT_TRSM) r = pipe_role_trsm<1>(ka, (pipe_lds_t)sm, tr, z, task.p, task.a);
                else if (task.type == PT_FINE) r = pipe_role_fine(ka, (pipe_lds_t)sm, tr, z, task.p, task.a, task.b);
                else r = pipe_role_coarse(ka, (pipe_lds_t)sm, tr, z, task.p, task.a, task.b, task.type == PT_COARSE_LAST ? 1 : 0);
            } else {
                const FlowBulkTask bt = flow_round_task(sh, w0, w1, w2 + rep);
                if (tr && tid == 0) {
                    tr[0] = bt.type | (bt.p0 << 8) | (z << 16) | ((long long)w0 << 24);
                    tr[1] = (unsigned)bt.I | ((long long)bt.J << 32);
                    tr[2] = wall_clock64();
                    tr[5] = __builtin_amdgcn_s_getreg((3 << 11) | 20) | ((long long)blockIdx.x << 8);
                }
                if (bt.type == PT_BULK) r = pipe_role_bulk(ka, (pipe_lds_t)sm, tr, z, w0, bt.I, bt.p0, bt.p1);
                else r = pipe_role_coarse(ka, (pipe_lds_t)sm, tr, z, bt.p0, bt.I, bt.J, 0);
            }
            if (tr && tid == 0) tr[4] = wall_clock64();
            if (r == 2) return 2;
        }
    }
}

// One workgroup per compute unit.  The first NP to START own the diagonal blocks 0 .. NP - 1 (start ticket, as in k_potrf_pipe):
// each works as everybody else until its block is near, factors it INLINE (rb_factor_block: no call, no spill), and is a
// worker again.  One matrix per launch (z = 0): lock-step batches keep the schedules of rounds 3 to 5.
__global__ __launch_bounds__(1024) void k_potrf_flow(FlowArgs f) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const PipeArgs &a = f.p;
    int *s_ctl = reinterpret_cast<int *>(sm + kPipeLdsDoubles);
    if (threadIdx.x == 0)
        s_ctl[3] = __hip_atomic_fetch_add(a.sync + kPipeHdr + a.NP, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int start = s_ctl[3];
    __syncthreads();
    const flow_kargs_t kf = (flow_kargs_t)__builtin_amdgcn_kernarg_segment_ptr();
    if (start < a.NP) {
        const int p = start, k0 = 256 * p;
        int rc = 3;
        if (p > f.lead_short) rc = flow_worker_loop(kf, (pipe_lds_t)sm, 0, p);
        if (rc == 2) return;
        if (rc == 3) {
            int *info = a.info;
            int *S = a.sync;
#ifdef EGX_PIPE_TRACE
            long long *tr = nullptr;
            if (a.trace) {
                if (threadIdx.x == 0) s_ctl[12] = __hip_atomic_fetch_add(a.sync + f.off_trace, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __syncthreads();
                const int slot = s_ctl[12];
                __syncthreads();
                if (slot < f.trace_cap) tr = a.trace + 8 * (int64_t)slot;
            }
            if (tr && threadIdx.x == 0) {
                tr[0] = PT_DIAG | (p << 8);
                tr[1] = 0;
                tr[2] = wall_clock64();
                tr[5] = __builtin_amdgcn_s_getreg((3 << 11) | 20) | ((long long)blockIdx.x << 8);
            }
#endif
            const int r = pipe_wg_wait(s_ctl, [&]() {
                int rr = load_flag(info) != 0 ? 1 : 0;
                const int *fc = S + a.off_fcnt + p * a.NC + (k0 >> 6);
                for (int i = 0; i < 4 && rr == 0 && p > 0; i++) rr = pipe_wait_ge(fc + i, i + 1, a.sync, info, a.timeout, (8 << 28) | (p << 20));
                return rr;
            });
            if (r == 2) return;
            if (r == 0) {
                RbPublish pub;
                pub.base = k0 >> 4;
                pub.strips = S + 1;
                if (a.stall && pub.base + 16 >= a.stall) pub.strips = S + 2;  // (test hook: this block's strips are never seen)
#ifdef EGX_PIPE_TRACE
                pub.trace = tr;
                if (tr && threadIdx.x == 0) tr[3] = wall_clock64();
#endif
                (void)rb_factor_block<16, true>(a.M + (int64_t)k0 * a.ld + k0, a.ld, 256, a.dinv + (int64_t)(k0 / 64) * 4096, info, f.col_base + k0,
                                                f.col_base + a.n_pad, sm, pub);
#ifdef EGX_PIPE_TRACE
                if (tr && threadIdx.x == 0) tr[4] = wall_clock64();
#endif
            } else if (threadIdx.x == 0) {
                // (the matrix lost a pivot earlier: the block is not factored, but its strips count as published so that the
                //  workgroups ahead of their own blocks leave their loops and the launch ends)
                __hip_atomic_fetch_max(S + 1, (k0 >> 4) + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
        }
    }
    (void)flow_worker_loop(kf, (pipe_lds_t)sm, 0, -1);
}

// =============================================================================================
// host side
// =============================================================================================
// Run-time settings (egx_set_tuning / environment): EGX_PIPE, EGX_PIPE_TIMEOUT_MS and EGX_PIPE_RETRY (gp_host.hip).  What else
// was switchable while the launch was developed is a constant now, with its measurement: workgroups of a launch (one per
// compute unit; 96 beside other launches of the same factorisation), the queue order of the coarse updates (16 panels ahead =
// right-looking), the size up to which the chain is one launch (schedule.h) -- profiles/r05_pipe_*.txt, docs/HISTORY.md §B.
static std::atomic<int> g_pipe{1};               // EGX_PIPE: 0 separate launches everywhere, 1 by the handle's schedule (schedule.h), 2 chain launches per group only
static std::atomic<int> g_pipe_timeout_ms{2000};  // EGX_PIPE_TIMEOUT_MS: bound of every wait inside the launch
constexpr int kPipeSharedWgs = 96;  // workgroups of a chain launch that runs beside other launches of its factorisation
constexpr int kPipeLookAhead = 16;  // how many panels ahead of their column's factorisation the coarse updates are queued
#ifdef EGX_TEST_HOOKS
// TEST HOOKS: compiled into egobox_amd/lib/_dev/libegx_gp_hip_testhooks.so (and tools/pipe_check) only -- the product library
// has neither the "pipe_stall" knob nor a way to set the grid or a trace buffer.
static std::atomic<int> g_pipe_stall{0};   // egx_set_tuning "pipe_stall": see PipeArgs::stall
static long long *g_pipe_trace = nullptr;  // profiling buffer of the NEXT launches (pipe_set_trace; tools/pipe_check)
static int g_pipe_test_wgs = 0;            // the grid of the next chain launches (the same bits on 3 workgroups and on 256)
static int g_pipe_trace_cap = 0;           // slots of that buffer (flow launches allocate trace slots from a counter)
static int g_flow_lead_short = -1, g_flow_lead_long = -1;  // tools/flow_check: FlowArgs::lead_* of the next flow launches
void pipe_set_trace(long long *buf) { g_pipe_trace = buf; }
void pipe_set_trace_cap(int slots) { g_pipe_trace_cap = slots; }
void pipe_test_set_workgroups(int wgs) { g_pipe_test_wgs = wgs; }
void flow_test_set_leads(int lead_short, int lead_long) { g_flow_lead_short = lead_short, g_flow_lead_long = lead_long; }
#endif

// one-time setup; EGX_SUCCESS, or EGX_ERR_HIP when the launch cannot get its dynamic LDS (the caller then takes separate launches)
static int pipe_init() {
    static std::once_flag once;
    static int rc_once = EGX_SUCCESS;
    std::call_once(once, [] {
        if (const char *e = std::getenv("EGX_PIPE")) g_pipe = std::atoi(e);
        if (const char *e = std::getenv("EGX_PIPE_TIMEOUT_MS")) g_pipe_timeout_ms = std::atoi(e) > 0 ? std::atoi(e) : 1;
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_potrf_pipe), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 kPipeLdsBytes);
        const hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_potrf_flow), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                  kPipeLdsBytes);
        if (e != hipSuccess || e2 != hipSuccess) {
            (void)hipGetLastError();
            rc_once = EGX_ERR_HIP;
        }
    });
    return rc_once;
}

int pipe_set_knob(const char *name, int value) {
    (void)pipe_init();
    struct { const char *n; std::atomic<int> *v; } tab[] = {{"pipe", &g_pipe}, {"pipe_timeout_ms", &g_pipe_timeout_ms},
#ifdef EGX_TEST_HOOKS
                                                           {"pipe_stall", &g_pipe_stall},
#endif
    };
    for (auto &e : tab)
        if (std::string(name) == e.n) return e.v->exchange(value);
    return INT_MIN;
}
long long pipe_timeout_ticks() { return pipe_init() == EGX_SUCCESS ? (long long)g_pipe_timeout_ms * 100000ll : 200000000ll; }
int pipe_enabled() { return pipe_init() == EGX_SUCCESS ? g_pipe.load() : 0; }

// compute units of a device (per device: a process may drive a partitioned and a whole GPU)
static int pipe_device_cus(int dev) {
    static std::mutex mu;
    static std::map<int, int> cus;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cus.find(dev);
    if (it != cus.end()) return it->second;
    hipDeviceProp_t prop;
    int n = 256;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) n = prop.multiProcessorCount;
    else (void)hipGetLastError();
    cus[dev] = n;
    return n;
}
// does a chain launch over `np` panels of `nz` matrices fit the CURRENT device?  (every diagonal block has a workgroup of its
// own, resident from the start, one workgroup per compute unit: launch_potrf takes separate launches when this says no)
bool pipe_fits(int nz, int np) {
    int dev = 0;
    if (pipe_init() != EGX_SUCCESS || hipGetDevice(&dev) != hipSuccess) return false;
    return (long long)nz * np + 1 <= pipe_device_cus(dev);
}

// (the task list of a launch and its order: pipe_tasks.h, host-testable)
// A plan = the task list of one (device, shape, group) on the device.  Built when a handle is created (pipe_prepare: no
// allocation and no copy inside the first launch), kept for the life of the process or until egx_trim (pipe_release_plans).
struct PipePlan {
    PipeTask *d_tasks = nullptr;
    int ntasks = 0;
};
static std::mutex g_plan_mu;
static std::map<std::tuple<int, int, int, int, int, int>, PipePlan> g_plans;

static int pipe_plan_get(int dev, int n_pad, int m_tot, int g0, int np, int rt, PipePlan &out) {
    std::lock_guard<std::mutex> lock(g_plan_mu);
    const auto key = std::make_tuple(dev, n_pad, m_tot, g0, np, rt);
    auto it = g_plans.find(key);
    if (it == g_plans.end()) {
        const std::vector<PipeTask> tasks = pipe_tasks(n_pad, m_tot, g0, np, rt, kPipeLookAhead);
        PipePlan pl;
        pl.ntasks = (int)tasks.size();
        if (pl.ntasks > 0) {
            EGX_HIP_CHECK(dev_malloc(&pl.d_tasks, sizeof(PipeTask) * tasks.size()));
            // (synchronous: the source is a temporary.  Handles prepare their plans at creation, so this is not on a launch path
            //  except for callers without a handle -- egx_potrf, the tools -- and after egx_trim)
            const hipError_t e = hipMemcpy(pl.d_tasks, tasks.data(), sizeof(PipeTask) * tasks.size(), hipMemcpyHostToDevice);
            if (e != hipSuccess) {
                (void)hipFree(pl.d_tasks);
                EGX_HIP_CHECK(e);
            }
        }
        it = g_plans.emplace(key, pl).first;
    }
    out = it->second;
    return EGX_SUCCESS;
}
// ---- flow launch (pipe_flow.h): the critical lists of all stages, one after the other, and where each stage starts
struct FlowPlan {
    PipeTask *d_tasks = nullptr;
    int *d_coff = nullptr;
    int ntasks = 0;
};
static std::map<std::tuple<int, int, int>, FlowPlan> g_flow_plans;  // (device, n_pad, m_tot); under g_plan_mu
static int flow_plan_get(int dev, int n_pad, int m_tot, FlowPlan &out) {
    std::lock_guard<std::mutex> lock(g_plan_mu);
    const auto key = std::make_tuple(dev, n_pad, m_tot);
    auto it = g_flow_plans.find(key);
    if (it == g_flow_plans.end()) {
        const int NP = n_pad / 256;
        std::vector<PipeTask> tasks;
        std::vector<int> coff((size_t)NP + 1, 0);  // first head task of every stage
        for (int s = 0; s < NP; s++) {
            const std::vector<PipeTask> st = flow_stage_tasks(n_pad, m_tot, s);
            coff[(size_t)s] = (int)tasks.size();
            tasks.insert(tasks.end(), st.begin(), st.end());
        }
        coff[(size_t)NP] = (int)tasks.size();
        FlowPlan pl;
        pl.ntasks = (int)tasks.size();
        EGX_HIP_CHECK(dev_malloc(&pl.d_tasks, sizeof(PipeTask) * (tasks.size() + 1)));
        hipError_t e = dev_malloc(&pl.d_coff, sizeof(int) * coff.size());
        if (e == hipSuccess) e = hipMemcpy(pl.d_tasks, tasks.data(), sizeof(PipeTask) * tasks.size(), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(pl.d_coff, coff.data(), sizeof(int) * coff.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(pl.d_tasks);
            if (pl.d_coff) (void)hipFree(pl.d_coff);
            EGX_HIP_CHECK(e);
        }
        it = g_flow_plans.emplace(key, pl).first;
    }
    out = it->second;
    return EGX_SUCCESS;
}
// the plans a handle of this shape and schedule will launch, on the current device (egx_gp_create / egx_gp_set_lockstep)
int pipe_prepare(int n_pad, int m_tot, const PotrfSchedule &sched) {
    if ((!sched.pipe && !sched.flow && !sched.flow_tail) || pipe_init() != EGX_SUCCESS) return EGX_SUCCESS;
    int dev = 0;
    EGX_HIP_CHECK(hipGetDevice(&dev));
    if (sched.flow) {
        FlowPlan fp;
        return flow_plan_get(dev, n_pad, m_tot, fp);
    }
    if (sched.flow_tail > 0 && sched.flow_tail < n_pad) {
        FlowPlan fp;
        return flow_plan_get(dev, sched.flow_tail, m_tot - (n_pad - sched.flow_tail), fp);
    }
    PipePlan pl;
    if (sched.whole) return pipe_plan_get(dev, n_pad, m_tot, 0, (n_pad + 255) / 256, 1, pl);
    const int GW = sched.group_panels * 256;
    for (int g0 = 0; g0 < n_pad; g0 += GW) {
        const int gw = (n_pad - g0 < GW) ? (n_pad - g0) : GW;
        EGX_RC_PIPE(pipe_plan_get(dev, n_pad, m_tot, g0, (gw + 255) / 256, 1, pl));
    }
    return EGX_SUCCESS;
}
// egx_trim: the task lists go with the pooled handles (every device is synchronised first: a launch in flight reads its list).
// Returns the bytes given back.
size_t pipe_release_plans() {
    std::lock_guard<std::mutex> lock(g_plan_mu);
    size_t bytes = 0;
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    int synced = -1;
    for (auto &kv : g_plans) {
        const int dev = std::get<0>(kv.first);
        if (!kv.second.d_tasks) continue;
        if (dev != synced) {
            if (hipSetDevice(dev) != hipSuccess) continue;
            (void)hipDeviceSynchronize();
            synced = dev;
        }
        (void)hipFree(kv.second.d_tasks);
        bytes += sizeof(PipeTask) * (size_t)kv.second.ntasks;
    }
    g_plans.clear();
    for (auto &kv : g_flow_plans) {
        const int dev = std::get<0>(kv.first);
        if (dev != synced) {
            if (hipSetDevice(dev) != hipSuccess) continue;
            (void)hipDeviceSynchronize();
            synced = dev;
        }
        if (kv.second.d_tasks) (void)hipFree(kv.second.d_tasks);
        if (kv.second.d_coff) (void)hipFree(kv.second.d_coff);
        bytes += sizeof(PipeTask) * (size_t)kv.second.ntasks;
    }
    g_flow_plans.clear();
    if (have_cur) (void)hipSetDevice(cur);
    (void)hipGetLastError();
    return bytes;
}

bool flow_fits(int n_pad) {  // one workgroup per diagonal block + workers on the CURRENT device; 256-column panels throughout
    int dev = 0;
    if (n_pad % 256 || n_pad / 256 > 254 || pipe_init() != EGX_SUCCESS || hipGetDevice(&dev) != hipSuccess) return false;
    return n_pad / 256 + 8 <= pipe_device_cus(dev);
}
// The whole factorisation of ONE matrix (pb.count == 1) as a flow launch.  The hand-off words (pb.sync: pipe_sync_ints ints,
// zeroed by launch_potrf) carry the chain launch's words and, behind them, the flow launch's (flow_layout).
int launch_potrf_flow(hipStream_t s, double *M, int64_t ld, int n_pad, int m_tot, double *dinv, int *info, const PotrfBatch &pb, int col_base) {
    if (pipe_init() != EGX_SUCCESS) {
        set_error("potrf_flow: the launch cannot get its dynamic LDS on this device");
        return EGX_ERR_HIP;
    }
    if (!pb.sync || n_pad % 256 || (m_tot - n_pad) % 128 || pb.count > 1) {
        set_error("potrf_flow: needs sync words, 256-column panels, padded right-hand-side rows and ONE matrix");
        return EGX_ERR_INVALID_VALUE;
    }
    int dev = 0;
    EGX_HIP_CHECK(hipGetDevice(&dev));
    FlowPlan plan;
    EGX_RC_PIPE(flow_plan_get(dev, n_pad, m_tot, plan));
    const PipeLayout l = pipe_layout(n_pad, m_tot);
    const FlowLayout fl = flow_layout(l.NP, l.total);
    FlowArgs f;
    PipeArgs &a = f.p;
    a.M = M, a.ld = ld, a.n_pad = n_pad, a.m_tot = m_tot, a.dinv = dinv, a.info = info, a.sync = pb.sync;
    a.sM = pb.sM, a.sD = pb.sD, a.sS = pb.sS, a.sI = pb.sI, a.nz = 1, a.g0 = 0, a.np = l.NP;
    a.tasks = plan.d_tasks, a.ntasks = plan.ntasks;
    a.NP = l.NP, a.NC = l.NC, a.NJ = l.NJ, a.off_rowT = l.off_rowT, a.off_fcnt = l.off_fcnt, a.off_cver = l.off_cver;
    a.rt = 1, a.ext_need = 0;
    a.timeout = (long long)g_pipe_timeout_ms * 100000ll;
    f.coff = plan.d_coff;
    f.NI = m_tot / 128;
    f.RMAX = fl.RMAX, f.off_open = fl.off_open, f.off_cnext = fl.off_cnext, f.off_rcur = fl.off_rcur, f.off_rcnt = fl.off_rcnt;
    f.off_lnext = fl.off_lnext, f.off_tnext = fl.off_tnext, f.off_trace = fl.off_trace;
    f.bulk_pairs = n_pad >= 8192 ? 1 : 0;  // (small matrices: one tile per claim keeps the last rounds balanced)
    f.col_base = col_base;
    f.lead_short = 1, f.lead_long = 3;
    f.trace_cap = 0;
    int test_wgs = 0;
#ifdef EGX_TEST_HOOKS
    a.stall = g_pipe_stall;
    a.trace = g_pipe_trace;
    f.trace_cap = g_pipe_trace ? g_pipe_trace_cap : 0;
    test_wgs = g_pipe_test_wgs;
    if (g_flow_lead_short >= 0) f.lead_short = g_flow_lead_short;
    if (g_flow_lead_long >= 0) f.lead_long = g_flow_lead_long;
    if (const char *e = std::getenv("EGX_DEV_FLOW_PAIRS")) f.bulk_pairs = std::atoi(e);
#else
    a.stall = 0;
    a.trace = nullptr;
#endif
    const int n_cu = pipe_device_cus(dev);
    int wgs = test_wgs > 0 ? test_wgs : n_cu;
    if (wgs > n_cu) wgs = n_cu;
    if (wgs < l.NP + 1) {
        set_error("potrf_flow: more diagonal blocks than compute units");
        return EGX_ERR_INVALID_VALUE;
    }
    hipLaunchKernelGGL(k_potrf_flow, dim3((unsigned)wgs), dim3(1024), kPipeLdsBytes, s, f);
    EGX_HIP_CHECK(hipGetLastError());
    return EGX_SUCCESS;
}

// word 3 of the batch's hand-off words <- value, in stream order (a one-thread kernel behind the launch it reports on)
__global__ void k_pipe_signal(int *word, int value) { store_flag(word, value); }
int pipe_signal(hipStream_t s, const PotrfBatch &pb, int value) {
    hipLaunchKernelGGL(k_pipe_signal, dim3(1), dim3(1), 0, s, pb.sync + 3, value);
    EGX_HIP_CHECK(hipGetLastError());
    return EGX_SUCCESS;
}

int launch_potrf_pipe(hipStream_t s, double *M, int64_t ld, int n_pad, int m_tot, double *dinv, int *info,
                      const PotrfBatch &pb, int g0, int gw, int ext_need, int shared_chip) {
    if (pipe_init() != EGX_SUCCESS) {
        set_error("potrf_pipe: the chain launch cannot get its dynamic LDS on this device");
        return EGX_ERR_HIP;
    }
    if (!pb.sync || g0 % 256 || gw <= 0 || (m_tot - n_pad) % 128 || n_pad % 128) {
        set_error("potrf_pipe: needs sync words, a group that starts on a panel boundary and padded sizes");
        return EGX_ERR_INVALID_VALUE;
    }
    const int np = (gw + 255) / 256;
    int dev = 0;
    EGX_HIP_CHECK(hipGetDevice(&dev));
    // (64 rows per solve task; the two-chunk form -- the fragments and the barriers serve two tiles per wave quadruple --
    //  spilled registers and was only meant for panels taller than the sizes chain launches are used for: removed)
    const int rt = 1;
    PipePlan plan;
    EGX_RC_PIPE(pipe_plan_get(dev, n_pad, m_tot, g0, np, rt, plan));
    const PipeLayout l = pipe_layout(n_pad, m_tot);
    PipeArgs a;
    a.M = M;
    a.ld = ld;
    a.n_pad = n_pad;
    a.m_tot = m_tot;
    a.dinv = dinv;
    a.info = info;
    a.sync = pb.sync;
    a.sM = pb.sM;
    a.sD = pb.sD;
    a.sS = pb.sS;
    a.sI = pb.sI;
    a.nz = pb.count > 0 ? pb.count : 1;
    a.g0 = g0;
    a.np = np;
    a.tasks = plan.d_tasks;
    a.ntasks = plan.ntasks;
    a.NP = l.NP;
    a.NC = l.NC;
    a.NJ = l.NJ;
    a.off_rowT = l.off_rowT;
    a.off_fcnt = l.off_fcnt;
    a.off_cver = l.off_cver;
    a.rt = rt;
    a.ext_need = ext_need;
    a.timeout = (long long)g_pipe_timeout_ms * 100000ll;
#ifdef EGX_TEST_HOOKS
    a.stall = g_pipe_stall;
    a.trace = g_pipe_trace;
    const int test_wgs = g_pipe_test_wgs;
#else
    a.stall = 0;
    a.trace = nullptr;
    const int test_wgs = 0;
#endif
    const int n_cu = pipe_device_cus(dev);
    // the DIAG workgroups (one per diagonal block and matrix, resident from the start) + workers, never more than one
    // workgroup per compute unit: every workgroup of the grid is resident, whatever the dispatch order
    const long long n_diag = (long long)a.nz * np, want = n_diag + (long long)a.nz * plan.ntasks;
    // (a chain launch that runs BESIDE the launches it waits for or shares the chip with -- look-ahead -- must leave them
    //  compute units: its workgroups fill one each and would otherwise spin on a launch that cannot start)
#ifdef EGX_DEV_KNOBS
    const int shared_wgs = dev_env("EGX_DEV_PIPE_SHARED_WGS", kPipeSharedWgs);
#else
    const int shared_wgs = kPipeSharedWgs;
#endif
    long long wgs = test_wgs > 0 ? test_wgs : (shared_chip ? (shared_wgs < n_cu ? shared_wgs : n_cu) : n_cu);
    if (wgs > n_cu) wgs = n_cu;
    if (wgs > want) wgs = want;
    if (wgs < n_diag + 1) wgs = n_diag + 1;
    if (n_diag + 1 > n_cu) {  // (launch_potrf asks pipe_fits first and never gets here)
        set_error("potrf_pipe: more diagonal blocks in one chain launch than compute units");
        return EGX_ERR_INVALID_VALUE;
    }
    hipLaunchKernelGGL(k_potrf_pipe, dim3((unsigned)wgs), dim3(1024), kPipeLdsBytes, s, a);
    EGX_HIP_CHECK(hipGetLastError());
    return EGX_SUCCESS;
}

}  // namespace egx
