// Blocked right-looking FP64 Cholesky + triangular solves for gfx950 (MI355X).
//
// Replaces, on the device, what the reference gets from linfa-linalg 0.2.1
// `cholesky()` / `solve_triangular()` (crates/gp/src/algorithm.rs:1004-1034, :337-367) or LAPACK
// dpotrf/dtrtrs with the `blas` feature (:1077-1115).
//
// Structure per 256-column block k (right-looking, lower, row-major, in place; blocks are grouped 2 or 4 to a trailing
// update, with look-ahead on side streams: launch_potrf):
//   A  k_potf2_reg    : ONE workgroup of 16 waves factors the 256x256 diagonal block with its 16x16 tiles resident in
//                       FP64-MFMA accumulator registers; 16-column strips; the 16x16 diagonal tile by a DPP-broadcast
//                       pivot chain on one wave, fused with its inverse.
//   B  k_panel_trsm16 : rows below the block:  X = P * L_kk^-T, one 16-row tile per workgroup, strip by strip with the
//                       16x16 inverses A left behind (+ one refinement step for ill-conditioned tiles); the solves after
//                       the factorisation (launch_trsm_rows) use it too, with the 16x16 diagonal blocks of the 64x64 tile
//                       inverses.
//   C  k_gemm_stream  : chip-filling trailing updates  C -= P P^T  (lower tiles only) -- the n^3/3 flops -- 128x256
//      k_gemm_nt_sub    tiles, LDS-DMA ring; smaller launches by the register-staged 128x128 / 64x64 kernel:
//                       v_mfma_f64_16x16x4_f64, A/B staged through LDS in 16-deep K chunks.
//   k_diag_tile_inverses, once at the end: inverses of all 64x64 diagonal tiles (for launch_trsm_rows / k_block_inv256).
// Right-hand-side rows appended below the square matrix ride along in B and C, so after the
// factorisation they hold (C^-1 [F | y])^T: the forward solves of algorithm.rs:1006,1028 are fused
// into the factorisation (classic augmented-matrix trick) and cost no extra pass over C.
//
// LOCK-STEP BATCHES (round 3).  Every kernel of the factorisation takes a batch dimension: workgroup (x, y, z) works on
// matrix z, whose every pointer is the pointer of matrix 0 plus a fixed element stride (the workspaces of a handle are
// carved out of one slab).  The candidates of a theta sweep all have the same n and therefore the same launch
// schedule, so `launch_potrf` factors nb of them through ONE launch sequence: the serial chain (diagonal block ->
// panel solve -> in-group update) costs its latency once for nb matrices, every launch has nb x the tiles (the
// look-ahead updates that used to be too small for the LDS-DMA stream kernel now fill the chip), and the per-matrix
// arithmetic is unchanged -- a matrix gets bit for bit the factor it gets alone.
//
// FP64 MFMA on gfx950 runs at the FP64 vector rate (78.6 TFLOP/s chip peak, 64 cycles per
// 16x16x4 instruction per SIMD): the matrix core is used because one instruction carries 2048
// flops with two 8-byte operands per lane, which keeps LDS and issue pressure negligible.
#include "egx_internal.h"
#include "potf2_blocks.h"
#include "mfma_gemm_core.h"

#include <atomic>
#include <cstdlib>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

namespace egx {


__device__ __forceinline__ double readlane_d(double v, int lane) {
    union {
        double d;
        int i[2];
    } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
    return u.d;
}

// (the MFMA GEMM core -- GemmShape, gemm_core -- lives in mfma_gemm_core.h, shared with kernels_pipe.hip)
// ---------------------------------------------------------------------------------------------
// C: trailing update / general  C -= A B^T.  grid = (M/128, N/128).
// ---------------------------------------------------------------------------------------------
using TrailShape = GemmShape<128, 128, 32, 64, 512>;  // 8 waves, 2 workgroups per CU -> 4 MFMA waves per SIMD
using SmallShape = GemmShape<64, 64, 32, 32, 256>;    // 4x lower per-tile latency: look-ahead column + small trailing matrices

// A failed pivot anywhere earlier in this factorisation (algorithm.rs:893-896: the candidate is +inf): nothing left to
// compute, every later kernel of the factorisation returns at once.  The flag may be SET while this kernel starts (the
// look-ahead chain of the next group runs beside the trailing update), so ONE lane reads it and the workgroup decides
// together: waves of a workgroup never part ways in front of a barrier.
__device__ __forceinline__ bool wg_failed_before(const int *info) {
    __shared__ int s_failed;
    if (threadIdx.x == 0) s_failed = (info != nullptr) ? __hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    __syncthreads();
    return s_failed != 0;
}

template <bool LOWER, int BM, int BN, int WM, int WN, int NTHREADS, bool SWZ>
__global__ __launch_bounds__(NTHREADS, (NTHREADS == 512 ? 4 : 2)) void k_gemm_nt_sub(
    double *__restrict__ C, int64_t ldc, const double *__restrict__ A, int64_t lda, const double *__restrict__ B,
    int64_t ldb, int K, int nbx, int nby, int ktri, const int *__restrict__ info, GemmBatch bt) {
    using S = GemmShape<BM, BN, WM, WN, NTHREADS>;
    {   // lock-step batch: matrix blockIdx.z
        const int64_t z = blockIdx.z;
        C += z * bt.sC;
        A += z * bt.sA;
        B += z * bt.sB;
        if (info != nullptr) info += z * bt.sInfo;
    }
    if (wg_failed_before(info)) return;
    // XCD-aware tile order (SWZ).  Workgroup w lands on XCD w % 8 (observed dispatch order, used for speed only,
    // never for correctness).  Tiles are grouped in 8x8 super-tiles; super-tile ST goes to XCD ST % 8 and its 64
    // tiles are the 64 workgroups resident on that XCD (2 per CU), which then share 8 A and 8 B panel blocks
    // (4 MiB = the XCD's L2) instead of streaming the whole 33 MB panel through every L2.  For the lower-triangular
    // update only super-tiles that touch the lower triangle are enumerated, so the eight XCDs stay balanced.
    int bx, by;
    if (SWZ) {
        const int w = blockIdx.x;
        const int xcd = w & 7, q = w >> 3;
        const int ST = (q >> 6) * 8 + xcd, local = q & 63;
        const int nsx = (nbx + 7) >> 3, nsy = (nby + 7) >> 3;
        int si, sj;
        if (LOWER) {
            const int ntri = nsy * (nsy + 1) / 2;  // rows si < nsy hold si + 1 super-tiles, later rows hold nsy
            if (ST < ntri) {
                si = (int)((sqrt(8.0 * ST + 1.0) - 1.0) * 0.5);
                while (si * (si + 1) / 2 > ST) si--;
                while ((si + 1) * (si + 2) / 2 <= ST) si++;
                sj = ST - si * (si + 1) / 2;
            } else {
                si = nsy + (ST - ntri) / nsy;
                sj = (ST - ntri) % nsy;
            }
        } else {
            si = ST / nsy;
            sj = ST % nsy;
        }
        if (si >= nsx) return;
        bx = si * 8 + (local & 7);
        by = sj * 8 + (local >> 3);
        if (bx >= nbx || by >= nby) return;
    } else if (LOWER && (ktri & 2)) {
        // 1-D grid over the tiles that touch the lower triangle only (column `by` holds rows bx >= by * BN / BM):
        // workgroups that exit at once would still be dealt round-robin to the XCDs / shader engines and leave the
        // real tiles unevenly spread (measured 5..12 tiles per CU with a 2-D grid and early exits).
        constexpr int R = BN / BM;
        const int t = blockIdx.x;
        // tiles before column c: c * nbx - R * c (c - 1) / 2
        const double bq = nbx + 0.5 * R;
        int c = (int)((bq - sqrt(bq * bq - 2.0 * R * t)) / R);
        if (c < 0) c = 0;
        if (c >= nby) c = nby - 1;
        while (c > 0 && c * nbx - R * c * (c - 1) / 2 > t) c--;
        while (c + 1 < nby && (c + 1) * nbx - R * (c + 1) * c / 2 <= t) c++;
        by = c;
        bx = R * c + (t - (c * nbx - R * c * (c - 1) / 2));
        if (bx >= nbx) return;
    } else {
        bx = blockIdx.x;
        by = blockIdx.y;
    }
    if (LOWER && (bx + 1) * BM <= by * BN) return;  // tile entirely above the diagonal
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x;
    double4_t acc[S::MT][S::NT];
#pragma unroll
    for (int mi = 0; mi < S::MT; mi++)
#pragma unroll
        for (int ni = 0; ni < S::NT; ni++) acc[mi][ni] = double4_t{0.0, 0.0, 0.0, 0.0};
    // ktri: both operands are upper triangular (row i is zero left of column i), C lower: the K range of tile
    // (bx, by), bx >= by, starts at the first row of the tile (used for R^-1 = C^-T C^-1 in the theta-gradient)
    const int koff = (ktri & 1) ? bx * BM : 0;
    gemm_core<BM, BN, WM, WN, NTHREADS>(A + (int64_t)bx * BM * lda + koff, lda, B + (int64_t)by * BN * ldb + koff, ldb,
                                        K - koff, acc, smem, tid);
    const int wave = tid >> 6, lane = tid & 63;
    const int r0 = bx * BM + (wave / S::WAVES_N) * WM + (lane >> 4);
    const int c0 = by * BN + (wave % S::WAVES_N) * WN + (lane & 15);
    // read-modify-write of the C tile in batches of independent loads (a naive `*p -= acc` chain
    // serialises on vmcnt(0) per element: measured 2x on the whole kernel); the loads of batch mi + 1 are issued
    // before batch mi is stored, so the epilogue pays one memory latency plus issue time instead of MT latencies
    double cv[2][S::NT][4];
#pragma unroll
    for (int ni = 0; ni < S::NT; ni++)
#pragma unroll
        for (int r = 0; r < 4; r++) cv[0][ni][r] = C[(int64_t)(r0 + 4 * r) * ldc + (c0 + ni * 16)];
#pragma unroll
    for (int mi = 0; mi < S::MT; mi++) {
        if (mi + 1 < S::MT) {
#pragma unroll
            for (int ni = 0; ni < S::NT; ni++)
#pragma unroll
                for (int r = 0; r < 4; r++)
                    cv[(mi + 1) & 1][ni][r] = C[(int64_t)(r0 + (mi + 1) * 16 + 4 * r) * ldc + (c0 + ni * 16)];
        }
#pragma unroll
        for (int ni = 0; ni < S::NT; ni++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                C[(int64_t)(r0 + mi * 16 + 4 * r) * ldc + (c0 + ni * 16)] = cv[mi & 1][ni][r] - acc[mi][ni][r];
    }
}

// ---------------------------------------------------------------------------------------------
// C': the chip-filling trailing updates as a STREAM kernel (k_gemm_stream).  Same 128x256 workgroup tile / 64x64
// wave tile as the wide k_gemm_nt_sub, but:
//   * A/B panel chunks go global -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave-instruction): no VGPR
//     staging hop, no ds_write pass.  The LDS image of a chunk is UNPADDED, [384 rows][16 doubles] = 48 KB, and
//     bank-conflict free through an XOR swizzle of the 16-byte slot (slot' = slot ^ ((row >> 1) & 7)) that is applied
//     on the SOURCE address (the LDS-DMA destination is lane-linear) and again in the fragment reads.
//   * three chunk stages (144 KB) form a ring with a prefetch distance of TWO chunks; the only wait in the K loop is
//     a counted `s_waitcnt vmcnt(6)` (the newest chunk stays in flight) followed by ONE raw s_barrier per chunk.
//   * a workgroup walks over several tiles (w, w + G, ...): the chunk stream runs across tile boundaries, so the next
//     tile's first two chunks are already landing while the current tile is stored.
//   * the accumulators start as -C (loaded while the first chunks land) and the epilogue is stores only
//     (C_new = -(-C + A B^T)): no read-modify-write round trip at the end of a tile.
// ---------------------------------------------------------------------------------------------
constexpr int ST_ROWS = 384;                  // 128 A rows + 256 B rows per chunk
constexpr int ST_STAGE = ST_ROWS * KC;        // doubles per stage (48 KB)
constexpr int ST_NSTAGE = 3;
constexpr int ST_LDS_BYTES = ST_NSTAGE * ST_STAGE * 8;

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *gbl_ptr_t;

#include "tile_walks.h"  // tile index -> (bx, by): column-major, the KTRI row walk

// KTRI (with LOWER, square, ONE tile per workgroup): A == B == W upper triangular (row i is zero left of column i); the
// accumulators start at ZERO and C <- -(W W^T) is written without being read (no memset of C, no read pass).
// TAG: no effect on the code -- the left-looking group updates get kernel symbols of their own (1: the long update, 2: the
// short one), so that `rocprofv3 --kernel-trace --stats` lists the launches bench.py's roofline times as their own row.
#ifndef EGX_STREAM_ILV
#define EGX_STREAM_ILV 1  // the LDS-DMA issues of a chunk interleaved with the first MFMAs behind its barrier (0: all six in a row, rounds 2-5)
#endif
#ifdef EGX_STREAM_TRACE
// Profiling builds only (tools/dev_build.sh trace -> egobox_amd/lib/_dev/libegx_gp_hip_trace.so; tools/long_update_attribution.py):
// every tile of the stream kernel (tag 1 / 2: the left-looking long / short group updates; 0: other LOWER launches; 4: rectangles --
// the in-group updates) leaves one record of kStraceWords words --
// where it ran, the 100-MHz wall clock at its start, around its K loop and at its end, the shader-clock counter around the K loop.
// [0] = records taken so far, [1] = capacity; records from word 16 on.
constexpr int kStraceWords = 10;
__device__ long long *g_strace = nullptr;
#endif
template <bool LOWER, bool KTRI = false, int TAG = 0>
__global__ __launch_bounds__(512, 2) void k_gemm_stream(double *__restrict__ C, int64_t ldc, const double *__restrict__ A,
                                                        int64_t lda, const double *__restrict__ B, int64_t ldb, int K,
                                                        int nbx, int nby, int ntiles, const int *__restrict__ info,
                                                        GemmBatch bt) {
    int G = gridDim.x, bid = (int)blockIdx.x;
    {   // lock-step batch: matrix blockIdx.z
        int64_t z = blockIdx.z;
        if (KTRI && !LOWER) {
            // the rectangle's tiles differ in their K ranges by two orders of magnitude: the matrices of a batch are
            // INTERLEAVED in the dispatch order (grid.x = tiles x matrices), so the heavy tiles of every matrix start first
            const int nzc = bt.count > 0 ? bt.count : 1;
            z = bid % nzc;
            bid = bid / nzc;
            G = ntiles;
        }
        C += z * bt.sC;
        A += z * bt.sA;
        B += z * bt.sB;
        if (info != nullptr) info += z * bt.sInfo;
    }
    if (wg_failed_before(info)) return;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto tile_of = [&](int t, int &bx, int &by) {  // tile index -> (bx, by)
        if (KTRI && LOWER) stream_tile_coords_ktri(t, bx, by);
        else if (KTRI) stream_tile_coords_ktri_rect(t, nby, bx, by);
        else stream_tile_coords<LOWER>(t, nbx, nby, bx, by);
    };
    int nch = K / KC;
    if (KTRI) {  // one tile per workgroup (G == ntiles): its K range starts at the tile's first row
        int bx, by;
        tile_of(bid, bx, by);
        const int koff = bx * 128;
        A += koff;
        B += koff;
        nch = (K - koff) / KC;
    }
    const int my_tiles = (ntiles - bid + G - 1) / G;
    const int total = my_tiles * nch;  // chunks this workgroup streams

    // ---- load side: wave w fills the 8-row groups w, w + 8 (A) and 16 + w + 8 i (B) of every stage
    const int lrow = lane >> 3;
    const int sw_ld = (4 * wave + (lane >> 4)) & 7;  // ((row >> 1) & 7) of this lane's row, the same for all 6 groups
    const unsigned lpart = (unsigned)(((lane & 7) ^ sw_ld) * 2);
    unsigned offA[2], offB[4];
#pragma unroll
    for (int i = 0; i < 2; i++) offA[i] = (unsigned)((8 * (wave + 8 * i) + lrow) * (int)lda) + lpart;
#pragma unroll
    for (int i = 0; i < 4; i++) offB[i] = (unsigned)((8 * (wave + 8 * i) + lrow) * (int)ldb) + lpart;
    int t_l = bid, ch_l = 0, issued = 0;
    const double *pA_l, *pB_l;
    {
        int bx, by;
        tile_of(t_l, bx, by);
        pA_l = A + (int64_t)bx * 128 * lda;
        pB_l = B + (int64_t)by * 256 * ldb;
    }
    // pieces [p0, p1) of the six (0, 1: A groups; 2..5: B groups) of the load cursor's chunk into `stage`
    auto issue_pieces = [&](int stage, int p0, int p1) {
        double *dst = smem + stage * ST_STAGE + wave * 128;  // group g starts at g * 8 rows * 16 doubles
        const double *ga = pA_l + ch_l * KC, *gb = pB_l + ch_l * KC;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            if (i < p0 || i >= p1) continue;
            if (i < 2)
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(ga + offA[i]), (lds_ptr_t)(dst + i * 1024), 16, 0, 0);
            else
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(gb + offB[i - 2]), (lds_ptr_t)(dst + 2048 + (i - 2) * 1024), 16,
                                                 0, 0);
        }
    };
    auto issue_done = [&]() {
        issued++;
        if (++ch_l == nch) {
            ch_l = 0;
            t_l += G;
            if (t_l < ntiles) {
                int bx, by;
                tile_of(t_l, bx, by);
                pA_l = A + (int64_t)bx * 128 * lda;
                pB_l = B + (int64_t)by * 256 * ldb;
            }
        }
    };
    auto issue = [&](int stage) {
        issue_pieces(stage, 0, 6);
        issue_done();
    };
    if (total > 0) issue(0);
    if (total > 1) issue(1);

    // ---- compute side
    const int wm0 = (wave >> 2) * 64, wn0 = (wave & 3) * 64;
    const int frow = lane & 15, fk = lane >> 4;
    // fragment reads are 16 bytes per lane (ds_read_b128: two consecutive k of one row), so a lane group fk feeds TWO
    // MFMAs from one read: within an 8-deep k block the first MFMA contracts k = 0, 2, 4, 6 and the second k = 1, 3, 5, 7
    // (the same permutation on A and B leaves the sum unchanged).  slot' = (4 kb + fk) ^ ((row >> 1) & 7) is
    // conflict free for the b128 lane groups.
    int foff[2];
    foff[0] = frow * KC + ((fk ^ ((frow >> 1) & 7)) << 1);
    foff[1] = foff[0] ^ 8;
    const int aoff = wm0 * KC, boff = (128 + wn0) * KC;
    unsigned coff[4];
#pragma unroll
    for (int r = 0; r < 4; r++) coff[r] = (unsigned)(((lane >> 4) + 4 * r) * (int)ldc + (lane & 15));

    int g = 0, stage = 0;  // global chunk counter of this workgroup, stage = g % 3
    d2_t a0[4], b0[4];     // fragments of the coming half chunk, read one half ahead (live across tiles)
#pragma unroll
    for (int i = 0; i < 4; i++) a0[i] = b0[i] = d2_t{0.0, 0.0};
    for (int t = bid; t < ntiles; t += G) {
        int bx, by;
        tile_of(t, bx, by);
        double *Ct = C + (int64_t)(bx * 128 + wm0) * ldc + by * 256 + wn0;
#ifdef EGX_STREAM_TRACE
        long long *trp = nullptr;
        if (tid == 0 && g_strace != nullptr) {
            const unsigned long long slot = atomicAdd(reinterpret_cast<unsigned long long *>(g_strace), 1ull);
            if ((long long)slot < g_strace[1]) trp = g_strace + 16 + kStraceWords * slot;
        }
        if (trp) {
            trp[0] = (long long)(C - (int64_t)blockIdx.z * bt.sC);  // the launch: its first matrix' C
            trp[1] = ((long long)(TAG | (LOWER ? 0 : 4)) << 60) | ((long long)blockIdx.z << 52) | ((long long)nch << 32) | (long long)t;
            trp[2] = (long long)__builtin_amdgcn_s_getreg(0xF804) | ((long long)__builtin_amdgcn_s_getreg(0xF814) << 32);  // HW_ID, XCC_ID
            trp[3] = wall_clock64();
            trp[9] = ((long long)bx << 32) | by;
        }
#endif
        double4_t acc[4][4];
#pragma unroll
        for (int mi = 0; mi < 4; mi++)
#pragma unroll
            for (int ni = 0; ni < 4; ni++)
#pragma unroll
                for (int r = 0; r < 4; r++) acc[mi][ni][r] = KTRI ? 0.0 : -Ct[(int64_t)mi * 16 * ldc + ni * 16 + coff[r]];
        // MID-CHUNK barrier (the survivor of five orderings, profiles/r02_run11_gemm_lab_variants_0_to_4.txt): the synchronisation for chunk g + 1 sits between the two 8-deep halves of chunk g, the
        // fragments of a half are read one half ahead, so the MFMA stream runs across the barrier and no LDS read
        // latency is exposed behind it:
        //   [read kb1(g)] [32 MFMA kb0(g)] [wait own loads of g + 1; barrier] [issue loads g + 2] [read kb0(g + 1)]
        //   [32 MFMA kb1(g)]
        // RAW: kb0/kb1(g + 1) are read after the barrier that follows every wave's wait for chunk g + 1.  WAR: the
        // loads of g + 2 overwrite the stage of g - 1, whose last reads were issued before the previous barrier.
        auto read_half = [&](int st, int kb, d2_t (&a)[4], d2_t (&b)[4]) {
            const double *As = smem + st * ST_STAGE + aoff, *Bs = smem + st * ST_STAGE + boff;
#pragma unroll
            for (int mi = 0; mi < 4; mi++) a[mi] = *reinterpret_cast<const d2_t *>(As + mi * 16 * KC + foff[kb]);
#pragma unroll
            for (int ni = 0; ni < 4; ni++) b[ni] = *reinterpret_cast<const d2_t *>(Bs + ni * 16 * KC + foff[kb]);
        };
        auto mma_quarter = [&](const d2_t (&a)[4], const d2_t (&b)[4], int h) {
#pragma unroll
            for (int mi = 0; mi < 4; mi++)
#pragma unroll
                for (int ni = 0; ni < 4; ni++)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi][h], b[ni][h], acc[mi][ni], 0, 0, 0);
        };
        auto mma_half = [&](const d2_t (&a)[4], const d2_t (&b)[4]) {
            mma_quarter(a, b, 0);
            mma_quarter(a, b, 1);
        };
        // MFMA i of a quarter, in mma_quarter's order (mi outer, ni inner)
#define EGX_ST_MMA(a, b, h, i) \
    acc[(i) >> 2][(i) & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[(i) >> 2][h], b[(i) & 3][h], acc[(i) >> 2][(i) & 3], 0, 0, 0)
        if (g == 0) {  // the workgroup's very first chunk: the classic wait + barrier in front of its first read
            if (issued > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            read_half(0, 0, a0, b0);
        }
#ifdef EGX_STREAM_TRACE
        if (trp) trp[4] = wall_clock64(), trp[5] = clock64();
#endif
        for (int ch = 0; ch < nch; ch++, g++) {
            d2_t a1[4], b1[4];
            read_half(stage, 1, a1, b1);
            mma_half(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            const bool next = g + 1 < total;   // another chunk follows (possibly the next tile's first)
            const bool more = issued < total;  // ... and one more to prefetch
            const int st1 = (stage == ST_NSTAGE - 1) ? 0 : stage + 1;
#if EGX_STREAM_ILV
            // round 6: behind the barrier every wave of the workgroup stands at the same instruction; six LDS-DMA issues in a row
            // (address, M0, wait state, load: ~25 instructions) left all four MFMA pipes of the CU empty for as long.  The pieces
            // of chunk g + 2 (-> stage of chunk g - 1) now go out one by one BETWEEN the first MFMAs behind the barrier, whose
            // fragments were read before it (in-situ trace: MFMA issue inside the K loop 0.930 before)
            if (next) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // own pieces of chunk g + 1 (the only ones in flight)
                __builtin_amdgcn_s_barrier();
            }
            __builtin_amdgcn_sched_barrier(0);
            {   // (the MFMAs stay outside the branches: inside, the accumulators met again as 128 phi copies -- spills)
                const bool ld = next && more;
                const int ist = stage == 0 ? 2 : stage - 1;
                // 3, 2, 2, 2, 2, 2 MFMAs in front of pieces 0 .. 5, three behind
#define EGX_ST_PIECE(q) __builtin_amdgcn_sched_barrier(0); if (ld) issue_pieces(ist, q, q + 1); __builtin_amdgcn_sched_barrier(0)
                EGX_ST_MMA(a1, b1, 0, 0); EGX_ST_MMA(a1, b1, 0, 1); EGX_ST_MMA(a1, b1, 0, 2);
                EGX_ST_PIECE(0);
                EGX_ST_MMA(a1, b1, 0, 3); EGX_ST_MMA(a1, b1, 0, 4);
                EGX_ST_PIECE(1);
                EGX_ST_MMA(a1, b1, 0, 5); EGX_ST_MMA(a1, b1, 0, 6);
                EGX_ST_PIECE(2);
                EGX_ST_MMA(a1, b1, 0, 7); EGX_ST_MMA(a1, b1, 0, 8);
                EGX_ST_PIECE(3);
                EGX_ST_MMA(a1, b1, 0, 9); EGX_ST_MMA(a1, b1, 0, 10);
                EGX_ST_PIECE(4);
                EGX_ST_MMA(a1, b1, 0, 11); EGX_ST_MMA(a1, b1, 0, 12);
                EGX_ST_PIECE(5);
                EGX_ST_MMA(a1, b1, 0, 13); EGX_ST_MMA(a1, b1, 0, 14); EGX_ST_MMA(a1, b1, 0, 15);
#undef EGX_ST_PIECE
            }
            __builtin_amdgcn_sched_barrier(0);
#else
            if (next) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // own pieces of chunk g + 1 (the only ones in flight)
                __builtin_amdgcn_s_barrier();
                if (more) issue_pieces(stage == 0 ? 2 : stage - 1, 0, 6);  // chunk g + 2 -> stage of chunk g - 1
            }
            __builtin_amdgcn_sched_barrier(0);
            mma_quarter(a1, b1, 0);  // (fragments read before the first half: nothing to wait for behind the barrier)
            __builtin_amdgcn_sched_barrier(0);
#endif
            if (next && more) issue_done();
            if (next) read_half(st1, 0, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            mma_quarter(a1, b1, 1);
            stage = st1;
        }
        // (the 16 row pointers of the C tile are rebuilt here instead of staying live through the K loop: the opaque
        //  pass through an empty asm keeps the compiler from carrying 32 address VGPRs across 2048 MFMAs)
#ifdef EGX_STREAM_TRACE
        if (trp) trp[6] = wall_clock64(), trp[7] = clock64();
#endif
#pragma unroll
        for (int r = 0; r < 4; r++) asm volatile("" : "+v"(coff[r]));
#pragma unroll
        for (int mi = 0; mi < 4; mi++)
#pragma unroll
            for (int ni = 0; ni < 4; ni++)
#pragma unroll
                for (int r = 0; r < 4; r++) Ct[(int64_t)mi * 16 * ldc + ni * 16 + coff[r]] = -acc[mi][ni][r];
#ifdef EGX_STREAM_TRACE
        if (trp) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            trp[8] = wall_clock64();
        }
#endif
    }
}

#ifdef EGX_STREAM_TRACE
}  // namespace egx
// mode 1: start recording into a fresh device buffer of `cap` records; mode 0: stop, copy the records taken (at most cap) to
// `out` (kStraceWords words each) and return their number
extern "C" long long egx_dev_stream_trace(int mode, long long *out, long long cap) {
    static long long *dbuf = nullptr;
    static long long dcap = 0;
    long long *null = nullptr;
    if (mode == 1) {
        if (dbuf) (void)hipFree(dbuf);
        dcap = cap;
        if (hipMalloc(&dbuf, sizeof(long long) * (16 + egx::kStraceWords * cap)) != hipSuccess) return -1;
        (void)hipMemset(dbuf, 0, sizeof(long long) * (16 + egx::kStraceWords * cap));
        (void)hipMemcpy(dbuf + 1, &cap, sizeof(long long), hipMemcpyHostToDevice);
        if (hipMemcpyToSymbol(HIP_SYMBOL(egx::g_strace), &dbuf, sizeof(dbuf)) != hipSuccess) return -2;
        return 0;
    }
    (void)hipDeviceSynchronize();
    (void)hipMemcpyToSymbol(HIP_SYMBOL(egx::g_strace), &null, sizeof(null));
    if (!dbuf) return 0;
    long long taken = 0;
    (void)hipMemcpy(&taken, dbuf, sizeof(long long), hipMemcpyDeviceToHost);
    if (taken > dcap) taken = dcap;
    if (taken > cap) taken = cap;
    if (out && taken > 0) (void)hipMemcpy(out, dbuf + 16, sizeof(long long) * egx::kStraceWords * taken, hipMemcpyDeviceToHost);
    (void)hipFree(dbuf);
    dbuf = nullptr;
    return taken;
}
namespace egx {
#endif

constexpr int TS = 64;
constexpr int TLD = 65;  // padded LDS row (doubles): per-lane row accesses are conflict free

// Workgroup: X = L^-1 (lower) for the factored 64x64 tile T (k_diag_tile_inverses); row `lane` of X per lane, into X
// (LDS, stride TLD).  Strips of 16 columns from the right; the contributions of the already finished strips go through
// the matrix cores, the short in-strip back substitution is done by wave 0.
template <int NW>
__device__ __forceinline__ void wg_inv_64(const double *T, const double *rd, double *X, int tid) {
    const int wave = tid >> 6, lane = tid & 63;
    const int frow = lane & 15, fk = lane >> 4;
#pragma unroll 1
    for (int cbk = 3; cbk >= 0; cbk--) {
        // bulk on the matrix cores: 16x16 tile (rt, cbk), rt > cbk:  - sum_{cbk < kb <= rt} X(rt, kb) L(kb, cbk);
        // the diagonal tile starts as the identity, tiles above it are zero
        for (int rt = wave; rt < 4; rt += NW) {
            double *xt = X + (rt * 16 + fk) * TLD + cbk * 16 + frow;
            double4_t acc = double4_t{0.0, 0.0, 0.0, 0.0};
            if (rt == cbk) {
#pragma unroll
                for (int r = 0; r < 4; r++) acc[r] = (fk + 4 * r == frow) ? 1.0 : 0.0;
            } else if (rt > cbk) {
                for (int kb = cbk + 1; kb <= rt; kb++) {
                    const double *arow = X + (rt * 16 + frow) * TLD + kb * 16 + fk;   // A[i][k] = X(rt*16+i, kb*16+k)
                    const double *bcol = T + (kb * 16 + fk) * TLD + cbk * 16 + frow;  // B[k][j] = L(kb*16+k, cbk*16+j)
#pragma unroll
                    for (int kk = 0; kk < 4; kk++)
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-arow[kk * 4], bcol[(kk * 4) * TLD], acc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; r++) xt[(4 * r) * TLD] = acc[r];
        }
        __syncthreads();
        if (wave == 0) {
            double x[16];
#pragma unroll
            for (int c = 0; c < 16; c++) x[c] = X[lane * TLD + cbk * 16 + c];
#pragma unroll
            for (int c = 15; c >= 0; c--) {
                double sacc = x[c];
#pragma unroll
                for (int k = c + 1; k < 16; k++)
                    sacc = __builtin_fma(-x[k], T[(cbk * 16 + k) * TLD + cbk * 16 + c], sacc);
                x[c] = sacc * rd[cbk * 16 + c];
            }
#pragma unroll
            for (int c = 0; c < 16; c++) X[lane * TLD + cbk * 16 + c] = x[c];
        }
        __syncthreads();
    }
}

// A: the register-resident diagonal-block factorisation -- tables, DPP pivot chain and the body of the kernel
// (rb_factor_block) live in potf2_blocks.h, shared with the pipelined chain kernel (kernels_pipe.hip)

template <int NW>  // waves: 1 chain wave + NW - 1 update waves (16 in the product; the tables also cover 8 and 12)
__global__ __launch_bounds__(64 * NW, 1) void k_potf2_reg(double *__restrict__ D, int64_t ld, int nbk, double *__restrict__ lin,
                                                          int *__restrict__ info, int col0, int n_valid, int64_t bsD,
                                                          int64_t bsL, int bsI) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    {   // lock-step batch: one workgroup per matrix
        const int64_t z = blockIdx.z;
        D += z * bsD;
        lin += z * bsL;
        info += z * bsI;
    }
    (void)rb_factor_block<NW, false>(D, ld, nbk, lin, info, col0, n_valid, sm, RbPublish());
}

// ---------------------------------------------------------------------------------------------
// B' (round 2): panel solve  X = P L^-T  below a diagonal block factored by k_potf2_reg, with the same 16x16 machinery:
// one workgroup = ONE 16-row tile of the panel x all (<= 16) column strips, as FP64-MFMA accumulators of its four waves
// (wave w owns the strips C = w mod 4: four tiles, 32 registers).  Per strip k: the wave that owns it turns its tile
// into X_k = T_k Linv_k^T (Linv_k and the refinement flag were left by k_potf2_reg in the tile-inverse slots; the
// refinement step is the diagonal-block kernel's), puts it into LDS (two buffers, one barrier per strip) and global;
// every wave then applies  T_C -= X_k L(C,k)^T  to its strips right of k, the L(C,k) fragments (rows pi(frow), 32
// contiguous bytes per lane) straight from the factored block in L2 and already in flight when the barrier opens.
// No global round trip between the strips (the round-1 kernel re-read its 64-row slab from global four times per block:
// 38 us whatever the panel's height); 16-row workgroups keep n / 16 of them in flight, which is what a
// 256-column panel of a small matrix needs to be spread over the chip at all.
// ---------------------------------------------------------------------------------------------
// POST = false: during a factorisation (lin = the slots k_potf2_reg filled: four 16x16 inverses of 256 doubles + four flags
// per 64x64 tile slot).  POST = true: the solves AFTER it (launch_trsm_rows: predict_var, theta-gradient, x-gradients),
// when the slots hold the 64x64 tile inverses: the inverse of a lower triangular matrix has the inverses of its diagonal
// blocks on its block diagonal, so strip C reads the 16x16 block (C % 4, C % 4) of tile C / 4 (row stride 64) and takes
// that tile's refinement flag (`tflags`, one per 64x64 tile, from k_diag_tile_inverses).
template <int RT, bool POST = false>  // RT: 16-row tiles per workgroup: 1 (panels of small matrices: n / 16 workgroups) or 2
                                      // (tall panels: the L fragments and the barrier serve two tiles, half the workgroups)
__global__ __launch_bounds__(256) void k_panel_trsm16(double *__restrict__ P, int64_t ldp, const double *__restrict__ L, int64_t ldl,
                                                      const double *__restrict__ lin, int nbk, const int *__restrict__ info,
                                                      int64_t bsP, int64_t bsM, int64_t bsL, int bsI,
                                                      const double *__restrict__ tflags) {
    __shared__ __attribute__((aligned(16))) double X[2][RT][16 * RB_LD];
    {   // lock-step batch: matrix blockIdx.z (bsP: between the panels' row blocks -- bsM inside a factorisation, the
        // right-hand-side buffers' own stride in the solves after it; bsM: between the factors; bsL: between the dinv blocks)
        const int64_t z = blockIdx.z;
        P += z * bsP;
        L += z * bsM;
        lin += z * bsL;
        if (tflags != nullptr) tflags += z * bsL;
        if (info != nullptr) info += z * bsI;
    }
    if (wg_failed_before(info)) return;  // failed pivot earlier: early exit (see k_gemm_nt_sub)
    __builtin_amdgcn_s_setprio(2);
    const int tid = threadIdx.x, lane = tid & 63, frow = lane & 15, fk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb16 = nbk >> 4;
    const int prow = ((frow & 3) << 2) | (frow >> 2);  // pi(frow), see k_potf2_reg
    double *Pw = P + (int64_t)(blockIdx.x * 16 * RT + frow) * ldp + 4 * fk;  // + 16 r ldp + 16 C: row tile r, strip C
    const double *Lp = L + (int64_t)prow * ldl + 4 * fk;                     // + (16 C) ldl + 16 k: row pi(frow) of tile (C, k)
    double acc[RT][16];                   // row tile r, slot t <-> strip C = 4 t + wave
    d2_t n01[4], n23[4], l01[4], l23[4];  // for the strips this wave solves: Linv fragments, masked fragments of L(C, C)
    bool refine[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int C = 4 * t + wave, Cc = C < nb16 ? C : 0;
#pragma unroll
        for (int r = 0; r < RT; r++) {
            const double *src = Pw + (int64_t)(16 * r) * ldp + 16 * Cc;
            const d2_t v0 = *reinterpret_cast<const d2_t *>(src), v1 = *reinterpret_cast<const d2_t *>(src + 2);
            acc[r][4 * t] = v0[0];
            acc[r][4 * t + 1] = v0[1];
            acc[r][4 * t + 2] = v1[0];
            acc[r][4 * t + 3] = v1[1];
        }
        // everything a solve needs besides its tile comes from L2 (~1 us): fetched here, not when the strip comes up
        constexpr int LS = POST ? 64 : 16;  // row stride of the 16x16 inverse
        const double *lk = lin + (int64_t)(Cc >> 2) * 4096 + (Cc & 3) * (POST ? 16 * 64 + 16 : 256);
        n01[t] = *reinterpret_cast<const d2_t *>(lk + prow * LS + 4 * fk);
        n23[t] = *reinterpret_cast<const d2_t *>(lk + prow * LS + 4 * fk + 2);
        refine[t] = POST ? (tflags != nullptr && tflags[Cc >> 2] != 0.0) : (lin[(int64_t)(Cc >> 2) * 4096 + 1024 + (Cc & 3)] != 0.0);
        const double *ld16 = Lp + (int64_t)(16 * Cc) * ldl + 16 * Cc;
        const d2_t r01 = *reinterpret_cast<const d2_t *>(ld16), r23 = *reinterpret_cast<const d2_t *>(ld16 + 2);
        l01[t] = d2_t{(4 * fk <= prow) ? r01[0] : 0.0, (4 * fk + 1 <= prow) ? r01[1] : 0.0};
        l23[t] = d2_t{(4 * fk + 2 <= prow) ? r23[0] : 0.0, (4 * fk + 3 <= prow) ? r23[1] : 0.0};
    }
    // solve strip k = 4 t + wave: X_k = T_k Linv_k^T (+ one refinement step, see k_potf2_reg) -> LDS buffer k & 1, global
    auto solve = [&](int t, int k) {
#pragma unroll
        for (int r = 0; r < RT; r++) {
            double4_t x = double4_t{0.0, 0.0, 0.0, 0.0};
            RB_MFMA4(x, 0, n01[t], n23[t], acc[r][4 * t], acc[r][4 * t + 1], acc[r][4 * t + 2], acc[r][4 * t + 3]);
            if (refine[t]) {
                double4_t rs = double4_t{acc[r][4 * t], acc[r][4 * t + 1], acc[r][4 * t + 2], acc[r][4 * t + 3]};
                RB_MFMA4(rs, 1, l01[t], l23[t], x[0], x[1], x[2], x[3]);
                RB_MFMA4(x, 0, n01[t], n23[t], rs[0], rs[1], rs[2], rs[3]);
            }
            double *xs = X[k & 1][r] + frow * RB_LD + 4 * fk;
            *reinterpret_cast<d2_t *>(xs) = d2_t{x[0], x[1]};
            *reinterpret_cast<d2_t *>(xs + 2) = d2_t{x[2], x[3]};
            double *dst = Pw + (int64_t)(16 * r) * ldp + 16 * k;
            *reinterpret_cast<d2_t *>(dst) = d2_t{x[0], x[1]};
            *reinterpret_cast<d2_t *>(dst + 2) = d2_t{x[2], x[3]};
        }
    };
    // The L(C, k) fragments (A' operands) of strip k for the four slots, in two sets (k & 1): fetched TWO strips ahead of
    // their use -- an L2 round trip (~1 us) is longer than a strip, and the set fetched during strip k - 1 would be
    // waited for at the top of strip k.
    d2_t a01[2][4], a23[2][4];
    auto fetch_l = [&](int k) {
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int C = 4 * t + wave;
            const double *src = Lp + (int64_t)(16 * ((C > k && C < nb16) ? C : k)) * ldl + 16 * k;
            a01[k & 1][t] = *reinterpret_cast<const d2_t *>(src);
            a23[k & 1][t] = *reinterpret_cast<const d2_t *>(src + 2);
        }
    };
    // T_C -= X_k L(C, k)^T for slot t (strip C = 4 t + wave), all row tiles
    auto update = [&](int t, int k, const d2_t (&b01)[RT], const d2_t (&b23)[RT]) {
#pragma unroll
        for (int r = 0; r < RT; r++) {
            double4_t c4 = double4_t{acc[r][4 * t], acc[r][4 * t + 1], acc[r][4 * t + 2], acc[r][4 * t + 3]};
            RB_MFMA4(c4, 1, a01[k & 1][t], a23[k & 1][t], b01[r][0], b01[r][1], b23[r][0], b23[r][1]);
            acc[r][4 * t] = c4[0];
            acc[r][4 * t + 1] = c4[1];
            acc[r][4 * t + 2] = c4[2];
            acc[r][4 * t + 3] = c4[3];
        }
    };
    fetch_l(0);
    if (nb16 > 1) fetch_l(1);
    if (wave == 0) solve(0, 0);
    __syncthreads();
    // Strip k: everybody applies X_k; the wave that owns strip k + 1 updates THAT tile first and solves it at once, so the
    // next X is on its way while the other updates of strip k still run (one barrier per strip, two X buffers).
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if (k < nb16) {
            d2_t b01[RT], b23[RT];
#pragma unroll
            for (int r = 0; r < RT; r++) {
                const double *xs = X[k & 1][r] + frow * RB_LD + 4 * fk;
                b01[r] = *reinterpret_cast<const d2_t *>(xs);
                b23[r] = *reinterpret_cast<const d2_t *>(xs + 2);
            }
            const int tn = (k + 1) >> 2;  // slot of strip k + 1 in the wave that owns it
            const bool next_owner = wave == ((k + 1) & 3) && k + 1 < nb16;
            if (next_owner) {
                update(tn, k, b01, b23);
                solve(tn, k + 1);
            }
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int C = 4 * t + wave;
                if (C > k && C < nb16 && !(next_owner && t == tn)) update(t, k, b01, b23);
            }
            if (k + 1 < nb16) {
                if (k + 2 < nb16) fetch_l(k + 2);  // into the set strip k just finished with
                __syncthreads();
            }
        }
    }
}

// Inverses of all 64x64 diagonal tiles of a GIVEN lower factor (at the end of a factorisation, and when a fitted model
// is loaded instead of factored here): one workgroup per tile, row-per-lane strips (wg_inv_64).
__global__ __launch_bounds__(256, 1) void k_diag_tile_inverses(const double *__restrict__ M, int64_t ld,
                                                               double *__restrict__ dinv, double *__restrict__ flags,
                                                               int64_t bsM, int64_t bsL) {
    __shared__ double Ls[TS * TLD];
    __shared__ double X[TS * TLD];
    __shared__ double rd[TS];
    __shared__ double red[2][256];
    {   // lock-step batch: matrix blockIdx.z
        const int64_t z = blockIdx.z;
        M += z * bsM;
        dinv += z * bsL;
        if (flags != nullptr) flags += z * bsL;
    }
    const int tid = threadIdx.x, t = blockIdx.x;
    const double *src = M + (int64_t)(t * TS) * ld + t * TS;
    for (int e = tid; e < TS * TS; e += 256) {
        const int row = e >> 6, c = e & 63;
        Ls[row * TLD + c] = (c <= row) ? src[(int64_t)row * ld + c] : 0.0;
    }
    __syncthreads();
    if (tid < TS) rd[tid] = 1.0 / Ls[tid * TLD + tid];
    __syncthreads();
    wg_inv_64<4>(Ls, rd, X, tid);
    double *dst = dinv + (int64_t)t * 4096;
    double xmax = 0.0;
    for (int e = tid; e < TS * TS; e += 256) {
        const double v = X[(e >> 6) * TLD + (e & 63)];
        dst[e] = v;
        xmax = fmax(xmax, fabs(v));
    }
    // flag of the tile for the solves that use its explicit inverse: max |Linv_ij| max L_ii (a stand-in for cond(L)) >= 32
    red[0][tid] = xmax;
    red[1][tid] = (tid < TS) ? Ls[tid * TLD + tid] : 0.0;
    __syncthreads();
    if (tid == 0 && flags != nullptr) {
        double a = 0.0, b = 0.0;
        for (int i = 0; i < 256; i++) {
            a = fmax(a, red[0][i]);
            b = fmax(b, red[1][i]);
        }
        flags[t] = (a * b >= 32.0 || !(a * b == a * b)) ? 1.0 : 0.0;
    }
}

// ---------------------------------------------------------------------------------------------
// Backward substitution  v <- C^-T v  (gamma = C^-T rho, algorithm.rs:1034), right-looking over
// 256-row blocks from the bottom.  The 64-step dependency chain is kept short by precomputing, once per
// fit and for all blocks in parallel, W_b = (L_bb^-1)^T of every 256x256 diagonal block (k_block_inv256,
// MFMA), so that the per-block solve is a single 256x256 mat-vec (k_trsv_w) followed by the transposed
// GEMV update of the rows above (k_gemv_t_update).
// ---------------------------------------------------------------------------------------------
// W (256 x 256 row-major per block, upper triangular) = Linv^T built from the 64x64 tile inverses:
//   W(c,c) = dinv_c^T ;  W(c,t) = -[ W(c, c..t-1) L(t, c..t-1)^T ] dinv_t^T   for t > c
__device__ __forceinline__ void block_inv256_body(const double *__restrict__ M, int64_t ld, int n_pad,
                                                  const double *__restrict__ dinv, double *__restrict__ Wall) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = blockIdx.x, k0 = b * 256;
    const int nbk = (n_pad - k0 < 256) ? (n_pad - k0) : 256, nt = nbk / 64;
    const double *D = M + (int64_t)k0 * ld + k0;
    const double *di = dinv + (int64_t)(k0 / 64) * 4096;
    double *W = Wall + (int64_t)b * 65536;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int e = tid; e < 65536; e += 512) {  // zero + transposed diagonal tiles
        const int j = e >> 8, i = e & 255;
        double v = 0.0;
        if ((j >> 6) == (i >> 6) && j < nbk) v = di[(int64_t)(j >> 6) * 4096 + (i & 63) * 64 + (j & 63)];
        W[e] = v;
    }
    __syncthreads();
    const int wrow = (wave >> 1) * 16 + (lane >> 4), wcol = (wave & 1) * 32 + (lane & 15);
    for (int dist = 1; dist < nt; dist++)
        for (int c = 0; c + dist < nt; c++) {
            const int t = c + dist;
            double4_t acc[1][2];
#pragma unroll
            for (int ni = 0; ni < 2; ni++) acc[0][ni] = double4_t{0.0, 0.0, 0.0, 0.0};
            gemm_core<64, 64, 16, 32, 512>(W + (int64_t)c * 64 * 256 + c * 64, 256, D + (int64_t)t * 64 * ld + c * 64, ld,
                                           dist * 64, acc, smem, tid);
            double *wt = W + (int64_t)(c * 64 + wrow) * 256 + t * 64 + wcol;
#pragma unroll
            for (int ni = 0; ni < 2; ni++)
#pragma unroll
                for (int r = 0; r < 4; r++) wt[(4 * r) * 256 + ni * 16] = acc[0][ni][r];
            __syncthreads();
#pragma unroll
            for (int ni = 0; ni < 2; ni++) acc[0][ni] = double4_t{0.0, 0.0, 0.0, 0.0};
            gemm_core<64, 64, 16, 32, 512>(W + (int64_t)c * 64 * 256 + t * 64, 256, di + (int64_t)t * 4096, 64, 64, acc,
                                           smem, tid);
#pragma unroll
            for (int ni = 0; ni < 2; ni++)
#pragma unroll
                for (int r = 0; r < 4; r++) wt[(4 * r) * 256 + ni * 16] = -acc[0][ni][r];
            __syncthreads();
        }
}
__global__ __launch_bounds__(512, 2) void k_block_inv256(const double *__restrict__ M, int64_t ld, int n_pad,
                                                         const double *__restrict__ dinv, double *__restrict__ Wall) {
    block_inv256_body(M, ld, n_pad, dinv, Wall);
}
// ... of several models at once (blockIdx.y = model; egx_gp_finalize_multi): the back-substitutions of eight experts are
// 8 x 65 launches of ~5 us that the command processor serialises however many streams and host threads issue them -- in
// lock-step they are 65 (profiles/r05_expert_group_*.txt)
__global__ __launch_bounds__(512, 2) void k_block_inv256_batch(SolveBatchPtrs b, int64_t ld, int n_pad) {
    block_inv256_body(b.M[blockIdx.y], ld, n_pad, b.dinv[blockIdx.y], b.dW[blockIdx.y]);
}

// x = W v for one 256-block (W upper triangular).  grid = 8 workgroups x 32 rows: each wave owns 8 rows and
// streams them with all loads of 4 rows in flight (one workgroup alone is HBM-latency bound: 24 us).
// x is written to xout (a separate buffer: other workgroups still read v).
__device__ __forceinline__ void trsv_w_body(const double *__restrict__ W, int nbk, const double *__restrict__ v,
                                            double *__restrict__ xout) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const double v0 = (lane < nbk) ? v[lane] : 0.0, v1 = (lane + 64 < nbk) ? v[lane + 64] : 0.0;
    const double v2 = (lane + 128 < nbk) ? v[lane + 128] : 0.0, v3 = (lane + 192 < nbk) ? v[lane + 192] : 0.0;
    const int jbase = blockIdx.x * 32 + wave * 8;
#pragma unroll
    for (int j0 = 0; j0 < 8; j0 += 4) {
        double acc[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const double *row = W + (int64_t)(jbase + j0 + u) * 256;
            acc[u] = row[lane] * v0 + row[lane + 64] * v1 + row[lane + 128] * v2 + row[lane + 192] * v3;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            double a = acc[u];
            for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
            if (lane == 0 && jbase + j0 + u < nbk) xout[jbase + j0 + u] = a;
        }
    }
}
__global__ __launch_bounds__(256) void k_trsv_w(const double *__restrict__ W, int nbk, const double *__restrict__ v,
                                                double *__restrict__ xout) {
    trsv_w_body(W, nbk, v, xout);
}
__global__ __launch_bounds__(256) void k_trsv_w_batch(SolveBatchPtrs b, int blk, int k0, int nbk) {
    const int z = blockIdx.y;
    trsv_w_body(b.dW[z] + (int64_t)blk * 65536, nbk, b.rhs[z] + k0, b.vec[z] + k0);
}

// v[j] -= sum_i Mrow[i*ld + j] * x[i]  for j < ncols ; grid = ncols/64
__device__ __forceinline__ void gemv_t_update_body(const double *__restrict__ Mrow, int64_t ld, int nbk,
                                                   const double *__restrict__ x, double *__restrict__ v) {
    __shared__ double xs[256];
    __shared__ double red[4][64];
    const int tid = threadIdx.x, g = tid >> 6, jl = tid & 63;
    const int j = blockIdx.x * 64 + jl;
    if (tid < nbk) xs[tid] = x[tid];
    __syncthreads();
    double part = 0.0;
    for (int i0 = g; i0 < nbk; i0 += 64) {  // nbk is a multiple of 64: 16 independent loads in flight
        double mv[16];
#pragma unroll
        for (int u = 0; u < 16; u++) mv[u] = Mrow[(int64_t)(i0 + 4 * u) * ld + j];
#pragma unroll
        for (int u = 0; u < 16; u++) part = __builtin_fma(mv[u], xs[i0 + 4 * u], part);
    }
    red[g][jl] = part;
    __syncthreads();
    if (g == 0) v[j] -= ((red[0][jl] + red[1][jl]) + red[2][jl]) + red[3][jl];
}
__global__ __launch_bounds__(256) void k_gemv_t_update(const double *__restrict__ Mrow, int64_t ld, int nbk,
                                                       const double *__restrict__ x, double *__restrict__ v) {
    gemv_t_update_body(Mrow, ld, nbk, x, v);
}
__global__ __launch_bounds__(256) void k_gemv_t_update_batch(SolveBatchPtrs b, int64_t ld, int k0, int nbk) {
    const int z = blockIdx.y;
    gemv_t_update_body(b.M[z] + (int64_t)k0 * ld, ld, nbk, b.vec[z] + k0, b.rhs[z]);
}

// ---------------------------------------------------------------------------------------------
// MFMA layout probe: C(16x16) = A(16x16) B(16x16) with asymmetric operands.
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// gamma = C^-T rho as ONE launch (round 6; launch_trsv_t / launch_trsv_t_batch): the block back-substitution above is
// 2 n / 256 launches of ~5 us in a row -- 0.81 ms of a 29.6-ms fit at n = 16384, 0.16 of 1.7 ms at n = 4096 -- for 1.07 GB of
// factor that stream in 0.2 ms.  Here one workgroup per 64 COLUMNS of the right-hand side (a segment) keeps that piece of
// rho in registers, takes the blocks' solutions x_b in order b = last ... as they appear (its 256 x 64 piece of L fetched
// BEFORE it waits: the factor is final, only x_b is not), and when its own block is next the (up to) four segment workgroups
// of that block exchange their finished pieces of the right-hand side, each applies 64 rows of W_b = (L_bb^-1)^T (fetched
// before the wait as well) and publishes them.  Exactly the sums of k_trsv_w / k_gemv_t_update in their order: the same bits
// as the launch-per-block form ("trsv_fused" = 0).
//   * segments are taken by a START ticket from the LAST one down, so a workgroup only ever waits for segments that started
//     before it (x_b of later blocks) or for the three other segments of its own block, whose tickets follow directly: no
//     deadlock on any device that holds four of these workgroups at once, whatever the dispatch order.
//   * hand-offs WITHOUT flags: the solution vector and the exchange buffer start as all-ones words (a NaN no arithmetic
//     produces); a value is one 8-byte agent-scope store, a consumer polls the very word it needs with agent-scope loads --
//     one trip through the fabric per hand-off instead of two (counter, then payload: 9 us per block measured, the launches
//     it replaced 12.6).
//   * every wait is bounded (the chain launch's EGX_PIPE_TIMEOUT_MS): a waiter whose time runs out goes on with a NaN of its
//     own, which reaches every later block -- nobody hangs, and the host's finite-check of gamma reports EGX_ERR_HIP instead
//     of returning a wrong model.
// tail (behind the inverse blocks, 0xFF-filled by the launcher): [0] the start ticket (an int, from -1), [4 ..] the finished
// right-hand side (n_pad doubles).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double trsv_poll(const double *p, long long limit) {
    long long t0 = 0;
    for (unsigned spins = 1;; spins++) {
        const double v = load_sc1(p);
        if (__double_as_longlong(v) != -1ll) return v;
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 63) == 0) {
            if (t0 == 0) t0 = wall_clock64();
            else if (wall_clock64() - t0 > limit) return __builtin_nan("");
        }
    }
}
__device__ __forceinline__ void trsv_fused_body(const double *__restrict__ M, int64_t ld, int n_pad, const double *__restrict__ Wall,
                                                const double *__restrict__ rhs, double *x, double *tail, long long limit, int stall) {
    __shared__ double xs[256];
    __shared__ double red[4][64];
    __shared__ int s_seg;
    const int tid = threadIdx.x, g = tid >> 6, jl = tid & 63;
    const int nblocks = (n_pad + kNB - 1) / kNB, nseg = n_pad / 64;
    double *rfin = tail + 4;
    if (tid == 0) s_seg = nseg - 2 - __hip_atomic_fetch_add(reinterpret_cast<int *>(tail), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int seg = s_seg;
    if (seg < 0) return;
    if (stall != 0 && seg == nseg - 1) return;  // (test builds, "trsv_stall": the last segment never publishes anything)
    const int blk = seg >> 2, j = seg * 64 + jl;
    double r = (g == 0) ? rhs[j] : 0.0;
    double mv[4][16];
    for (int b = nblocks - 1; b > blk; b--) {
        const int k0 = b * kNB;
        const int nbk = (n_pad - k0 < kNB) ? (n_pad - k0) : kNB;  // a multiple of 64
        const double *Mrow = M + (int64_t)k0 * ld;
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (q * 64 < nbk) {
#pragma unroll
                for (int u = 0; u < 16; u++) mv[q][u] = Mrow[(int64_t)(g + 64 * q + 4 * u) * ld + j];
            }
        if (tid < nbk) xs[tid] = trsv_poll(x + k0 + tid, limit);
        __syncthreads();
        double part = 0.0;  // (the order of k_gemv_t_update: rows g, g + 4, ... of each 64-row quarter, quarter by quarter)
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (q * 64 < nbk) {
#pragma unroll
                for (int u = 0; u < 16; u++) part = __builtin_fma(mv[q][u], xs[g + 64 * q + 4 * u], part);
            }
        red[g][jl] = part;
        __syncthreads();
        if (g == 0) r -= ((red[0][jl] + red[1][jl]) + red[2][jl]) + red[3][jl];
    }
    // this segment of the right-hand side is final: publish it, then 64 rows of x_blk = W_blk r_blk
    const int k0 = blk * kNB;
    const int nbk = (n_pad - k0 < kNB) ? (n_pad - k0) : kNB;
    if (g == 0) __hip_atomic_store(rfin + j, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int lane = jl, wave = g;
    const int row0 = (seg & 3) * 64 + wave * 16;  // this wave's 16 rows of the block
    const double *W = Wall + (int64_t)blk * 65536;
#pragma unroll
    for (int u = 0; u < 16; u++) {
        const double *row = W + (int64_t)(row0 + u) * 256;
        mv[0][u] = row[lane], mv[1][u] = row[lane + 64], mv[2][u] = row[lane + 128], mv[3][u] = row[lane + 192];
    }
    __syncthreads();  // (xs: the last update's reads are done)
    xs[tid] = (tid < nbk) ? trsv_poll(rfin + k0 + tid, limit) : 0.0;
    __syncthreads();
    const double v0 = xs[lane], v1 = xs[lane + 64], v2 = xs[lane + 128], v3 = xs[lane + 192];
#pragma unroll
    for (int u = 0; u < 16; u++) {
        double a = mv[0][u] * v0 + mv[1][u] * v1 + mv[2][u] * v2 + mv[3][u] * v3;  // (k_trsv_w's expression)
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
        if (lane == 0 && row0 + u < nbk) __hip_atomic_store(x + k0 + row0 + u, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__global__ __launch_bounds__(256) void k_trsv_t_fused(const double *__restrict__ M, int64_t ld, int n_pad, const double *__restrict__ Wall,
                                                      const double *__restrict__ rhs, double *x, double *tail, long long limit, int stall) {
    trsv_fused_body(M, ld, n_pad, Wall, rhs, x, tail, limit, stall);
}
__global__ __launch_bounds__(256) void k_trsv_t_fused_batch(SolveBatchPtrs b, int64_t ld, int n_pad, long long limit, int stall) {
    const int z = blockIdx.y;
    trsv_fused_body(b.M[z], ld, n_pad, b.dW[z], b.rhs[z], b.vec[z], b.dW[z] + (int64_t)((n_pad + kNB - 1) / kNB) * 65536, limit, stall);
}

__global__ void k_mfma_probe(const double *A, const double *B, double *C) {
    const int lane = threadIdx.x;
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
    for (int k4 = 0; k4 < 4; k4++) {
        const double a = A[(lane & 15) * 16 + k4 * 4 + (lane >> 4)];  // A[i][k]
        const double b = B[(k4 * 4 + (lane >> 4)) * 16 + (lane & 15)];  // B[k][j]
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; r++) C[((lane >> 4) + 4 * r) * 16 + (lane & 15)] = acc[r];
}

// ---------------------------------------------------------------------------------------------
// Split-K Gram matrix (sparse GP: G = W W^T with a SMALL output, rows x rows, and a HUGE contraction length K = n):
// grid.x enumerates the lower 64x64 tiles, grid.y the K splits; every workgroup writes its partial tile to
// P[split] (no read-modify-write, deterministic), k_gram_reduce sums the splits into Gneg = -sum_s P[s].
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_gram_splitk(const double *__restrict__ W, int64_t ldw, int rows, int kc,
                                                        int K, double *__restrict__ P) {
    using S = GemmShape<64, 64, 32, 32, 256>;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int t = blockIdx.x;
    int bx = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while (bx * (bx + 1) / 2 > t) bx--;
    while ((bx + 1) * (bx + 2) / 2 <= t) bx++;
    const int by = t - bx * (bx + 1) / 2;
    const int k0 = blockIdx.y * kc;
    int klen = K - k0;
    if (klen > kc) klen = kc;
    const int tid = threadIdx.x;
    double4_t acc[S::MT][S::NT];
#pragma unroll
    for (int mi = 0; mi < S::MT; mi++)
#pragma unroll
        for (int ni = 0; ni < S::NT; ni++) acc[mi][ni] = double4_t{0.0, 0.0, 0.0, 0.0};
    if (klen > 0)
        gemm_core<64, 64, 32, 32, 256>(W + (int64_t)bx * 64 * ldw + k0, ldw, W + (int64_t)by * 64 * ldw + k0, ldw, klen,
                                       acc, smem, tid);
    const int wave = tid >> 6, lane = tid & 63;
    const int r0 = bx * 64 + (wave / S::WAVES_N) * 32 + (lane >> 4);
    const int c0 = by * 64 + (wave % S::WAVES_N) * 32 + (lane & 15);
    double *out = P + (int64_t)blockIdx.y * rows * rows;
#pragma unroll
    for (int mi = 0; mi < S::MT; mi++)
#pragma unroll
        for (int ni = 0; ni < S::NT; ni++)
#pragma unroll
            for (int r = 0; r < 4; r++) out[(int64_t)(r0 + mi * 16 + 4 * r) * rows + (c0 + ni * 16)] = acc[mi][ni][r];
}

__global__ void k_gram_reduce(const double *__restrict__ P, int rows, int nsplit, double *__restrict__ Gneg, int64_t ldg) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= rows || (j >> 6) > (i >> 6)) return;  // lower 64x64 tiles only
    double sacc = 0.0;
    for (int sp = 0; sp < nsplit; sp++) sacc += P[((int64_t)sp * rows + i) * rows + j];
    Gneg[(int64_t)i * ldg + j] = -sacc;
}

// =============================================================================================
// host launchers
// =============================================================================================
// The run-time settings that survive (environment, read once; egx_set_tuning for A/B runs inside one process).  With EGX_PIPE
// and EGX_PIPE_TIMEOUT_MS (kernels_pipe.hip) that makes eight; everything else that was ever switchable is a constant now,
// its A/B in profiles/ (docs/HISTORY.md §B lists them).  Atomics: a setting may change while another thread evaluates --
// that thread then sees the old or the new value per read, never a torn one (schedules are per handle and not affected).
static std::atomic<int> g_potrf_group{0};         // EGX_POTRF_GROUP: panels per trailing update, 1..8 (0 = by size: 4 from n_pad 14336, else 2)
static std::atomic<int> g_stream_min_tiles{128};  // EGX_STREAM_MIN: launches with at least this many 128x256 tiles PER MATRIX go to k_gemm_stream
static std::atomic<int> g_gemm_small_max{1024};   // EGX_GEMM_SMALL: below this many 128x128 tiles the 64x64-tile kernel is used
static std::atomic<int> g_look_min_cols{3072};    // EGX_LOOK_MIN: look-ahead while at least this many columns trail the next group
static std::atomic<int> g_lur_side{1};            // EGX_LUR_SIDE=0: the look-ahead columns' update stays in front of RU (bench.py's roofline leg)
#ifdef EGX_TEST_HOOKS
static std::atomic<int> g_trsv_stall{0};  // egx_set_tuning "trsv_stall" (test build only): the one-launch back-substitution's last segment never publishes
#define EGX_TRSV_STALL g_trsv_stall.load()
#else
#define EGX_TRSV_STALL 0
#endif
static std::atomic<int> g_trsv_fused{1};          // EGX_TRSV_FUSED=0: the back-substitution gamma = C^-T rho as one launch PER BLOCK (rounds 1-5) instead of one launch
static std::atomic<int> g_potrf_left{1};          // EGX_POTRF_LEFT: left-looking group updates (factor and C^-T rider) 0 never, 1 by schedule.h, 2 always
constexpr int kTrsmGroupPanels = 4;   // panels per update in the solves after the factorisation
constexpr int kLurSideMinCols = 6144; // LUr on the side stream only while at least this many columns trail the look-ahead group

int chol_init() {
    static std::once_flag once;
    static int rc_once = EGX_SUCCESS;
    std::call_once(once, [] {
        if (const char *e = std::getenv("EGX_POTRF_GROUP")) {
            const int g = std::atoi(e);
            if (g >= 1 && g <= 8) g_potrf_group = g;
        }
        if (const char *e = std::getenv("EGX_GEMM_SMALL")) g_gemm_small_max = std::atoi(e);
        if (const char *e = std::getenv("EGX_STREAM_MIN")) g_stream_min_tiles = std::atoi(e) > 0 ? std::atoi(e) : 1;
        if (const char *e = std::getenv("EGX_LOOK_MIN")) g_look_min_cols = std::atoi(e);
        if (const char *e = std::getenv("EGX_LUR_SIDE")) g_lur_side = std::atoi(e);
        if (const char *e = std::getenv("EGX_TRSV_FUSED")) g_trsv_fused = std::atoi(e);
        if (const char *e = std::getenv("EGX_POTRF_LEFT")) g_potrf_left = std::atoi(e);
        auto set = [](const void *fn, int bytes) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e != hipSuccess && rc_once == EGX_SUCCESS) {
                set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
                rc_once = EGX_ERR_HIP;
            }
        };
        set(reinterpret_cast<const void *>(&k_gemm_nt_sub<true, 128, 128, 32, 64, 512, true>), TrailShape::LDS_BYTES);
        set(reinterpret_cast<const void *>(&k_gemm_nt_sub<false, 128, 128, 32, 64, 512, true>), TrailShape::LDS_BYTES);
        set(reinterpret_cast<const void *>(&k_gemm_stream<true>), ST_LDS_BYTES);
        set(reinterpret_cast<const void *>(&k_gemm_stream<false>), ST_LDS_BYTES);
        set(reinterpret_cast<const void *>(&k_gemm_stream<true, true>), ST_LDS_BYTES);
        set(reinterpret_cast<const void *>(&k_gemm_stream<false, true>), ST_LDS_BYTES);
        set(reinterpret_cast<const void *>(&k_gemm_stream<true, false, 1>), ST_LDS_BYTES);
        set(reinterpret_cast<const void *>(&k_gemm_stream<true, false, 2>), ST_LDS_BYTES);
    });
    return rc_once;
}

static ScheduleKnobs schedule_knobs() {
    (void)chol_init();
    ScheduleKnobs k;
    k.potrf_left = g_potrf_left;
    k.pipe = pipe_enabled();
    k.potrf_group = g_potrf_group;
    return k;
}
int potrf_group_panels(int n_pad) { return schedule_table(n_pad, 1, 1, schedule_knobs()).group_panels; }
// the one place where a handle's schedule is decided: schedule.h
PotrfSchedule schedule_for(int n_pad, int lockstep, int n_workspaces) {
    return schedule_table(n_pad, lockstep, n_workspaces, schedule_knobs());
}
int potrf_left_for(int n_pad, int lockstep) { return schedule_for(n_pad, lockstep, 1).left; }
int w_left_for(int n_pad, int lockstep) { return schedule_for(n_pad, lockstep, 1).w_left; }

// run-time access to the knobs above (egx_set_tuning): A/B measurements inside ONE process (a gpurun call is minutes, a
// launch sequence milliseconds) and the serialised profiling mode of bench.py's roofline leg.  Returns the previous
// value, or INT_MIN for an unknown name.  Not to be called while evaluations are in flight.
int set_knob(const char *name, int value) {
    (void)chol_init();  // the environment is read first, once; a later call here wins
    struct { const char *n; std::atomic<int> *v; } tab[] = {{"potrf_group", &g_potrf_group}, {"stream_min", &g_stream_min_tiles},
                                                           {"gemm_small", &g_gemm_small_max}, {"look_min", &g_look_min_cols},
                                                           {"lur_side", &g_lur_side},         {"potrf_left", &g_potrf_left},
                                                           {"trsv_fused", &g_trsv_fused}};
    for (auto &e : tab)
        if (std::string(name) == e.n) return e.v->exchange(value);
#ifdef EGX_TEST_HOOKS
    if (std::string(name) == "trsv_stall") return g_trsv_stall.exchange(value);
#endif
    return -2147483647 - 1;
}

// Kernel choice by the size of ONE matrix' launch (never by the batch count: a matrix is factored by the same kernels
// alone and in a lock-step batch, hence to the same bits):
//   k_gemm_stream (128x256 tile, LDS-DMA ring)   >= g_stream_min_tiles wide tiles, N % 256 == 0, no ktri
//   k_gemm_nt_sub 64x64 (register staged)        latency-bound launches: a 128x128xK tile is one wave chain of
//                                                K/4*16 MFMAs (~43 us at K = 256)
//   k_gemm_nt_sub 128x128, XCD-swizzled          the rest (the theta-gradient's R^-1 = W W^T with ktri, odd widths)
int launch_gemm_nt_sub(hipStream_t s, double *C, int64_t ldc, const double *A, int64_t lda,
                       const double *B, int64_t ldb, int M, int N, int K, int lower, int ktri, bool *used_big_tile,
                       const int *info, const GemmBatch *batch, int tag) {
    if (used_big_tile) *used_big_tile = false;
    if (M <= 0 || N <= 0 || K <= 0) return EGX_SUCCESS;
    if (M % 128 || N % 128 || K % KC) {
        set_error("gemm_nt_sub: M,N must be multiples of 128 and K of 16");
        return EGX_ERR_INVALID_VALUE;
    }
    const GemmBatch bt = batch ? *batch : GemmBatch();
    const unsigned nz = (unsigned)(bt.count > 0 ? bt.count : 1);
    const int64_t big_tiles = lower ? ((int64_t)(M / 128) * (N / 128) - (int64_t)(N / 128) * (N / 128 - 1) / 2)
                                    : (int64_t)(M / 128) * (N / 128);
    const bool small = big_tiles < g_gemm_small_max;  // fewer than 2 waves of workgroups over 256 CUs x 2
    int64_t wide_tiles = 0;
    if (N % 256 == 0 && !ktri && (!lower || M >= N)) {
        if (lower)
            for (int c = 0; c < N / 256; c++) wide_tiles += (M / 128 - 2 * c > 0) ? M / 128 - 2 * c : 0;
        else
            wide_tiles = (int64_t)(M / 128) * (N / 256);
    }
    // (long K loops amortise a tile's prologue / epilogue and the launch: the left-looking group updates, K = all earlier
    //  columns, stay on the stream kernel down to a handful of tiles.  The rule is by shape, for EVERY caller: the solves
    //  after the factorisation and the prediction paths with K >= 2048 take the same kernel as the factorisation would)
    const int64_t min_tiles = (K >= 2048) ? 8 : g_stream_min_tiles.load();
    if (K >= 2 * KC && wide_tiles >= min_tiles) {
        // the launches that fill the chip on their own are the ones the roofline trace follows
        if (used_big_tile) *used_big_tile = wide_tiles >= 512;
        const int nbx = M / 128, nby = N / 256;  // LOWER: column c holds nbx - 2 c tiles (M >= N in the factorisation)
        const int nt = (int)wide_tiles;
        const dim3 grid((unsigned)nt, 1, nz);  // one tile per workgroup: CUs turn over, a chain beside it squeezes in
        if (lower && tag == 1)
            hipLaunchKernelGGL((k_gemm_stream<true, false, 1>), grid, dim3(512), ST_LDS_BYTES, s, C, ldc, A, lda, B, ldb, K, nbx,
                               nby, nt, info, bt);
        else if (lower && tag == 2)
            hipLaunchKernelGGL((k_gemm_stream<true, false, 2>), grid, dim3(512), ST_LDS_BYTES, s, C, ldc, A, lda, B, ldb, K, nbx,
                               nby, nt, info, bt);
        else if (lower)
            hipLaunchKernelGGL((k_gemm_stream<true>), grid, dim3(512), ST_LDS_BYTES, s, C, ldc, A, lda, B, ldb, K, nbx, nby, nt,
                               info, bt);
        else
            hipLaunchKernelGGL((k_gemm_stream<false>), grid, dim3(512), ST_LDS_BYTES, s, C, ldc, A, lda, B, ldb, K, nbx, nby, nt,
                               info, bt);
        EGX_HIP_CHECK(hipGetLastError());
        return EGX_SUCCESS;
    }
    if (small) {
        const dim3 grid(M / 64, N / 64, nz);
        if (lower)
            hipLaunchKernelGGL((k_gemm_nt_sub<true, 64, 64, 32, 32, 256, false>), grid, dim3(256), SmallShape::LDS_BYTES, s,
                               C, ldc, A, lda, B, ldb, K, M / 64, N / 64, ktri, info, bt);
        else
            hipLaunchKernelGGL((k_gemm_nt_sub<false, 64, 64, 32, 32, 256, false>), grid, dim3(256), SmallShape::LDS_BYTES,
                               s, C, ldc, A, lda, B, ldb, K, M / 64, N / 64, ktri, info, bt);
    } else {
        const int nbx = M / 128, nby = N / 128;
        const int nsx = (nbx + 7) / 8, nsy = (nby + 7) / 8;
        int nst;
        if (lower) {
            const int tri_rows = nsx < nsy ? nsx : nsy;
            nst = tri_rows * (tri_rows + 1) / 2 + (nsx > nsy ? (nsx - nsy) * nsy : 0);
            if (nsx < nsy) nst = nsy * (nsy + 1) / 2;  // (not used by the factorisation: M >= N there)
        } else {
            nst = nsx * nsy;
        }
        const int per_xcd = (nst + 7) / 8;
        const dim3 grid(8 * per_xcd * 64, 1, nz);
        if (lower)
            hipLaunchKernelGGL((k_gemm_nt_sub<true, 128, 128, 32, 64, 512, true>), grid, dim3(512), TrailShape::LDS_BYTES, s,
                               C, ldc, A, lda, B, ldb, K, nbx, nby, ktri, info, bt);
        else
            hipLaunchKernelGGL((k_gemm_nt_sub<false, 128, 128, 32, 64, 512, true>), grid, dim3(512), TrailShape::LDS_BYTES,
                               s, C, ldc, A, lda, B, ldb, K, nbx, nby, ktri, info, bt);
    }
    EGX_HIP_CHECK(hipGetLastError());
    return EGX_SUCCESS;
}

// Gneg (rows x rows, ldg; lower 64-tiles) <- -(W W^T), W (rows x K, ldw) K-contiguous; rows % 64 == 0, K % 128 == 0.
// P is scratch for the K splits: gram_scratch_doubles(rows, K) doubles.
static int gram_nsplit(int rows, int K) {
    const int nt = rows / 64, tiles = nt * (nt + 1) / 2;
    int nsplit = (2048 + tiles - 1) / tiles;
    const int maxs = K / 512 > 0 ? K / 512 : 1;
    if (nsplit > maxs) nsplit = maxs;
    if (nsplit < 1) nsplit = 1;
    return nsplit;
}
size_t gram_scratch_doubles(int rows, int K) { return (size_t)gram_nsplit(rows, K) * rows * rows; }
int launch_gram_lower(hipStream_t s, const double *W, int64_t ldw, int rows, int K, double *Gneg, int64_t ldg, double *P) {
    if (rows % 64 || K % KC) {
        set_error("gram: rows must be a multiple of 64 and K of 16");
        return EGX_ERR_INVALID_VALUE;
    }
    const int nt = rows / 64, tiles = nt * (nt + 1) / 2;
    int nsplit = gram_nsplit(rows, K);
    int kc = (int)round_up((K + nsplit - 1) / nsplit, 128);
    nsplit = (K + kc - 1) / kc;
    hipLaunchKernelGGL(k_gram_splitk, dim3(tiles, nsplit), dim3(256), SmallShape::LDS_BYTES, s, W, ldw, rows, kc, K, P);
    hipLaunchKernelGGL(k_gram_reduce, dim3((rows + 255) / 256, rows), dim3(256), 0, s, (const double *)P, rows, nsplit, Gneg,
                       ldg);
    EGX_HIP_CHECK(hipGetLastError());
    return EGX_SUCCESS;
}

// Right-looking factorisation, two-level blocking (kNB-wide panels inside groups of panels, ONE trailing update per
// group with K = group width) with a look-ahead whose CRITICAL PATH is kept narrow:
//   * the critical path of a right-looking Cholesky is  potf2(k) -> trsm(k) -> update of panel k+1 -> potf2(k+1) ...
//     Only the DIAGONAL block of panel k+1 has to be up to date before potf2(k+1) may start, so every update on that
//     path is split into the diagonal block (a 256x256 launch on the chain's stream) and the rest (on a side stream,
//     overlapping the next diagonal-block factorisation); the panel solve that follows waits for the rest.
//   * as soon as the next group's first diagonal block has been updated (LUd) that group's chain -- diagonal blocks,
//     panel solves, in-group updates -- runs on the auxiliary high-priority streams (lk->s2 / lk->s3), concurrently with
//     the rest of the look-ahead columns (LUr) and the trailing update (RU) on `s`.
// lk == nullptr (or lk->s2 == nullptr) runs everything in order on `s`.
// batch != nullptr: batch->count matrices in lock-step (matrix z at M + z sM, dinv + z sD, info + z sI); every launch
// below covers all of them (grid.z), the schedule is the one a single matrix of this size gets.
int launch_potrf(hipStream_t s, double *M, int64_t ld, int n_pad, int m_tot, double *dinv, int *info,
                 const PotrfLookahead *lk, GemmTrace *trace, const PotrfBatch *batch, const PotrfInverse *inv) {
    if (trace) trace->used = 0;
    int rc = chol_init();
    if (rc) return rc;
    if (n_pad % kTile || m_tot % kTile || m_tot < n_pad) {
        set_error("potrf: padded sizes must be multiples of 128");
        return EGX_ERR_INVALID_VALUE;
    }
    const PotrfBatch pb = batch ? *batch : PotrfBatch();
    const unsigned nz = (unsigned)(pb.count > 0 ? pb.count : 1);
    GemmBatch gb;
    gb.count = (int)nz;
    gb.sC = gb.sA = gb.sB = pb.sM;
    gb.sInfo = pb.sI;
    hipStream_t s2 = lk ? lk->s2 : nullptr, s3 = lk ? lk->s3 : nullptr;
    auto potf2 = [&](hipStream_t st, int k0, int nbk) {
        double *diag = M + (int64_t)k0 * ld + k0;
        double *dtiles = dinv + (int64_t)(k0 / 64) * 4096;
        // (the 64x64 tile inverses `dinv` -- for the solves AFTER the factorisation -- are built at the end, all at once;
        //  during the factorisation their slots carry the 16x16 inverses + refinement flags to the panel solve)
        hipLaunchKernelGGL(k_potf2_reg<16>, dim3(1, 1, nz), dim3(1024), RB_LDS_BYTES, st, diag, ld, nbk, dtiles, info, k0, n_pad,
                           pb.sM, pb.sD, pb.sI);
    };
    auto trsm = [&](hipStream_t st, int k0, int nbk) {
        const int below = m_tot - (k0 + nbk);  // a multiple of 64
        if (below <= 0) return;
        double *P = M + (int64_t)(k0 + nbk) * ld + k0;
        const double *L = M + (int64_t)k0 * ld + k0;
        const double *lin = dinv + (int64_t)(k0 / 64) * 4096;
        if (below >= 4096)  // tall panels: two 16-row tiles per workgroup (+1 %, profiles/r02_run34_*)
            hipLaunchKernelGGL((k_panel_trsm16<2, false>), dim3(below / 32, 1, nz), dim3(256), 0, st, P, ld, L, ld, lin, nbk,
                               (const int *)info, pb.sM, pb.sM, pb.sD, pb.sI, (const double *)nullptr);
        else
            hipLaunchKernelGGL((k_panel_trsm16<1, false>), dim3(below / 16, 1, nz), dim3(256), 0, st, P, ld, L, ld, lin, nbk,
                               (const int *)info, pb.sM, pb.sM, pb.sD, pb.sI, (const double *)nullptr);
    };
    // C[r0.., c0..c0+N) -= P[r0.., k0..k0+K) P[c0..c0+N, k0..k0+K)^T for the M rows from r0
    auto update = [&](hipStream_t st, int r0, int c0, int Mr, int N, int k0, int K, int lower, bool *big, int tag = 0) -> int {
        return launch_gemm_nt_sub(st, M + (int64_t)r0 * ld + c0, ld, M + (int64_t)r0 * ld + k0, ld,
                                  M + (int64_t)c0 * ld + k0, ld, Mr, N, K, lower, 0, big, info, &gb, tag);
    };
    // groups of four panels (K = 1024 per trailing update: half the C tile traffic and tile boundaries of K = 512) pay
    // from n ~ 14000 on (measured, profiles/r02_run13_group_by_size.txt: n = 16384 -2 %, n = 12288 even, n = 8192 +5 %:
    // the chain of a group gets longer while its trailing update shrinks)
    // the serial chain as ONE persistent launch per group of panels (kernels_pipe.hip) when the caller provides the hand-off
    // words; for handles whose schedule says so (schedule.h) the whole factorisation is one such launch (every update inside it)
    // (pb.pipe / pb.whole / pb.group_panels come from the HANDLE's schedule, decided when it was created: nothing here reads the
    //  process-wide knobs, so egx_set_tuning never changes the arithmetic of a handle that exists -- egx_gp_get_schedule stays
    //  true.  A device on which the launch does not fit -- one workgroup per diagonal block and matrix: a partitioned GPU -- takes
    //  the separate launches for the whole factorisation.)
    const int GW = (pb.group_panels > 0 ? pb.group_panels : potrf_group_panels(n_pad)) * kNB;
    // (not with the theta-gradient's rider: a flow launch holds every compute unit, so the rider's substitution -- as many flops as
    //  the factorisation, which otherwise fill the chain's bubbles group by group -- would run behind it instead of beside it.
    //  Measured, one candidate, flow / separate launches: n = 8192 13.5 / 12.2 ms, 12288 38.6 / 35.3, 16384 77.8 / 76.5:
    //  profiles/r06_gradient_flow_ab.txt.  The likelihood egx_gp_likelihood_grad returns on such a handle is egx_gp_likelihood's
    //  to rounding, 1e-12, not bit for bit.)
    const bool flow = pb.sync != nullptr && pb.flow != 0 && nz == 1 && inv == nullptr && flow_fits(n_pad);
    // (a flow launch for the LAST columns of a right-looking factorisation: the switch happens on a group boundary)
    const int tail_cols = (pb.sync != nullptr && pb.flow_tail > 0 && nz == 1 && inv == nullptr && !pb.left && pb.flow_tail < n_pad && (n_pad - pb.flow_tail) % GW == 0 &&
                           flow_fits(pb.flow_tail))
                              ? pb.flow_tail
                              : 0;
    const bool have_sync = flow || tail_cols > 0 || (pb.sync != nullptr && pb.pipe != 0 &&
                                    pipe_fits((int)nz, ((pb.whole || GW > n_pad ? n_pad : GW) + kNB - 1) / kNB));
    const bool pipe = have_sync && !flow && tail_cols == 0;
    if (have_sync) EGX_HIP_CHECK(hipMemsetAsync(pb.sync, 0, sizeof(int) * (nz > 1 ? (size_t)pb.sS * nz : pipe_sync_ints(n_pad, m_tot)), s));
    auto gwidth = [&](int g0) { return (n_pad - g0 < GW) ? (n_pad - g0) : GW; };
    // panels of one group on stream `st`; `side` != nullptr splits every in-group update into the next diagonal block
    // (on st) and the rest (on side); `first_wait` is waited for before the FIRST panel solve (the rest of the update
    // that brought this group's columns up to date)
    auto inner_factor = [&](hipStream_t st, int g0, int gw, hipStream_t side, hipEvent_t first_wait) -> int {
        if (pipe) {  // one launch: diagonal blocks, panel solves and in-group updates hand over to each other on the device
            // (first_wait -- the rest of the update that brought this group's columns up to date, on another stream -- is
            //  waited for ON THE DEVICE by the first panel's solve tasks: pipe_signal behind that update)
            //  With more than two launch sequences of the handle in flight the workgroups that spin on that word -- 96 per launch,
            //  one compute unit each -- could leave the update they wait for no unit to run on (3 x 96 > 256): the STREAM waits then.
            const bool on_device = first_wait != nullptr && pb.seqs <= 2;
            if (first_wait && !on_device) EGX_HIP_CHECK(hipStreamWaitEvent(st, first_wait, 0));
            const int rc2 = launch_potrf_pipe(st, M, ld, n_pad, m_tot, dinv, info, pb, g0, gw, on_device ? g0 / kNB + 1 : 0, st != s);
            // (... and by the stream behind the launch: a matrix that lost a pivot skips its tasks without waiting for anything)
            if (rc2 == EGX_SUCCESS && on_device) EGX_HIP_CHECK(hipStreamWaitEvent(st, first_wait, 0));
            return rc2;
        }
        hipEvent_t pending = first_wait;
        for (int k0 = g0; k0 < g0 + gw; k0 += kNB) {
            const int nbk = (g0 + gw - k0 < kNB) ? (g0 + gw - k0) : kNB;
            potf2(st, k0, nbk);
            if (pending) {
                EGX_HIP_CHECK(hipStreamWaitEvent(st, pending, 0));
                pending = nullptr;
            }
            trsm(st, k0, nbk);
            const int r1 = k0 + nbk;
            if (r1 >= g0 + gw) break;
            const int ncols = g0 + gw - r1;
            const int nb1 = ncols < kNB ? ncols : kNB;  // width of the next panel
            const int rest_rows = m_tot - r1 - nb1;
            if (side && rest_rows >= 128) {
                int rc2 = update(st, r1, r1, nb1, nb1, k0, nbk, 1, nullptr);  // the next diagonal block only
                if (rc2) return rc2;
                EGX_HIP_CHECK(hipEventRecord(lk->ev_a, st));
                EGX_HIP_CHECK(hipStreamWaitEvent(side, lk->ev_a, 0));
                // everything below it (blocks above the diagonal inside this rectangle are written but never read)
                rc2 = update(side, r1 + nb1, r1, rest_rows, ncols, k0, nbk, 0, nullptr);
                if (rc2) return rc2;
                EGX_HIP_CHECK(hipEventRecord(lk->ev_b, side));
                pending = lk->ev_b;
            } else {
                int rc2 = update(st, r1, r1, m_tot - r1, ncols, k0, nbk, 1, nullptr);
                if (rc2) return rc2;
            }
        }
        if (pending) EGX_HIP_CHECK(hipStreamWaitEvent(st, pending, 0));
        return EGX_SUCCESS;
    };
    // The rows of W = C^-T through the group of panels [g0, g0 + gw), which is final on `s` when this is called (PotrfInverse,
    // egx_internal.h): on inv->sw, beside whatever the factorisation does next.  Row i of the identity is zero left of
    // column i, so panel k0 only has the rows [0, k0 + nbk) to solve and the updates as many rows to carry.
    GemmBatch gbw;
    if (inv) {
        gbw.count = (int)nz;
        gbw.sC = gbw.sA = inv->sW;
        gbw.sB = pb.sM;
        gbw.sInfo = pb.sI;
    }
    const bool w_left = pb.w_left != 0 && n_pad % 256 == 0;  // one decision for the whole factorisation (GW % 256 == 0)
    auto inverse_group = [&](hipStream_t sfin, int g0, int gw) -> int {
        if (!inv) return EGX_SUCCESS;
        hipStream_t sw = inv->sw;
        EGX_HIP_CHECK(hipEventRecord(inv->ev_grp, sfin));
        EGX_HIP_CHECK(hipStreamWaitEvent(sw, inv->ev_grp, 0));
        const int gend = g0 + gw;
        for (int k0 = g0; k0 < gend; k0 += kNB) {
            const int nbk = (gend - k0 < kNB) ? (gend - k0) : kNB;
            const int m_eff = k0 + nbk;
            double *P = inv->W + k0;
            const double *L = M + (int64_t)k0 * ld + k0;
            const double *lin = dinv + (int64_t)(k0 / 64) * 4096;
            if (m_eff >= 4096)
                hipLaunchKernelGGL((k_panel_trsm16<2, false>), dim3(m_eff / 32, 1, nz), dim3(256), 0, sw, P, inv->ldw, L, ld, lin, nbk,
                                   (const int *)info, inv->sW, pb.sM, pb.sD, pb.sI, (const double *)nullptr);
            else
                hipLaunchKernelGGL((k_panel_trsm16<1, false>), dim3(m_eff / 16, 1, nz), dim3(256), 0, sw, P, inv->ldw, L, ld, lin, nbk,
                                   (const int *)info, inv->sW, pb.sM, pb.sD, pb.sI, (const double *)nullptr);
            const int ncols = gend - (k0 + nbk);
            if (ncols > 0) {
                int rc2 = launch_gemm_nt_sub(sw, inv->W + (k0 + nbk), inv->ldw, inv->W + k0, inv->ldw,
                                             M + (int64_t)(k0 + nbk) * ld + k0, ld, m_eff, ncols, nbk, 0, 0, nullptr, info, &gbw);
                if (rc2) return rc2;
            }
        }
        const int ncols = n_pad - gend;
        if (ncols > 0 && w_left) {
            // left-looking rider (round 4; w_left_for): the NEXT group's columns of W receive all earlier columns at once,
            //     W[0:gend, next) = -W[0:gend, 0:gend) L[next, 0:gend)^T      (rows below gend: the identity, untouched)
            // by one launch of the stream kernel with per-tile K ranges (row i of W is zero left of column i) and
            // zero-initialised accumulators: a tile of W is written once by a long K loop, then read and written once by
            // its group's own solves -- instead of one read-modify-write pass (K = 1024) per earlier group.  It only reads
            // rows of the factor that are final with THIS group, so it is issued here, ahead of the next group's chain.
            const int gwn = gwidth(gend);
            const int nbx = gend / 128, nby = gwn / 256, nt = nbx * nby;
            hipLaunchKernelGGL((k_gemm_stream<false, true>), dim3((unsigned)nt * nz, 1, 1), dim3(512), ST_LDS_BYTES, sw, inv->W + gend,
                               inv->ldw, (const double *)inv->W, inv->ldw, (const double *)(M + (int64_t)gend * ld), ld, gend, nbx,
                               nby, nt, (const int *)info, gbw);
            EGX_HIP_CHECK(hipGetLastError());
        } else if (ncols > 0) {
            // (round 6 measured this update SPLIT into the next group's columns on this stream and the rest on a second rider stream
            //  beside the next group's solves -- the timeline shows the rider six groups behind the factorisation with ~0.7 ms of
            //  small launches exposed per group: one candidate 76.5 -> 79.4 ms at n = 16384, 35.3 -> 39.5 at 12288, 12.2 -> 14.2 at 8192
            //  (a third stream competing with the factorisation, two launches' tails for one): profiles/r06_rider_split_ab.txt; gone)
            int rc2 = launch_gemm_nt_sub(sw, inv->W + gend, inv->ldw, inv->W + g0, inv->ldw, M + (int64_t)gend * ld + g0, ld, gend,
                                         ncols, gw, 0, 0, nullptr, info, &gbw);
            if (rc2) return rc2;
        }
        return EGX_SUCCESS;
    };
    // (decided per HANDLE, never by the number of matrices in a launch: a matrix gets the same bits in any batch -- and with or
    //  without the theta-gradient's rider: its substitution then follows the launch group by group instead of riding along)
    if (flow || (pipe && pb.whole)) {
        rc = flow ? launch_potrf_flow(s, M, ld, n_pad, m_tot, dinv, info, pb) : launch_potrf_pipe(s, M, ld, n_pad, m_tot, dinv, info, pb, 0, n_pad);
        if (rc) return rc;
        if (inv) {
            for (int g0 = 0; g0 < n_pad; g0 += GW) {
                rc = inverse_group(s, g0, gwidth(g0));
                if (rc) return rc;
            }
            EGX_HIP_CHECK(hipEventRecord(inv->ev_done, inv->sw));
            EGX_HIP_CHECK(hipStreamWaitEvent(s, inv->ev_done, 0));
        }
        hipLaunchKernelGGL(k_diag_tile_inverses, dim3(n_pad / 64, 1, nz), dim3(256), 0, s, (const double *)M, ld, dinv,
                           dinv + (int64_t)(n_pad / 64) * 4096, pb.sM, pb.sD);
        EGX_HIP_CHECK(hipGetLastError());
        return EGX_SUCCESS;
    }
    if (pb.left) {
        // LEFT-looking over the groups of panels (round 4): the columns of group J receive the contributions of ALL earlier
        // columns in two updates -- a LONG one, K = every column before the previous group, and a SHORT one, K = the previous
        // group (1024) -- instead of one read-modify-write pass per earlier group: a tile of the factor is read and written
        // twice in the whole factorisation, by long K loops (what makes the theta-gradient's R^-1 launch run at 0.85 of
        // peak; the long updates of a lock-step group of eight measured 0.82, the right-looking trailing updates 0.72).
        // Look-ahead: the chain of group J (diagonal blocks, panel solves, in-group updates; stream s2) runs beside the
        // long update of group J + 1 (stream s), which only needs the groups before J:
        //     s :  ... U_long(J) | wait C(J-1) | U_short(J) | rec U(J) | U_long(J+1) | wait C(J) | U_short(J+1) ...
        //     s2:                                            wait U(J) | chain(J) | rec C(J)
        // A lone matrix' late long updates have few tiles with very long K loops and leave the chip underfilled: the mode is
        // chosen per HANDLE (n_pad >= 14336 and a lock-step width of at least eight, PotrfBatch::left), never by the number of
        // matrices in a launch, so a candidate's bits do not depend on its companions.
        const bool la = s2 != nullptr;
        hipStream_t sc = la ? s2 : s;  // the chain's stream
        rc = inner_factor(s, 0, gwidth(0), s3, nullptr);
        if (rc) return rc;
        rc = inverse_group(s, 0, gwidth(0));
        if (rc) return rc;
        int gprev = 0;  // start of the previous group
        for (int g0 = gwidth(0); g0 < n_pad; g0 += GW) {
            const int gw = gwidth(g0);
            if (gprev > 0) {  // U_long: columns [0, gprev), all final (C(J-2) was waited for before U_short(J-1))
                const bool timed = trace && trace->ready && trace->used < GemmTrace::kMax;
                if (timed) EGX_HIP_CHECK(hipEventRecord(trace->e0[trace->used], s));
                rc = update(s, g0, g0, m_tot - g0, gw, 0, gprev, 1, nullptr, 1);
                if (rc) return rc;
                if (timed) {
                    EGX_HIP_CHECK(hipEventRecord(trace->e1[trace->used], s));
                    const double nr = (double)(n_pad - g0);
                    trace->flops[trace->used] = (double)nz * 2.0 * gprev * (nr * gw - 0.5 * gw * (gw - 1.0));
                    trace->used++;
                }
            }
            if (la && gprev > 0) EGX_HIP_CHECK(hipStreamWaitEvent(s, lk->ev_panel, 0));  // C(J-1): the previous group is final
            rc = update(s, g0, g0, m_tot - g0, gw, gprev, g0 - gprev, 1, nullptr, 2);     // U_short
            if (rc) return rc;
            if (la) {
                EGX_HIP_CHECK(hipEventRecord(lk->ev_lu, s));
                EGX_HIP_CHECK(hipStreamWaitEvent(sc, lk->ev_lu, 0));
            }
            rc = inner_factor(sc, g0, gw, la ? s3 : nullptr, nullptr);
            if (rc) return rc;
            if (la) EGX_HIP_CHECK(hipEventRecord(lk->ev_panel, sc));
            rc = inverse_group(sc, g0, gw);
            if (rc) return rc;
            gprev = g0;
        }
        if (la && gprev > 0) EGX_HIP_CHECK(hipStreamWaitEvent(s, lk->ev_panel, 0));
        if (inv) {
            EGX_HIP_CHECK(hipEventRecord(inv->ev_done, inv->sw));
            EGX_HIP_CHECK(hipStreamWaitEvent(s, inv->ev_done, 0));
        }
        hipLaunchKernelGGL(k_diag_tile_inverses, dim3(n_pad / 64, 1, nz), dim3(256), 0, s, (const double *)M, ld, dinv,
                           dinv + (int64_t)(n_pad / 64) * 4096, pb.sM, pb.sD);
        EGX_HIP_CHECK(hipGetLastError());
        return EGX_SUCCESS;
    }
    rc = inner_factor(s, 0, gwidth(0), s3, nullptr);
    if (rc) return rc;
    rc = inverse_group(s, 0, gwidth(0));
    if (rc) return rc;
    for (int g0 = 0; g0 < n_pad; g0 += GW) {
        const int gw = gwidth(g0);
        const int r1 = g0 + gw;  // first row/col of the trailing matrix
        if (r1 >= n_pad) break;  // (right-hand-side rows below the last block were solved by its panel)
        const int gw1 = gwidth(r1);
        if (tail_cols > 0 && n_pad - r1 == tail_cols) {
            // the flow tail: ONE update of the whole trailing matrix by this group, then the trailing matrix -- every update of the
            // columns before it applied -- is a factorisation of its own, as one flow launch (pipe_flow.h)
            rc = update(s, r1, r1, m_tot - r1, n_pad - r1, g0, gw, 1, nullptr);
            if (rc) return rc;
            rc = launch_potrf_flow(s, M + (int64_t)r1 * ld + r1, ld, n_pad - r1, m_tot - r1, dinv + (int64_t)(r1 / 64) * 4096, info, pb, r1);
            if (rc) return rc;
            for (int t0 = r1; t0 < n_pad; t0 += GW) {
                rc = inverse_group(s, t0, gwidth(t0));
                if (rc) return rc;
            }
            break;
        }
        // the cross-stream hand-off costs ~2 x 10 us; below ~3k trailing columns RU is shorter than that (and with
        // several fits in flight the extra hand-offs cost more than they hide: profiles/r02_run23_tail_lookahead_ab.txt)
        const bool look = (s2 != nullptr) && (n_pad - r1 - gw1 >= g_look_min_cols);
        const int nb1 = gw1 < kNB ? gw1 : kNB;
        if (look && s3) {
            // LUd: the next group's first diagonal block; its chain may start.  LUr: the rest of the next group's columns.
            rc = update(s, r1, r1, nb1, nb1, g0, gw, 1, nullptr);
            if (rc) return rc;
            EGX_HIP_CHECK(hipEventRecord(lk->ev_lu, s));
            EGX_HIP_CHECK(hipStreamWaitEvent(s2, lk->ev_lu, 0));
            // LUr writes the next group's columns, RU the columns right of them, both only READ this group's panel: LUr goes
            // to the (high-priority) side stream, so that RU fills the CUs LUr's last, partly filled round of tiles leaves idle
            // (+0.8 % on the sweep, +1 % on a lone fit: profiles/r03_run4_lur_side_ab.txt, r03_run8_*).  Round 3 kept a lone
            // matrix' LUr in front of RU because the driver line's roofline timed those RU launches one by one; since round 4
            // that leg switches the side stream off for itself (egx_set_tuning "lur_side" = 0).  From n_pad 14336 on only
            // (n = 16384 lone: 31.5 -> 30.9 ms; n = 8192: one fit 6.24 -> 6.44 ms, twelve in lock-step 260 -> 258 fits/s:
            // profiles/r04_run11_lur_side_lone_ab.txt).  In the last ~6000 columns the chain is what a factorisation waits for and
            // its first panel solve waits for LUr: there LUr runs ALONE in front of RU (250 us instead of 520-770 beside it;
            // n = 16384 lone: 29.85 -> 29.6 ms, profiles/r04_run14_*).  Streams only: the arithmetic is the same either way.
            hipStream_t slu = (g_lur_side && n_pad >= 14336 && n_pad - r1 - gw1 >= kLurSideMinCols) ? s3 : s;
            if (slu != s) EGX_HIP_CHECK(hipStreamWaitEvent(slu, lk->ev_lu, 0));
            rc = update(slu, r1 + nb1, r1, m_tot - r1 - nb1, gw1, g0, gw, 0, nullptr);
            if (rc) return rc;
            EGX_HIP_CHECK(hipEventRecord(lk->ev_lur, slu));
            if (pipe) {
                rc = pipe_signal(slu, pb, r1 / kNB + 1);
                if (rc) return rc;
            }
            rc = inner_factor(s2, r1, gw1, s3, lk->ev_lur);
            if (rc) return rc;
            EGX_HIP_CHECK(hipEventRecord(lk->ev_panel, s2));
        } else if (!look) {
            // no look-ahead (small matrices, the last ~3000 columns): nothing runs beside the trailing update, so the next
            // group's columns and the rest are ONE launch (round 4: one ramp-up and one tail instead of two)
            const bool timed = trace && trace->ready && trace->used < GemmTrace::kMax;
            if (timed) EGX_HIP_CHECK(hipEventRecord(trace->e0[trace->used], s));
            bool big = false;
            rc = update(s, r1, r1, m_tot - r1, n_pad - r1, g0, gw, 1, &big);
            if (rc) return rc;
            if (timed && big) {
                EGX_HIP_CHECK(hipEventRecord(trace->e1[trace->used], s));
                const double nc = (double)(n_pad - r1);
                trace->flops[trace->used] = (double)nz * 2.0 * gw * nc * (nc + 1.0) / 2.0;
                trace->used++;
            }
        } else {
            // LU: the next group's columns only (look-ahead without a side stream)
            rc = update(s, r1, r1, m_tot - r1, gw1, g0, gw, 1, nullptr);
            if (rc) return rc;
            EGX_HIP_CHECK(hipEventRecord(lk->ev_lu, s));
            EGX_HIP_CHECK(hipStreamWaitEvent(s2, lk->ev_lu, 0));
            rc = inner_factor(s2, r1, gw1, nullptr, nullptr);
            if (rc) return rc;
            EGX_HIP_CHECK(hipEventRecord(lk->ev_panel, s2));
        }
        // RU: the rest of the trailing matrix
        const int r2 = r1 + gw1;
        if (r2 < n_pad && look) {
            const bool timed = trace && trace->ready && trace->used < GemmTrace::kMax;
            if (timed) EGX_HIP_CHECK(hipEventRecord(trace->e0[trace->used], s));
            bool big = false;
            rc = update(s, r2, r2, m_tot - r2, n_pad - r2, g0, gw, 1, &big);
            if (rc) return rc;
            if (timed && big) {
                EGX_HIP_CHECK(hipEventRecord(trace->e1[trace->used], s));
                const double nc = (double)(n_pad - r2);
                trace->flops[trace->used] = (double)nz * 2.0 * gw * nc * (nc + 1.0) / 2.0;
                trace->used++;
            }
        }
        if (look) {
            EGX_HIP_CHECK(hipStreamWaitEvent(s, lk->ev_panel, 0));
        } else {
            rc = inner_factor(s, r1, gw1, nullptr, nullptr);
            if (rc) return rc;
        }
        rc = inverse_group(s, r1, gw1);  // the group that has just become final
        if (rc) return rc;
    }
    if (inv) {  // W is complete; its panel solves read the 16 x 16 inverses the launch below overwrites
        EGX_HIP_CHECK(hipEventRecord(inv->ev_done, inv->sw));
        EGX_HIP_CHECK(hipStreamWaitEvent(s, inv->ev_done, 0));
    }
    // every stream has been joined into `s`: the 64x64 tile inverses of the complete factor(s), one launch
    hipLaunchKernelGGL(k_diag_tile_inverses, dim3(n_pad / 64, 1, nz), dim3(256), 0, s, (const double *)M, ld, dinv,
                       dinv + (int64_t)(n_pad / 64) * 4096, pb.sM, pb.sD);
    EGX_HIP_CHECK(hipGetLastError());
    return EGX_SUCCESS;
}

int launch_trsm_rows(hipStream_t s, const double *M, int64_t ldm, int n_pad, const double *dinv, double *RT,
                     int64_t ldr, int m, int tri_rows, const TrsmBatch *batch) {
    int rc = chol_init();
    if (rc) return rc;
    const TrsmBatch tb = batch ? *batch : TrsmBatch();
    const unsigned nz = (unsigned)(tb.count > 0 ? tb.count : 1);
    GemmBatch gb;  // the updates: C and A in the right-hand sides' buffers, B in the factors
    gb.count = (int)nz;
    gb.sC = gb.sA = tb.sR;
    gb.sB = tb.sM;
    if (m % kTile || n_pad % kTile) {
        set_error("trsm_rows: sizes must be multiples of 128");
        return EGX_ERR_INVALID_VALUE;
    }
    // same two-level blocking as launch_potrf: block forward substitution inside a group of panels, then ONE update
    // of the remaining columns per group (K = group width)
    // (groups of four panels: K = 1024 per update of the remaining columns.  The factorisation pays for wide groups with
    //  a longer serial chain; here every row tile is independent, so the width is bounded by the in-group work only)
    const int GW = kTrsmGroupPanels * kNB;
    // LEFT-looking over the groups (round 4; dense right-hand sides only): the columns of group J receive
    // the contributions of ALL earlier columns in one update with K = g0 before the group's own block substitution --
    // every tile of RT is read and written once by a long K loop instead of once per earlier group (K = 1024).  With
    // m >= 128 rows a launch always has m / 128 x 4 tiles: none of the underfill a lone factorisation's late updates have.
    const bool left = !tri_rows;
    for (int g0 = 0; g0 < n_pad; g0 += GW) {
        const int gw = (n_pad - g0 < GW) ? (n_pad - g0) : GW;
        const int gend = g0 + gw;
        if (left && g0 > 0) {
            rc = launch_gemm_nt_sub(s, RT + g0, ldr, RT, ldr, M + (int64_t)g0 * ldm, ldm, m, gw, g0, 0, 0, nullptr, nullptr, &gb);
            if (rc) return rc;
        }
        for (int k0 = g0; k0 < gend; k0 += kNB) {
            const int nbk = (gend - k0 < kNB) ? (gend - k0) : kNB;
            const double *diag = M + (int64_t)k0 * ldm + k0;
            const double *dtiles = dinv + (int64_t)(k0 / 64) * 4096;
            // tri_rows: the right-hand sides are the rows of the identity, so the solution (C^-T) is upper triangular:
            // rows below the current block are still zero in these columns and are skipped
            const int m_eff = tri_rows ? ((k0 + nbk < m) ? (k0 + nbk) : m) : m;
            // the same register-resident 16-row-tile solve as inside the factorisation (POST layout of the tile inverses)
            const double *tfl = dinv + (int64_t)(n_pad / 64) * 4096 + k0 / 64;
            if (m_eff >= 4096)
                hipLaunchKernelGGL((k_panel_trsm16<2, true>), dim3(m_eff / 32, 1, nz), dim3(256), 0, s, RT + k0, ldr, diag, ldm,
                                   dtiles, nbk, (const int *)nullptr, tb.sR, tb.sM, tb.sD, 0, tfl);
            else
                hipLaunchKernelGGL((k_panel_trsm16<1, true>), dim3(m_eff / 16, 1, nz), dim3(256), 0, s, RT + k0, ldr, diag, ldm,
                                   dtiles, nbk, (const int *)nullptr, tb.sR, tb.sM, tb.sD, 0, tfl);
            const int ncols = gend - (k0 + nbk);
            if (ncols > 0) {
                rc = launch_gemm_nt_sub(s, RT + (k0 + nbk), ldr, RT + k0, ldr, M + (int64_t)(k0 + nbk) * ldm + k0, ldm,
                                        m_eff, ncols, nbk, 0, 0, nullptr, nullptr, &gb);
                if (rc) return rc;
            }
        }
        const int ncols = n_pad - gend;
        if (ncols > 0 && !left) {
            const int m_eff = tri_rows ? ((gend < m) ? gend : m) : m;
            rc = launch_gemm_nt_sub(s, RT + gend, ldr, RT + g0, ldr, M + (int64_t)gend * ldm + g0, ldm, m_eff, ncols, gw, 0,
                                    0, nullptr, nullptr, &gb);
            if (rc) return rc;
        }
    }
    EGX_HIP_CHECK(hipGetLastError());
    return EGX_SUCCESS;
}

// W <- the rows of the identity, as far as launch_trsm_rows(tri_rows) and launch_syrk_uptri_neg ever read or update
// them: every element (i, j) with i < end of the panel group that holds column j (the upper triangle plus the lower
// parts of the diagonal groups); the rest of the buffer is never touched.  One 128 x 128 tile per workgroup.
__global__ __launch_bounds__(256) void k_identity_rows(double *__restrict__ W, int64_t ld, int n_pad, int gw, int64_t bsW) {
    const int bi = blockIdx.y, bj = blockIdx.x;
    int gend = (bj * 128 / gw + 1) * gw;
    if (gend > n_pad) gend = n_pad;
    if (bi * 128 >= gend) return;
    W += (int64_t)blockIdx.z * bsW + (int64_t)bi * 128 * ld + bj * 128;
    const int tid = threadIdx.x, c = (tid & 63) * 2, r0 = tid >> 6;
    for (int r = r0; r < 128; r += 4) {
        d2_t v = d2_t{0.0, 0.0};
        if (bi == bj) {
            if (r == c) v[0] = 1.0;
            if (r == c + 1) v[1] = 1.0;
        }
        *reinterpret_cast<d2_t *>(W + (int64_t)r * ld + c) = v;
    }
}

int trsm_group_cols() { return kTrsmGroupPanels * kNB; }

int launch_identity_rows(hipStream_t s, double *W, int64_t ld, int n_pad, int count, int64_t stride) {
    int rc = chol_init();
    if (rc) return rc;
    hipLaunchKernelGGL(k_identity_rows, dim3(n_pad / 128, n_pad / 128, (unsigned)(count > 0 ? count : 1)), dim3(256), 0, s, W, ld,
                       n_pad, trsm_group_cols(), stride);
    EGX_HIP_CHECK(hipGetLastError());
    return EGX_SUCCESS;
}

// C (lower; n x n, ldc) <- -(W W^T) for the upper triangular W = C^-T (n x n, ldw): the theta-gradient's -R^-1.
// n % 256 == 0: the LDS-DMA stream kernel with per-tile K ranges and zero-initialised accumulators (C is not read);
// other sizes (n_pad < 4096 only): C is zeroed and the register-staged 128 x 128 kernel subtracts.
int launch_syrk_uptri_neg(hipStream_t s, double *C, int64_t ldc, const double *W, int64_t ldw, int n, const GemmBatch *batch) {
    int rc = chol_init();
    if (rc) return rc;
    if (n % 128) {
        set_error("syrk_uptri_neg: n must be a multiple of 128");
        return EGX_ERR_INVALID_VALUE;
    }
    const GemmBatch bt = batch ? *batch : GemmBatch();
    const unsigned nz = (unsigned)(bt.count > 0 ? bt.count : 1);
    if (n % 256 == 0) {
        const int nbx = n / 128, m = nbx / 2;
        const int nt = m * (m + 1);
        hipLaunchKernelGGL((k_gemm_stream<true, true>), dim3((unsigned)nt, 1, nz), dim3(512), ST_LDS_BYTES, s, C, ldc, W, ldw, W,
                           ldw, n, nbx, m, nt, (const int *)nullptr, bt);
        EGX_HIP_CHECK(hipGetLastError());
        return EGX_SUCCESS;
    }
    for (unsigned z = 0; z < nz; z++)
        EGX_HIP_CHECK(hipMemset2DAsync(C + (int64_t)z * bt.sC, sizeof(double) * ldc, 0, sizeof(double) * n, n, s));
    return launch_gemm_nt_sub(s, C, ldc, W, ldw, W, ldw, n, n, n, 1, 1, nullptr, nullptr, batch);
}

int launch_diag_tile_inverses(hipStream_t s, const double *M, int64_t ld, int n_pad, double *dinv) {
    hipLaunchKernelGGL(k_diag_tile_inverses, dim3(n_pad / 64), dim3(256), 0, s, M, ld, dinv, dinv + (int64_t)(n_pad / 64) * 4096,
                       (int64_t)0, (int64_t)0);
    EGX_HIP_CHECK(hipGetLastError());
    return EGX_SUCCESS;
}

int launch_block_inverse(hipStream_t s, const double *M, int64_t ld, int n_pad, const double *dinv, double *Wall) {
    int rc = chol_init();
    if (rc) return rc;
    constexpr int lds = GemmShape<64, 64, 16, 32, 512>::LDS_BYTES;
    hipLaunchKernelGGL(k_block_inv256, dim3((n_pad + kNB - 1) / kNB), dim3(512), lds, s, M, ld, n_pad, dinv, Wall);
    EGX_HIP_CHECK(hipGetLastError());
    return EGX_SUCCESS;
}

int launch_trsv_t(hipStream_t s, const double *M, int64_t ld, int n_pad, const double *Wall, double *v,
                  double *xout, bool per_block) {
    // the one-launch form leaves v alone; the launch-per-block form updates it in place above the current block
    const int nblocks = (n_pad + kNB - 1) / kNB;
    if (!per_block && g_trsv_fused.load() != 0 && n_pad % 64 == 0) {  // one launch (k_trsv_t_fused); its hand-off words sit behind the inverse blocks
        double *tail = const_cast<double *>(Wall) + (int64_t)nblocks * 65536;
        EGX_HIP_CHECK(hipMemsetAsync(tail, 0xFF, sizeof(double) * trsv_tail_doubles(n_pad), s));
        EGX_HIP_CHECK(hipMemsetAsync(xout, 0xFF, sizeof(double) * (size_t)n_pad, s));
        hipLaunchKernelGGL(k_trsv_t_fused, dim3((unsigned)(n_pad / 64)), dim3(256), 0, s, M, ld, n_pad, Wall, (const double *)v, xout, tail,
                           pipe_timeout_ticks(), EGX_TRSV_STALL);
        EGX_HIP_CHECK(hipGetLastError());
        return EGX_SUCCESS;
    }
    for (int b = nblocks - 1; b >= 0; b--) {
        const int k0 = b * kNB;
        const int nbk = (n_pad - k0 < kNB) ? (n_pad - k0) : kNB;
        hipLaunchKernelGGL(k_trsv_w, dim3(8), dim3(256), 0, s, Wall + (int64_t)b * 65536, nbk,
                           (const double *)(v + k0), xout + k0);
        if (k0 > 0)
            hipLaunchKernelGGL(k_gemv_t_update, dim3(k0 / 64), dim3(256), 0, s, M + (int64_t)k0 * ld, ld, nbk,
                               (const double *)(xout + k0), v);
    }
    EGX_HIP_CHECK(hipGetLastError());
    return EGX_SUCCESS;
}

// launch_block_inverse / launch_trsv_t for `count` models of one shape in lock-step (blockIdx.y = model):
// dW[j] <- the 256 x 256 inverse blocks of factor j;  vec[j] (n_pad) <- C_j^-T rhs[j]
int launch_block_inverse_batch(hipStream_t s, const SolveBatchPtrs &b, int count, int64_t ld, int n_pad) {
    int rc = chol_init();
    if (rc) return rc;
    constexpr int lds = GemmShape<64, 64, 16, 32, 512>::LDS_BYTES;
    hipLaunchKernelGGL(k_block_inv256_batch, dim3((unsigned)((n_pad + kNB - 1) / kNB), (unsigned)count), dim3(512), lds, s, b, ld, n_pad);
    EGX_HIP_CHECK(hipGetLastError());
    return EGX_SUCCESS;
}
int launch_trsv_t_batch(hipStream_t s, const SolveBatchPtrs &b, int count, int64_t ld, int n_pad) {
    const int nblocks = (n_pad + kNB - 1) / kNB;
    if (g_trsv_fused.load() != 0 && n_pad % 64 == 0) {
        for (int z = 0; z < count; z++) {
            EGX_HIP_CHECK(hipMemsetAsync(b.dW[z] + (int64_t)nblocks * 65536, 0xFF, sizeof(double) * trsv_tail_doubles(n_pad), s));
            EGX_HIP_CHECK(hipMemsetAsync(b.vec[z], 0xFF, sizeof(double) * (size_t)n_pad, s));
        }
        hipLaunchKernelGGL(k_trsv_t_fused_batch, dim3((unsigned)(n_pad / 64), (unsigned)count), dim3(256), 0, s, b, ld, n_pad,
                           pipe_timeout_ticks(), EGX_TRSV_STALL);
        EGX_HIP_CHECK(hipGetLastError());
        return EGX_SUCCESS;
    }
    for (int blk = nblocks - 1; blk >= 0; blk--) {
        const int k0 = blk * kNB;
        const int nbk = (n_pad - k0 < kNB) ? (n_pad - k0) : kNB;
        hipLaunchKernelGGL(k_trsv_w_batch, dim3(8, (unsigned)count), dim3(256), 0, s, b, blk, k0, nbk);
        if (k0 > 0) hipLaunchKernelGGL(k_gemv_t_update_batch, dim3((unsigned)(k0 / 64), (unsigned)count), dim3(256), 0, s, b, ld, k0, nbk);
    }
    EGX_HIP_CHECK(hipGetLastError());
    return EGX_SUCCESS;
}

int mfma_probe(double *max_abs_err) {
    double hA[256], hB[256], hC[256];
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) {
            hA[i * 16 + j] = 1.0 + i * 0.37 - j * 0.11 + (i * j % 5) * 0.013;
            hB[i * 16 + j] = -0.5 + i * 0.29 + j * j * 0.017 - (i % 3) * 0.21;
        }
    double *dA, *dB, *dC;
    EGX_HIP_CHECK(hipMalloc(&dA, sizeof hA));
    EGX_HIP_CHECK(hipMalloc(&dB, sizeof hB));
    EGX_HIP_CHECK(hipMalloc(&dC, sizeof hC));
    EGX_HIP_CHECK(hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice));
    EGX_HIP_CHECK(hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_mfma_probe, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    EGX_HIP_CHECK(hipGetLastError());
    EGX_HIP_CHECK(hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost));
    hipFree(dA);
    hipFree(dB);
    hipFree(dC);
    double worst = 0.0;
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) {
            double ref = 0.0;
            for (int k = 0; k < 16; k++) ref += hA[i * 16 + k] * hB[k * 16 + j];
            double e = ref - hC[i * 16 + j];
            if (e < 0) e = -e;
            if (e > worst) worst = e;
        }
    *max_abs_err = worst;
    return EGX_SUCCESS;
}

}  // namespace egx
