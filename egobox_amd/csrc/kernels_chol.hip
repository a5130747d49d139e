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

#include <cstdlib>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

namespace egx {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));

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

// =============================================================================================
// MFMA GEMM core:  acc += A_tile (BM x K) * B_tile (BN x K)^T, both operands K-contiguous.
// 256 threads = 4 waves laid out (BM/WM) x (BN/WN); each wave owns a WM x WN sub-tile made of
// 16x16 MFMA tiles.  v_mfma_f64_16x16x4_f64 operand layout (lane l):
//   A: A[i = l & 15][k = l >> 4]      B: B[k = l >> 4][j = l & 15]
//   C/D reg r: row = (l >> 4) + 4 r, col = l & 15
// =============================================================================================
constexpr int KC = 16;      // K chunk staged per iteration
constexpr int LDS_LD = 18;  // doubles per staged tile row (16 + 2 pad): conflict-free ds_read_b64

// Global -> register staging of one K chunk of a tile.  `g` is the wave-uniform tile base (+ k0): the
// per-thread part of the address is a 32-bit element offset so the loads use the SGPR-base + VGPR-offset form
// (64-bit per-thread addresses cost 2 VGPRs per load and pushed the 128-VGPR trailing-update kernel into scratch).
template <int ROWS, int NT>
__device__ __forceinline__ void tile_load_regs(const double *__restrict__ g, const unsigned (&off)[ROWS * 8 / NT],
                                               d2_t (&r)[ROWS * 8 / NT]) {
#pragma unroll
    for (int i = 0; i < ROWS * 8 / NT; i++) r[i] = *reinterpret_cast<const d2_t *>(g + off[i]);
}
template <int ROWS, int NT>
__device__ __forceinline__ void tile_offsets(int64_t ld, unsigned (&off)[ROWS * 8 / NT], int tid) {
#pragma unroll
    for (int i = 0; i < ROWS * 8 / NT; i++) {
        int p = tid + NT * i;
        int row = p >> 3, part = p & 7;
        off[i] = (unsigned)(row * (int)ld + part * 2);
    }
}
template <int ROWS, int NT>
__device__ __forceinline__ void tile_store_lds(double *s, const d2_t (&r)[ROWS * 8 / NT], int tid) {
#pragma unroll
    for (int i = 0; i < ROWS * 8 / NT; i++) {
        int p = tid + NT * i;
        int row = p >> 3, part = p & 7;
        *reinterpret_cast<d2_t *>(s + row * LDS_LD + part * 2) = r[i];
    }
}

// A single wave can issue v_mfma_f64_16x16x4_f64 only at ~46 % of the pipe rate; two MFMA-ready waves on a
// SIMD reach 99 % (tools/fp64_peak.hip, profiles/r01_fp64_peak_microbench.txt).  The big trailing-update tile
// therefore uses 8 waves per workgroup with 32x64 wave tiles (<= 128 VGPRs) so that 4 waves share a SIMD.
template <int BM, int BN, int WM, int WN, int NTHREADS = 256>
struct GemmShape {
    static constexpr int MT = WM / 16, NT = WN / 16;
    static constexpr int WAVES_N = BN / WN;
    static constexpr int A_TILE = BM * LDS_LD, B_TILE = BN * LDS_LD;
    static constexpr int STAGE = A_TILE + B_TILE;
    static constexpr int LDS_BYTES = 2 * STAGE * 8;
    static_assert((BM / WM) * (BN / WN) * 64 == NTHREADS, "one wave per wave tile");
    static_assert((BM * 8) % NTHREADS == 0 && (BN * 8) % NTHREADS == 0, "tile rows per load pass");
};

template <int BM, int BN, int WM, int WN, int NTHREADS = 256>
__device__ __forceinline__ void gemm_core(const double *__restrict__ A, int64_t lda,
                                          const double *__restrict__ B, int64_t ldb, int K,
                                          double4_t (&acc)[WM / 16][WN / 16], double *smem, int tid) {
    using S = GemmShape<BM, BN, WM, WN, NTHREADS>;
    const int nchunks = K / KC;
    if (nchunks <= 0) return;
    const int wave = tid >> 6, lane = tid & 63;
    const int wm0 = (wave / S::WAVES_N) * WM, wn0 = (wave % S::WAVES_N) * WN;
    const int frow = lane & 15, fk = lane >> 4;

    d2_t ra[BM * 8 / NTHREADS], rb[BN * 8 / NTHREADS];
    unsigned oa[BM * 8 / NTHREADS], ob[BN * 8 / NTHREADS];
    tile_offsets<BM, NTHREADS>(lda, oa, tid);
    tile_offsets<BN, NTHREADS>(ldb, ob, tid);
    tile_load_regs<BM, NTHREADS>(A, oa, ra);
    tile_load_regs<BN, NTHREADS>(B, ob, rb);
    tile_store_lds<BM, NTHREADS>(smem, ra, tid);
    tile_store_lds<BN, NTHREADS>(smem + S::A_TILE, rb, tid);
    __syncthreads();
    for (int c = 0; c < nchunks; c++) {
        double *As = smem + (c & 1) * S::STAGE;
        double *Bs = As + S::A_TILE;
        const bool more = (c + 1 < nchunks);
        if (more) {
            tile_load_regs<BM, NTHREADS>(A + (c + 1) * KC, oa, ra);
            tile_load_regs<BN, NTHREADS>(B + (c + 1) * KC, ob, rb);
        }
#pragma unroll
        for (int kk = 0; kk < KC / 4; kk++) {
            double a[S::MT], b[S::NT];
#pragma unroll
            for (int mi = 0; mi < S::MT; mi++)
                a[mi] = As[(wm0 + mi * 16 + frow) * LDS_LD + kk * 4 + fk];
#pragma unroll
            for (int ni = 0; ni < S::NT; ni++)
                b[ni] = Bs[(wn0 + ni * 16 + frow) * LDS_LD + kk * 4 + fk];
#pragma unroll
            for (int mi = 0; mi < S::MT; mi++)
#pragma unroll
                for (int ni = 0; ni < S::NT; ni++)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
        if (more) {
            double *An = smem + ((c + 1) & 1) * S::STAGE;
            tile_store_lds<BM, NTHREADS>(An, ra, tid);
            tile_store_lds<BN, NTHREADS>(An + S::A_TILE, rb, tid);
        }
        __syncthreads();
    }
}

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
        if (g == 0) {  // the workgroup's very first chunk: the classic wait + barrier in front of its first read
            if (issued > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            read_half(0, 0, a0, b0);
        }
        for (int ch = 0; ch < nch; ch++, g++) {
            d2_t a1[4], b1[4];
            read_half(stage, 1, a1, b1);
            mma_half(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            const bool next = g + 1 < total;   // another chunk follows (possibly the next tile's first)
            const bool more = issued < total;  // ... and one more to prefetch
            const int st1 = (stage == ST_NSTAGE - 1) ? 0 : stage + 1;
            if (next) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // own pieces of chunk g + 1 (the only ones in flight)
                __builtin_amdgcn_s_barrier();
                if (more) issue_pieces(stage == 0 ? 2 : stage - 1, 0, 6);  // chunk g + 2 -> stage of chunk g - 1
            }
            __builtin_amdgcn_sched_barrier(0);
            mma_quarter(a1, b1, 0);  // (fragments read before the first half: nothing to wait for behind the barrier)
            __builtin_amdgcn_sched_barrier(0);
            if (next && more) issue_done();
            if (next) read_half(st1, 0, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            mma_quarter(a1, b1, 1);
            stage = st1;
        }
        // (the 16 row pointers of the C tile are rebuilt here instead of staying live through the K loop: the opaque
        //  pass through an empty asm keeps the compiler from carrying 32 address VGPRs across 2048 MFMAs)
#pragma unroll
        for (int r = 0; r < 4; r++) asm volatile("" : "+v"(coff[r]));
#pragma unroll
        for (int mi = 0; mi < 4; mi++)
#pragma unroll
            for (int ni = 0; ni < 4; ni++)
#pragma unroll
                for (int r = 0; r < 4; r++) Ct[(int64_t)mi * 16 * ldc + ni * 16 + coff[r]] = -acc[mi][ni][r];
    }
}

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

// ---------------------------------------------------------------------------------------------
// A: REGISTER-RESIDENT diagonal block factorisation (nbk x nbk, nbk a multiple of 16 up to 256) by ONE workgroup.  The lower triangle of the 256x256 block is 136
// tiles of 16x16; they live for the whole kernel as FP64-MFMA accumulators in the registers of seven "update" waves
// (<= 20 tiles = 160 VGPRs per wave), wave 0 is the "chain" wave.  Per 16-column strip k, two workgroup barriers:
//   phase A (waves 1-7)  TRSM_k: X_R = T(R,k) Linv_k^T for the tiles below the diagonal tile: ONE 4-MFMA chain per tile
//                        with the accumulator registers themselves as the B operand -> X to the LDS panel (18-double
//                        rows) and to global.  The wave that owns (k+1,k) also owns (k+1,k+1): it updates that tile with
//                        its own X_{k+1} at once and hands it to the chain wave (LDS), so that the next diagonal tile
//                        never waits for the trailing update.
//   phase B              wave 0 factors + inverts tile (k+1,k+1) (the chain) WHILE waves 1-7 apply  -= X_R X_C^T  to all
//                        other tiles {C > k} (look-ahead inside the kernel; operands from the LDS panel).
// The chain: one matrix row per lane of a 16-lane DPP row (replicated in the four rows of the wave); right-looking
// Cholesky whose column broadcasts are DPP operands (v_fmac_f64_dpp row_newbcast: no LDS, no v_readlane), fused with
// the inverse of the factor (forward elimination of the column-scaled unit triangle, four columns per DPP row).  One
// wave issues one FP64 VALU instruction per ~8.5 cycles whether dependent or not (profiles/r02_run20_dp_latency.txt),
// so the chain is written for INSTRUCTION COUNT: v_rsq_f64 (2^-24) + one Halley step instead of two Newton steps, the
// diagonal entry from the same multiply as the column (p * rsqrt(p)), no selects on the factor (entries right of the
// diagonal carry garbage that only ever meets other such entries),
// the pivots checked once per strip (a failed one leaves NaNs), 1/L_ii through an LDS side buffer: ~21 VALU
// instructions per column.
// Accumulator convention: acc holds the tile transposed and with the columns permuted, so that no operand ever needs a
// transposition (and the one subtraction is the MFMA's own neg:[1,0,0]):
//     lane (frow, fk), register q   <->   T[row = frow][col = 4 fk + q]         (MFMA D[i][j], i = fk + 4 q, j = frow)
// which is at once the D layout of v_mfma_f64_16x16x4 and its A/B operand layout (k slot fk of MFMA step q <-> k =
// 4 fk + q; the same k permutation is used for the other operand, read as two ds_read_b128 of 4 consecutive doubles).
// With pi(i) = 4 (i % 4) + i / 4 (the column of D row i):   D[i][j] += sum_k A'[i][k] B'[k][j]  gives
//     update:  A' = -X_C[pi(i)][k], B' = X_R[j][k]                     -> T(R,C) -= X_R X_C^T
//     TRSM:    A' = Linv[pi(i)][k], B' = acc (= T(R,k)[j][k])          -> D = X_R in the same lane layout.
// The 64x64 tile inverses (`dinv`, used by the panel solve and the triangular solves) are built from the stored factor
// by k_diag_tile_inverses at the end of the factorisation; what the panel solve below the block needs instead -- the 16x16
// inverses and refinement flags -- is left in those tile-inverse slots, see store_diag.
// ---------------------------------------------------------------------------------------------
#ifndef RB_REFINE_LOG2
#define RB_REFINE_LOG2 5  // refine the strip's TRSM when log2(max |Linv| max L_ii) reaches this (-100: always, 100: never)
#endif
constexpr int RB_LD = 18;                 // LDS row stride (doubles): b128 reads of 16 rows hit 16 distinct 4-bank groups
constexpr int RB_P_DOUBLES = 256 * RB_LD;
constexpr int RB_RD_DOUBLES = 16 + 64 + 16;  // 1 / L_jj of the current strip + a dump area (see rb_chain_run)
constexpr int RB_LDS_BYTES = (RB_P_DOUBLES + 3 * 16 * RB_LD + RB_RD_DOUBLES) * 8 + 16;
// Tile -> (update wave, slot) for NU update waves, and per strip k the slots each phase touches (bit masks examined
// by the scalar unit: two instructions per slot instead of ten).  The pairs {(r, r-1), (r, r)} are dealt first (same
// wave, consecutive slots), then the other tiles in column-major order, each to the least loaded wave: every trailing
// set {C > k} is balanced to +-1 tile (offsets / strides of the pair dealing found by exhaustive search; the 16-wave
// kernel has its own table, below).
constexpr int rb_slots(int nu) { return nu == 7 ? 20 : (nu == 11 ? 13 : 9); }
template <int NU>
struct RbTab {
    int rc[NU][rb_slots(NU)];             // R * 16 + C, 0x100 = none (32-bit: fetched by scalar loads, not through vmcnt)
    unsigned int trsm[NU][16];            // phase A of strip k: slots with C == k, R > k
    unsigned int pair[NU][16];            // ... of those, the slot of (k+1, k) (slot + 1 is tile (k+1, k+1))
    unsigned int upd[NU][16];             // phase B of strip k: slots with C > k except (k+1, k+1)
};
// The 16-wave kernel's dealing (tools/rb_deal.py): update waves 3, 7, 11 (= waves 4, 8, 12, the ones that share the chain
// wave's SIMD: wave -> SIMD was measured as [0 2 1 3][w % 4] or [1 3 0 2][w % 4]) own tiles of columns 0..4 only, so that
// from strip 4 on -- where the strip time is the chain's -- no FP64 MFMA holds the DP pipe the chain's VALU instructions
// need (an MFMA blocks it for 64 cycles: the chain ran 2 - 2.8x slower beside the trailing update, profiles/r02_run19_*).
constexpr unsigned char rb_rc15[15][9] = {
    {0x32, 0x33, 0xfe, 0xff, 0x53, 0x94, 0xe5, 0xd7, 0xe9}, {0x21, 0x22, 0xed, 0xee, 0x72, 0xa4, 0xf5, 0xe7, 0xf9},
    {0x10, 0x11, 0xdc, 0xdd, 0x82, 0xb4, 0xa6, 0xf7, 0xca}, {0xf0, 0xc0, 0xf1, 0xc1, 0xf2, 0xc2, 0xf3, 0xc3, 0xf4},
    {0xcb, 0xcc, 0x20, 0x41, 0x92, 0xc4, 0xb6, 0xa8, 0xda}, {0xba, 0xbb, 0x30, 0x51, 0x63, 0x85, 0xc6, 0xb8, 0xdb},
    {0xa9, 0xaa, 0x40, 0x61, 0x73, 0x95, 0xd6, 0xc8, 0xea}, {0xe0, 0xb0, 0xe1, 0xb1, 0xe2, 0xb2, 0xe3, 0xb3, 0xe4},
    {0x98, 0x99, 0x50, 0x71, 0x83, 0xa5, 0xe6, 0xb9, 0xfa}, {0x87, 0x88, 0x60, 0x81, 0x93, 0xb5, 0xf6, 0xd8, 0xeb},
    {0x76, 0x77, 0x70, 0x91, 0x64, 0xc5, 0x97, 0xe8, 0xfb}, {0xd0, 0xa0, 0xd1, 0xa1, 0xd2, 0xa2, 0xd3, 0xa3, 0xd4},
    {0x65, 0x66, 0x80, 0x42, 0x74, 0x86, 0xa7, 0xf8, 0xec}, {0x54, 0x55, 0x90, 0x52, 0x75, 0x96, 0xb7, 0xc9, 0xfc},
    {0x43, 0x44, 0x31, 0x62, 0x84, 0xd5, 0xc7, 0xd9, 0xfd}};
template <int NU>
constexpr RbTab<NU> rb_make_tab() {
    constexpr int NS = rb_slots(NU);
    constexpr int off = NU == 7 ? 2 : 1, step = NU == 7 ? 6 : 3;
    RbTab<NU> t{};
    int cnt[NU] = {};
    for (int w = 0; w < NU; w++)
        for (int s = 0; s < NS; s++) t.rc[w][s] = 0x100;
    if (NU == 15) {
        for (int w = 0; w < NU; w++)
            for (int s = 0; s < NS; s++) t.rc[w][s] = rb_rc15[w % 15][s % 9];
    } else {
        for (int r = 1; r < 16; r++) {
            const int w = (off + step * (r - 1)) % NU;
            t.rc[w][cnt[w]++] = r * 16 + r - 1;
            t.rc[w][cnt[w]++] = r * 16 + r;
        }
        for (int c = 0; c < 16; c++)
            for (int r = c + 2; r < 16; r++) {
                int w = 0;
                for (int v = 1; v < NU; v++)
                    if (cnt[v] < cnt[w]) w = v;
                t.rc[w][cnt[w]++] = r * 16 + c;
            }
    }
    // Slots in the order of their first use (column of the tile; a pair counts as its tile (r, r-1) and stays together):
    // the block is loaded slot by slot, and what strip 0 needs should not queue behind what strip 9 needs.
    for (int w = 0; w < NU; w++) {
        int first[NS] = {}, len[NS] = {}, units = 0;
        for (int s0 = 0; s0 < NS && t.rc[w][s0] < 0x100;) {
            const int R = t.rc[w][s0] >> 4, C = t.rc[w][s0] & 15;
            const bool pair = C + 1 == R && s0 + 1 < NS && t.rc[w][s0 + 1] == R * 16 + R;
            first[units] = s0;
            len[units++] = pair ? 2 : 1;
            s0 += pair ? 2 : 1;
        }
        int sorted[NS] = {};
        bool used[NS] = {};
        int out = 0;
        for (int n = 0; n < units; n++) {  // selection sort, stable
            int best = -1;
            for (int v = 0; v < units; v++)
                if (!used[v] && (best < 0 || (t.rc[w][first[v]] & 15) < (t.rc[w][first[best]] & 15))) best = v;
            used[best] = true;
            for (int e = 0; e < len[best]; e++) sorted[out++] = t.rc[w][first[best] + e];
        }
        for (int s0 = 0; s0 < out; s0++) t.rc[w][s0] = sorted[s0];
    }
    for (int w = 0; w < NU; w++)
        for (int k = 0; k < 16; k++) {
            t.trsm[w][k] = t.pair[w][k] = t.upd[w][k] = 0;
            for (int s = 0; s < NS; s++) {
                if (t.rc[w][s] >= 0x100) continue;
                const int R = t.rc[w][s] >> 4, C = t.rc[w][s] & 15;
                if (C == k && R > k) t.trsm[w][k] |= 1u << s;
                if (C == k && R == k + 1) t.pair[w][k] |= 1u << s;
                if (C > k && !(C == k + 1 && R == k + 1)) t.upd[w][k] |= 1u << s;
            }
        }
    return t;
}
// every tile of the lower triangle except (0,0) exactly once; (r, r) right behind (r, r-1); masks consistent
template <int NU>
constexpr bool rb_tab_ok() {
    constexpr int NS = rb_slots(NU);
    const RbTab<NU> t = rb_make_tab<NU>();
    int seen[16][16] = {};
    for (int w = 0; w < NU; w++)
        for (int s = 0; s < NS; s++) {
            if (t.rc[w][s] >= 0x100) continue;
            const int R = t.rc[w][s] >> 4, C = t.rc[w][s] & 15;
            if (C > R || (R == 0 && C == 0)) return false;
            seen[R][C]++;
            if (R == C && (s == 0 || t.rc[w][s - 1] != R * 16 + R - 1)) return false;
            if (s > 0 && R != C) {  // sorted by column (a pair counts as its first tile)
                const int pr = t.rc[w][s - 1] >> 4, pc = t.rc[w][s - 1] & 15;
                if ((pr == pc ? pc - 1 : pc) > C) return false;
            }
        }
    for (int r = 0; r < 16; r++)
        for (int c = 0; c <= r; c++)
            if (seen[r][c] != ((r == 0 && c == 0) ? 0 : 1)) return false;
    for (int k = 0; k < 15; k++) {
        int trsm = 0, pair = 0, upd = 0;
        for (int w = 0; w < NU; w++)
            for (int s = 0; s < NS; s++) {
                trsm += (t.trsm[w][k] >> s) & 1;
                pair += (t.pair[w][k] >> s) & 1;
                upd += (t.upd[w][k] >> s) & 1;
            }
        if (trsm != 15 - k || pair != 1 || upd != (15 - k) * (16 - k) / 2 - 1) return false;
    }
    return true;
}
static_assert(rb_tab_ok<7>() && rb_tab_ok<11>() && rb_tab_ok<15>(), "tile tables of the register-resident diagonal-block kernel");
template <int NU>
__constant__ const RbTab<NU> c_rb_tab = rb_make_tab<NU>();

// DPP row broadcasts of FP64 operands (gfx90a+: row_newbcast is the one DPP control the DP ALU accepts).  Inline asm
// is invisible to the compiler's hazard recogniser, so the two wait states a DPP read needs behind a VALU write of
// its source are part of the statement.
template <int L>
__device__ __forceinline__ double dpp_bcast(double v) {
    double r;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(L));
    return r;
}
// acc += (-src[lane L of this row]) * mul
template <int L>
__device__ __forceinline__ void dpp_fnmac(double &acc, double src, double mul) {
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc)
                 : "v"(src), "v"(mul), "n"(L));
}
// the same without the wait states: for a `src` that was written at least two instructions earlier (volatile asm
// statements keep their program order, so "behind two other DPP statements" is such a place)
template <int L>
__device__ __forceinline__ void dpp_fnmac_late(double &acc, double src, double mul) {
    asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc)
                 : "v"(src), "v"(mul), "n"(L));
}

// Chain state of wave 0 (its own code path in the kernel: wave 0 owns no tiles, the update waves no chain state)
struct RbChain {
    double a[16];    // matrix row `i` of the 16x16 tile, then of its factor (entries right of the diagonal: garbage)
    double z[4];     // row i of the inverse of the column-scaled factor, columns fk + 4 q
};

template <int J, int... Cs>
__device__ __forceinline__ void rb_chain_trailing(RbChain &ch, double l, std::integer_sequence<int, Cs...>) {
    (dpp_fnmac_late<J + 2 + Cs>(ch.a[J + 2 + Cs], l, l), ...);  // behind fnmac<J+1> and bcast<J+1>
}
template <int J, int... Qs>
__device__ __forceinline__ void rb_chain_inverse(RbChain &ch, double m, std::integer_sequence<int, Qs...>) {
    (dpp_fnmac_late<J>(ch.z[Qs], ch.z[Qs], m), ...);  // z was written a whole column earlier
}
// column J of the factorisation; `p` = the pivot, already broadcast.  Returns the broadcast pivot of column J + 1,
// which is formed first.
template <int J>
__device__ __forceinline__ double rb_chain_step(RbChain &ch, double p, int i, double *rd_lane) {
    // 1 / sqrt(p): hardware estimate (relative error <= 2^-24, measured) + one Halley step (cubic: <= 2^-70)
    const double y0 = __builtin_amdgcn_rsq(p);
    const double t = p * y0;
    const double e = __builtin_fma(-t, y0, 1.0);
    const double h = __builtin_fma(e, 0.375, 0.5);
    const double ye = y0 * e;
    const double y = __builtin_fma(ye, h, y0);
    // Column J: L_iJ = a_iJ / sqrt(p); in lane J this is p / sqrt(p) = L_JJ itself.  Lanes above the diagonal (i < J)
    // carry garbage (their a[C], C > i, received the updates of the columns <= i): it only ever meets other entries
    // right of the diagonal, is masked out of the inverse below and zeroed when the factor is stored.
    const double l = ch.a[J] * y;
    ch.a[J] = l;
    // (a pivot that is not positive and finite makes L_JJ and everything after it NaN: looked for once per strip, in
    // rb_chain_run -- the NaNs stay inside this workgroup, and every later kernel returns on entry)
    double pnext = 0.0;
    if constexpr (J < 15) {
        dpp_fnmac<J + 1>(ch.a[J + 1], l, l);
        pnext = dpp_bcast<J + 1>(ch.a[J + 1]);
    }
    if constexpr (J < 14) rb_chain_trailing<J>(ch, l, std::make_integer_sequence<int, 14 - J>{});
    rd_lane[J] = y;  // lane 0 -> rd[J] = 1 / L_JJ, every other lane into its own dump slot (no exec change, no conflict)
    // forward elimination step J of the inverse:  z_i -= (L_iJ / L_JJ) z_J  for the rows below
    const double m = (i > J) ? l * y : 0.0;
    rb_chain_inverse<J>(ch, m, std::make_integer_sequence<int, J / 4 + 1>{});
    return pnext;
}
template <int... Js>
__device__ __forceinline__ void rb_chain_all(RbChain &ch, int i, double *rd_lane, std::integer_sequence<int, Js...>) {
    double p = dpp_bcast<0>(ch.a[0]);
    ((p = rb_chain_step<Js>(ch, p, i, rd_lane)), ...);
}

// wave 0: factor + invert the diagonal tile of a strip (rows in ch.a); publishes Linv and the raw factor rows in LDS
__device__ __forceinline__ void rb_chain_run(RbChain &ch, double *NL, double *LR, double *rd, int lane, int frow, int fk,
                                             int *info, int gcol0, int n_valid, int *fail_flag, int *refine_flag) {
#pragma unroll
    for (int q = 0; q < 4; q++) ch.z[q] = (fk + 4 * q == frow) ? 1.0 : 0.0;
    double *rd_lane = (lane == 0) ? rd : rd + 16 + lane;
    rb_chain_all(ch, frow, rd_lane, std::make_integer_sequence<int, 16>{});
    const double r_own = rd[frow];
    // log2 of  max |Linv_ij| * max L_ii  -- a cheap stand-in for cond(L) of this tile (the spread of the diagonal alone
    // is not one: the factor of a noise-level Schur complement has a flat diagonal and cond ~ 1e6).  The high words of
    // positive doubles order like the doubles, so the maxima / minima are integer ones; the four DPP rows hold the same
    // rows, a rotate-and-combine over the 16 lanes of a row finishes them.  The update waves refine X = A Linv^T once
    // when this is >= RB_REFINE_LOG2, see the TRSM phase of the kernel.
    int hmax = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const double nl = r_own * ch.z[q];
        NL[frow * RB_LD + fk + 4 * q] = nl;
        hmax = max(hmax, __double2hiint(nl) & 0x7fffffff);
    }
    {
        int ymin = __double2hiint(r_own);
#define RB_ROR(v, n) __builtin_amdgcn_update_dpp(v, v, 0x120 + n, 0xf, 0xf, false)
        hmax = max(hmax, RB_ROR(hmax, 1)), ymin = min(ymin, RB_ROR(ymin, 1));
        hmax = max(hmax, RB_ROR(hmax, 2)), ymin = min(ymin, RB_ROR(ymin, 2));
        hmax = max(hmax, RB_ROR(hmax, 4)), ymin = min(ymin, RB_ROR(ymin, 4));
        hmax = max(hmax, RB_ROR(hmax, 8)), ymin = min(ymin, RB_ROR(ymin, 8));
#undef RB_ROR
        // this DPP row saw the columns fk + 4 q only: combine the four rows through the scalar unit
        int hall = hmax;
        hall = max(hall, __builtin_amdgcn_readlane(hmax, 16));
        hall = max(hall, __builtin_amdgcn_readlane(hmax, 32));
        hall = max(hall, __builtin_amdgcn_readlane(hmax, 48));
        if (lane == 0) *refine_flag = ((hall >> 20) - (ymin >> 20) >= RB_REFINE_LOG2) ? 1 : 0;
    }
    if (fk == 0) {
#pragma unroll
        for (int c = 0; c < 16; c += 2) *reinterpret_cast<d2_t *>(LR + frow * RB_LD + c) = d2_t{ch.a[c], ch.a[c + 1]};
    }
    // the diagonal of the factor, lane i its own L_ii (read back from the rows just written): all positive and finite,
    // or the first one that is not is the first pivot that failed
    const unsigned int ok = (unsigned int)__builtin_amdgcn_ballot_w64(__builtin_amdgcn_class(LR[frow * RB_LD + frow], 0x180)) & 0xffffu;
    if (ok != 0xffffu && lane == 0) {
        const int first_bad = __builtin_ctz(~ok);
        if (gcol0 + first_bad < n_valid) atomicCAS(info, 0, gcol0 + first_bad + 1);
        *fail_flag = 1;
    }
}

// one 16x16x16 product on the matrix cores: c +-= A' B' with both operands as 4 consecutive doubles per lane.  NEG = 1
// negates A' in the instruction (the FP64 MFMAs of gfx940+ read their BLGP field as neg:[a,b,c]).
#define RB_MFMA4(c, NEG, a01, a23, b0, b1, b2, b3)                               \
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a01[0], b0, c, 0, 0, NEG);           \
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a01[1], b1, c, 0, 0, NEG);           \
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a23[0], b2, c, 0, 0, NEG);           \
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a23[1], b3, c, 0, 0, NEG)

template <int NW>  // waves: 1 chain wave + NW - 1 update waves (16 in the product; the tables also cover 8 and 12)
__global__ __launch_bounds__(64 * NW, 1) void k_potf2_reg(double *__restrict__ D, int64_t ld, int nbk, double *__restrict__ lin,
                                                          int *__restrict__ info, int col0, int n_valid, int64_t bsD,
                                                          int64_t bsL, int bsI) {
    constexpr int NU = NW - 1, RB_NS = rb_slots(NU);
    extern __shared__ __attribute__((aligned(16))) double sm[];
    {   // lock-step batch: one workgroup per matrix
        const int64_t z = blockIdx.z;
        D += z * bsD;
        lin += z * bsL;
        info += z * bsI;
    }
    // examined once the loads of the block are on their way (below).  Only diagonal-block kernels SET the flag and the
    // chain runs them one after the other, so every wave of this workgroup reads the same value.
    const int failed_before = *info;
    __builtin_amdgcn_s_setprio(2);    // above the trailing-update workgroups this kernel may share its CU with
    double *P = sm;                   // X panel of the current strip, row = row of the block
    double *Dg = sm + RB_P_DOUBLES;   // hand-off of the next diagonal tile (update wave -> chain wave)
    double *NL = Dg + 16 * RB_LD;     // Linv of the current strip (chain wave -> update waves)
    double *LR = NL + 16 * RB_LD;     // raw rows of the strip's 16x16 factor (chain wave -> the update wave that stores it)
    double *rd = LR + 16 * RB_LD;     // 1 / L_jj of the strip being factored
    int *flag = reinterpret_cast<int *>(rd + RB_RD_DOUBLES);
    const int tid = threadIdx.x, lane = tid & 63, frow = lane & 15, fk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb16 = nbk >> 4;
    const int prow = ((frow & 3) << 2) | (frow >> 2);  // pi(frow)
    if (tid == 0) *flag = 0;
    if (wave == 0) {
        // ================= chain wave: no tiles, no MFMA; two barriers per strip like the update waves
        __builtin_amdgcn_s_setprio(3);
        RbChain ch;
        const double *src = D + (int64_t)frow * ld;
#pragma unroll
        for (int c = 0; c < 16; c += 2) {
            const d2_t v = *reinterpret_cast<const d2_t *>(src + c);
            ch.a[c] = v[0];
            ch.a[c + 1] = v[1];
        }
        if (failed_before != 0) return;  // a previous block of this factorisation already failed: early exit
        rb_chain_run(ch, NL, LR, rd, lane, frow, fk, info, col0, n_valid, flag, flag + 1);
        __syncthreads();
        if (*flag) return;
        for (int k = 0; k + 1 < nb16; k++) {
            __syncthreads();  // phase A of strip k done: tile (k+1, k+1) is in Dg
#pragma unroll
            for (int c = 0; c < 16; c += 2) {
                const d2_t v = *reinterpret_cast<const d2_t *>(Dg + frow * RB_LD + c);
                ch.a[c] = v[0];
                ch.a[c + 1] = v[1];
            }
            rb_chain_run(ch, NL, LR, rd, lane, frow, fk, info, col0 + (k + 1) * 16, n_valid, flag, flag + 1);
            __syncthreads();
            if (*flag) return;
        }
        return;
    }
    // ===================== update waves 1..NU
    const RbTab<NU> &tab = c_rb_tab<NU>;
    double acc[RB_NS * 4];  // slot s, register q: acc[4 s + q]
    int rc[RB_NS];          // (wave-uniform) tile of slot s: R * 16 + C
    unsigned int valid = 0;  // slots whose tile exists in an nbk x nbk block
#pragma unroll
    for (int s = 0; s < RB_NS; s++) {
        int t = __builtin_amdgcn_readfirstlane(tab.rc[wave - 1][s]);
        asm volatile("" : "+s"(t));  // keep it in its SGPR: re-loading it from the table costs a memory round trip per use
        rc[s] = t;
        // The loads are unconditional (a slot without a tile re-reads tile (0,0) and never uses it) and the loaded
        // registers are not touched before the tile's first MFMA: straight-line loads let the compiler wait for exactly
        // the ones a phase needs (s_waitcnt vmcnt(n) counts in issue order = order of first use), and nothing is waited
        // for before the first barrier -- 31k cycles, a fifth of the kernel, when the tiles were negated on arrival.
        const bool have = t < 0x100 && (t >> 4) < nb16;
        if (have) valid |= 1u << s;
        const int tt = have ? t : 0;
        const double *src = D + (int64_t)((tt >> 4) * 16 + frow) * ld + (tt & 15) * 16 + 4 * fk;
        const d2_t v0 = *reinterpret_cast<const d2_t *>(src), v1 = *reinterpret_cast<const d2_t *>(src + 2);
        acc[4 * s] = v0[0];
        acc[4 * s + 1] = v0[1];
        acc[4 * s + 2] = v1[0];
        acc[4 * s + 3] = v1[1];
    }
    if (failed_before != 0) return;  // a previous block of this factorisation already failed: early exit
    // the strictly upper 16x16 tiles of the 64x64 diagonal tiles are part of the factor's contract: zeros
    for (int zt = wave - 1; zt < 24; zt += NU) {
        const int t = zt / 6, e = zt % 6;
        const int r = (e < 3) ? 0 : (e < 5 ? 1 : 2), c = (e < 3) ? e + 1 : (e < 5 ? e - 1 : 3);
        if (4 * t + c < nb16) {
            double *dst = D + (int64_t)((4 * t + r) * 16 + frow) * ld + (4 * t + c) * 16 + 4 * fk;
            *reinterpret_cast<d2_t *>(dst) = d2_t{0.0, 0.0};
            *reinterpret_cast<d2_t *>(dst + 2) = d2_t{0.0, 0.0};
        }
    }
    // the 16x16 factor of strip k (raw rows in LR): zero right of the diagonal, to global
    auto store_diag = [&](int k) {
        const double *lr = LR + frow * RB_LD + 4 * fk;
        const d2_t v0 = *reinterpret_cast<const d2_t *>(lr), v1 = *reinterpret_cast<const d2_t *>(lr + 2);
        double *dst = D + (int64_t)(k * 16 + frow) * ld + k * 16 + 4 * fk;
        *reinterpret_cast<d2_t *>(dst) = d2_t{(4 * fk <= frow) ? v0[0] : 0.0, (4 * fk + 1 <= frow) ? v0[1] : 0.0};
        *reinterpret_cast<d2_t *>(dst + 2) = d2_t{(4 * fk + 2 <= frow) ? v1[0] : 0.0, (4 * fk + 3 <= frow) ? v1[1] : 0.0};
        // ... and its inverse + refinement flag, for the solve of the rows below the block (k_panel_trsm16): they travel in
        // the slot of the 64x64 tile inverse this strip belongs to (tile j = k % 4 at doubles [256 j, 256 j + 256), flags at
        // 1024 + j), which k_diag_tile_inverses overwrites with the 64x64 inverses once the factorisation is complete
        const double *nl = NL + frow * RB_LD + 4 * fk;
        double *dl = lin + (int64_t)(k >> 2) * 4096 + (k & 3) * 256 + frow * 16 + 4 * fk;
        *reinterpret_cast<d2_t *>(dl) = *reinterpret_cast<const d2_t *>(nl);
        *reinterpret_cast<d2_t *>(dl + 2) = *reinterpret_cast<const d2_t *>(nl + 2);
        if (lane == 0) lin[(int64_t)(k >> 2) * 4096 + 1024 + (k & 3)] = flag[1] ? 1.0 : 0.0;
    };
    unsigned int m_trsm = tab.trsm[wave - 1][0] & valid, m_pair = tab.pair[wave - 1][0] & valid, m_upd = tab.upd[wave - 1][0] & valid;
    __syncthreads();
    if (*flag) return;
    // One strip.  It is a lambda so that strip 0 can be run OUTSIDE the loop: the compiler drains every outstanding load
    // before it enters a loop that uses them (s_waitcnt vmcnt(0) at the preheader), which made the TRSM of strip 0 --
    // it needs one or two of a wave's nine tiles -- wait for the whole block.
    auto strip = [&](int k) -> bool {
        // the slot masks of the next strip: scalar loads, a strip ahead of their use
        const unsigned int n_trsm = tab.trsm[wave - 1][k + 1] & valid, n_pair = tab.pair[wave - 1][k + 1] & valid,
                           n_upd = tab.upd[wave - 1][k + 1] & valid;
        // Per-slot addresses are recomputed per strip from these three: left to itself the compiler hoists 20 slots' worth
        // of loop-invariant row offsets out of the strip loop and spills them (measured: 52 VGPR spills).
        int fr = frow, pr = prow, f4 = 4 * fk;
        asm volatile("" : "+v"(fr), "+v"(pr), "+v"(f4));
        // ---- phase A: TRSM_k (the tiles (R, k), R > k, become X_R); tile (k+1, k+1) is updated by its owner at once
        if (wave == 1 + k % NU) store_diag(k);
        if (m_trsm) {
            const d2_t n01 = *reinterpret_cast<const d2_t *>(NL + pr * RB_LD + f4);
            const d2_t n23 = *reinterpret_cast<const d2_t *>(NL + pr * RB_LD + f4 + 2);
            // X = A Linv^T with an explicitly inverted triangle leaves a residual A - X L^T of order eps cond(L) |A| -- not
            // the backward-stable eps |A| of a substitution -- and that residual lands in the Schur complement (a
            // numerically rank-deficient correlation matrix, kept positive definite by its nugget alone, then loses a
            // pivot).  One step of iterative refinement, X += (A - X L^T) Linv^T, restores eps |A| as long as
            // eps cond(L)^2 < 1; it is skipped for well-conditioned diagonal tiles (the chain wave's estimate, flag[1]).
            const bool refine = flag[1] != 0;
            d2_t l01 = d2_t{0.0, 0.0}, l23 = d2_t{0.0, 0.0};
            if (refine) {  // A' operand of X L^T: rows pi(frow) of the raw factor, zero right of the diagonal
                const d2_t r01 = *reinterpret_cast<const d2_t *>(LR + pr * RB_LD + f4);
                const d2_t r23 = *reinterpret_cast<const d2_t *>(LR + pr * RB_LD + f4 + 2);
                l01 = d2_t{(f4 <= pr) ? r01[0] : 0.0, (f4 + 1 <= pr) ? r01[1] : 0.0};
                l23 = d2_t{(f4 + 2 <= pr) ? r23[0] : 0.0, (f4 + 3 <= pr) ? r23[1] : 0.0};
            }
#pragma unroll
            for (int s = 0; s < RB_NS; s++) {
                if (m_trsm & (1u << s)) {
                    const int R = rc[s] >> 4;
                    double4_t x = double4_t{0.0, 0.0, 0.0, 0.0};
                    RB_MFMA4(x, 0, n01, n23, acc[4 * s], acc[4 * s + 1], acc[4 * s + 2], acc[4 * s + 3]);
                    if (refine) {
                        double4_t r = double4_t{acc[4 * s], acc[4 * s + 1], acc[4 * s + 2], acc[4 * s + 3]};  // A
                        RB_MFMA4(r, 1, l01, l23, x[0], x[1], x[2], x[3]);                                      // A - X L^T
                        RB_MFMA4(x, 0, n01, n23, r[0], r[1], r[2], r[3]);                                      // X + (A - X L^T) Linv^T
                    }
                    double *px = P + (R * 16 + fr) * RB_LD + f4;
                    *reinterpret_cast<d2_t *>(px) = d2_t{x[0], x[1]};
                    *reinterpret_cast<d2_t *>(px + 2) = d2_t{x[2], x[3]};
                    double *dst = D + (int64_t)(R * 16 + fr) * ld + k * 16 + f4;
                    *reinterpret_cast<d2_t *>(dst) = d2_t{x[0], x[1]};
                    *reinterpret_cast<d2_t *>(dst + 2) = d2_t{x[2], x[3]};
                    if (s + 1 < RB_NS && (m_pair & (1u << s))) {  // slot s + 1 is tile (k+1, k+1): T -= X X^T, A' = own LDS rows
                        const double *pa = P + (R * 16 + pr) * RB_LD + f4;
                        const d2_t a01 = *reinterpret_cast<const d2_t *>(pa), a23 = *reinterpret_cast<const d2_t *>(pa + 2);
                        double4_t c4 = double4_t{acc[4 * s + 4], acc[4 * s + 5], acc[4 * s + 6], acc[4 * s + 7]};
                        RB_MFMA4(c4, 1, a01, a23, x[0], x[1], x[2], x[3]);
                        double *dg = Dg + fr * RB_LD + f4;
                        *reinterpret_cast<d2_t *>(dg) = d2_t{c4[0], c4[1]};
                        *reinterpret_cast<d2_t *>(dg + 2) = d2_t{c4[2], c4[3]};
                    }
                }
            }
        }
        __syncthreads();
        // ---- phase B: all other tiles right of column k, while wave 0 factors tile (k+1, k+1)
#pragma unroll
        for (int s = 0; s < RB_NS; s++) {
            if (m_upd & (1u << s)) {
                const int R = rc[s] >> 4, C = rc[s] & 15;
                const double *pa = P + (C * 16 + pr) * RB_LD + f4;
                const double *pb = P + (R * 16 + fr) * RB_LD + f4;
                const d2_t a01 = *reinterpret_cast<const d2_t *>(pa), a23 = *reinterpret_cast<const d2_t *>(pa + 2);
                const d2_t b01 = *reinterpret_cast<const d2_t *>(pb), b23 = *reinterpret_cast<const d2_t *>(pb + 2);
                double4_t c4 = double4_t{acc[4 * s], acc[4 * s + 1], acc[4 * s + 2], acc[4 * s + 3]};
                RB_MFMA4(c4, 1, a01, a23, b01[0], b01[1], b23[0], b23[1]);
                acc[4 * s] = c4[0];
                acc[4 * s + 1] = c4[1];
                acc[4 * s + 2] = c4[2];
                acc[4 * s + 3] = c4[3];
            }
        }
        __syncthreads();
        if (*flag) return false;
        m_trsm = n_trsm;
        m_pair = n_pair;
        m_upd = n_upd;
            return true;
    };
    if (nb16 > 1 && !strip(0)) return;
    for (int k = 1; k + 1 < nb16; k++)
        if (!strip(k)) return;
    if (wave == 1 + (nb16 - 1) % NU) store_diag(nb16 - 1);
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
__global__ __launch_bounds__(512, 2) void k_block_inv256(const double *__restrict__ M, int64_t ld, int n_pad,
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

// x = W v for one 256-block (W upper triangular).  grid = 8 workgroups x 32 rows: each wave owns 8 rows and
// streams them with all loads of 4 rows in flight (one workgroup alone is HBM-latency bound: 24 us).
// x is written to xout (a separate buffer: other workgroups still read v).
__global__ __launch_bounds__(256) void k_trsv_w(const double *__restrict__ W, int nbk, const double *__restrict__ v,
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

// v[j] -= sum_i Mrow[i*ld + j] * x[i]  for j < ncols ; grid = ncols/64
__global__ __launch_bounds__(256) void k_gemv_t_update(const double *__restrict__ Mrow, int64_t ld, int nbk,
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

// ---------------------------------------------------------------------------------------------
// MFMA layout probe: C(16x16) = A(16x16) B(16x16) with asymmetric operands.
// ---------------------------------------------------------------------------------------------
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
// Tuning knobs that survive (each read once; everything else that was ever switchable has its A/B in profiles/ and
// is gone -- DESIGN.md section 4 lists both):
static int g_potrf_group = 0;         // EGX_POTRF_GROUP: panels per trailing update, 1..8 (0 = by size: 4 from n_pad 14336, else 2)
static int g_stream_min_tiles = 128;  // EGX_STREAM_MIN: launches with at least this many 128x256 tiles PER MATRIX go to k_gemm_stream
static int g_stream_tpw = 1;          // EGX_STREAM_TPW: tiles a workgroup of k_gemm_stream walks (1: CUs turn over, the chain squeezes in)
static int g_gemm_small_max = 1024;   // EGX_GEMM_SMALL: below this many 128x128 tiles the 64x64-tile kernel is used
static int g_look_min_cols = 3072;    // EGX_LOOK_MIN: look-ahead while at least this many columns trail the next group
static int g_trsm_group = 0;          // EGX_TRSM_GROUP: panels per update in the solves after the factorisation (0 = 4)
static int g_lur_side = 1;            // EGX_LUR_SIDE=0: the look-ahead columns' update stays in front of RU also in lock-step batches
static int g_trsm_left = 1;           // EGX_TRSM_LEFT=0: the solves after the factorisation update right-looking (one pass over all later columns per group)
static int g_tail_merge = 1;          // EGX_TAIL_MERGE=0: without look-ahead the next group's columns and the rest are updated by two launches
static int g_lur_side_min = 6144;     // EGX_LUR_SIDE_MIN: ... only while at least this many columns trail the look-ahead group
static int g_w_left = 1;              // EGX_W_LEFT: left-looking update of the C^-T rider 0 never, 1 per handle (w_left_for), 2 always
static int g_potrf_left = 1;          // EGX_POTRF_LEFT: left-looking group updates (launch_potrf) 0 never, 1 for handles with n_pad >= 14336
                                      // and a lock-step width >= 8, 2 always

int chol_init() {
    static std::once_flag once;
    static int rc_once = EGX_SUCCESS;
    std::call_once(once, [] {
        if (const char *e = std::getenv("EGX_POTRF_GROUP")) {
            const int g = std::atoi(e);
            if (g >= 1 && g <= 8) g_potrf_group = g;
        }
        if (const char *e = std::getenv("EGX_GEMM_SMALL")) g_gemm_small_max = std::atoi(e);
        if (const char *e = std::getenv("EGX_STREAM_TPW")) g_stream_tpw = std::atoi(e) > 0 ? std::atoi(e) : 1;
        if (const char *e = std::getenv("EGX_STREAM_MIN")) g_stream_min_tiles = std::atoi(e) > 0 ? std::atoi(e) : 1;
        if (const char *e = std::getenv("EGX_LOOK_MIN")) g_look_min_cols = std::atoi(e);
        if (const char *e = std::getenv("EGX_TRSM_GROUP")) g_trsm_group = std::atoi(e);
        if (const char *e = std::getenv("EGX_LUR_SIDE")) g_lur_side = std::atoi(e);
        if (const char *e = std::getenv("EGX_POTRF_LEFT")) g_potrf_left = std::atoi(e);
        if (const char *e = std::getenv("EGX_TAIL_MERGE")) g_tail_merge = std::atoi(e);
        if (const char *e = std::getenv("EGX_TRSM_LEFT")) g_trsm_left = std::atoi(e);
        if (const char *e = std::getenv("EGX_W_LEFT")) g_w_left = std::atoi(e);
        if (const char *e = std::getenv("EGX_LUR_SIDE_MIN")) g_lur_side_min = std::atoi(e);
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

// does a handle of this shape factor left-looking?  (per handle, never per launch: see launch_potrf)
int potrf_left_for(int n_pad, int lockstep) {
    (void)chol_init();
    // measured (profiles/r04_run4_*, r04_run5_ab_small_*): n = 16384 with groups of eight +2.3 % on the sweep; n = 8192 (groups
    // of two panels, K <= 7k, <= 128 tiles per matrix and launch) -14 %; a lone matrix 2.4x slower -- hence the narrow rule
    return g_potrf_left >= 2 || (g_potrf_left == 1 && n_pad >= 14336 && lockstep >= 8);
}

// does the C^-T rider (the theta-gradient's W, PotrfInverse) of a handle of this shape update left-looking?  Per handle like
// potrf_left_for.  Measured (profiles/r04_run12_w_left_ab.txt; likelihood + gradient, ms per candidate, right- -> left-looking):
//   n = 16384   lock-step 8: 74.4 -> 71.9    lock-step 4: 75.8 -> 73.4    one workspace: 78.1 -> 78.4
//   n = 8192    lock-step 12: 10.4 -> 9.9    ONE candidate: 12.6 -> 21.7   (n = 4096: 1.79 -> 1.69, 3.4 -> 5.5)
// a lone matrix' launch has few tiles with K ranges from 128 to all earlier columns, and the next group's solves wait for
// the longest: from n_pad 14336 on (where one candidate on such a handle neither gains nor loses) and lock-step widths >= 4.
// Both forms add the same products in the same order wherever every group is 1024 columns wide: the same bits at n = 16384
// (checked by the A/B); a narrower last group's right-looking update is another kernel's (3e-12 at n = 14400).
int w_left_for(int n_pad, int lockstep) {
    (void)chol_init();
    return n_pad % 256 == 0 && (g_w_left >= 2 || (g_w_left == 1 && n_pad >= 14336 && lockstep >= 4));
}

// run-time access to the knobs above (egx_set_tuning): A/B measurements inside ONE process (a gpurun call is minutes, a
// launch sequence milliseconds) and the serialised profiling mode of bench.py's roofline leg.  Returns the previous
// value, or INT_MIN for an unknown name.  Not to be called while evaluations are in flight.
int set_knob(const char *name, int value) {
    (void)chol_init();  // the environment is read first, once; a later call here wins
    struct { const char *n; int *v; } tab[] = {{"potrf_group", &g_potrf_group}, {"stream_min", &g_stream_min_tiles},
                                              {"stream_tpw", &g_stream_tpw},   {"gemm_small", &g_gemm_small_max},
                                              {"look_min", &g_look_min_cols},  {"trsm_group", &g_trsm_group},
                                              {"lur_side", &g_lur_side},
                                              {"potrf_left", &g_potrf_left},   {"tail_merge", &g_tail_merge},
                                              {"trsm_left", &g_trsm_left},     {"w_left", &g_w_left},
                                              {"lur_side_min", &g_lur_side_min}};
    for (auto &e : tab)
        if (std::string(name) == e.n) {
            const int old = *e.v;
            *e.v = value;
            return old;
        }
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
    //  columns, stay on the stream kernel down to a handful of tiles)
    const int64_t min_tiles = (K >= 2048) ? 8 : g_stream_min_tiles;
    if (K >= 2 * KC && wide_tiles >= min_tiles) {
        // the launches that fill the chip on their own are the ones the roofline trace follows
        if (used_big_tile) *used_big_tile = wide_tiles >= 512;
        const int nbx = M / 128, nby = N / 256;  // LOWER: column c holds nbx - 2 c tiles (M >= N in the factorisation)
        const int nt = (int)wide_tiles;
        const dim3 grid((unsigned)((nt + g_stream_tpw - 1) / g_stream_tpw), 1, nz);
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
    const int GW = (g_potrf_group ? g_potrf_group : (n_pad >= 14336 ? 4 : 2)) * kNB;
    auto gwidth = [&](int g0) { return (n_pad - g0 < GW) ? (n_pad - g0) : GW; };
    // panels of one group on stream `st`; `side` != nullptr splits every in-group update into the next diagonal block
    // (on st) and the rest (on side); `first_wait` is waited for before the FIRST panel solve (the rest of the update
    // that brought this group's columns up to date)
    auto inner_factor = [&](hipStream_t st, int g0, int gw, hipStream_t side, hipEvent_t first_wait) -> int {
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
            int rc2 = launch_gemm_nt_sub(sw, inv->W + gend, inv->ldw, inv->W + g0, inv->ldw, M + (int64_t)gend * ld + g0, ld, gend,
                                         ncols, gw, 0, 0, nullptr, info, &gbw);
            if (rc2) return rc2;
        }
        return EGX_SUCCESS;
    };
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
            hipStream_t slu = (g_lur_side && n_pad >= 14336 && n_pad - r1 - gw1 >= g_lur_side_min) ? s3 : s;
            if (slu != s) EGX_HIP_CHECK(hipStreamWaitEvent(slu, lk->ev_lu, 0));
            rc = update(slu, r1 + nb1, r1, m_tot - r1 - nb1, gw1, g0, gw, 0, nullptr);
            if (rc) return rc;
            EGX_HIP_CHECK(hipEventRecord(lk->ev_lur, slu));
            rc = inner_factor(s2, r1, gw1, s3, lk->ev_lur);
            if (rc) return rc;
            EGX_HIP_CHECK(hipEventRecord(lk->ev_panel, s2));
        } else if (!look && g_tail_merge) {
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
            // LU: the next group's columns only
            rc = update(s, r1, r1, m_tot - r1, gw1, g0, gw, 1, nullptr);
            if (rc) return rc;
            if (look) {
                EGX_HIP_CHECK(hipEventRecord(lk->ev_lu, s));
                EGX_HIP_CHECK(hipStreamWaitEvent(s2, lk->ev_lu, 0));
                rc = inner_factor(s2, r1, gw1, nullptr, nullptr);
                if (rc) return rc;
                EGX_HIP_CHECK(hipEventRecord(lk->ev_panel, s2));
            }
        }
        // RU: the rest of the trailing matrix
        const int r2 = r1 + gw1;
        if (r2 < n_pad && (look || !g_tail_merge)) {
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
    const int GW = (g_trsm_group > 0 ? g_trsm_group : 4) * kNB;
    // LEFT-looking over the groups (round 4, g_trsm_left; dense right-hand sides only): the columns of group J receive
    // the contributions of ALL earlier columns in one update with K = g0 before the group's own block substitution --
    // every tile of RT is read and written once by a long K loop instead of once per earlier group (K = 1024).  With
    // m >= 128 rows a launch always has m / 128 x 4 tiles: none of the underfill a lone factorisation's late updates have.
    const bool left = g_trsm_left != 0 && !tri_rows;
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

int trsm_group_cols() { return (g_trsm_group > 0 ? g_trsm_group : 4) * kNB; }

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
                  double *xout) {
    // v holds the right-hand side and is updated in place above the current block; the solution goes to xout
    const int nblocks = (n_pad + kNB - 1) / kNB;
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
