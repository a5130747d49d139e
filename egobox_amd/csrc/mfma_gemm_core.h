// The register-staged FP64 MFMA GEMM core (v_mfma_f64_16x16x4_f64, operands through LDS in 16-deep K chunks) shared by
// the update kernels of kernels_chol.hip and the update roles of the pipelined chain kernel (kernels_pipe.hip).
#pragma once
#include "potf2_blocks.h"  // double4_t, d2_t

namespace egx {

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
// SC1: agent-scope loads (they bypass the L1 of the compute unit): operands another workgroup of the SAME launch has just
// published (kernels_pipe.hip) need no acquire fence in front of them
template <int ROWS, int NT, bool SC1 = false>
__device__ __forceinline__ void tile_load_regs(const double *__restrict__ g, const unsigned (&off)[ROWS * 8 / NT],
                                               d2_t (&r)[ROWS * 8 / NT]) {
#pragma unroll
    for (int i = 0; i < ROWS * 8 / NT; i++) {
        if constexpr (SC1) r[i] = load_d2_sc1(g + off[i]);
        else r[i] = *reinterpret_cast<const d2_t *>(g + off[i]);
    }
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

template <int BM, int BN, int WM, int WN, int NTHREADS = 256, bool SC1 = false>
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
    tile_load_regs<BM, NTHREADS, SC1>(A, oa, ra);
    tile_load_regs<BN, NTHREADS, SC1>(B, ob, rb);
    tile_store_lds<BM, NTHREADS>(smem, ra, tid);
    tile_store_lds<BN, NTHREADS>(smem + S::A_TILE, rb, tid);
    __syncthreads();
    for (int c = 0; c < nchunks; c++) {
        double *As = smem + (c & 1) * S::STAGE;
        double *Bs = As + S::A_TILE;
        const bool more = (c + 1 < nchunks);
        if (more) {
            tile_load_regs<BM, NTHREADS, SC1>(A + (c + 1) * KC, oa, ra);
            tile_load_regs<BN, NTHREADS, SC1>(B + (c + 1) * KC, ob, rb);
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

}  // namespace egx
