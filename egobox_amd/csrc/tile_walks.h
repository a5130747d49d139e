// Tile walks of k_gemm_stream (kernels_chol.hip): workgroup / tile index -> (bx, by) of the 128 x 256 tiling.  Plain
// integer + double arithmetic, compiled for the device by hipcc and for the HOST by tests/c_host/tile_walks_test.cpp, which
// checks that every walk visits every tile of its region exactly once.
#pragma once
#include <math.h>
#ifdef __HIPCC__
#define EGX_HD __device__ __forceinline__
#else
#define EGX_HD static inline
#endif

// tile index -> (bx, by) of the 128x256 tiling; LOWER: column c holds the nbx - 2c tiles bx >= 2c
template <bool LOWER>
EGX_HD void stream_tile_coords(int t, int nbx, int nby, int &bx, int &by) {
    if (LOWER) {
        const double bq = nbx + 1.0;
        int c = (int)((bq - sqrt(bq * bq - 4.0 * t)) * 0.5);
        if (c < 0) c = 0;
        if (c >= nby) c = nby - 1;
        while (c > 0 && c * nbx - c * (c - 1) > t) c--;
        while (c + 1 < nby && (c + 1) * nbx - (c + 1) * c <= t) c++;
        by = c;
        bx = 2 * c + (t - (c * nbx - c * (c - 1)));
    } else {
        by = t / nbx;
        bx = t - by * nbx;
    }
}

// XCD-aware walk (round 4, EGX_STREAM_WALK / egx_set_tuning "stream_walk" = 1): workgroup w lands on XCD w % 8 (observed
// dispatch order; used for speed only, any walk yields the same bits), and the 32 workgroups q = 32 j .. 32 j + 31 of an
// XCD (one per CU) take the 8 x 4 tiles of ONE super-tile: they share 8 A blocks and 4 B blocks of the panel in the XCD's
// L2 -- 12 block streams for 32 tiles, where the column-major walk has 32 tiles of one column (33 streams: every A block
// is fetched by one CU only).  Super-tiles go round-robin to the XCDs; LOWER enumerates the super-tiles that touch the lower
// triangle (si >= sj; rows si >= nsy hold nsy each), tiles outside the matrix or above the diagonal exit at once.
template <bool LOWER>
EGX_HD bool stream_tile_coords_xcd(int t, int nbx, int nby, int &bx, int &by) {
    const int xcd = t & 7, q = t >> 3;
    const int ST = (q >> 5) * 8 + xcd, local = q & 31;
    const int nsx = (nbx + 7) >> 3, nsy = (nby + 3) >> 2;
    int si, sj;
    if (LOWER) {
        const int ntri = nsy * (nsy + 1) / 2;
        if (ST < ntri) {
            si = (int)((sqrt(8.0 * ST + 1.0) - 1.0) * 0.5);
            while (si > 0 && si * (si + 1) / 2 > ST) si--;
            while ((si + 1) * (si + 2) / 2 <= ST) si++;
            sj = ST - si * (si + 1) / 2;
        } else {
            si = nsy + (ST - ntri) / nsy;
            sj = (ST - ntri) % nsy;
        }
    } else {
        si = ST / nsy;
        sj = ST - si * nsy;
    }
    if (si >= nsx) return false;
    bx = si * 8 + (local & 7);
    by = sj * 4 + (local >> 3);
    if (bx >= nbx || by >= nby) return false;
    if (LOWER && bx < 2 * by) return false;
    return true;
}
// workgroups of the XCD walk (a multiple of 8 x 32)
static inline int stream_xcd_grid(bool lower, int nbx, int nby) {
    const int nsx = (nbx + 7) / 8, nsy = (nby + 3) / 4;
    const int nst = lower ? (nsy * (nsy + 1) / 2 + (nsx > nsy ? (nsx - nsy) * nsy : 0)) : nsx * nsy;
    return 8 * ((nst + 7) / 8) * 32;
}

// KTRI walk (R^-1 = W W^T of the theta-gradient, W = C^-T upper triangular): the K range of tile (bx, by) starts at the
// tile's first row, so the work per tile falls with bx -- tiles are enumerated ROW by row (row bx holds by = 0 .. bx / 2),
// the heaviest first.  Tiles before row bx: m (m + 1) for bx = 2 m, (m + 1)^2 for bx = 2 m + 1.
EGX_HD void stream_tile_coords_ktri(int t, int &bx, int &by) {
    int m = (int)((sqrt(4.0 * t + 1.0) - 1.0) * 0.5);
    while (m > 0 && m * (m + 1) > t) m--;
    while ((m + 1) * (m + 2) <= t) m++;
    if (t < (m + 1) * (m + 1)) {
        bx = 2 * m;
        by = t - m * (m + 1);
    } else {
        bx = 2 * m + 1;
        by = t - (m + 1) * (m + 1);
    }
}

