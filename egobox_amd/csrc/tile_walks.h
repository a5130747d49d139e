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

// (An XCD-aware 8 x 4 super-tile walk -- the 32 workgroups of an XCD sharing 12 panel-block streams -- was built and measured
//  in round 4: L2 hit rate 0.55 -> 0.66, but -11 % on the sweep; profiles/r04_run2_stream_walk_xcd_supertiles_ab_discarded.txt,
//  profiles/r04_pmc_update_kernel_walk1.json; the code is in the history, commit 54a9160.)

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

// KTRI walk of a RECTANGLE (the left-looking update of the C^-T rider, launch_potrf: columns of one group of panels, all
// rows above the group): again the K range of tile (bx, by) starts at the tile's first row -- row by row from the top, the
// heaviest first, nby tiles per row.
EGX_HD void stream_tile_coords_ktri_rect(int t, int nby, int &bx, int &by) {
    bx = t / nby;
    by = t - bx * nby;
}
