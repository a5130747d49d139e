// Host check of the tile walks of k_gemm_stream (egobox_amd/csrc/tile_walks.h): every walk must visit every tile of its
// region exactly once and nothing else.  Built and run by tests/test_tile_tables_cpu.py (g++, no GPU).
#include <cstdio>
#include <map>
#include <utility>

#include "../../egobox_amd/csrc/tile_walks.h"

static int fails = 0;
#define CHECK(cond, ...)                 \
    do {                                 \
        if (!(cond)) {                   \
            std::printf(__VA_ARGS__);    \
            std::printf("\n");           \
            fails++;                     \
        }                                \
    } while (0)

template <bool LOWER>
static void check_rect(int nbx, int nby) {
    // the region: LOWER -> tiles (bx, by) with bx >= 2 by (the 128 x 256 tiles that touch the lower triangle)
    std::map<std::pair<int, int>, int> want, col;
    for (int by = 0; by < nby; by++)
        for (int bx = 0; bx < nbx; bx++)
            if (!LOWER || bx >= 2 * by) want[{bx, by}] = 0;
    const int nt = (int)want.size();
    for (int t = 0; t < nt; t++) {
        int bx = -1, by = -1;
        stream_tile_coords<LOWER>(t, nbx, nby, bx, by);
        col[{bx, by}]++;
    }
    for (auto &kv : want) {
        CHECK(col[kv.first] == 1, "column walk: tile (%d, %d) visited %d times (nbx %d nby %d lower %d)", kv.first.first,
              kv.first.second, col[kv.first], nbx, nby, (int)LOWER);
    }
    CHECK(col.size() == want.size(), "the column walk left its region (nbx %d nby %d lower %d): %zu / %zu", nbx, nby, (int)LOWER,
          col.size(), want.size());
}

static void check_ktri(int nbx) {  // square, nbx even: rows bx hold by = 0 .. bx / 2, heaviest (smallest bx) first
    const int m = nbx / 2, nt = m * (m + 1);
    std::map<std::pair<int, int>, int> seen;
    int last_bx = 0;
    for (int t = 0; t < nt; t++) {
        int bx = -1, by = -1;
        stream_tile_coords_ktri(t, bx, by);
        CHECK(bx >= last_bx, "ktri walk not row by row at t = %d", t);
        last_bx = bx;
        CHECK(bx >= 0 && bx < nbx && by >= 0 && by <= bx / 2, "ktri walk: tile (%d, %d) outside (nbx %d)", bx, by, nbx);
        seen[{bx, by}]++;
    }
    CHECK((int)seen.size() == nt, "ktri walk: %zu distinct tiles of %d (nbx %d)", seen.size(), nt, nbx);
}

static void check_ktri_rect(int nbx, int nby) {  // all nbx x nby tiles, row by row from the top (the longest K ranges first)
    std::map<std::pair<int, int>, int> seen;
    int last_bx = 0;
    for (int t = 0; t < nbx * nby; t++) {
        int bx = -1, by = -1;
        stream_tile_coords_ktri_rect(t, nby, bx, by);
        CHECK(bx >= last_bx, "ktri rectangle walk not row by row at t = %d", t);
        last_bx = bx;
        CHECK(bx >= 0 && bx < nbx && by >= 0 && by < nby, "ktri rectangle walk: tile (%d, %d) outside (%d x %d)", bx, by, nbx, nby);
        seen[{bx, by}]++;
    }
    CHECK((int)seen.size() == nbx * nby, "ktri rectangle walk: %zu distinct tiles of %d", seen.size(), nbx * nby);
}

int main() {
    for (int nby = 1; nby <= 70; nby++) {
        check_rect<true>(2 * nby, nby);           // the trailing matrix of a factorisation
        check_rect<true>(2 * nby + 1, nby);       // + the right-hand-side rows below it
        check_rect<true>(2 * nby + 7, nby);
        check_ktri(2 * nby);
    }
    for (int nbx = 1; nbx <= 130; nbx += 3)
        for (int nby = 1; nby <= 64; nby += 5) check_rect<false>(nbx, nby);
    for (int nbx = 1; nbx <= 128; nbx += 7)
        for (int nby = 1; nby <= 4; nby++) check_ktri_rect(nbx, nby);  // the C^-T rider's left-looking update: nby = 4
    if (fails) std::printf("%d failures\n", fails);
    else std::printf("tile walks ok\n");
    return fails ? 1 : 0;
}
