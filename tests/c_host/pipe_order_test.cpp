// Host check of the chain launch's task list (egobox_amd/csrc/pipe_tasks.h): the property its deadlock freedom rests on.
// Tickets are handed out in list order to resident workgroups, so the launch can always make progress iff every task's
// producers come EARLIER in the list (or are diagonal blocks, which have workgroups of their own).  The dependency rules below
// restate what the device roles wait for (kernels_pipe.hip):
//   TRSM(p, chunk c)      all FINE tiles of panel p - 1 in row chunk c (column tiles 0 .. width / 64 - 1); strips of DIAG(p)
//   FINE(p, chunk c, j)   TRSM(p) of row chunk c and of the row chunk that holds the tile's columns; every COARSE of an earlier
//                         panel on the 128 x 128 tile that contains it (cver)
//   COARSE(p, I, J)       TRSM(p) of the four 64-row chunks of row tiles I and J; COARSE(p', I, J) for every earlier p' that has one
//   DIAG(s), s > 0        the lower-triangle FINE tiles of panel s - 1 in its own block: they must exist
// plus: no task twice, and the number of tasks of every kind is the blocked algorithm's.  Built and run by
// tests/test_tile_tables_cpu.py (g++, no GPU).
#include <cstdio>
#include <set>
#include <tuple>

#include "../../egobox_amd/csrc/pipe_tasks.h"

using namespace egx;
typedef std::tuple<int, int, int, int> Key;

static int check(int n_pad, int rhs_rows, int g0, int np, int la) {
    const int m_tot = n_pad + rhs_rows, rt = 1;
    const std::vector<PipeTask> v = pipe_tasks(n_pad, m_tot, g0, np, rt, la);
    auto width = [&](int p) { const int k0 = g0 + 256 * p; return (n_pad - k0 < 256) ? (n_pad - k0) : 256; };
    int bad = 0;
    std::set<Key> all, done;
    for (const PipeTask &t : v)
        if (!all.insert(Key(t.type, t.p, t.a, t.b)).second) bad++, std::printf("task twice: type %d p %d a %d b %d\n", t.type, t.p, t.a, t.b);
    for (const PipeTask &t : v) {
        if (t.type == PT_TRSM) {
            if (t.p > 0)
                for (int j = 0; j < width(t.p) / 64; j++)
                    if (!done.count(Key(PT_FINE, t.p - 1, t.a, j))) bad++, std::printf("TRSM p %d chunk %d ahead of FINE tile %d\n", t.p, t.a, j);
        } else if (t.type == PT_FINE) {
            const int C0 = g0 + 256 * t.p + 256 + 64 * t.b, c_col = C0 / 64;
            if (!done.count(Key(PT_TRSM, t.p, t.a, 0)) || !done.count(Key(PT_TRSM, t.p, c_col, 0)))
                bad++, std::printf("FINE p %d chunk %d tile %d ahead of its TRSM\n", t.p, t.a, t.b);
            for (int pp = 0; pp < t.p; pp++) {
                const Key c(PT_COARSE, pp, 64 * t.a / 128, C0 / 128);
                if (all.count(c) && !done.count(c)) bad++, std::printf("FINE p %d chunk %d tile %d ahead of COARSE of panel %d\n", t.p, t.a, t.b, pp);
            }
        } else if (t.type == PT_COARSE) {
            for (int c : {2 * t.a, 2 * t.a + 1, 2 * t.b, 2 * t.b + 1})
                if (64 * c < m_tot && !done.count(Key(PT_TRSM, t.p, c, 0))) bad++, std::printf("COARSE p %d (%d, %d) ahead of TRSM chunk %d\n", t.p, t.a, t.b, c);
            for (int pp = 0; pp < t.p; pp++) {
                const Key c(PT_COARSE, pp, t.a, t.b);
                if (all.count(c) && !done.count(c)) bad++, std::printf("COARSE p %d (%d, %d) ahead of panel %d's\n", t.p, t.a, t.b, pp);
            }
        } else {
            bad++, std::printf("unknown task type %d\n", t.type);
        }
        done.insert(Key(t.type, t.p, t.a, t.b));
    }
    for (int s = 1; s < np; s++) {  // what DIAG(s) waits for exists
        const int c0 = (g0 + 256 * s) / 64;
        for (int c = c0; c < c0 + width(s) / 64; c++)
            for (int j = 0; j <= c - c0; j++)
                if (!all.count(Key(PT_FINE, s - 1, c, j))) bad++, std::printf("no FINE tile (%d, %d) for diagonal block %d\n", c, j, s);
    }
    // completeness: every 64 x 64 tile of block column s + 1 at or below its diagonal gets FINE(s), every 128 x 128 tile of the
    // block columns q >= p + 2 at or below the diagonal gets COARSE(p), every 64-row chunk below block p gets TRSM(p)
    size_t n_trsm = 0, n_fine = 0, n_coarse = 0;
    for (int p = 0; p < np; p++) {
        n_trsm += (size_t)(m_tot - (g0 + 256 * p + width(p))) / 64;
        if (p + 1 < np) {
            const int c0 = (g0 + 256 * (p + 1)) / 64, nb = width(p + 1) / 64;
            for (int c = c0; c < m_tot / 64; c++) n_fine += (size_t)((c - c0 + 1 < nb) ? (c - c0 + 1) : nb);
        }
        for (int q = p + 2; q < np; q++) {
            const int cq = g0 + 256 * q;
            for (int J = cq / 128; J < (cq + width(q)) / 128; J++) n_coarse += (size_t)(m_tot / 128 - J);
        }
    }
    size_t c_trsm = 0, c_fine = 0, c_coarse = 0;
    for (const PipeTask &t : v) (t.type == PT_TRSM ? c_trsm : t.type == PT_FINE ? c_fine : c_coarse)++;
    if (c_trsm != n_trsm || c_fine != n_fine || c_coarse != n_coarse)
        bad++, std::printf("counts: TRSM %zu (want %zu) FINE %zu (%zu) COARSE %zu (%zu)\n", c_trsm, n_trsm, c_fine, n_fine, c_coarse, n_coarse);
    if (bad) std::printf("n_pad %d rhs %d g0 %d np %d la %d: %d violations in %zu tasks\n", n_pad, rhs_rows, g0, np, la, bad, v.size());
    return bad;
}

int main() {
    int bad = 0;
    // whole factorisations (g0 = 0, np = all panels), the product's queue order (16 panels ahead) and the orders measured beside it
    for (int n_pad : {128, 256, 384, 512, 1024, 1152, 2048, 4096, 4352, 7168})
        for (int rhs : {128, 256})
            for (int la : {1, 2, 4, 16, 99}) bad += check(n_pad, rhs, 0, (n_pad + 255) / 256, la);
    // chain launches per group of two / four panels inside a larger matrix
    for (int n_pad : {1024, 4096, 4224})
        for (int gp : {2, 4})
            for (int g0 = 0; g0 < n_pad; g0 += 256 * gp) {
                const int left = (n_pad - g0 + 255) / 256;
                bad += check(n_pad, 128, g0, left < gp ? left : gp, 16);
            }
    if (bad) {
        std::printf("%d violations\n", bad);
        return 1;
    }
    std::printf("chain launch task order ok\n");
    return 0;
}
