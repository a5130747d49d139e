// CPU test of the sweep's payload layout (csrc/sweep_shard.h): every rank of a simulated world packs the results of
// its share into a full-length payload, the payloads are concatenated in rank order (what ncclAllGather delivers),
// every rank unpacks, and the outcome must be the serial result -- for ragged k (k < world, k % world != 0, k = 0),
// for the static (c mod world) and for an arbitrary "dynamic" assignment, and with one rank POISONED (its local work
// failed): the survivors' candidates stay valid, the failed rank's come back as missing, the verdict names the rank.
#include <cstdio>
#include <cstring>

#include "../../egobox_amd/csrc/sweep_shard.h"
using namespace egx;

static int owner_of(int64_t c, int world, int mode) {
    if (mode == 0) return (int)(c % world);                       // static
    return (int)(((c * 2654435761u) >> 7) % (unsigned)world);      // some other partition (what a dynamic pull gives)
}

int main() {
    int checked = 0;
    for (int world : {1, 2, 3, 4, 8})
        for (int64_t k : {0, 1, 2, 5, 7, 8, 9, 16, 31, 512, 513})
            for (int mode : {0, 1})
                for (int bad_rank : {-1, 0, world - 1}) {
                    std::vector<double> truth_lk(k);
                    std::vector<int32_t> truth_st(k);
                    for (int64_t c = 0; c < k; c++) {
                        truth_st[c] = (int32_t)(c % 5 == 3 ? 1 : (c % 11 == 7 ? 4 : 0));
                        truth_lk[c] = truth_st[c] ? -INFINITY : 1000.0 + 0.25 * c;
                    }
                    std::vector<double> gathered((size_t)k * 2 * world);
                    int64_t static_total = 0;
                    for (int r = 0; r < world; r++) {
                        static_total += sweep_count_of_rank(k, r, world);
                        for (int64_t j = 0; j < sweep_count_of_rank(k, r, world); j++) {
                            const int64_t c = sweep_candidate(r, j, world);
                            if (c < 0 || c >= k || c % world != r) { printf("bad candidate index\n"); return 1; }
                        }
                        std::vector<double> send = (r == bad_rank) ? sweep_poison(k, 3) : sweep_payload(k);
                        if ((int64_t)send.size() != k * 2) { printf("bad payload size\n"); return 1; }
                        if (r != bad_rank)
                            for (int64_t c = 0; c < k; c++)
                                if (owner_of(c, world, mode) == r) sweep_put(send, c, truth_lk[c], truth_st[c]);
                        if (k) std::memcpy(&gathered[(size_t)r * k * 2], send.data(), sizeof(double) * k * 2);
                    }
                    if (static_total != k) { printf("static shards do not cover the candidates\n"); return 1; }
                    std::vector<double> lk(k, -1.0);
                    std::vector<int32_t> st(k, -1);
                    const SweepVerdict v = sweep_unpack(gathered.data(), k, world, lk.data(), st.data(), 5);
                    int64_t expect_missing = 0;
                    for (int64_t c = 0; c < k; c++) {
                        const bool lost = owner_of(c, world, mode) == bad_rank;
                        expect_missing += lost;
                        const int32_t est = lost ? 5 : truth_st[c];
                        const double elk = lost ? -INFINITY : truth_lk[c];
                        if (st[c] != est || !(lk[c] == elk)) {
                            printf("world %d k %ld mode %d bad %d candidate %ld: (%g, %d) vs (%g, %d)\n", world, (long)k, mode,
                                   bad_rank, (long)c, lk[c], st[c], elk, est);
                            return 1;
                        }
                    }
                    const int expect_failed = (bad_rank >= 0 && k > 0) ? bad_rank : -1;
                    if (v.failed_rank != expect_failed || (expect_failed >= 0 && v.failed_rc != 3) || v.missing != expect_missing ||
                        v.duplicate != 0) {
                        printf("world %d k %ld mode %d bad %d: verdict rank %d rc %d missing %ld duplicate %ld\n", world, (long)k,
                               mode, bad_rank, v.failed_rank, v.failed_rc, (long)v.missing, (long)v.duplicate);
                        return 1;
                    }
                    int64_t sum = 0;
                    for (int r = 0; r < world; r++) sum += v.per_rank[r];
                    if (sum != k - expect_missing) { printf("per-rank counts do not add up\n"); return 1; }
                    checked++;
                }
    // two ranks claiming the same candidate (inconsistent assignment modes) must be noticed
    {
        std::vector<double> a = sweep_payload(4), b = sweep_payload(4), g;
        sweep_put(a, 1, 1.0, 0);
        sweep_put(b, 1, 1.0, 0);
        g.insert(g.end(), a.begin(), a.end());
        g.insert(g.end(), b.begin(), b.end());
        double lk[4];
        int32_t st[4];
        const SweepVerdict v = sweep_unpack(g.data(), 4, 2, lk, st, 5);
        if (v.duplicate != 1 || v.missing != 3) { printf("duplicate / missing not detected\n"); return 1; }
    }
    printf("OK %d (world, k, assignment, failed rank) combinations\n", checked);
    return 0;
}
