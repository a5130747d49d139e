// CPU test of the sweep's partition arithmetic (csrc/sweep_shard.h): every rank of a simulated world packs the
// results of its shard, the payloads are concatenated in rank order (what ncclAllGather delivers), every rank unpacks,
// and the outcome must be the serial result -- for ragged k (k < world, k % world != 0, k = 0) too.
#include <cstdio>
#include <cstring>

#include "../../egobox_amd/csrc/sweep_shard.h"
using namespace egx;

int main() {
    int checked = 0;
    for (int world : {1, 2, 3, 4, 8})
        for (int64_t k : {0, 1, 2, 5, 7, 8, 9, 16, 31, 512, 513}) {
            std::vector<double> truth_lk(k);
            std::vector<int32_t> truth_st(k);
            for (int64_t c = 0; c < k; c++) {
                truth_st[c] = (int32_t)(c % 5 == 3 ? 1 : (c % 11 == 7 ? 4 : 0));
                truth_lk[c] = truth_st[c] ? -INFINITY : 1000.0 + 0.25 * c;
            }
            const int64_t per = sweep_slots_per_rank(k, world);
            std::vector<double> gathered((size_t)per * 2 * world);
            int64_t total = 0;
            for (int r = 0; r < world; r++) {
                const int64_t mine = sweep_count_of_rank(k, r, world);
                total += mine;
                std::vector<double> lk(mine);
                std::vector<int32_t> st(mine);
                for (int64_t j = 0; j < mine; j++) {
                    const int64_t c = sweep_candidate(r, j, world);
                    if (c < 0 || c >= k || c % world != r) { printf("bad candidate index\n"); return 1; }
                    lk[j] = truth_lk[c];
                    st[j] = truth_st[c];
                }
                std::vector<double> send = sweep_pack(lk.data(), st.data(), mine, per);
                if ((int64_t)send.size() != per * 2) { printf("bad payload size\n"); return 1; }
                if (per) std::memcpy(&gathered[(size_t)r * per * 2], send.data(), sizeof(double) * per * 2);
            }
            if (total != k) { printf("shards do not cover the candidates: %ld of %ld\n", (long)total, (long)k); return 1; }
            std::vector<double> lk(k, -1.0);
            std::vector<int32_t> st(k, -1);
            sweep_unpack(gathered.data(), k, world, lk.data(), st.data());
            for (int64_t c = 0; c < k; c++)
                if (st[c] != truth_st[c] || !(lk[c] == truth_lk[c])) {
                    printf("world %d k %ld candidate %ld: (%g, %d) vs (%g, %d)\n", world, (long)k, (long)c, lk[c], st[c],
                           truth_lk[c], truth_st[c]);
                    return 1;
                }
            checked++;
        }
    printf("OK %d (world, k) combinations\n", checked);
    return 0;
}
