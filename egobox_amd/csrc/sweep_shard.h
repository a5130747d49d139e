// Index arithmetic of the theta sweep's static partition (sweep.hip), kept free of HIP / RCCL so that the CPU test
// tests/c_host/sweep_shard_test.cpp can run every rank's pack / unpack against each other.
//   candidate c of k  ->  rank c % world, slot c / world of that rank's payload
//   payload of a rank =  per = ceil(k / world) slots of {likelihood, status as double}; unused slots hold NaN
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

namespace egx {

inline int64_t sweep_slots_per_rank(int64_t k, int world) { return (k + world - 1) / world; }
inline int64_t sweep_count_of_rank(int64_t k, int rank, int world) { return (k > rank) ? (k - rank + world - 1) / world : 0; }
inline int64_t sweep_candidate(int rank, int64_t slot, int world) { return rank + slot * (int64_t)world; }

// this rank's payload from its shard results (mine = sweep_count_of_rank entries)
inline std::vector<double> sweep_pack(const double *lk, const int32_t *st, int64_t mine, int64_t per) {
    std::vector<double> send((size_t)per * 2, std::numeric_limits<double>::quiet_NaN());
    for (int64_t j = 0; j < mine; j++) {
        send[2 * j] = lk[j];
        send[2 * j + 1] = (double)st[j];
    }
    return send;
}

// all ranks' payloads concatenated in rank order (what ncclAllGather delivers) -> per-candidate arrays
inline void sweep_unpack(const double *recv, int64_t k, int world, double *lkh, int32_t *status) {
    const int64_t per = sweep_slots_per_rank(k, world);
    for (int r = 0; r < world; r++)
        for (int64_t j = 0; sweep_candidate(r, j, world) < k; j++) {
            const int64_t c = sweep_candidate(r, j, world);
            lkh[c] = recv[((size_t)r * per + j) * 2];
            status[c] = (int32_t)recv[((size_t)r * per + j) * 2 + 1];
        }
}

}  // namespace egx
