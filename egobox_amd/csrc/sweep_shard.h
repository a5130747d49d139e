// Index arithmetic and payload layout of the theta sweep (sweep.hip), kept free of HIP / RCCL so that the CPU test
// tests/c_host/sweep_shard_test.cpp can run every rank's pack / unpack against each other.
//
// Assignment of candidate c of k:
//   static   rank c % world (the order of the reference's rayon par_iter does not matter: the evaluations are
//            independent, crates/gp/src/algorithm.rs:928-945)
//   dynamic  whoever pulls c from the node-wide counter first (sweep.hip; candidates that are not positive definite
//            return ~10x sooner than the others, so a static shard can be badly unbalanced)
// Either way the payload of a rank is FULL LENGTH: k slots of {likelihood, status as double}; a slot the rank did not
// evaluate holds {NaN, kSweepEmpty}.  A rank whose local work failed (HIP error, out of memory ...) still takes part
// in the all-gather with a POISONED payload -- every slot {NaN, -(kSweepPoison + rc)} -- so that no rank is left
// waiting in the collective; after unpacking every rank reports the failure.
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

namespace egx {

constexpr int kSweepEmpty = -1;     // status of a slot this rank did not evaluate
constexpr int kSweepPoison = 1000;  // status of every slot of a failed rank: -(kSweepPoison + egx_rc)

inline int64_t sweep_slots_per_rank(int64_t k, int world) { return (k + world - 1) / world; }
inline int64_t sweep_count_of_rank(int64_t k, int rank, int world) { return (k > rank) ? (k - rank + world - 1) / world : 0; }
inline int64_t sweep_candidate(int rank, int64_t slot, int world) { return rank + slot * (int64_t)world; }

// an empty full-length payload (k slots)
inline std::vector<double> sweep_payload(int64_t k) {
    std::vector<double> send((size_t)k * 2);
    for (int64_t c = 0; c < k; c++) {
        send[2 * c] = std::numeric_limits<double>::quiet_NaN();
        send[2 * c + 1] = (double)kSweepEmpty;
    }
    return send;
}
inline void sweep_put(std::vector<double> &send, int64_t c, double lk, int32_t st) {
    send[2 * c] = lk;
    send[2 * c + 1] = (double)st;
}
// the payload of a rank whose local work failed with egx_rc `rc`
inline std::vector<double> sweep_poison(int64_t k, int rc) {
    std::vector<double> send((size_t)k * 2);
    for (int64_t c = 0; c < k; c++) {
        send[2 * c] = std::numeric_limits<double>::quiet_NaN();
        send[2 * c + 1] = -(double)(kSweepPoison + rc);
    }
    return send;
}

struct SweepVerdict {
    int failed_rank = -1;   // lowest rank that sent a poisoned payload (-1: none)
    int failed_rc = 0;      // its egx_rc
    int64_t missing = 0;    // candidates nobody evaluated (only possible next to a failed rank)
    int64_t duplicate = 0;  // candidates more than one rank evaluated (never, by construction)
    std::vector<int64_t> per_rank;  // candidates each rank evaluated (the balance of a dynamic sweep)
};

// all ranks' payloads concatenated in rank order (what ncclAllGather delivers) -> per-candidate arrays.
// Candidates without a result get {-inf, status_missing}.
inline SweepVerdict sweep_unpack(const double *recv, int64_t k, int world, double *lkh, int32_t *status,
                                 int32_t status_missing) {
    SweepVerdict v;
    v.per_rank.assign(world, 0);
    for (int r = 0; r < world && k > 0; r++) {
        const double s0 = recv[(size_t)r * k * 2 + 1];
        if (s0 <= -(double)kSweepPoison && v.failed_rank < 0) {
            v.failed_rank = r;
            v.failed_rc = (int)(-s0) - kSweepPoison;
        }
    }
    for (int64_t c = 0; c < k; c++) {
        int found = 0;
        for (int r = 0; r < world; r++) {
            const double *slot = recv + ((size_t)r * k + c) * 2;
            if (slot[1] >= 0.0) {
                if (!found) {
                    lkh[c] = slot[0];
                    status[c] = (int32_t)slot[1];
                }
                found++;
                v.per_rank[r]++;
            }
        }
        if (!found) {
            lkh[c] = -std::numeric_limits<double>::infinity();
            status[c] = status_missing;
            v.missing++;
        } else if (found > 1) {
            v.duplicate++;
        }
    }
    return v;
}

}  // namespace egx
