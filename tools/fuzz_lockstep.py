"""Lock-step batches against single candidates on random problems: n 40..3300 (every padding / panel / group shape of the
factorisation), d 1..10, every mean x kernel, 3..13 candidates with thetas over three decades (some NaN, some tiny: not
positive definite), random workspace counts and lock-step widths.  A candidate must get the SAME BITS whatever the width,
its slot or its companions; statuses must agree too.
    python tools/fuzz_lockstep.py <seed> <seconds> [mid]      (mid: n 4100..9500 -- the sizes whose default width became one slot of
                                                               up to 12 in round 6 -- and the default width among the widths tried)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import egobox_amd as egx  # noqa: E402

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
mid = len(sys.argv) > 3 and sys.argv[3] == "mid"
t0 = time.time()
cases = cand = bad = notpd = 0
while time.time() - t0 < budget:
    n = int(rng.choice([rng.integers(40, 400), rng.integers(400, 1500), rng.integers(1500, 3300)]))
    if mid:
        n = int(rng.choice([rng.integers(4100, 5400), rng.integers(5400, 7200), rng.integers(7200, 9500)]))
    d = int(rng.integers(1, 11))
    mean = int(rng.integers(0, 3))
    corr = int(rng.integers(0, 4))
    if (1, 1 + d, 1 + d + d * (d + 1) // 2)[mean] >= n // 2:
        mean = 0
    x = rng.random((n, d)) * rng.uniform(0.5, 5.0, size=d)
    y = np.sin(x @ rng.standard_normal(d)) + 0.05 * rng.standard_normal(n)
    k = int(rng.integers(3, 14))
    thetas = 10.0 ** rng.uniform(-1.5, 1.2, size=(k, d))
    if rng.random() < 0.5:
        thetas[int(rng.integers(0, k))] = np.nan
    if rng.random() < 0.5:
        thetas[int(rng.integers(0, k))] = 10.0 ** rng.uniform(-6, -3)
    nws = int(rng.integers(2, 13))
    with egx.GpHandle(x, y, mean=mean, corr=corr, n_workspaces=nws) as h:
        h.set_lockstep(1)
        ref_lk, ref_st = h.likelihood_batch(thetas)
        for trial in range(2):
            w = int(rng.integers(2, nws + 1))
            if mid and trial == 0:
                w = 0  # the default
            w = h.set_lockstep(w)
            lk, st = h.likelihood_batch(thetas)
            ok = ref_st == 0
            if not (np.array_equal(st, ref_st) and np.array_equal(lk[ok], ref_lk[ok])):
                bad += 1
                print("MISMATCH n", n, "d", d, "mean", mean, "corr", corr, "nws", nws, "width", w, "status", st.tolist(), ref_st.tolist(),
                      "max rel", float(np.max(np.abs(lk[ok] / ref_lk[ok] - 1.0))) if ok.any() else None)
    cases += 1
    cand += k
    notpd += int((ref_st == 1).sum())
print("cases", cases, "candidates", cand, "not positive definite among them", notpd, "mismatches", bad)
