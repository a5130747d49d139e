"""egx_potrf against LAPACK dpotrf on random matrices of four families (see tests/test_gpu_diag_block.py::test_fuzz_against_lapack).
    python tools/fuzz_potrf.py <seed> <seconds>"""
import numpy as np, sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import egobox_amd as egx
import scipy.linalg as sl
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
eps = np.finfo(float).eps
worst = 0.0; fails = 0; t0 = time.time(); cases = 0
while time.time() - t0 < float(sys.argv[2]) if len(sys.argv) > 2 else 60:
    n = int(rng.integers(1, 1600))
    kind = rng.integers(0, 4)
    if kind == 0:
        g = rng.standard_normal((n, n)); a = g @ g.T / n + 10.0 ** rng.uniform(-6, 0) * np.eye(n)
    elif kind == 1:
        t = np.sort(rng.random(n)) * rng.uniform(0.5, 20); ls = 10.0 ** rng.uniform(-3, 0)
        a = np.exp(-((t[:, None] - t[None, :]) ** 2) / ls) + 10.0 ** rng.uniform(-13, -6) * np.eye(n)
    elif kind == 2:
        d = int(rng.integers(1, 6)); x = rng.random((n, d)); th = 10.0 ** rng.uniform(-2, 1, d)
        dd = np.abs(x[:, None, :] - x[None, :, :]) * th
        a = np.exp(-dd.sum(-1)) + 1e-12 * np.eye(n)
    else:
        g = rng.standard_normal((n, max(1, n // 2))); a = g @ g.T / n + 10.0 ** rng.uniform(-14, -10) * np.eye(n)  # rank deficient + tiny shift
    p = rng.permutation(n); a = a[np.ix_(p, p)]
    _, li = sl.lapack.dpotrf(a, lower=1)
    l, info = egx.potrf(a)
    cases += 1
    if li == 0 and info == 0:
        r = np.abs(l @ l.T - a).max() / np.abs(a).max() / eps
        worst = max(worst, r)
        if r > 200 or not np.all(np.triu(l, 1) == 0):
            print("RESIDUAL", n, kind, r); fails += 1
    elif (li == 0) != (info == 0):
        # only a disagreement when LAPACK's smallest pivot is far from the rounding level
        lw = np.linalg.cholesky(a) if li == 0 else None
        margin = (np.diag(lw).min() ** 2 / np.abs(a).max()) if lw is not None else 0.0
        print("STATUS", n, kind, "lapack", li, "gpu", info, "lapack min pivot / max|a| %.2e" % margin)
        if margin > 1e-11: fails += 1
    elif li != info:
        print("INFO index", n, kind, li, info)
print("cases", cases, "worst residual/eps", worst, "fails", fails)
