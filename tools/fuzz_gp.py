"""Random GP configurations (n 5..2600, d 1..8, every mean x kernel, theta over 1.5 decades) on the GPU against the CPU oracle:
likelihood (bar 1e-8), predictions and variances (bar 1e-6) wherever the oracle's smallest pivot is above 1e-3.
    python tools/fuzz_gp.py <seed> <seconds>"""
import numpy as np, sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import egobox_amd as egx
from oracle import gp_oracle as O
KINDS = ["SquaredExponential", "AbsoluteExponential", "Matern32", "Matern52"]; MEANS = ["Constant", "Linear", "Quadratic"]
rng = np.random.default_rng(int(sys.argv[1])); t0 = time.time(); budget = float(sys.argv[2])
checked = skipped = 0; worst_lk = worst_v = 0.0; bad = 0
while time.time() - t0 < budget:
    n = int(rng.integers(5, 2600)); d = int(rng.integers(1, 9))
    mean = int(rng.integers(0, 3)); corr = int(rng.integers(0, 4))
    if O.regression_value(MEANS[mean], np.zeros((1, d))).shape[1] >= n // 2: mean = 0
    x = rng.random((n, d)) * rng.uniform(0.5, 20.0, size=d) + rng.uniform(-5, 5, size=d)
    y = np.sin(x @ rng.standard_normal(d) / np.sqrt(d)) + 0.1 * (x[:, 0] - x[:, 0].mean()) ** 2
    theta = 10.0 ** rng.uniform(-0.5, 1.0, size=d) * (3.0 if corr == 0 else 1.0)
    try:
        ref = O.fit_fixed(x, y, theta, mean=MEANS[mean], corr=KINDS[corr])
    except Exception:
        skipped += 1; continue
    if np.min(np.diag(ref.inner.r_chol)) < 1e-3: skipped += 1; continue
    xq = rng.random((9, d)) * (x.max(axis=0) - x.min(axis=0)) + x.min(axis=0)
    with egx.GpHandle(x, y, mean=mean, corr=corr) as h:
        lk, st = h.likelihood(theta)
        e = abs(lk / ref.likelihood - 1.0) if st == 0 else 1.0
        h.finalize(theta)
        yv, vv = h.predict_valvar(xq)
    ev = np.abs(vv - ref.predict_var(xq)).max() / max(1e-300, np.abs(ref.predict_var(xq)).max())
    ey = np.abs(yv - ref.predict(xq)).max() / max(1e-300, np.abs(ref.predict(xq)).max())
    worst_lk = max(worst_lk, e); worst_v = max(worst_v, ev, ey); checked += 1
    if e > 1e-8 or ev > 1e-6 or ey > 1e-6:
        bad += 1; print("BAD", n, d, mean, corr, st, e, ev, ey, "min pivot", np.min(np.diag(ref.inner.r_chol)))
print("checked", checked, "skipped", skipped, "worst likelihood rel", worst_lk, "worst prediction rel", worst_v, "bad", bad)
