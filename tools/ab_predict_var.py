#!/usr/bin/env python3
"""A/B of run-time knobs on predict_var (BASELINE config 5's expert: n = 8192, d = 16, 100 000 points; and n = 16384, d = 32,
32 768 points), interleaved:   python tools/ab_predict_var.py "trsm_left=0" "trsm_left=1" """
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egobox_amd as egx  # noqa: E402

settings = sys.argv[1:] or ["trsm_left=0", "trsm_left=1"]


def apply(setting):
    for kv in setting.split(","):
        k, v = kv.split("=")
        egx.set_tuning(k.strip(), int(v))


for n, d, m in ((8192, 16, 100000), (16384, 32, 32768)):
    x, y = egx.workload.make_training_set(n, d, 7)
    h = egx.GpHandle(x, y, corr=0)
    h.finalize(egx.workload.default_theta(d))
    xq = np.random.default_rng(7).random((m, d))
    h.predict_var(xq[:20000])
    res = {s: [] for s in settings}
    ref = None
    for r in range(3):
        for s in settings:
            apply(s)
            t0 = time.perf_counter()
            v = h.predict_var(xq)
            res[s].append(m / (time.perf_counter() - t0))
            if ref is None:
                ref = v
            else:
                assert np.allclose(v, ref, rtol=1e-7, atol=1e-12 * np.abs(ref).max())
    for s in settings:
        best = max(res[s])
        print(f"n={n} d={d} m={m} {s}: {' '.join(f'{q / 1e3:.1f}' for q in res[s])} k points/s; best = "
              f"{float(n) * n * best / 1e12:.1f} TFLOP/s = {float(n) * n * best / 78.6e12:.3f} of peak", flush=True)
    h.close()
