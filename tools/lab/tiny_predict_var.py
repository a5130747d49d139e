import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import egobox_amd as egx
for n in (60, 100, 128, 129, 200, 256, 300):
    d = 4
    x, y = egx.workload.make_training_set(n, d, 1)
    h = egx.GpHandle(x, y, corr=3)
    h.finalize(np.full(d, 1.0))
    xq = np.random.default_rng(0).random((1, d))
    for name, fn in (("predict", lambda: h.predict(xq)), ("predict_var", lambda: h.predict_var(xq)), ("valvar", lambda: h.predict_valvar(xq))):
        fn()
        t0 = time.perf_counter()
        for _ in range(50):
            fn()
        print(f"n={n} {name}: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms", flush=True)
    h.close()
