#!/usr/bin/env python3
"""ONE lock-step group of eight n = 16384 fits, the look-ahead columns' update serialised in front of the trailing update
(egx_set_tuning lur_side = 0): the command behind bench.py's `roofline` (launch shape of the timed region, clean per-launch
durations).  `rocprofv3 --kernel-trace --stats -- python tools/group_roofline.py` + tools/rocpd_stats.py print the same
launches as their own row ("... 8 matrices per launch")."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egobox_amd as egx  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
d = int(sys.argv[2]) if len(sys.argv) > 2 else 32
gl = int(sys.argv[3]) if len(sys.argv) > 3 else 8
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
x, y = egx.workload.make_training_set(n, d, 42)
base = egx.workload.default_theta(d)
h = egx.GpHandle(x, y, corr=0, n_workspaces=gl)
h.set_lockstep(gl)
egx.set_tuning("lur_side", 0)
for j in range(reps):
    lk, st = h.likelihood_batch(np.stack([base * (1.0 + 0.01 * (j * gl + c)) for c in range(gl)]))
    t = h.timings()
    print(f"group {j}: potrf {t['potrf_ms']:.2f} ms, {t['syrk_launches']} chip-filling launches {t['potrf_syrk_ms']:.3f} ms = "
          f"{t['syrk_flops'] / t['potrf_syrk_ms'] / 1e9:.2f} TFLOP/s, statuses {np.bincount(st)}", flush=True)
h.close()
