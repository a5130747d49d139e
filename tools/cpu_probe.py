#!/usr/bin/env python3
"""Where does the host's dpotrf rate come from?  Times LAPACK dpotrf (scipy/OpenBLAS) at n = 8192 under several
thread settings, with and without torch imported first, and prints the host topology.  Diagnostic for the
`cpu_baseline` leg of bench.py (round 1 saw 25 GFLOP/s on a 128-core host)."""
import json
import os
import subprocess
import sys
import time

CHILD = r"""
import os, sys, time, json
pre = sys.argv[1]
if pre == "torch":
    import torch
import numpy as np
from scipy.linalg import lapack
from threadpoolctl import threadpool_info, threadpool_limits
n = int(sys.argv[2]); lim = int(sys.argv[3])
if lim > 0:
    threadpool_limits(limits=lim, user_api="blas")
rng = np.random.default_rng(0)
a = rng.standard_normal((n, 64))
r = a @ a.T + n * np.eye(n)
best = 1e9
for rep in range(2):
    m = np.asfortranarray(r)
    t0 = time.perf_counter()
    c, info = lapack.dpotrf(m, lower=1, overwrite_a=1, clean=0)
    best = min(best, time.perf_counter() - t0)
blas = [p for p in threadpool_info() if p["user_api"] == "blas"]
print(json.dumps({"pre": pre, "n": n, "limit": lim, "env_openblas": os.environ.get("OPENBLAS_NUM_THREADS"),
                  "dpotrf_s": best, "gflops": n ** 3 / 3 / best / 1e9,
                  "blas_threads": [p["num_threads"] for p in blas], "affinity": len(os.sched_getaffinity(0))}))
"""


def run(pre, n, lim, env_threads=None):
    env = dict(os.environ)
    env.pop("OPENBLAS_NUM_THREADS", None)
    env.pop("OMP_NUM_THREADS", None)
    if env_threads:
        env["OPENBLAS_NUM_THREADS"] = str(env_threads)
    out = subprocess.run([sys.executable, "-c", CHILD, pre, str(n), str(lim)], env=env, capture_output=True, text=True)
    print(out.stdout.strip() or out.stderr[-300:], flush=True)


if __name__ == "__main__":
    print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), flush=True)
    print(subprocess.run("lscpu | grep -E 'Model name|Socket|Core|Thread|NUMA node\\(s\\)|MHz' | head -12", shell=True,
                         capture_output=True, text=True).stdout, flush=True)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    for pre in ("none", "torch"):
        run(pre, n, 0)
    for lim in (8, 16, 32, 64):
        run("none", n, lim)
    run("none", n, 0, env_threads=32)
