#!/bin/bash
# A/B: two-level potrf group size x wide GEMM tile threshold
cd "$GRAFT_REPO_ROOT"
for g in 1 2 4; do
  EGX_POTRF_GROUP=$g EGX_GEMM_WIDE=512 python -m pytest tests -q -m gpu -x -k "potrf or golden or full_size" 2>&1 | tail -1
done
for g in 1 2 4; do for w in 0 512 1536; do
  echo "== GROUP=$g WIDE=$w"
  EGX_POTRF_GROUP=$g EGX_GEMM_WIDE=$w python bench.py --no-cpu-baseline --steps 20 --warmup 4 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['achieved'], d['roofline'].get('launch_ms_avg'), d['config'].get('cholesky_tflops_single_fit'))"
done; done
