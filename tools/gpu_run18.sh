#!/bin/bash
# A/B: tile-shape thresholds (which kernel serves the look-ahead / small trailing updates)
cd "$GRAFT_REPO_ROOT"
for w in 96 192 512; do for sm in 384 1024; do
  for b in 1 2; do
  echo -n "WIDE=$w SMALL=$sm batch=$b: "
  EGX_GEMM_WIDE=$w EGX_GEMM_SMALL=$sm python bench.py --no-cpu-baseline --steps 20 --warmup 4 --batch $b 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print(round(d['value'],2), round(d['roofline']['achieved'],2), d['roofline']['launches_per_fit'], round(d['stage_ms_single_fit']['potrf_fused_fwd_solve'],2))"
done; done; done
