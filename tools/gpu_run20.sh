#!/bin/bash
# same-box A/B of the software-pipelined K loop (EGX_GEMM_PIPE)
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for p in 0 1; do
  echo "== PIPE=$p gemm_prof n=16384 K=512"; EGX_GEMM_PIPE=$p timeout 100 tools/gemm_prof 16384 r 512 | grep -E "rep 5|shader|K loop"
  echo -n "PIPE=$p bench: "
  EGX_GEMM_PIPE=$p python bench.py --no-cpu-baseline --steps 20 --warmup 4 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print(round(d['value'],2), round(d['roofline']['achieved'],2), d['roofline']['launches_per_fit'], round(d['stage_ms_single_fit']['potrf_fused_fwd_solve'],2))"
done; done
