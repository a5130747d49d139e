#!/bin/bash
# same-box A/B of the split-K tail of the chip-filling trailing updates (EGX_GEMM_TAIL)
cd "$GRAFT_REPO_ROOT"
python -m pytest tests -q -m gpu -x -k "potrf or golden or full_size or ornstein or config or n20000" 2>&1 | tail -1
for rep in 1 2; do for t in 0 1; do for b in 2 1; do
  echo -n "TAIL=$t batch=$b: "
  EGX_GEMM_TAIL=$t python bench.py --no-cpu-baseline --steps 20 --warmup 4 --batch $b 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print(round(d['value'],2), round(d['roofline']['achieved'],2), d['roofline']['launches_per_fit'], round(d['stage_ms_single_fit']['potrf_fused_fwd_solve'],2))"
done; done; done
