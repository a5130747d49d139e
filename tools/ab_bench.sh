#!/bin/bash
# A/B of environment-switchable kernel variants through bench.py (same box, interleaved):
#   tools/ab_bench.sh "EGX_GEMM_STREAM=0" "EGX_GEMM_STREAM=1" "EGX_GEMM_STREAM=1 EGX_STREAM_TPW=2" ...
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
OUT=gpurun_out/${EGX_TAG:-r02}_ab.txt
: > $OUT
for rep in 1 2; do
  for cfg in "$@"; do
    line=$(env $cfg timeout 300 python bench.py --steps ${AB_STEPS:-6} --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%.2f fits/s | single-fit %.2f fits/s | update kernel %.2f TFLOP/s (%.3f ms x %d) | potrf %.2f ms' % (d['value'], d['single_fit_in_flight_fits_per_s'], d['roofline']['achieved'], d['roofline']['launch_ms_avg'], d['roofline']['launches_per_fit'], d['stage_ms_single_fit']['potrf_fused_fwd_solve']))")
    echo "[$cfg] $line" | tee -a $OUT
  done
done
