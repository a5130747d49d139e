#!/bin/bash
# A/B of run-time variants through bench.py on one box, interleaved, two repetitions:
#   tools/ab3.sh "ENV=.. ENV=.. @ --in-flight 6 --lockstep 3" "..."
# (left of '@': environment, right of it: extra bench.py arguments)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
OUT=gpurun_out/${EGX_TAG:-r03}_ab.txt
: > $OUT
for rep in $(seq 1 ${AB_REPS:-2}); do
  for cfg in "$@"; do
    envs="${cfg%%@*}"; args="${cfg#*@}"
    [ "$envs" == "$cfg" ] && args=""
    line=$(env $envs timeout 300 python bench.py --steps ${AB_STEPS:-5} --warmup 1 --no-cpu-baseline $args 2>gpurun_out/ab_err.log | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%.2f fits/s | single-fit %.2f fits/s | update kernel %.2f TFLOP/s (%.3f ms x %d) | potrf %.2f ms | one-shot first %.1f pooled %.1f fits/s' % (d['value'], d['single_fit_in_flight_fits_per_s'], d['roofline']['achieved'], d['roofline']['launch_ms_avg'], d['roofline']['launches_per_fit'], d['stage_ms_single_fit']['potrf_fused_fwd_solve'], d['pcie_inclusive']['first_handle_in_process']['fits_per_s'], d['pcie_inclusive']['fits_per_s_cold_handle']))" 2>&1)
    [ -z "$line" ] && line="FAILED: $(tail -3 gpurun_out/ab_err.log | tr '\n' ' ')"
    echo "[$cfg] $line" | tee -a $OUT
  done
done
