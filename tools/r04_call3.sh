#!/bin/bash
# round 4, GPU call 3: whole suite (mixture gradients, d > 64, ...), left-looking A/B, PMC passes, gradient kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $O/r04c3_all_tests.log
timeout 400 python tools/ab_knobs.py "potrf_left=0,stream_min=128" "potrf_left=1,stream_min=8" "potrf_left=1,stream_min=8,potrf_group=8" --rounds 2 > $O/r04c3_ab_left_looking.log 2>&1
EGX_POTRF_LEFT=1 EGX_STREAM_MIN=8 timeout 600 python -m pytest tests -m gpu -q -k "lockstep or potrf or fixed_theta or gradient" 2>&1 | tail -6 > $O/r04c3_left_tests.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_grad" -o prof -- python "$GRAFT_REPO_ROOT/tools/one_grad.py" 16384 32 4 3 > "$GRAFT_REPO_ROOT/$O/r04c3_one_grad.log" 2>&1)
python tools/rocpd_stats.py "$(find $O/prof_grad -name '*_results.db' | head -1)" > $O/r04c3_grad_kernel_stats.txt 2>&1
rm -rf $O/prof_grad
bash tools/gpu_pmc_r04.sh > $O/r04c3_pmc.log 2>&1
tail -12 $O/r04c3_all_tests.log; cat $O/r04c3_ab_left_looking.log; tail -3 $O/r04c3_left_tests.log; cat $O/r04c3_one_grad.log | tail -4; head -14 $O/r04c3_grad_kernel_stats.txt | cut -c1-150; tail -12 $O/r04c3_pmc.log
python - <<'PY'
import json
for f in ("r04_pmc_update_kernel.json", "r04_pmc_update_kernel_walk1.json"):
    try:
        r = json.load(open("gpurun_out/" + f))
        print(f, {k: r.get(k) for k in ("fetch_bytes_per_launch", "write_bytes_per_launch", "l2_hit_rate", "mfma_busy_frac")})
    except Exception as e:
        print(f, "failed", e)
PY
