#!/bin/bash
# round 4, GPU call 5: whole suite with the left-looking default, default bench, group roofline kernel stats, small-n A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -30 > $O/r04c5_all_tests.log
timeout 300 python tools/ab_small.py "tail_merge=0" "tail_merge=1" "tail_merge=1,potrf_left=2" "tail_merge=1,look_min=0" > $O/r04c5_ab_small_n4096.log 2>&1
timeout 300 python tools/ab_small.py "tail_merge=0" "tail_merge=1" "tail_merge=1,potrf_left=2" --n 8192 --d 16 --rounds 2 > $O/r04c5_ab_small_n8192.log 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_grp" -o prof -- python "$GRAFT_REPO_ROOT/tools/group_roofline.py" > "$GRAFT_REPO_ROOT/$O/r04c5_group_roofline.log" 2>&1)
python tools/rocpd_stats.py "$(find $O/prof_grp -name '*_results.db' | head -1)" > $O/r04c5_group_roofline_kernel_stats.txt 2>&1
rm -rf $O/prof_grp
timeout 900 python bench.py > $O/r04c5_bench_default.json 2> $O/r04c5_bench_default.err
tail -6 $O/r04c5_all_tests.log; cat $O/r04c5_ab_small_n4096.log $O/r04c5_ab_small_n8192.log; grep "^group" $O/r04c5_group_roofline.log; grep -A1 "left-looking\|matrices per launch" $O/r04c5_group_roofline_kernel_stats.txt | cut -c1-160
python - <<'PY'
import json
try:
    r = json.loads(open("gpurun_out/r04c5_bench_default.json").read().strip().splitlines()[-1])
    print("value", r["value"], "tflops/gpu", r["cholesky_tflops_per_gpu_in_timed_region"])
    print("roofline", {k: r["roofline"].get(k) for k in ("achieved", "frac", "launch_ms_avg", "launches_per_group", "share_of_potrf_flops")})
    print("group alone", r.get("lockstep_group_alone"))
    oc = r.get("other_configs", {})
    c3 = oc.get("config3_matern52_n16384_d32", {})
    print("config3", c3.get("likelihood_plus_theta_gradient_ms"), c3.get("gradient_roofline_lockstep_batch_of_8"))
    print("config4", oc.get("config4_sweep_512"))
    print("tuned", oc.get("tuned_fit_11_starts_sharded"))
except Exception as e:
    print("bench parse failed:", e)
PY
