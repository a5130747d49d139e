#!/bin/bash
# round 4, GPU call 1: the rebuilt theta-gradient -- parity tests, timings, kernel profile; the whole GPU suite; group roofline
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests -m gpu -q -x -k "gradient or grad or lbfgs" 2>&1 | tail -15 > $O/r04c1_grad_tests.log
timeout 300 python tools/grad_bench.py 16384 32 3 8 > $O/r04c1_grad_bench.log 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_grad" -o prof -- python "$GRAFT_REPO_ROOT/tools/grad_bench.py" 16384 32 3 1 > "$GRAFT_REPO_ROOT/$O/r04c1_grad_prof_run.log" 2>&1)
python tools/rocpd_stats.py "$(find $O/prof_grad -name '*_results.db' | head -1)" > $O/r04c1_grad_kernel_stats.txt 2>&1
python tools/timeline.py "$(find $O/prof_grad -name '*_results.db' | head -1)" $O/r04c1_grad_timeline.txt >> $O/r04c1_grad_prof_run.log 2>&1
rm -rf $O/prof_grad
timeout 200 python tools/group_roofline.py > $O/r04c1_group_roofline.log 2>&1
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/r04c1_all_tests.log
tail -5 $O/r04c1_grad_tests.log; cat $O/r04c1_grad_bench.log; head -30 $O/r04c1_grad_kernel_stats.txt; cat $O/r04c1_group_roofline.log; tail -8 $O/r04c1_all_tests.log
