#!/bin/bash
# rocprofv3 kernel stats of `bench.py --batch 1` (one fit in flight: kernel durations are their own, comparable with
# the HIP-event launch_ms_avg of the roofline leg)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
LOG=gpurun_out/run26.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_r26_b1" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --batch 1 2>&1 | grep -v "simple_timer\|generateRocpd" > "$GRAFT_REPO_ROOT/$LOG")
python tools/rocpd_stats.py gpurun_out/prof_r26_b1/bench_results.db >> $LOG 2>&1
cat $LOG | cut -c1-2600
