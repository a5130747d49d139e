#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
LOG=gpurun_out/run13.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_r13" -o cfg5 -- python "$GRAFT_REPO_ROOT/bench_configs.py" --only 5 2>&1 | grep -v "simple_timer\|generateRocpd" > "$GRAFT_REPO_ROOT/$LOG")
python tools/rocpd_stats.py gpurun_out/prof_r13/cfg5_results.db >> $LOG 2>&1
cat $LOG | cut -c1-400
