#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_gb
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_gb" -o prof -- python "$R/tools/grad_bench.py" 16384 32 3 8 > "$R/gpurun_out/r04_call13_grad_bench.log" 2>&1)
python tools/rocpd_stats.py "$(find gpurun_out/prof_gb -name '*_results.db' | head -1)" > gpurun_out/r04_call13_grad_batch_kernel_stats.txt
rm -rf $R/gpurun_out/prof_gb
cat gpurun_out/r04_call13_grad_bench.log | tail -8; head -30 gpurun_out/r04_call13_grad_batch_kernel_stats.txt | cut -c1-170
