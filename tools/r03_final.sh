#!/bin/bash
# Round-3 closing measurements on the GPU box (one gpurun call): see profiles/r03_final_*.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r03_final
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/${T}_gpu_tests.log; tail -3 gpurun_out/${T}_gpu_tests.log
timeout 300 python -m pytest tests -m gpu -q -s -k "config2_dense_sqexp_n4096_d8 or config4_sweep_candidates or collinear or two_ranks or one_shot_fits" 2>&1 | grep -E "config2 at|corner theta|theta 0.00|eps_col|dynamic balance|resident fit|passed|failed" > gpurun_out/${T}_gpu_tests_printed_errors.log; cat gpurun_out/${T}_gpu_tests_printed_errors.log
timeout 900 python bench.py > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err; tail -c 600 gpurun_out/${T}_bench_default.json; echo
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_${T}_default" -o prof -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs > "$GRAFT_REPO_ROOT/gpurun_out/${T}_prof_default_bench.json" 2>/dev/null)
python tools/rocpd_stats.py "$(find gpurun_out/prof_${T}_default -name '*_results.db' | head -1)" > gpurun_out/${T}_bench_default_kernel_stats.txt; head -12 gpurun_out/${T}_bench_default_kernel_stats.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_${T}_b1" -o prof -- python "$GRAFT_REPO_ROOT/bench.py" --sweep-batch 1 --in-flight 1 --lockstep 1 --steps 8 --warmup 2 --no-cpu-baseline --no-extra-configs > "$GRAFT_REPO_ROOT/gpurun_out/${T}_prof_b1_bench.json" 2>/dev/null)
DB=$(find gpurun_out/prof_${T}_b1 -name '*_results.db' | head -1)
python tools/rocpd_stats.py "$DB" > gpurun_out/${T}_bench_sweepbatch1_kernel_stats.txt; tail -4 gpurun_out/${T}_bench_sweepbatch1_kernel_stats.txt
python tools/timeline.py "$DB" gpurun_out/${T}_timeline_n16384_one_fit.txt
# one lock-step group in flight: kernel stats of the batched launches without another group's kernels beside them
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_${T}_g4" -o prof -- python "$GRAFT_REPO_ROOT/bench.py" --sweep-batch 4 --in-flight 4 --lockstep 4 --steps 8 --warmup 2 --no-cpu-baseline --no-extra-configs > "$GRAFT_REPO_ROOT/gpurun_out/${T}_prof_g4_bench.json" 2>/dev/null)
python tools/rocpd_stats.py "$(find gpurun_out/prof_${T}_g4 -name '*_results.db' | head -1)" > gpurun_out/${T}_bench_one_group_kernel_stats.txt; tail -6 gpurun_out/${T}_bench_one_group_kernel_stats.txt
rm -rf gpurun_out/prof_${T}_g4
timeout 300 python tools/small_n_latency.py > gpurun_out/${T}_small_n_latency.jsonl 2>/dev/null; tail -2 gpurun_out/${T}_small_n_latency.jsonl
rm -rf gpurun_out/prof_${T}_default gpurun_out/prof_${T}_b1
timeout 600 python bench_configs.py > gpurun_out/${T}_bench_configs.jsonl 2>/dev/null; wc -l gpurun_out/${T}_bench_configs.jsonl
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
