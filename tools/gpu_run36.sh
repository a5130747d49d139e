#!/bin/bash
# full verification: all GPU tests, smoke, bench (default + batch 1), bench_configs, rocprof kernel stats
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
LOG=gpurun_out/run36.log
{
echo "=== pytest gpu"; timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -3
echo "=== smoke"; python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -2
echo "=== bench default"; timeout 900 python bench.py --steps 10 --warmup 3
echo "=== bench batch 1"; timeout 900 python bench.py --steps 10 --warmup 3 --batch 1 --no-cpu-baseline
echo "=== bench_configs"; timeout 900 python bench_configs.py
} > $LOG 2>&1
for B in 2; do
echo "=== rocprof batch $B" >> $LOG
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_r36_b$B" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --batch $B 2>&1 | grep -v "simple_timer\|generateRocpd" >> "$GRAFT_REPO_ROOT/$LOG")
python tools/rocpd_stats.py gpurun_out/prof_r36_b$B/bench_results.db >> $LOG 2>&1
done
cat $LOG | cut -c1-1500
