#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
LOG=gpurun_out/run11.log
{
echo "=== potf2 profile"; timeout 120 ./tools/potf2_prof | tail -5
echo "=== pytest gpu"
timeout 1800 python -m pytest tests -m gpu -q --timeout=900 --tb=short -rf 2>&1 | tail -30
for B in 1 2; do echo "=== bench batch=$B"; timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --batch $B; done
} > $LOG 2>&1
echo "=== rocprof" >> $LOG
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_r11" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --batch 1 2>&1 | grep -v "simple_timer\|generateRocpd\|^{" >> "$GRAFT_REPO_ROOT/$LOG")
python tools/rocpd_stats.py gpurun_out/prof_r11/bench_results.db >> $LOG 2>&1
cat $LOG | cut -c1-330
