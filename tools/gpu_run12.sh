#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
LOG=gpurun_out/run12.log
{
echo "=== pytest gpu (256-thread diag kernel)"
timeout 1800 python -m pytest tests -m gpu -q --timeout=900 --tb=short -rf 2>&1 | tail -5
for T in 256 512; do for B in 1 2; do echo "=== bench potf2_threads=$T batch=$B"; EGX_POTF2_THREADS=$T timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --batch $B | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms_single_fit'])"; done; done
} > $LOG 2>&1
cat $LOG | cut -c1-330
