#!/bin/bash
# First GPU pass: layout probe, parity tests, bench, rocprof kernel trace.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "=== rocminfo"; rocminfo | grep -E "Marketing|gfx|Compute Unit" | head -8
echo "=== nproc"; nproc; lscpu | grep "Model name"
echo "=== probe"
timeout 300 python -c "
import egobox_amd as egx, numpy as np
print('mfma probe err', egx.mfma_probe())
"
echo "=== pytest gpu"
timeout 1800 python -m pytest tests -m gpu -q --timeout=900 --tb=short -rf 2>&1 | tail -150
echo "=== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()"
echo "=== bench"
timeout 900 python bench.py --steps 4 --warmup 1
} > gpurun_out/run1.log 2>&1
echo "=== rocprof" >> gpurun_out/run1.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_r1" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline >> "$GRAFT_REPO_ROOT/gpurun_out/run1.log" 2>&1)
ls -R gpurun_out/prof_r1 | head -20 >> gpurun_out/run1.log
tail -60 gpurun_out/run1.log
