#!/bin/bash
# Closing run of a round on the tree as it is: GPU suite, smoke, the driver-shaped bench line, the same command under rocprofv3.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-closing}
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/${TAG}_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> gpurun_out/${TAG}_gpu_tests.log 2>&1
timeout 900 python bench.py 2>gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench_default.json
cp gpurun_out/bench_last_run_details.json gpurun_out/${TAG}_bench_details.json 2>/dev/null
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}" -o prof -- python "$GRAFT_REPO_ROOT/bench.py" > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_bench_default_profiled.json" 2>/dev/null)
python tools/rocpd_stats.py "$(find gpurun_out/prof_${TAG} -name '*_results.db' | head -1)" 2>&1 | head -40 > gpurun_out/${TAG}_kernel_stats.txt
rm -rf gpurun_out/prof_${TAG}
tail -3 gpurun_out/${TAG}_gpu_tests.log; wc -c gpurun_out/${TAG}_bench_default.json; head -c 400 gpurun_out/${TAG}_bench_default.json; echo; head -12 gpurun_out/${TAG}_kernel_stats.txt
