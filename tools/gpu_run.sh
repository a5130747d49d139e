#!/bin/bash
# One parametrised GPU-box script (run through gpurun):  tools/gpu_run.sh <stage> [...]
#   tests [pytest args]   python -m pytest tests -m gpu -q <args>
#   bench [bench args]    python bench.py <args>
#   prof  [bench args]    rocprofv3 --kernel-trace --stats of bench.py <args>  -> gpurun_out/prof_<tag>/
#   cpu                   tools/cpu_probe.py + oracle/cpu_baseline.py at full size
# Several stages can be chained with '--':  tools/gpu_run.sh tests -x -- bench --steps 10
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${EGX_TAG:-r02}
stage() {
    local s=$1; shift
    case "$s" in
    tests) timeout 1500 python -m pytest tests -m gpu -q "$@" 2>&1 | tail -25 | tee gpurun_out/${TAG}_tests.log ;;
    bench) timeout 900 python bench.py "$@" 2>gpurun_out/${TAG}_bench.err | tee gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.err ;;
    prof)
        (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}" -o prof -- \
            python "$GRAFT_REPO_ROOT/bench.py" "$@" > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_bench.json" 2>/dev/null)
        python tools/rocpd_stats.py "$(find gpurun_out/prof_${TAG} -name '*_results.db' | head -1)" 2>&1 | head -30 | tee gpurun_out/${TAG}_kernel_stats.txt ;;
    cpu)
        python tools/cpu_probe.py 8192 2>&1 | tee gpurun_out/${TAG}_cpu_probe.log
        timeout 900 python -m oracle.cpu_baseline --n 16384 --d 32 2>&1 | tee gpurun_out/${TAG}_cpu_baseline.json ;;
    *) echo "unknown stage $s"; return 2 ;;
    esac
}
args=()
for a in "$@"; do
    if [ "$a" == "--" ]; then stage "${args[@]}"; args=(); else args+=("$a"); fi
done
[ ${#args[@]} -gt 0 ] && stage "${args[@]}"
exit 0
