#!/bin/bash
# Round-5 closing measurements on one MI355X: GPU test suite, the default bench line, rocprofv3 kernel stats of the same command,
# the roofline leg's own command, the size-ladder legs and the expert group (all under gpurun_out/final/).
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/final; export TMPDIR=/tmp
O=gpurun_out/final
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gpu_tests.log 2>&1
tail -4 $O/gpu_tests.log
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
grep -c '^{' $O/bench_default.json
prof() {  # name command...
  local name=$1; shift
  rm -rf /tmp/prof_$name
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o $name -- "$@" ) > $O/${name}_run.txt 2>&1
  local db=$(find /tmp/prof_$name -name "*_results.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_stats.py $db > $O/${name}_kernel_stats.txt 2>&1
  echo $db
}
prof bench_default python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > /dev/null
grep '^{' $O/bench_default_run.txt > $O/bench_default_under_rocprof.json
prof group_roofline python $GRAFT_REPO_ROOT/tools/group_roofline.py 16384 32 8 4 > /dev/null
for leg in "n4096_one_fit 4096 8 6 0 1" "n4096_lockstep12 4096 8 5 0 12" "n8192_one_fit 8192 16 5 0 1" "n8192_lockstep12 8192 16 4 0 12" "n16384_one_fit 16384 32 4 0 1" "n2048_one_fit 2048 8 6 0 1" "n6144_one_fit 6144 16 5 0 1"; do
  set -- $leg; name=$1; shift
  db=$(prof $name python $GRAFT_REPO_ROOT/tools/one_fit.py "$@")
  [ -n "$db" ] && python tools/timeline.py $db $O/${name}_timeline.txt >> $O/${name}_run.txt 2>&1
done
prof expert_group python $GRAFT_REPO_ROOT/tools/expert_group.py 8 8192 16 5 > /dev/null
grep -E "^finalize|^likelihood" $O/expert_group_run.txt | tail -3
ls $O | wc -l
