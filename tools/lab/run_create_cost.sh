cd /tmp && export TMPDIR=/tmp
python $GRAFT_REPO_ROOT/tools/lab/create_group_cost.py 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/create_cost.txt
rocprofv3 --kernel-trace --memory-copy-trace --hip-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/cc -o cc -- python $GRAFT_REPO_ROOT/tools/lab/create_group_cost.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/cc -name "*hip_api_stats.csv" | head -1); echo "== hip api stats" >> gpurun_out/create_cost.txt; head -25 $f >> gpurun_out/create_cost.txt
f=$(find gpurun_out/cc -name "*memory_copy_stats.csv" | head -1); echo "== memcpy stats" >> gpurun_out/create_cost.txt; head -10 $f >> gpurun_out/create_cost.txt
rm -rf gpurun_out/cc
cat gpurun_out/create_cost.txt
