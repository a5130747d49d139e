cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/tl_expert -o tl -- python $GRAFT_REPO_ROOT/tools/expert_group.py 8 8192 16 2 > $GRAFT_REPO_ROOT/gpurun_out/tl_expert.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find gpurun_out/tl_expert -name "*_results.db" | head -1)
python tools/timeline.py $db gpurun_out/tl_expert_group.txt
rm -rf gpurun_out/tl_expert
tail -5 gpurun_out/tl_expert.log
