cd /tmp && export TMPDIR=/tmp
python $GRAFT_REPO_ROOT/tools/lab/tiny_predict_var.py 2>&1 | grep -v amdgpu | tee $GRAFT_REPO_ROOT/gpurun_out/tiny_pv.txt
timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/tpv -o tpv -- python $GRAFT_REPO_ROOT/tools/lab/tiny_predict_var.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find gpurun_out/tpv -name "*_results.db" | head -1) 2>&1 | head -30 | tee -a gpurun_out/tiny_pv.txt
rm -rf gpurun_out/tpv
