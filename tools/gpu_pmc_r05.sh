#!/bin/bash
# Round-5 PMC passes (counters only beside --kernel-trace; every counter group in its own run, as gpurun requires):
#   the launch bench.py's `roofline` describes (ONE lock-step group of eight, left-looking long update)
#       -> gpurun_out/r05_pmc_lockstep_group_left_looking_pass<i>.json + _summary.json (tools/pmc_group_summary.py)
#   the chain launch k_potrf_pipe of a lone n = 4096 fit (whole factorisation as one launch)
#       -> gpurun_out/r05_pmc_chain_launch_n4096_pass<i>.json
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
pass() {  # pass <dir> <counters...> -- <command...>
  local dir=$1; shift; local ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "${ctrs[@]}" -d "$GRAFT_REPO_ROOT/gpurun_out/$dir" -o pmc -- "$@" > "$GRAFT_REPO_ROOT/gpurun_out/$dir.log" 2>&1)
  find "$GRAFT_REPO_ROOT/gpurun_out/$dir" -name "*_results.db" | head -1
}
i=0
for CTRS in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F64"; do
  i=$((i+1))
  db=$(pass pmc5_g_$i $CTRS -- python "$GRAFT_REPO_ROOT/tools/group_roofline.py" 16384 32 8 3)
  [ -n "$db" ] && python tools/pmc_kernels.py gpurun_out/r05_pmc_lockstep_group_left_looking_pass$i.json "$db" "rocprofv3 --kernel-trace --pmc $CTRS -- python tools/group_roofline.py 16384 32 8 3 (per-dispatch averages)" > /dev/null || { echo "group pass $i: no database"; tail -3 gpurun_out/pmc5_g_$i.log; }
  db=$(pass pmc5_c_$i $CTRS -- python "$GRAFT_REPO_ROOT/tools/one_fit.py" 4096 8 6 0)
  [ -n "$db" ] && python tools/pmc_kernels.py gpurun_out/r05_pmc_chain_launch_n4096_pass$i.json "$db" "rocprofv3 --kernel-trace --pmc $CTRS -- python tools/one_fit.py 4096 8 6 0 (one fit in flight, the whole factorisation one k_potrf_pipe launch; per-dispatch averages)" > /dev/null || { echo "chain pass $i: no database"; tail -3 gpurun_out/pmc5_c_$i.log; }
done
python tools/pmc_group_summary.py gpurun_out/r05_pmc_lockstep_group_left_looking_summary.json gpurun_out/r05_pmc_lockstep_group_left_looking_pass1.json gpurun_out/r05_pmc_lockstep_group_left_looking_pass2.json gpurun_out/r05_pmc_lockstep_group_left_looking_pass3.json
rm -rf gpurun_out/pmc5_*
ls -la gpurun_out/r05_pmc_*
