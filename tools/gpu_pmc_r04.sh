#!/bin/bash
# Round-4 PMC passes (counters only beside --kernel-trace; every counter group in its own run):
#   update kernel of a lone n = 16384 fit, column-major walk and -- while that code existed (commit 54a9160 .. its removal) -- the XCD
#   super-tile walk (EGX_STREAM_WALK=1; today the variable does nothing)
#       FETCH_SIZE | WRITE_SIZE + L2 hits / misses | MFMA busy     -> gpurun_out/r04_pmc_update_kernel[_walk1].json
#   the theta-gradient (tools/one_grad.py, config 3)                -> gpurun_out/r04_pmc_theta_gradient_pass<i>.json
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
pass() {  # pass <dir> <counters...> -- <command...>
  local dir=$1; shift; local ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "${ctrs[@]}" -d "$GRAFT_REPO_ROOT/gpurun_out/$dir" -o pmc -- "$@" > "$GRAFT_REPO_ROOT/gpurun_out/$dir.log" 2>&1)
  find "$GRAFT_REPO_ROOT/gpurun_out/$dir" -name "*_results.db" | head -1
}
for WALK in 0 1; do
  dbs=""; i=0
  for CTRS in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F64"; do
    i=$((i+1))
    db=$(EGX_STREAM_WALK=$WALK pass pmc4_w${WALK}_$i $CTRS -- python "$GRAFT_REPO_ROOT/tools/one_fit.py" 16384 32 3 0)
    [ -n "$db" ] && dbs="$dbs $db" || { echo "walk $WALK pass $i: no database"; tail -3 gpurun_out/pmc4_w${WALK}_$i.log; }
  done
  sfx=""; [ "$WALK" == "1" ] && sfx="_walk1"
  python tools/pmc_update_kernel.py gpurun_out/r04_pmc_update_kernel$sfx.json 16384 $dbs > /dev/null
done
i=0
for CTRS in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F64" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES"; do
  i=$((i+1))
  db=$(pass pmc4_g_$i $CTRS -- python "$GRAFT_REPO_ROOT/tools/one_grad.py" 16384 32 3 3)
  [ -n "$db" ] && python tools/pmc_kernels.py gpurun_out/r04_pmc_theta_gradient_pass$i.json "$db" "rocprofv3 --kernel-trace --pmc $CTRS -- python tools/one_grad.py 16384 32 3 3 (Matern-5/2, one candidate in flight; per-dispatch averages)" > /dev/null
done
rm -rf gpurun_out/pmc4_*
ls -la gpurun_out/r04_pmc_*
