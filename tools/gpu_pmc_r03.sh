#!/bin/bash
# Round-3 PMC passes (counters only, with --kernel-trace: gpurun refuses --pmc beside the other trace domains), each
# counter group in its own run of `tools/one_fit.py 16384 32 3 [corr]` (lone fixed-theta fits, one in flight):
#   update kernel   FETCH_SIZE | WRITE_SIZE + L2 hits/misses | MFMA busy         -> r03_pmc_update_kernel.json
#   K1 k_corr_sym   VALU instruction / busy counters (is it VALU-issue bound?)   -> r03_pmc_per_kernel_<i>.json
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
CORR=${1:-0}
i=0; dbs=""
for CTRS in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F64" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $CTRS -d "$GRAFT_REPO_ROOT/gpurun_out/pmc3_$i" -o pmc -- python "$GRAFT_REPO_ROOT/tools/one_fit.py" 16384 32 3 $CORR > "$GRAFT_REPO_ROOT/gpurun_out/pmc3_$i.log" 2>&1)
  db=$(find "$GRAFT_REPO_ROOT/gpurun_out/pmc3_$i" -name "*_results.db" | head -1)
  if [ -n "$db" ]; then
    python tools/pmc_chain_kernels.py gpurun_out/r03_pmc_per_kernel_corr${CORR}_$i.json "$db" > /dev/null
    [ $i -le 3 ] && dbs="$dbs $db"
  else
    echo "pass $i ($CTRS) produced no database"; tail -3 gpurun_out/pmc3_$i.log
  fi
done
[ "$CORR" == "0" ] && python tools/pmc_update_kernel.py gpurun_out/r03_pmc_update_kernel.json 16384 $dbs > /dev/null
rm -rf gpurun_out/pmc3_*
ls -la gpurun_out/r03_pmc_*
