#!/bin/bash
# PMC pass over the serial path's kernels in a lone n = 4096 fit (MFMA busy, LDS bank conflicts): counters only, with
# --kernel-trace (gpurun refuses --pmc together with the other trace domains).  -> gpurun_out/r02_pmc_chain_kernels*.json
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
i=0
for CTRS in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F64" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $CTRS -d "$GRAFT_REPO_ROOT/gpurun_out/pmcc_$i" -o pmc -- python "$GRAFT_REPO_ROOT/tools/one_fit.py" 4096 8 5 > /dev/null 2>&1)
  db=$(find "$GRAFT_REPO_ROOT/gpurun_out/pmcc_$i" -name "*_results.db" | head -1)
  [ -n "$db" ] && python tools/pmc_chain_kernels.py gpurun_out/r02_pmc_chain_kernels_$i.json "$db" | head -60
  rm -rf gpurun_out/pmcc_$i
done
