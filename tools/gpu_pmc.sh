#!/bin/bash
# PMC passes for the trailing-update kernel (HBM/fabric bytes, L2 hit rate, MFMA busy), each in its own run with
# --kernel-trace only (gpurun refuses --pmc together with the other trace domains).  Writes the raw sqlite summaries to
# gpurun_out/pmc_<i>/ and the per-launch JSON to gpurun_out/r02_pmc_update_kernel.json (copy it to profiles/).
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
i=0; dbs=""
for CTRS in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F64"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $CTRS -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$i" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --sweep-batch 2 > /dev/null 2>&1)
  db=$(find "$GRAFT_REPO_ROOT/gpurun_out/pmc_$i" -name "*_results.db" | head -1)
  dbs="$dbs $db"
done
python tools/pmc_update_kernel.py gpurun_out/r02_pmc_update_kernel.json 16384 $dbs
# keep the merge small: the databases stay on the box
rm -rf gpurun_out/pmc_1 gpurun_out/pmc_2 gpurun_out/pmc_3
