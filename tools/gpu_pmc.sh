#!/bin/bash
# ONE driver for every PMC measurement committed under profiles/ (counters only beside --kernel-trace, every counter group in a
# run of its own, as the MI355X guide's HBM / rocprofv3 section prescribes and gpurun requires).  Usage, on the GPU box:
#     tools/gpu_pmc.sh <target> [round-tag]
# writes gpurun_out/<tag>_pmc_<target>_pass<i>.json (per-dispatch averages of every kernel: tools/pmc_kernels.py) and, where a
# summariser exists for the target, its summary.  Targets (what each one profiles, the round that first committed it):
#   group    ONE lock-step group of eight, left-looking long update -- the launch bench.py's `roofline` describes (r04, r05)
#            -> + <tag>_pmc_lockstep_group_left_looking_summary.json (tools/pmc_group_summary.py; bench.py attaches it as `traffic`)
#   chain    the whole-factorisation chain launch k_potrf_pipe of a lone n = 4096 fit (r05)
#   flow     the flow launch k_potrf_flow of a lone n = 8192 fit (r06)
#   lone     the update kernels of a lone n = 16384 fit (r02, r03, r04) -> + <tag>_pmc_update_kernel.json (tools/pmc_update_kernel.py)
#   grad     the theta-gradient at config 3, one candidate (r04)
#   predict  the predict-side kernels at config 5's expert (r03): VALU issue counters
#   k1       the correlation build: VALU issue counters of a lone n = 16384 fit, arg 3 = kernel id (r03)
cd "$GRAFT_REPO_ROOT" || exit 1; mkdir -p gpurun_out; export TMPDIR=/tmp
TARGET=${1:?target}; TAG=${2:-r06}
MEM=("FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F64")
VALU=("SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES")
LDS=("SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS")
case "$TARGET" in
  group)   CMD=(python tools/group_roofline.py 16384 32 8 3); CTRSETS=("${MEM[@]}");;
  chain)   CMD=(python tools/one_fit.py 4096 8 6 0);          CTRSETS=("${MEM[@]}" "${LDS[@]}");;
  flow)    CMD=(python tools/one_fit.py 8192 16 6 0);         CTRSETS=("${MEM[@]}" "${LDS[@]}");;
  lone)    CMD=(python tools/one_fit.py 16384 32 3 0);        CTRSETS=("${MEM[@]}");;
  grad)    CMD=(python tools/one_grad.py 16384 32 3 3);       CTRSETS=("${MEM[@]}" "${VALU[@]}");;
  predict) CMD=(python tools/predict_only.py);                CTRSETS=("${VALU[@]}");;
  k1)      CMD=(python tools/one_fit.py 16384 32 3 "${3:-0}"); CTRSETS=("${VALU[@]}");;
  *) echo "unknown target $TARGET"; exit 2;;
esac
i=0; dbs=""
for CTRS in "${CTRSETS[@]}"; do
  i=$((i+1)); dir="gpurun_out/pmc_${TARGET}_$i"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $CTRS -d "$GRAFT_REPO_ROOT/$dir" -o pmc -- "${CMD[0]}" "$GRAFT_REPO_ROOT/${CMD[1]}" "${CMD[@]:2}" > "$GRAFT_REPO_ROOT/$dir.log" 2>&1)
  db=$(find "$dir" -name "*_results.db" | head -1)
  if [ -n "$db" ]; then
    python tools/pmc_kernels.py "gpurun_out/${TAG}_pmc_${TARGET}_pass$i.json" "$db" "rocprofv3 --kernel-trace --pmc $CTRS -- ${CMD[*]} (per-dispatch averages)" > /dev/null
    dbs="$dbs $db"
  else
    echo "$TARGET pass $i ($CTRS): no database"; tail -3 "$dir.log"
  fi
done
case "$TARGET" in
  group) python tools/pmc_group_summary.py "gpurun_out/${TAG}_pmc_lockstep_group_left_looking_summary.json" gpurun_out/${TAG}_pmc_group_pass1.json gpurun_out/${TAG}_pmc_group_pass2.json gpurun_out/${TAG}_pmc_group_pass3.json;;
  lone)  python tools/pmc_update_kernel.py "gpurun_out/${TAG}_pmc_update_kernel.json" 16384 $dbs > /dev/null;;
esac
rm -rf gpurun_out/pmc_${TARGET}_*
ls -la gpurun_out/${TAG}_pmc_*
