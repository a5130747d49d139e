#!/bin/bash
# LDS bank-conflict share (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE) kernel by kernel: separate launches at n = 4096 (EGX_PIPE=0), a lone n = 16384 fit
cd "$GRAFT_REPO_ROOT" || exit 1; export TMPDIR=/tmp
run() {  # tag, env, args...
  tag=$1; shift; envs=$1; shift
  (cd /tmp && env $envs timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -d "$GRAFT_REPO_ROOT/gpurun_out/lds_$tag" -o pmc -- python "$GRAFT_REPO_ROOT/tools/one_fit.py" "$@" > /dev/null 2>&1)
  db=$(find gpurun_out/lds_$tag -name "*_results.db" | head -1)
  python tools/pmc_kernels.py gpurun_out/lds_conflicts_$tag.json "$db" "rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -- $envs python tools/one_fit.py $*" > /dev/null
  rm -rf gpurun_out/lds_$tag
}
run sep4096 EGX_PIPE=0 4096 8 6 0
run lone16384 EGX_PIPE=1 16384 32 3 0
python - <<'PY'
import json
for t in ("sep4096","lone16384"):
    o=json.load(open(f"gpurun_out/lds_conflicts_{t}.json"))
    print("==",o["source"])
    for k,v in sorted(o["kernels"].items()):
        if v.get("SQ_LDS_IDX_ACTIVE_per_dispatch",0)>0:
            print(f"  {k:28s} dispatches {v['dispatches']:4d}  conflict cycles / LDS active cycles {v['SQ_LDS_BANK_CONFLICT_per_dispatch']/v['SQ_LDS_IDX_ACTIVE_per_dispatch']:.3f}  LDS instructions per dispatch {v['SQ_INSTS_LDS_per_dispatch']:.3g}")
PY
