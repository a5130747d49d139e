#!/bin/bash
# PMC pass over the predict-side kernels (k_predict_mean, k_cross_corr, k_gemm_stream of the solves): VALU counters only, with
# --kernel-trace (gpurun refuses --pmc beside the other trace domains).  -> gpurun_out/r03_pmc_predict_kernels.json
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES -d "$GRAFT_REPO_ROOT/gpurun_out/pmcp" -o pmc -- python "$GRAFT_REPO_ROOT/tools/predict_only.py" > "$GRAFT_REPO_ROOT/gpurun_out/pmcp.log" 2>&1)
db=$(find "$GRAFT_REPO_ROOT/gpurun_out/pmcp" -name "*_results.db" | head -1)
python - "$db" <<'PY'
import json, sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
q = """select s.kernel_name, p.name, e.value, d.event_id from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
       join rocpd_kernel_dispatch d on e.event_id = d.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id"""
acc = {}
for name, ctr, val, ev in cur.execute(q):
    key = next((k for k in ("k_predict_mean_srow", "k_predict_mean", "k_normalize_queries", "k_cross_corr", "k_row_reduce") if k in name), None)
    if key is None:
        continue
    a = acc.setdefault(key, {"dispatches": set()})
    a[ctr] = a.get(ctr, 0.0) + val
    a["dispatches"].add(ev)
res = {}
for k, a in acc.items():
    n = len(a.pop("dispatches"))
    r = {"dispatches": n, **{c: v / n for c, v in a.items()}}
    if "SQ_INSTS_VALU" in a and "GRBM_GUI_ACTIVE" in a:
        # issue slots = 256 CUs x 4 SIMDs x (cycles per XCD) / 4 cycles per FP64 wave instruction; GRBM_GUI_ACTIVE is summed over 8 XCDs
        r["valu_issue_utilisation"] = a["SQ_INSTS_VALU"] / (256 * 4 * (a["GRBM_GUI_ACTIVE"] / 8.0) / 4.0)
    res[k] = r
json.dump({"source": "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU ... -- python tools/predict_only.py (n = 8192, d = 16, m = 100000; per-dispatch averages)",
           "kernels": res}, open("gpurun_out/r03_pmc_predict_kernels.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf gpurun_out/pmcp
