#!/usr/bin/env python3
"""Per-kernel PMC sums of the serial path's kernels (k_potf2_reg, k_panel_trsm16, k_gemm_nt_sub 64x64) from one
rocprofv3 --kernel-trace --pmc pass of tools/one_fit.py (see tools/gpu_pmc.sh chain):
    python tools/pmc_chain_kernels.py out.json results.db"""
import json
import sqlite3
import sys

out, db = sys.argv[1], sys.argv[2]
cur = sqlite3.connect(db).cursor()
q = """select s.kernel_name, p.name, e.value, d.event_id from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
       join rocpd_kernel_dispatch d on e.event_id = d.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id"""
acc = {}
for name, ctr, val, ev in cur.execute(q):
    key = next((k for k in ("k_potf2_reg", "k_panel_trsm16", "k_gemm_nt_sub", "k_gemm_stream", "k_corr_sym") if k in name), None)
    if key is None:
        continue
    a = acc.setdefault(key, {"dispatches": set()})
    a[ctr] = a.get(ctr, 0.0) + val
    a["dispatches"].add(ev)
res = {}
for k, a in acc.items():
    n = len(a.pop("dispatches"))
    r = {"dispatches": n, **{c: v / n for c, v in a.items()}}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in a and "SQ_BUSY_CYCLES" in a and a["SQ_BUSY_CYCLES"]:
        r["mfma_busy_over_sq_busy"] = a["SQ_VALU_MFMA_BUSY_CYCLES"] / a["SQ_BUSY_CYCLES"]
    if "SQ_INSTS_VALU" in a and a.get("SQ_WAVES"):
        r["valu_insts_per_wave"] = a["SQ_INSTS_VALU"] / a["SQ_WAVES"]
    if "SQ_ACTIVE_INST_VALU" in a and a.get("SQ_WAVE_CYCLES"):
        r["valu_active_over_wave_cycles"] = a["SQ_ACTIVE_INST_VALU"] / a["SQ_WAVE_CYCLES"]
    if "SQ_ACTIVE_INST_VALU" in a and a.get("SQ_BUSY_CYCLES"):
        r["valu_active_over_sq_busy"] = a["SQ_ACTIVE_INST_VALU"] / a["SQ_BUSY_CYCLES"]
    if "SQ_LDS_BANK_CONFLICT" in a and a.get("SQ_LDS_IDX_ACTIVE"):
        r["lds_bank_conflict_frac"] = a["SQ_LDS_BANK_CONFLICT"] / a["SQ_LDS_IDX_ACTIVE"]
    res[k] = r
json.dump({"source": "rocprofv3 --kernel-trace --pmc -- python tools/one_fit.py 4096 8 5 (per-dispatch averages)", "kernels": res}, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
