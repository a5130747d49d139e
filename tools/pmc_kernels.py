#!/usr/bin/env python3
"""Per-kernel PMC sums from one rocprofv3 --kernel-trace --pmc pass (rocpd database), kernels keyed by their short name
plus the template flags that distinguish their roles:   python tools/pmc_kernels.py out.json results.db "source note"
FETCH_SIZE / WRITE_SIZE are in units of 1 KB (raw, before the guide's gfx950 16 B/lane correction)."""
import json
import re
import sqlite3
import sys

out, db = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ""
cur = sqlite3.connect(db).cursor()
q = """select s.kernel_name, p.name, e.value, d.event_id, d.grid_size_x, d.workgroup_size_x, d.grid_size_z
       from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
       join rocpd_kernel_dispatch d on e.event_id = d.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id"""


def short(nm, gx, wx):
    m = re.search(r"egx\d+(k_[a-z0-9_]+)", nm)
    base = m.group(1) if m else nm[:40]
    if base == "k_gemm_stream":
        t = re.search(r"k_gemm_streamILb([01])ELb([01])ELi(\d)E", nm)
        if t:
            base += "<lower>" if t.group(1) == "1" and t.group(2) == "0" else ("<ktri>" if t.group(2) == "1" else "<rect>")
            base += {"1": "<left-long>", "2": "<left-short>"}.get(t.group(3), "")
        base += ">=512wg" if gx // max(1, wx) >= 512 else "<512wg"
    elif base == "k_gemm_nt_sub":
        t = re.search(r"ILb([01])ELi(\d+)ELi(\d+)", nm)
        if t:
            base += f"<{t.group(2)}x{t.group(3)}>"
    elif base == "k_panel_trsm16":
        t = re.search(r"ILi(\d)ELb([01])E", nm)
        if t:
            base += "<post>" if t.group(2) == "1" else "<factor>"
    return base


acc = {}
for name, ctr, val, ev, gx, wx, gz in cur.execute(q):
    key = short(name, gx or 0, wx or 1)
    a = acc.setdefault(key, {"dispatches": set()})
    a[ctr] = a.get(ctr, 0.0) + val
    a["dispatches"].add(ev)
res = {}
for k, a in sorted(acc.items()):
    n = len(a.pop("dispatches"))
    r = {"dispatches": n, **{c + "_per_dispatch": v / n for c, v in a.items()}}
    if "TCC_HIT_sum" in a and "TCC_MISS_sum" in a and a["TCC_HIT_sum"] + a["TCC_MISS_sum"] > 0:
        r["l2_hit_rate"] = a["TCC_HIT_sum"] / (a["TCC_HIT_sum"] + a["TCC_MISS_sum"])
    if "SQ_VALU_MFMA_BUSY_CYCLES" in a and a.get("SQ_BUSY_CYCLES"):
        r["mfma_busy_over_sq_busy"] = a["SQ_VALU_MFMA_BUSY_CYCLES"] / a["SQ_BUSY_CYCLES"]
    if "SQ_ACTIVE_INST_VALU" in a and a.get("SQ_BUSY_CYCLES"):
        r["valu_active_over_sq_busy"] = a["SQ_ACTIVE_INST_VALU"] / a["SQ_BUSY_CYCLES"]
    if "SQ_INSTS_VALU" in a and a.get("SQ_WAVES"):
        r["valu_insts_per_wave"] = a["SQ_INSTS_VALU"] / a["SQ_WAVES"]
    res[k] = r
json.dump({"source": note, "kernels": res}, open(out, "w"), indent=1)
print(json.dumps({k: {c: (round(v, 4) if isinstance(v, float) else v) for c, v in r.items()} for k, r in res.items()}, indent=1))
