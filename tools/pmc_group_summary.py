#!/usr/bin/env python3
"""The `roofline.traffic` summary bench.py attaches (profiles/r0X_pmc_lockstep_group_left_looking_summary.json) from the three
per-pass files tools/pmc_kernels.py wrote for `python tools/group_roofline.py 16384 32 8 3`:
    python tools/pmc_group_summary.py out.json pass1.json pass2.json pass3.json
Corrections exactly as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE / WRITE_SIZE in KB; FETCH_SIZE
counts the 16 B/lane reads (the panel operands through global_load_lds) at half."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out, p1, p2, p3 = sys.argv[1:5]
n, d, gl, K_group = 16384, 32, 8, 1024
KEY = "k_gemm_stream<lower><left-long>"


def row(path):
    with open(path) as f:
        js = json.load(f)
    rows = [v for k, v in js["kernels"].items() if k.startswith(KEY)]
    tot = sum(r["dispatches"] for r in rows)
    acc = {}
    for r in rows:
        for k, v in r.items():
            if k.endswith("_per_dispatch"):
                acc[k] = acc.get(k, 0.0) + v * r["dispatches"] / tot
    acc["dispatches"] = tot
    return acc, js.get("source", "")


a1, src = row(p1)
a2, _ = row(p2)
a3, _ = row(p3)
# the 14 long updates of a factorisation: group J (columns g0 = 1024 J) receives K = g0 - 1024 earlier columns, J = 2 .. 15
launches = []
for J in range(2, n // K_group):
    g0, K = K_group * J, K_group * (J - 1)
    rows_ = n - g0
    c_elems = rows_ * K_group - K_group * (K_group - 1) // 2
    launches.append({"K": K, "c_bytes": 8 * c_elems, "operand_bytes": 8 * rows_ * K, "flops": 2.0 * K * c_elems})
c_read = gl * sum(l["c_bytes"] for l in launches) / len(launches)
alg = gl * sum(2 * l["c_bytes"] + l["operand_bytes"] for l in launches) / len(launches)
flops = gl * sum(l["flops"] for l in launches) / len(launches)
# (KB = 1000 B, as in tools/pmc_update_kernel.py: calibrated on the write side -- WRITE_SIZE x 1000 agrees with the algorithmic size
#  of the C regions to 1 %)
fetch = a1["FETCH_SIZE_per_dispatch"] * 1000.0
write = a2["WRITE_SIZE_per_dispatch"] * 1000.0
with open(os.path.join(ROOT, "egobox_amd", "csrc", "kernels_chol.hip")) as f:
    txt = f.read()
a = txt.rfind("template", 0, txt.index("void k_gemm_stream("))
sig = hashlib.sha256(txt[a:txt.index("\n}\n", a) + 3].encode()).hexdigest()
rec = {
    "source": f"{os.path.basename(p1)} .. {os.path.basename(p3)}: rocprofv3 --kernel-trace --pmc <one counter set per pass> -- python "
              f"tools/group_roofline.py 16384 32 8 3 (ONE lock-step group of eight in flight, left-looking; averages over the long-update "
              f"dispatches, a dispatch = 8 matrices); {src}",
    "kernel": "k_gemm_stream<LOWER, TAG 1>: the long left-looking group update, 8 matrices per launch",
    "dispatches": a1["dispatches"], "launches_per_factorisation": len(launches),
    "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "c_read_bytes_per_launch": c_read,
    "corrected_traffic_bytes_per_launch": write + c_read + 2.0 * (fetch - c_read),
    "correction": "gfx950 FETCH_SIZE counts 16 B/lane reads (the panel operands through global_load_lds) at half: traffic = WRITE + "
                  "C_read + 2 * (FETCH - C_read); C_read = the algorithmic size of the 8 C regions",
    "algorithmic_bytes_per_launch": alg,
    "algorithmic_note": "every row of the factor the launch contracts over read once (rows x K x 8 B per matrix) + the C region read and "
                        "written once; averages over the 14 launches of a factorisation",
    "algorithmic_flops_per_launch": flops,
    "l2_hit_rate": a2["TCC_HIT_sum_per_dispatch"] / (a2["TCC_HIT_sum_per_dispatch"] + a2["TCC_MISS_sum_per_dispatch"]),
    # MFMA-busy cycles summed over the chip's 1024 SIMDs / (GRBM_GUI_ACTIVE summed over the 8 XCDs / 8 x 1024)
    "mfma_busy_frac": a3["SQ_VALU_MFMA_BUSY_CYCLES_per_dispatch"] / (a3["GRBM_GUI_ACTIVE_per_dispatch"] / 8.0 * 1024.0)
    if "GRBM_GUI_ACTIVE_per_dispatch" in a3 else None,
    "mfma_flops_executed_per_launch": a3.get("SQ_INSTS_VALU_MFMA_MOPS_F64_per_dispatch", 0.0) * 512.0,
    "kernel_source_sha256": sig,
    "kernel_source_sha256_note": "sha256 of the source text of the k_gemm_stream template in egobox_amd/csrc/kernels_chol.hip: bench.py "
                                 "attaches these counters to its `roofline` only while the current source still has this hash and the live "
                                 "launch shape matches",
}
with open(out, "w") as f:
    json.dump(rec, f, indent=1)
print(json.dumps({k: rec[k] for k in ("fetch_bytes_per_launch", "write_bytes_per_launch", "corrected_traffic_bytes_per_launch",
                                      "algorithmic_bytes_per_launch", "algorithmic_flops_per_launch", "l2_hit_rate", "mfma_busy_frac")}))
