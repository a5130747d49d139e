#!/usr/bin/env python3
"""Summarise the separate rocprofv3 --pmc passes of `bench.py --steps 1` for the chip-filling trailing-update launches
(k_gemm_stream / k_gemm_nt_sub<128x256> with >= 512 tiles) into profiles/r02_pmc_update_kernel.json, the file bench.py's
`roofline.traffic` is read from.  FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1 KB by rocprofv3 (round 1:
1.144e8 "KB" over 140 launches); raw per-launch BYTES are stored here, the gfx950 16 B/lane correction is applied by
bench.py and stated there."""
import json
import sqlite3
import sys


def rows(db):
    con = sqlite3.connect(db)
    cur = con.cursor()
    q = """select s.kernel_name, p.name, e.value, d.grid_size_x, d.workgroup_size_x, d.id
           from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
           join rocpd_kernel_dispatch d on e.event_id = d.event_id
           join rocpd_info_kernel_symbol s on d.kernel_id = s.id"""
    try:
        return list(cur.execute(q))
    except sqlite3.OperationalError:
        q = q.replace("d.grid_size_x, d.workgroup_size_x, d.id", "0, 1, d.event_id")
        return list(cur.execute(q))


def corr_sym_summary(dbs, n, d=32):
    """K1 (k_corr_sym): bytes per launch from the same passes; algorithmic = 8 n d + 8 n (n + 1) / 2."""
    acc, nd = {}, {}
    for db in dbs:
        for name, ctr, val, gx, wx, did in rows(db):
            if "k_corr_sym" not in name:
                continue
            acc[ctr] = acc.get(ctr, 0.0) + val
            nd.setdefault(ctr, set()).add((db, did))
    res = {"algorithmic_bytes_per_launch": 8.0 * n * d + 8.0 * n * (n + 1) / 2}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        if c in acc:
            res[c.lower() + "_bytes_per_launch_raw"] = acc[c] * 1000.0 / max(1, len(nd[c]))
    if "SQ_BUSY_CYCLES" in acc and "GRBM_GUI_ACTIVE" in acc:
        res["launches"] = len(nd["GRBM_GUI_ACTIVE"])
    return res


def main(out, n, dbs):
    agg = {}       # counter -> (sum, dispatches)
    per_xcd = {}   # counter -> values count
    ndisp = {}
    for db in dbs:
        for name, ctr, val, gx, wx, did in rows(db):
            if "k_gemm_stream" not in name and "k_gemm_nt_subILb1ELi128ELi256" not in name:
                continue
            if gx and wx and gx // max(1, wx) < 512:
                continue  # not a chip-filling launch
            s, _ = agg.get(ctr, (0.0, 0))
            agg[ctr] = (s + val, 0)
            per_xcd[ctr] = per_xcd.get(ctr, 0) + 1
            ndisp.setdefault(ctr, set()).add((db, did))
    res = {"source": "rocprofv3 --kernel-trace --pmc (separate passes) -- python tools/one_fit.py 16384 32 3 0 (lone fits, one in flight; "
                     "tools/gpu_pmc.sh lone; EGX_STREAM_WALK selected the tile walk while that code existed)",
           "kernel": "k_gemm_stream<LOWER> (>= 512 tiles of 128x256)", "n": n}
    def per_launch(c):
        if c not in agg:
            return None
        return agg[c][0] / max(1, len(ndisp[c]))
    for c in agg:
        res[f"{c}_sum"] = agg[c][0]
        res[f"{c}_dispatches"] = len(ndisp[c])
    if per_launch("FETCH_SIZE") is not None:
        res["fetch_bytes_per_launch"] = per_launch("FETCH_SIZE") * 1000.0
    if per_launch("WRITE_SIZE") is not None:
        res["write_bytes_per_launch"] = per_launch("WRITE_SIZE") * 1000.0
    # algorithmic C tile read of the chip-filling launches of one fit (lower triangle, one launch per group), averaged
    gw = 1024 if n >= 14336 else 512  # launch_potrf: groups of four 256-wide panels from n_pad >= 14336
    ncs = [n - gw * (g + 2) for g in range(0, 64) if n - gw * (g + 2) > 0]
    # (the 128 appended right-hand-side rows ride along: nbx = (nc + 128) / 128 row tiles)
    ncs = [c for c in ncs if ((c + 128) // 128) * (c // 256) - (c // 256) * (c // 256 - 1) >= 512]
    res["c_read_bytes_per_launch"] = 8.0 * sum(c * (c + 1) / 2 + 128 * c for c in ncs) / max(1, len(ncs))
    res["launches_per_fit_counted"] = len(ncs)
    if "TCC_HIT_sum" in agg and "TCC_MISS_sum" in agg:
        res["l2_hit_rate"] = agg["TCC_HIT_sum"][0] / (agg["TCC_HIT_sum"][0] + agg["TCC_MISS_sum"][0])
    if "SQ_VALU_MFMA_BUSY_CYCLES" in agg and "GRBM_GUI_ACTIVE" in agg:
        gui_avg = agg["GRBM_GUI_ACTIVE"][0] / per_xcd["GRBM_GUI_ACTIVE"]  # per dispatch and XCD
        nd = len(ndisp["SQ_VALU_MFMA_BUSY_CYCLES"])
        res["mfma_busy_frac"] = agg["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (gui_avg * nd * 1024.0)
        res["grbm_gui_active_avg_per_dispatch"] = gui_avg
    if "SQ_INSTS_VALU_MFMA_MOPS_F64" in agg:
        res["mfma_flops_executed_per_launch"] = agg["SQ_INSTS_VALU_MFMA_MOPS_F64"][0] * 512.0 / max(1, len(ndisp["SQ_INSTS_VALU_MFMA_MOPS_F64"]))
    res["k_corr_sym"] = corr_sym_summary(dbs, n)
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), sys.argv[3:])
