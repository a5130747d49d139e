#!/usr/bin/env python3
"""Dispatch timeline of the LAST fit in a rocprofv3 --kernel-trace rocpd database: one line per kernel dispatch
(start us, duration us, queue, grid, short name), relative to the fit's first dispatch (k_corr_sym)."""
import re
import sqlite3
import sys


def main(path, out):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    gcol = "grid_size_x" if "grid_size_x" in cols else None
    wcol = "workgroup_size_x" if "workgroup_size_x" in cols else None
    sel = "d.start, d.end, s.kernel_name" + (f", d.{qcol}" if qcol else ", 0") + (f", d.{gcol}" if gcol else ", 0") + \
          (f", d.{wcol}" if wcol else ", 1")
    rows = list(cur.execute(f"select {sel} from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                            "on d.kernel_id = s.id order by d.start"))
    starts = [i for i, r in enumerate(rows) if "k_corr_sym" in r[2]]
    if not starts:
        print("no k_corr_sym dispatch found; columns:", cols)
        return
    rows = rows[starts[-1]:]
    t0 = rows[0][0]

    def short(nm):
        m = re.search(r"egx\d+(k_[a-z0-9_]+)", nm)
        base = m.group(1) if m else nm[:40]
        t = re.search(r"ILb[01]ELi(\d+)ELi(\d+)", nm)
        if t:
            base += f"<{t.group(1)}x{t.group(2)}>"
        return base

    with open(out, "w") as f:
        f.write(f"# {path}: last fit, {len(rows)} dispatches; start_us dur_us queue wgs kernel\n")
        for st, en, nm, q, g, w in rows:
            f.write(f"{(st - t0) / 1e3:10.1f} {(en - st) / 1e3:9.1f} {q!s:>4} {int(g) // max(1, int(w)):6d} {short(nm)}\n")
    print(f"wrote {out}: {len(rows)} dispatches, span {(rows[-1][1] - t0) / 1e6:.3f} ms")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
