#!/usr/bin/env python3
"""Per-kernel summary (count / total / avg / min / max) from a rocprofv3 rocpd sqlite database
(`rocprofv3 --kernel-trace --stats` writes <name>_results.db on ROCm 7.2)."""
import sqlite3
import sys


def main(path, per_fit=1):
    con = sqlite3.connect(path)
    cur = con.cursor()
    q = """select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3,
                  min(d.end-d.start)/1e3, max(d.end-d.start)/1e3
           from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           group by s.kernel_name order by 3 desc"""
    rows = list(cur.execute(q))
    tot = sum(r[2] for r in rows)
    print(f"# {path}: total kernel time {tot:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    print(f"{'kernel':<72} {'calls':>6} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>10} {'pct':>6}")
    for r in rows:
        print(f"{r[0][:72]:<72} {r[1]:>6} {r[2]:>10.3f} {r[3]:>10.1f} {r[4]:>9.1f} {r[5]:>10.1f} {100 * r[2] / tot:>6.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
