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
    # The roofline of bench.py follows the launches of the trailing-update kernel that fill the chip on their own
    # (>= 512 workgroups of one 128x256 tile each, ONE matrix per launch): the same selection here, so that the average
    # duration below is the one its HIP events must agree with.  (Since round 3 the kernel also serves smaller launches.)
    try:
        q2 = """select count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, max(d.end-d.start)/1e3
                from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
                where s.kernel_name like '%k_gemm_streamILb1ELb0ELi0E%' and d.grid_size_x / d.workgroup_size_x >= 512
                      and d.grid_size_z <= 1"""
        c, t, a, lo, hi = list(cur.execute(q2))[0]
        if c:
            print(f"{'k_gemm_stream<LOWER>, launches of >= 512 workgroups, one matrix (the roofline set)':<72}"[:100])
            print(f"{'':<72} {c:>6} {t:>10.3f} {a:>10.1f} {lo:>9.1f} {hi:>10.1f} {100 * t / tot:>6.1f}")
        # ... and the same launches for a lock-step group (grid.z matrices per launch).  Not a kernel rate: inside a group the
        # rest-of-group update runs on a side stream and overlaps these launches (and with several groups in flight, so do the
        # other groups' kernels).
        q3 = """select d.grid_size_z, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3,
                       max(d.end-d.start)/1e3
                from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
                where s.kernel_name like '%k_gemm_streamILb1ELb0ELi0E%' and d.grid_size_x / d.workgroup_size_x >= 512
                      and d.grid_size_z > 1 group by d.grid_size_z"""
        for z, c, t, a, lo, hi in cur.execute(q3):
            print(f"k_gemm_stream<LOWER>, launches of >= 512 workgroups per matrix, {z} matrices per launch")
            print(f"{'':<72} {c:>6} {t:>10.3f} {a:>10.1f} {lo:>9.1f} {hi:>10.1f} {100 * t / tot:>6.1f}")
        # left-looking group updates (own kernel symbols, k_gemm_stream's TAG): the LONG update (K = all columns before the
        # previous group) is what bench.py's `roofline` times in a lock-step group; the short one has K = 1024
        for tag, what in ((1, "long update (bench.py roofline)"), (2, "short update, K = one group")):
            qt = f"""select d.grid_size_z, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3,
                            max(d.end-d.start)/1e3
                     from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
                     where s.kernel_name like '%k_gemm_streamILb1ELb0ELi{tag}E%' group by d.grid_size_z"""
            for z, c, t, a, lo, hi in cur.execute(qt):
                print(f"k_gemm_stream<LOWER>, left-looking {what}, {z} matrices per launch")
                print(f"{'':<72} {c:>6} {t:>10.3f} {a:>10.1f} {lo:>9.1f} {hi:>10.1f} {100 * t / tot:>6.1f}")
        # the theta-gradient's R^-1 = C^-T C^-1 launches (per-tile K ranges), by matrices per launch
        q4 = """select d.grid_size_z, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3,
                       max(d.end-d.start)/1e3
                from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
                where s.kernel_name like '%k_gemm_streamILb1ELb1E%' group by d.grid_size_z"""
        for z, c, t, a, lo, hi in cur.execute(q4):
            print(f"k_gemm_stream<LOWER, KTRI> (R^-1 = W W^T of the theta-gradient), {z} matrices per launch")
            print(f"{'':<72} {c:>6} {t:>10.3f} {a:>10.1f} {lo:>9.1f} {hi:>10.1f} {100 * t / tot:>6.1f}")
    except sqlite3.OperationalError as e:  # older rocpd schema without grid sizes
        print(f"# (no grid-size columns in this database: {e})")


if __name__ == "__main__":
    main(sys.argv[1])
