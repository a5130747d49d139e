#!/bin/bash
# PMC passes for the trailing-update kernel: HBM/fabric bytes, L2 hit rate, MFMA busy.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
LOG=gpurun_out/pmc.log
: > $LOG
(rocprofv3 -L 2>/dev/null | grep -E "^\s*(Name|name)|TCC_HIT|TCC_MISS|FETCH_SIZE|WRITE_SIZE|MFMA|GRBM_GUI|TCC_EA0_RDREQ|TCC_REQ|TCP_TCC" | head -60) >> $LOG 2>&1
i=0
for CTRS in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F64"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $CTRS -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$i" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep -v "simple_timer\|generateRocpd\|^{" | tail -3 >> "$GRAFT_REPO_ROOT/$LOG")
  python - "$GRAFT_REPO_ROOT/gpurun_out/pmc_$i/pmc_results.db" >> $LOG 2>&1 <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
try:
    q = """select s.kernel_name, p.name, count(*), sum(e.value), avg(e.value)
           from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
           join rocpd_kernel_dispatch d on e.event_id = d.event_id
           join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           group by s.kernel_name, p.name order by 4 desc"""
    for r in cur.execute(q):
        print("%-60s %-34s n=%4d sum=%.6g avg=%.6g" % (r[0][:60], r[1], r[2], r[3], r[4]))
except Exception as ex:
    print("query failed", ex)
    print([r[1] for r in cur.execute("pragma table_info(rocpd_pmc_event)")])
    print([r[1] for r in cur.execute("pragma table_info(rocpd_info_pmc)")])
PY
done
cat $LOG | cut -c1-220
