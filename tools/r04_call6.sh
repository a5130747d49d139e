#!/bin/bash
# round 4, GPU call 6: the whole suite on the final defaults, default bench + its kernel stats, PMC of the left-looking updates
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -30 > $O/r04c6_all_tests.log
timeout 900 python bench.py > $O/r04c6_bench_default.json 2> $O/r04c6_bench_default.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_b" -o prof -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-extra-configs > "$GRAFT_REPO_ROOT/$O/r04c6_bench_under_rocprof.json" 2>/dev/null)
python tools/rocpd_stats.py "$(find $O/prof_b -name '*_results.db' | head -1)" > $O/r04c6_bench_default_kernel_stats.txt 2>&1
rm -rf $O/prof_b
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_g" -o prof -- python "$GRAFT_REPO_ROOT/tools/group_roofline.py" > "$GRAFT_REPO_ROOT/$O/r04c6_group_roofline.log" 2>&1)
python tools/rocpd_stats.py "$(find $O/prof_g -name '*_results.db' | head -1)" > $O/r04c6_group_roofline_kernel_stats.txt 2>&1
rm -rf $O/prof_g
i=0
for CTRS in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F64"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $CTRS -d "$GRAFT_REPO_ROOT/$O/pmc6_$i" -o pmc -- python "$GRAFT_REPO_ROOT/tools/group_roofline.py" 16384 32 8 3 > "$GRAFT_REPO_ROOT/$O/pmc6_$i.log" 2>&1)
  db=$(find "$GRAFT_REPO_ROOT/$O/pmc6_$i" -name "*_results.db" | head -1)
  [ -n "$db" ] && python tools/pmc_kernels.py $O/r04_pmc_lockstep_group_left_looking_pass$i.json "$db" "rocprofv3 --kernel-trace --pmc $CTRS -- python tools/group_roofline.py 16384 32 8 3 (one lock-step group of eight, left-looking; per-dispatch averages, a dispatch = 8 matrices)" > /dev/null
done
rm -rf $O/pmc6_*
tail -6 $O/r04c6_all_tests.log; tail -3 $O/r04c6_bench_default.err; grep "^group" $O/r04c6_group_roofline.log
python - <<'PY'
import json
try:
    r = json.loads(open("gpurun_out/r04c6_bench_default.json").read().strip().splitlines()[-1])
    print("value", r["value"], "tflops/gpu", r["cholesky_tflops_per_gpu_in_timed_region"], "in flight", r["config"]["fits_in_flight_per_gpu"], "lockstep", r["config"]["lockstep_width"])
    print("roofline", {k: r["roofline"].get(k) for k in ("achieved", "frac", "launch_ms_avg", "launches_per_group", "share_of_potrf_flops")})
    print("single", r["roofline_single_matrix"]["frac"], "single fit", r["single_fit_in_flight_fits_per_s"], "group alone", r.get("lockstep_group_alone"))
    oc = r.get("other_configs", {})
    print("config2", oc.get("config2_sqexp_n4096_d8"))
    c3 = oc.get("config3_matern52_n16384_d32", {})
    print("config3", c3.get("likelihood_plus_theta_gradient_ms"), c3.get("gradient_roofline_lockstep_batch_of_8"))
    print("config5", oc.get("config5_expert_n8192_d16_m100000"))
    print("config4", oc.get("config4_sweep_512"))
    print("speedups", {k: v for k, v in r.items() if k.startswith("speedup")})
    print("pcie", r.get("pcie_inclusive", {}).get("fits_per_s_cold_handle"))
    ru = json.loads(open("gpurun_out/r04c6_bench_under_rocprof.json").read().strip().splitlines()[-1])
    print("under rocprof value", ru["value"], "roofline launch_ms_avg", ru["roofline"]["launch_ms_avg"])
except Exception as e:
    print("bench parse failed:", e)
for i in (1, 2, 3):
    try:
        k = json.load(open(f"gpurun_out/r04_pmc_lockstep_group_left_looking_pass{i}.json"))["kernels"]
        for name, v in k.items():
            if "left" in name:
                print(i, name, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
    except Exception as e:
        print("pmc", i, e)
PY
grep -A1 "left-looking" $O/r04c6_bench_default_kernel_stats.txt | cut -c1-150; grep -A1 "left-looking" $O/r04c6_group_roofline_kernel_stats.txt | cut -c1-150
