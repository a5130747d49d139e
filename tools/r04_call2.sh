#!/bin/bash
# round 4, GPU call 2: C^-T riding along the factorisation + prescaled trace kernel; XCD walk A/B; whole suite; default bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests -m gpu -q -x -k "gradient or grad or lbfgs" 2>&1 | tail -15 > $O/r04c2_grad_tests.log
timeout 300 python tools/grad_bench.py 16384 32 3 8 > $O/r04c2_grad_bench.log 2>&1
timeout 400 python tools/ab_knobs.py stream_walk 0 1 --rounds 3 > $O/r04c2_ab_stream_walk.log 2>&1
EGX_STREAM_WALK=1 timeout 600 python -m pytest tests -m gpu -q -k "lockstep or potrf or predict_var or gradient" 2>&1 | tail -6 > $O/r04c2_walk1_tests.log
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/r04c2_all_tests.log
timeout 900 python bench.py > $O/r04c2_bench_default.json 2> $O/r04c2_bench_default.err
tail -3 $O/r04c2_grad_tests.log; cat $O/r04c2_grad_bench.log; cat $O/r04c2_ab_stream_walk.log; tail -3 $O/r04c2_walk1_tests.log; tail -8 $O/r04c2_all_tests.log; tail -5 $O/r04c2_bench_default.err
python - <<'PY'
import json
try:
    r = json.loads(open("gpurun_out/r04c2_bench_default.json").read().strip().splitlines()[-1])
    print("value", r["value"], "roofline", {k: r["roofline"].get(k) for k in ("achieved", "frac", "launch_shape", "launch_ms_avg")})
    print("single", r["roofline_single_matrix"]["frac"], "alone", (r.get("roofline_kernel_alone") or {}).get("frac"))
    print("yardstick", r.get("vendor_yardstick"))
    oc = r.get("other_configs", {})
    print("config3", {k: oc.get("config3_matern52_n16384_d32", {}).get(k) for k in ("likelihood_plus_theta_gradient_ms", "gradient_roofline_lockstep_batch_of_8")})
    print("config4", oc.get("config4_sweep_512"))
    cb = r.get("cpu_baseline", {})
    print("cpu", cb.get("value"), cb.get("dpotrf_gflops"), cb.get("dpotrf_frac_of_host_fp64_peak"), cb.get("thread_settings_tried"))
except Exception as e:
    print("bench parse failed:", e)
PY
