#!/bin/bash
# round 4, second closing run: whole GPU suite, smoke, and bench.py exactly as the driver calls it (20 steps, 5 warm-up) with its wall time
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/r04_close2_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r04_close2_smoke.log 2>&1
S=$(date +%s.%N)
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_close2_bench_driver_shape.json 2> $O/r04_close2_bench_driver_shape.err
E=$(date +%s.%N)
echo "bench.py --gpus 1 --steps 20 --warmup 5: wall $(python -c "print(round($E-$S,1))") s" | tee $O/r04_close2_bench_wall.txt
tail -3 $O/r04_close2_gpu_tests.log; tail -1 $O/r04_close2_smoke.log
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r04_close2_bench_driver_shape.json").read().strip().splitlines()[-1])
print("value", r["value"], "ms_per_step", r["ms_per_step"], "roofline", r["roofline"]["frac"], "single", r["roofline_single_matrix"]["frac"])
oc = r["other_configs"]
print({k: (v.get("error") if isinstance(v, dict) and "error" in v else "ok") for k, v in oc.items()})
PY
