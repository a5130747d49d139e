#!/bin/bash
# round 4, closing run on the final tree: whole GPU suite, smoke, default bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/r04_final_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r04_final_smoke.log 2>&1
timeout 900 python bench.py > $O/r04_final_bench_default.json 2> $O/r04_final_bench_default.err
tail -3 $O/r04_final_gpu_tests.log; tail -1 $O/r04_final_smoke.log
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r04_final_bench_default.json").read().strip().splitlines()[-1])
print("value", r["value"], "roofline", r["roofline"]["frac"], "single", r["roofline_single_matrix"]["frac"], "alone", (r.get("roofline_kernel_alone") or {}).get("frac"), (r.get("roofline_kernel_alone") or {}).get("error"))
oc = r["other_configs"]
print({k: (v.get("error") if isinstance(v, dict) and "error" in v else "ok") for k, v in oc.items()})
PY
