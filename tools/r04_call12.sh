#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in 0 240; do
  rm -rf $R/gpurun_out/tl_cap$C
  (cd /tmp && EGX_STREAM_CAP=$C timeout 300 rocprofv3 --kernel-trace -d "$R/gpurun_out/tl_cap$C" -o tl -- python "$R/tools/one_fit.py" 16384 32 3 > "$R/gpurun_out/r04_call12_cap$C.log" 2>&1)
  python tools/timeline.py "$(find gpurun_out/tl_cap$C -name '*_results.db' | head -1)" gpurun_out/r04_call12_timeline_cap$C.txt
  rm -rf $R/gpurun_out/tl_cap$C
  grep "fit 2" gpurun_out/r04_call12_cap$C.log | cut -c1-120
done
