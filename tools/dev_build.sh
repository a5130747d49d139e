#!/bin/bash
# Scratch builds of the library under egobox_amd/lib/_dev/ (objects in /tmp), never the product:
#   tools/dev_build.sh                 -DEGX_DEV_KNOBS    -> libegx_gp_hip.so        environment overrides of schedule constants (schedule.h / kernels_pipe.hip)
#   tools/dev_build.sh trace           -DEGX_STREAM_TRACE -> libegx_gp_hip_trace.so  per-tile stamps in k_gemm_stream (kernels_chol.hip): what
#                                       tools/long_update_attribution.py loads (EGX_TEST_LIBRARY=trace)
#   tools/dev_build.sh <name> <defs>   <defs>             -> libegx_gp_hip_<name>.so  any A/B variant (EGX_TEST_LIBRARY=<name>; tools/ab_lib.py)
set -e
cd "$(dirname "$0")/../egobox_amd/csrc"
MODE=${1:-knobs}
if [ "$MODE" == "trace" ]; then DEF=-DEGX_STREAM_TRACE; OUT=libegx_gp_hip_trace.so
elif [ "$MODE" == "knobs" ]; then DEF=-DEGX_DEV_KNOBS; OUT=libegx_gp_hip.so
else shift; DEF="$*"; OUT=libegx_gp_hip_$MODE.so; fi
OBJ=/tmp/devobj_$MODE
mkdir -p $OBJ ../lib/_dev
for f in kernels_chol kernels_pipe kernels_corr gp_host gp_predict gp_fit sgp_host sweep; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $DEF -Wno-unused-value -Wno-unused-result -c $f.hip -o $OBJ/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ/*.o -ldl -o ../lib/_dev/$OUT
ls -la ../lib/_dev/
