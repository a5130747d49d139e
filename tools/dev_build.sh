#!/bin/bash
# A/B builds: the library with -DEGX_DEV_KNOBS (environment overrides of schedule constants, schedule.h / kernels_pipe.hip) of the library into egobox_amd/lib/_dev/libegx_gp_hip.so (objects in /tmp)
set -e
cd /root/repo/egobox_amd/csrc
mkdir -p /tmp/devobj ../lib/_dev
for f in kernels_chol kernels_pipe kernels_corr gp_host gp_predict gp_fit sgp_host sweep; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DEGX_DEV_KNOBS -Wno-unused-value -Wno-unused-result -c $f.hip -o /tmp/devobj/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/devobj/*.o -ldl -o ../lib/_dev/libegx_gp_hip.so
ls -la ../lib/_dev/
