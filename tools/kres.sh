#!/bin/bash
# kernel resource summary of one HIP translation unit: tools/kres.sh <file.hip> [name filter]
cd "$(dirname "$0")/../egobox_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -c "$1" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 |
  awk -v pat="${2:-.}" '/Function Name:/ {name=$5} /VGPRs:/ {v=$4} /AGPRs:/ {a=$4} /ScratchSize/ {s=$5} /Occupancy/ {o=$5} /VGPRs Spill/ {vs=$5} /LDS Size/ { if (name ~ pat) printf "%-90s vgpr %3s agpr %3s scratch %4s occ %s vspill %s lds %s\n", substr(name,1,90), v, a, s, o, vs, $6 }'
