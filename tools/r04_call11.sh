#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out/r04_call11.txt
{
timeout 400 python tools/ab_small.py "lur_side_min=0,stream_cap=0" "lur_side_min=4096,stream_cap=0" "lur_side_min=6144,stream_cap=0" "lur_side_min=8192,stream_cap=0" "lur_side_min=4096,stream_cap=240" "lur_side_min=6144,stream_cap=240" "lur_side_min=6144,stream_cap=224" "lur_side_min=8192,stream_cap=240" --n 16384 --d 32 --rounds 2
} > $O 2>&1
cat $O
