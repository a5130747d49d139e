#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
{
timeout 300 python tools/ab_w_left.py 16384 32 3 16 8 1 0
timeout 300 python tools/ab_w_left.py 16384 32 3 24 8 1
} > $O/r04_call8.txt 2>&1
cat $O/r04_call8.txt
