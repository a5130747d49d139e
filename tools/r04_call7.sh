#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
{
timeout 300 python tools/ab_w_left.py 16384 32 3 1 0 2 0
timeout 300 python tools/ab_w_left.py 16384 32 3 8 4 2 0
timeout 300 python tools/ab_w_left.py 16384 32 3 8 8 2 1 0
timeout 300 python tools/ab_w_left.py 8192 16 3 1 0 2 0
timeout 300 python tools/ab_w_left.py 8192 16 0 12 12 2 0
timeout 300 python tools/ab_w_left.py 4096 8 0 1 0 2 0
timeout 300 python tools/ab_w_left.py 4096 8 0 12 12 2 0
} > $O/r04_call7_ab_w_left.txt 2>&1
cat $O/r04_call7_ab_w_left.txt
