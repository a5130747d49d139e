#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
EGX_PIPE_TIMEOUT_MS=300 timeout 200 python tools/ab_small.py --n 8192 --d 16 --rounds 2 "pipe=1" > $O/r05c7_ab_n8192.log 2>&1
EGX_PIPE_TIMEOUT_MS=300 timeout 200 python tools/ab_small.py --n 8192 --d 16 --rounds 2 "pipe=1" >> $O/r05c7_ab_n8192.log 2>&1
cat $O/r05c7_ab_n8192.log
