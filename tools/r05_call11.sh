#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $O/r05c11_tests.log
timeout 300 tools/pipe_check 4096 2>&1 | grep -v "^PASS" > $O/r05c11_pipe_check.txt 2>&1
timeout 200 python tools/ab_small.py "pipe=0" "pipe=1" > $O/r05c11_ab_n4096.log 2>&1
cat $O/r05c11_tests.log $O/r05c11_pipe_check.txt $O/r05c11_ab_n4096.log
