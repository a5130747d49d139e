#!/bin/bash
# round 5, GPU call 6: start tickets (DIAG role by arrival), device-side wait for the look-ahead columns' update
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
timeout 300 tools/pipe_check 8192 2>&1 | grep -v "^PASS" > $O/r05c6_pipe_check.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_diag_block.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -8 > $O/r05c6_tests.log
timeout 200 python tools/ab_small.py --n 8192 --d 16 "pipe=0" "pipe=1" "pipe=1,pipe_shared_wgs=64" "pipe=1,pipe_shared_wgs=160" > $O/r05c6_ab_n8192.log 2>&1
timeout 400 python tools/ab_knobs.py --no-group --in-flight 16 --lockstep 8 --rounds 2 "pipe=0" "pipe=1" "pipe=1,pipe_shared_wgs=64" "pipe=1,pipe_shared_wgs=160" > $O/r05c6_ab_n16384.log 2>&1
cat $O/r05c6_pipe_check.txt $O/r05c6_tests.log $O/r05c6_ab_n8192.log $O/r05c6_ab_n16384.log
