#!/bin/bash
# round 5, GPU call 4: the chain kernel inside the library -- parity subset, then A/B against the separate-launch chain
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_diag_block.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -8 > $O/r05c4_tests.log
timeout 200 python tools/ab_small.py "pipe=0" "pipe=1,pipe_whole=0" "pipe=1,pipe_whole=4096" > $O/r05c4_ab_n4096.log 2>&1
timeout 200 python tools/ab_small.py --n 8192 --d 16 "pipe=0" "pipe=1" > $O/r05c4_ab_n8192.log 2>&1
timeout 200 python tools/ab_small.py --n 2048 --d 8 "pipe=0" "pipe=1" > $O/r05c4_ab_n2048.log 2>&1
timeout 400 python tools/ab_knobs.py --no-group --in-flight 16 --lockstep 8 --rounds 2 "pipe=0" "pipe=1,pipe_wgs=0" "pipe=1,pipe_wgs=64" "pipe=1,pipe_wgs=128" > $O/r05c4_ab_n16384.log 2>&1
cat $O/r05c4_tests.log $O/r05c4_ab_n4096.log $O/r05c4_ab_n8192.log $O/r05c4_ab_n2048.log $O/r05c4_ab_n16384.log
