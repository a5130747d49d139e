#!/bin/bash
# round 5, GPU call 12: lock-step across models (groups), the rest of the suite
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_pipe.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -30 > $O/r05c12_tests.log
cat $O/r05c12_tests.log
