#!/bin/bash
# round 5, GPU call 10: whole GPU suite with the chain launches in the library
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/r05c10_tests.log
cat $O/r05c10_tests.log
