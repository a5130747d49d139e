#!/bin/bash
# round 5, GPU call 1: the pipelined chain kernel's checker (bits / groups / batch / pivot / timeout / timings)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
timeout 240 tools/pipe_check 8192 > $O/r05c1_pipe_check.log 2>&1
echo "pipe_check rc=$?" >> $O/r05c1_pipe_check.log
cat $O/r05c1_pipe_check.log
