#!/bin/bash
# round 5, GPU call 2: task timelines of the chain kernel (where does a chain launch spend its time?)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
timeout 60 tools/pipe_check trace 1024 1 1 > $O/r05c2_trace_n1024_whole.txt 2>&1
timeout 60 tools/pipe_check trace 4096 1 4 > $O/r05c2_trace_n4096_whole_la4.txt 2>&1
timeout 60 tools/pipe_check trace 512 0 1 > $O/r05c2_trace_n512_group.txt 2>&1
head -120 $O/r05c2_trace_n1024_whole.txt; tail -5 $O/r05c2_trace_n4096_whole_la4.txt
