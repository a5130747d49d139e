#!/bin/bash
# round 5, GPU call 9: hop costs -- FINE without an acquire fence, DIAG workgroups touching their rows while they wait, queue depth
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
( for k in "pipe_warm=0" "pipe_warm=1"; do echo "== $k"; PIPE_KNOBS=$k timeout 60 tools/_pipe_check_tr trace 512 0 1 2>&1 | grep "DIAG\|launch_potrf\|FINE   p=0 z=0 a=4 b=0\|TRSM   p=0 z=0 a=4 "; done
  for la in 4 6 8 16; do timeout 60 tools/_pipe_check_tr trace 4096 1 $la 2>&1 | grep "launch_potrf.*traced\|^# [A-Z]"; done
  timeout 300 tools/pipe_check 4096 2>&1 | grep -v "^PASS" ) > $O/r05c9_hops.txt 2>&1
cat $O/r05c9_hops.txt
