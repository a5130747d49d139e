#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
( timeout 60 tools/_pipe_check_tr trace 512 0 1 2>&1 | grep "DIAG\|launch_potrf\|FINE   p=0 z=0 a=4 b=0\|TRSM   p=0 z=0 a=4 "
  timeout 60 tools/_pipe_check_tr trace 4096 1 4 2>&1 | grep "DIAG\|launch_potrf\|^# "
  timeout 200 tools/pipe_check 8192 2>&1 | grep -v "^PASS" ) > $O/r05c3_diag_variants.txt 2>&1
cat $O/r05c3_diag_variants.txt
