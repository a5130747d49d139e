#!/bin/bash
# A/B of the deferred updates of the diagonal-block factorisation (potf2_blocks.h RB_DEFER / RB_DEFER_P): pipe_check built per
# variant (tools/lab/bin/pc_d<RB_DEFER>p<RB_DEFER_P>); the block alone, and launch_potrf of one matrix up to 4096 columns with
# all of pipe_check's checks.
OUT=gpurun_out/diag_defer_ab.txt
: > $OUT
for b in tools/lab/bin/pc_d*; do
  echo "===== $b" >> $OUT
  timeout 120 $b diag 2>&1 | tail -2 >> $OUT
  timeout 300 $b 4096 2>&1 | grep -E "^time|CHECKS|FAIL" >> $OUT
done
cat $OUT
