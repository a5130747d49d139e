#!/bin/bash
# A/B of a padded leading dimension of the factorisation's matrix (ld = n_pad + pad doubles) through tools/pipe_check:
# launch_potrf of one matrix by separate launches, chain launches per group, whole / flow launches.
OUT=gpurun_out/ld_pad_ab.txt
: > $OUT
for pad in 0 16 32 64 272; do
  echo "===== EGX_LD_PAD=$pad" >> $OUT
  EGX_LD_PAD=$pad timeout 300 tools/pipe_check 16384 2>&1 | grep -E "^time|CHECKS|FAILED|FAIL" >> $OUT
  EGX_LD_PAD=$pad timeout 300 tools/pipe_check flow 4096 8192 16384 2>&1 | tail -25 >> $OUT
done
cat $OUT
