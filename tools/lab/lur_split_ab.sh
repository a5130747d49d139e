#!/bin/bash
# A/B of LUr in two pieces (EGX_LUR_SPLIT=0/1, read once per process): lone fits, a theta-gradient, lock-step batches at n = 8192, the sweep
OUT=gpurun_out/lur_split_ab.txt
: > $OUT
for v in 0 1 0 1; do
  echo "===== EGX_LUR_SPLIT=$v" >> $OUT
  EGX_LUR_SPLIT=$v python tools/one_fit.py 16384 32 5 0 2>&1 | grep -E "^fit" | tail -3 | cut -c1-200 >> $OUT
  EGX_LUR_SPLIT=$v python tools/one_fit.py 12288 24 5 0 2>&1 | tail -2 | cut -c1-200 >> $OUT
  EGX_LUR_SPLIT=$v EGX_PIPE=2 python tools/one_fit.py 8192 16 5 0 2>&1 | tail -2 | cut -c1-200 >> $OUT
  EGX_LUR_SPLIT=$v python tools/one_grad.py 16384 32 3 3 2>&1 | tail -2 >> $OUT
  EGX_LUR_SPLIT=$v python tools/one_fit.py 8192 16 4 0 12 12 2>&1 | tail -2 | cut -c1-120 >> $OUT
  EGX_LUR_SPLIT=$v python tools/expert_group.py 8 8192 16 4 2>&1 | tail -3 >> $OUT
done
cat $OUT
