#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out/r04_call11.txt
{
timeout 400 python tools/ab_small.py "tail_cols=0" "tail_cols=3072" "tail_cols=4096" "tail_cols=6144" "tail_cols=8192" "tail_cols=6144,tail_group=1" --n 16384 --d 32 --rounds 1
timeout 400 python tools/ab_small.py "tail_cols=0,tail_group=2" "tail_cols=6144,tail_group=2" --n 16384 --d 32 --rounds 2
} > $O 2>&1
cat $O
