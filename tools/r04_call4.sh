#!/bin/bash
# round 4, GPU call 4: left-looking group updates with look-ahead -- correctness under EGX_POTRF_LEFT=2, A/B in the sweep
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests -m gpu -q -k "gmx_responsibility or beyond_64 or kpls_with_100 or beyond_256" 2>&1 | tail -5 > $O/r04c4_fixed_tests.log
EGX_POTRF_LEFT=2 timeout 900 python -m pytest tests -m gpu -q -k "lockstep or potrf or fixed_theta or gradient or predict_var or ornstein or config2 or config3" 2>&1 | tail -8 > $O/r04c4_left_tests.log
timeout 400 python tools/ab_knobs.py "potrf_left=0" "potrf_left=1" --rounds 3 > $O/r04c4_ab_left_lookahead.log 2>&1
timeout 300 python tools/ab_knobs.py "potrf_left=0" "potrf_left=1" --rounds 2 --in-flight 32 --cands 64 --no-group > $O/r04c4_ab_left_inflight32.log 2>&1
timeout 300 python tools/ab_knobs.py "potrf_left=1" "potrf_left=1,potrf_group=2" "potrf_left=1,potrf_group=8" --rounds 2 --no-group > $O/r04c4_ab_left_group.log 2>&1
EGX_POTRF_LEFT=1 timeout 300 python tools/grad_bench.py 16384 32 3 8 > $O/r04c4_grad_bench_left.log 2>&1
tail -3 $O/r04c4_fixed_tests.log; tail -4 $O/r04c4_left_tests.log; cat $O/r04c4_ab_left_lookahead.log $O/r04c4_ab_left_inflight32.log $O/r04c4_ab_left_group.log; cat $O/r04c4_grad_bench_left.log
