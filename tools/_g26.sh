#!/bin/bash
cd $GRAFT_REPO_ROOT
for c in cfgT cfg2 cfg3 cfg4 cfg5; do echo "== $c"; BENCH_ARGS="--config $c --no-traffic" bash tools/sweep.sh 2 REFIL_DW4_MIN_OUT=30000 REFIL_DW4_MIN_OUT=-1 REFIL_DW4_MIN_OUT=2000; done
timeout 900 python -m pytest tests/test_gpu_learner.py tests/test_gpu_ops.py -m gpu -q 2>&1 | grep -E "passed|failed|rror" | tail -3
