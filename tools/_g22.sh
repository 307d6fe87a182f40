#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_early.py -m gpu -x -q 2>&1 | tail -5
for c in cfgT cfg2; do echo "== $c"; BENCH_ARGS="--config $c --no-traffic" bash tools/sweep.sh 2 REFIL_EARLY=0 REFIL_EARLY=1,REFIL_EARLY_ON=1 REFIL_EARLY=1,REFIL_EARLY_ON=2 REFIL_EARLY=1,REFIL_EARLY_ON=3; done
