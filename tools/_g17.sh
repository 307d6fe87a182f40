#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_early.py -m gpu -x -q 2>&1 | tail -15
python tools/probes/determinism.py 20 cfgT
for c in cfgT cfg2 cfg3 cfg5 cfg4; do echo "== $c"; BENCH_ARGS="--config $c --no-traffic" bash tools/sweep.sh 3 REFIL_EARLY=0 REFIL_EARLY=1; done
