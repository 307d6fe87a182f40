#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_learner.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
for c in cfg2 cfg3 cfg5; do echo "== $c"; BENCH_ARGS="--config $c --no-traffic" bash tools/sweep.sh 2 REFIL_LIB_PATH=$PWD/tools/_libs/base.so REFIL_LIB_PATH=$PWD/refil_amd/librefil_hip.so; done
