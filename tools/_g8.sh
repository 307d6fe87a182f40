#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_ops.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 1500 python -m pytest tests/test_gpu_learner.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
python tools/probes/determinism.py 20 cfgT
for c in cfgT cfg2 cfg3 cfg5; do echo "== $c"; BENCH_ARGS="--config $c --no-traffic" bash tools/sweep.sh 3 REFIL_LIB_PATH=$PWD/tools/_libs/base.so REFIL_LIB_PATH=$PWD/refil_amd/librefil_hip.so; done
python bench.py --no-cpu-baseline --no-traffic > gpurun_out/g8_new.json 2>/dev/null
python bench.py --config cfg2 --no-cpu-baseline --no-traffic > gpurun_out/g8_new_cfg2.json 2>/dev/null
