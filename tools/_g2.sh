#!/bin/bash
cd $GRAFT_REPO_ROOT
for m in 0 1 2; do timeout 60 tools/_bin/capture_refork $m; echo "exit=$?"; done
REFIL_GEMM_LOG=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-traffic 2>&1 | grep "deferred" | tail -1
for c in cfg2 cfg3 cfg5; do REFIL_GEMM_LOG=1 python bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-traffic 2>&1 | grep "deferred" | tail -1; done
timeout 900 python -m pytest tests/test_gpu_learner.py tests/test_gpu_dp.py -m gpu -x -q 2>&1 | tail -5
python tools/probes/determinism.py 20 cfgT
for c in cfgT cfg2 cfg3 cfg5; do echo "== $c"; BENCH_ARGS="--config $c --no-traffic" bash tools/sweep.sh 3 REFIL_DEFER_REDUCE=0 REFIL_DEFER_REDUCE=1; done
