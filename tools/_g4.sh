#!/bin/bash
cd $GRAFT_REPO_ROOT
for m in 0 1 2 3 4; do timeout 60 tools/_bin/capture_refork $m; echo "exit=$?"; done
timeout 1500 python -m pytest tests/test_gpu_learner.py tests/test_gpu_dp.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -5
python tools/probes/determinism.py 20 cfgT
python tools/probes/determinism.py 20 cfg2
