#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python tools/probes/determinism.py 20 cfgT
for c in cfg3 cfg4 cfg5; do echo "== $c"; BENCH_ARGS="--config $c --no-traffic" bash tools/sweep.sh 2 REFIL_EARLY=0 REFIL_EARLY=1; done
python bench.py --fresh-batches 8 --no-cpu-baseline --no-profile --no-traffic 2>/dev/null | cut -c1-300
