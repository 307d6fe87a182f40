#!/bin/bash
cd $GRAFT_REPO_ROOT
for c in cfgT cfg2 cfg3 cfg5 cfg4; do echo "== $c"; BENCH_ARGS="--config $c --no-traffic" bash tools/sweep.sh 3 REFIL_DEFER_REDUCE=0 REFIL_DEFER_REDUCE=1 REFIL_DEFER_REDUCE=2; done
