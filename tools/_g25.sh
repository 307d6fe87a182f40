#!/bin/bash
cd $GRAFT_REPO_ROOT
for c in cfgT cfg2; do echo "== $c"; BENCH_ARGS="--config $c --no-traffic" bash tools/sweep.sh 2 REFIL_DW4_MIN_OUT=30000 REFIL_DW4_MIN_OUT=16000 REFIL_DW4_MIN_OUT=12000 REFIL_DW4_MIN_OUT=10000 REFIL_DW4_MIN_OUT=2000; done
