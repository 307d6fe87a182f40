#!/bin/bash
cd $GRAFT_REPO_ROOT
for c in cfgT cfg2; do echo "== $c"; BENCH_ARGS="--config $c --no-traffic" bash tools/sweep.sh 2 REFIL_EARLY=0 REFIL_EARLY=1 REFIL_EARLY=1,GPU_MAX_HW_QUEUES=6 REFIL_EARLY=1,GPU_MAX_HW_QUEUES=8 REFIL_EARLY=0,GPU_MAX_HW_QUEUES=8; done
