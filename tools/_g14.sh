#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/probes/wres_bench.py
REFIL_LIB_PATH=$PWD/tools/_libs/line.so python tools/probes/wres_bench.py
for c in cfgT; do echo "== $c"; BENCH_ARGS="--config $c --no-traffic" bash tools/sweep.sh 3 REFIL_LIB_PATH=$PWD/tools/_libs/base.so REFIL_LIB_PATH=$PWD/tools/_libs/line.so; done
