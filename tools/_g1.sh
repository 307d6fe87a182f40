#!/bin/bash
# capture matrix: which schedule survives hipStreamEndCapture
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
run() { echo "=== $*" ; env "$@" timeout 300 python tools/probes/graph_capture.py 30 cfg2 2>&1 | tail -25; echo "exit=$?"; }
run REFIL_GRADSTREAM=0
run REFIL_HIPGRAPH_4S=1 REFIL_MW_SIDE=0
run REFIL_HIPGRAPH_4S=1
