#!/bin/bash
cd $GRAFT_REPO_ROOT
for l in base line base line; do REFIL_LIB_PATH=$PWD/tools/_libs/$l.so python tools/probes/wres_bench.py 2>&1 | tail -1; done
for l in base line; do REFIL_LIB_PATH=$PWD/tools/_libs/$l.so python tools/probes/wres_bench.py 185856 128 84 2>&1 | tail -1; done
