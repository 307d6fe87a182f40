#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/probes/timeline.py > gpurun_out/tl_cfgT.txt 2>&1
python tools/probes/timeline.py --config cfg2 > gpurun_out/tl_cfg2.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_graph.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error" | tail -5
REFIL_HIPGRAPH=1 python bench.py --config cfg2 --no-cpu-baseline --no-profile --no-traffic 2>&1 | tail -1 | cut -c1-400
python bench.py --config cfg2 --no-cpu-baseline --no-profile --no-traffic 2>&1 | tail -1 | cut -c1-400
