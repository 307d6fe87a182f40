#!/bin/bash
cd $GRAFT_REPO_ROOT
for c in cfgT cfg2; do for e in 0 1; do echo "== $c EARLY=$e"; REFIL_EARLY=$e python bench.py --config $c --no-cpu-baseline --no-profile --no-traffic 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print(j['ms_per_step'], j['median_ms_per_step'], j['host_enqueue_ms_per_step'])"; done; done
REFIL_EARLY=1 python tools/probes/timeline.py --config cfg2 > gpurun_out/tl_early_cfg2.txt 2>&1
REFIL_EARLY=1 python tools/probes/timeline.py > gpurun_out/tl_early_cfgT.txt 2>&1
