#!/bin/bash
cd $GRAFT_REPO_ROOT
REFIL_GRU_VALU=3 timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k gru 2>&1 | grep -E "passed|failed|rror" | tail -3
for c in cfgT cfg2 cfg4 cfg5; do echo "== $c"; BENCH_ARGS="--config $c --no-traffic" bash tools/sweep.sh 3 REFIL_GRU_VALU=0 REFIL_GRU_VALU=1; done
for v in 0 1; do REFIL_GRU_VALU=$v python bench.py --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.readline()); print(j['ms_per_step'], [(k['name'][:14],k['avg_us'],k['avg_us_isolated']) for k in j['kernels'] if 'gru' in k['name']])"; done
