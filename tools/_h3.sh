#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_early.py tests/test_replay_buffer.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -8
for i in 1 2; do
for e in 0 1; do echo "fresh EARLY=$e"; REFIL_EARLY=$e python bench.py --fresh-batches 8 --no-cpu-baseline --no-profile --no-traffic 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print(j['ms_per_step'], j['median_ms_per_step'], j['host_enqueue_ms_per_step'])"; done
done
for e in 0 1; do echo "cfg2 fresh EARLY=$e"; REFIL_EARLY=$e python bench.py --config cfg2 --fresh-batches 8 --no-cpu-baseline --no-profile --no-traffic 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print(j['ms_per_step'], j['median_ms_per_step'], j['host_enqueue_ms_per_step'])"; done
