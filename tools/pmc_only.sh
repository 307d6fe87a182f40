#!/bin/bash
TAG=r02x; ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_pmc_$c -o p -- \
        python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile --serial > $OUT/${TAG}_pmc_$c.log 2>&1
done
cd $ROOT
python tools/pmc_summarize.py $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE $OUT/${TAG}_pmc_traffic.json
rm -rf $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE
python - <<'PY'
import json
t=json.load(open('gpurun_out/r02x_pmc_traffic.json'))
print(t['hbm_bytes_per_step_all_kernels'])
for k,v in t['kernels'].items():
    if k in ('attn_fwd_mfma','attn_bwd_mfma'): print(k, v['hbm_bytes_per_launch']/1e6, v['fetch_bytes_per_launch_raw']/1e6, v['write_bytes_per_launch_raw']/1e6)
PY
