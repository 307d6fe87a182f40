#!/bin/bash
# Run ON THE GPU BOX from the repo root (gpurun -- 'bash tools/collect_profiles.sh r01'):
#   1. default bench (JSON line with roofline + cpu_baseline)      -> gpurun_out/<tag>_bench_n1.json
#   2. rocprofv3 --kernel-trace --stats of the same command        -> gpurun_out/<tag>_stats/
#   2b. the same under --serial (streams serialised: bench's HIP-event durations == rocprofv3's)
#   3. PMC passes FETCH_SIZE / WRITE_SIZE (kernel-trace only, each in its own run) -> gpurun_out/<tag>_pmc_*/
# Copy the summaries into profiles/ afterwards (tools/pmc_summarize.py folds step 3).
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/${TAG}_bench_n1.json 2> $OUT/${TAG}_bench_n1.err
tail -c 600 $OUT/${TAG}_bench_n1.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o p -- \
    python $ROOT/bench.py --no-cpu-baseline > $OUT/${TAG}_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats_serial -o p -- \
    python $ROOT/bench.py --no-cpu-baseline --serial > $OUT/${TAG}_stats_serial.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_pmc_$c -o p -- \
        python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile > $OUT/${TAG}_pmc_$c.log 2>&1
done
cd $ROOT
find $OUT/${TAG}_stats $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE -name "*.csv" | head -20
