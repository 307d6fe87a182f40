#!/bin/bash
# Run ON THE GPU BOX: MFMA-busy counters, calibrated. tools/pmc_mfma.sh <tag>
#  1. tools/probes/mfma_probe (fp32 MFMA issue at ~99 % of peak, known by its own timer) under the same --pmc set: the
#     counter values of a kernel that keeps every matrix pipe busy = the normalisation of SQ_VALU_MFMA_BUSY_CYCLES
#  2. the serialised bench under the same counters (kernel-trace + pmc only, separate passes per counter group)
TAG=${1:-r03}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${TAG}_mfma; mkdir -p $OUT
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_probe.hip -o /tmp/mfma_probe 2> $OUT/probe_build.log
/tmp/mfma_probe > $OUT/probe_timer.txt 2>&1
(cd /tmp && rocprofv3 -L > $OUT/counters_list.txt 2>&1)
grep -i -E "mfma|GRBM_GUI_ACTIVE|SQ_BUSY_CU|SQ_WAVE_CYCLES" $OUT/counters_list.txt | head -60 > $OUT/counters_mfma.txt
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  g=$(echo $grp | tr ' ' '_')
  (cd /tmp && rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/probe_$g -o p -- /tmp/mfma_probe > $OUT/probe_$g.log 2>&1)
  (cd /tmp && rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/bench_$g -o p -- \
      python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile --serial --no-traffic --no-dense-region > $OUT/bench_$g.log 2>&1)
done
python tools/pmc_mfma.py $OUT > $OUT/summary.txt 2>&1
cp $OUT/mfma.json $ROOT/gpurun_out/${TAG}_pmc_mfma.json; cp $OUT/summary.txt $ROOT/gpurun_out/${TAG}_pmc_mfma_summary.txt 2>/dev/null
find $OUT -name "*.csv" ! -name "*counter_collection.csv" ! -name "*kernel_trace.csv" -delete
tail -50 $OUT/summary.txt
