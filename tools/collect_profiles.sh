#!/bin/bash
# Run ON THE GPU BOX from the repo root (gpurun -- 'bash tools/collect_profiles.sh r02'):
#   1. rocprofv3 --kernel-trace --stats of the default bench, overlapped and --serial      -> gpurun_out/<tag>_stats*/
#   2. PMC passes FETCH_SIZE / WRITE_SIZE (kernel-trace only, each in its own run)          -> gpurun_out/<tag>_pmc_*/
#   3. per-dispatch timeline of one step (serial and overlapped)                            -> gpurun_out/<tag>_timeline_*.txt
#   3b. the same step without a tracer (HIP events of the library profiler)                 -> gpurun_out/<tag>_timeline_untraced.txt
#   4. the bench line of every BASELINE.json config (cfgT with the CPU baseline and the PMC traffic of step 2)
# Copy the summaries into profiles/ afterwards (tools/pmc_summarize.py folds step 2).
TAG=${1:-r05}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
# the schedule knobs QLearner autotunes on its first call are measured ONCE per config here (plain runs, no tracer) and reused by
# every profiled run below: a tracer changes what overlaps, and the tuner's own launches would be counted into the profiled steps
export REFIL_AUTOTUNE_CACHE=$OUT/${TAG}_autotune_cache.json
rm -f $REFIL_AUTOTUNE_CACHE
for c in cfgT cfg2 cfg3 cfg4 cfg5; do python $ROOT/bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-profile --no-traffic > /dev/null 2>&1; done
for b in 4 8 16; do python $ROOT/bench.py --batch $b --steps 5 --warmup 2 --no-cpu-baseline --no-profile --no-traffic > /dev/null 2>&1; done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o p -- \
    python $ROOT/bench.py --no-cpu-baseline --no-traffic --no-dense-region > $OUT/${TAG}_bench_n1_under_rocprofv3.json 2> $OUT/${TAG}_stats.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats_serial -o p -- \
    python $ROOT/bench.py --no-cpu-baseline --no-traffic --no-dense-region --serial > $OUT/${TAG}_bench_n1_serial_under_rocprofv3.json 2> $OUT/${TAG}_stats_serial.log
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_pmc_$c -o p -- \
        python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile --no-traffic --no-dense-region --serial > $OUT/${TAG}_pmc_$c.log 2>&1
done
cd $ROOT
python tools/trace_step.py $(find $OUT/${TAG}_stats_serial -name "*kernel_trace.csv" | head -1) > $OUT/${TAG}_timeline_serial.txt
python tools/trace_step.py $(find $OUT/${TAG}_stats -name "*kernel_trace.csv" | head -1) > $OUT/${TAG}_timeline_overlapped.txt
cp $(find $OUT/${TAG}_stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_rocprofv3_kernel_stats.csv
cp $(find $OUT/${TAG}_stats_serial -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_rocprofv3_kernel_stats_serial.csv
python tools/pmc_summarize.py $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE $OUT/${TAG}_pmc_traffic.json
rm -rf $OUT/${TAG}_stats $OUT/${TAG}_stats_serial $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE
python bench.py --traffic-json $OUT/${TAG}_pmc_traffic.json > $OUT/${TAG}_bench_cfgT.json 2> $OUT/${TAG}_bench_cfgT.err
# the driver's command line (bench.py collects the PMC traffic itself), a fresh minibatch per step, unpadded data
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench_cfgT_driver_cmd.json 2> /dev/null
python bench.py --fresh-batches 8 --no-cpu-baseline --no-traffic > $OUT/${TAG}_bench_cfgT_fresh.json 2> /dev/null
python bench.py --dense-data --no-cpu-baseline --no-traffic > $OUT/${TAG}_bench_cfgT_dense.json 2> /dev/null
for c in cfg2 cfg3 cfg4 cfg5; do
    python bench.py --config $c --no-traffic > $OUT/${TAG}_bench_$c.json 2> $OUT/${TAG}_bench_$c.err
    # per-config rocprofv3 summary with the chains serialised (each kernel alone on the GPU: the kernel's own duration)
    (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats_$c -o p -- \
        python $ROOT/bench.py --config $c --no-cpu-baseline --no-traffic --no-dense-region --serial --steps 20 --warmup 5 > /dev/null 2>&1)
    cp $(find $OUT/${TAG}_stats_$c -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_${c}_kernel_stats_serial.csv
    rm -rf $OUT/${TAG}_stats_$c
done
python bench.py --serial --no-cpu-baseline --no-traffic > $OUT/${TAG}_bench_cfgT_serial.json 2>/dev/null
# calibrated MFMA-busy counters (tools/pmc_mfma.sh: probe at ~99 % of peak under the same counters, then the serialised bench)
bash tools/pmc_mfma.sh $TAG > $OUT/${TAG}_pmc_mfma.log 2>&1
# one step WITHOUT a tracer: HIP events around every launch (library profiler), stream + start + duration per launch
python tools/probes/timeline.py --out $OUT/${TAG}_timeline_untraced.txt > /dev/null 2>&1; rm -f $OUT/${TAG}_timeline_untraced.txt.raw
python tools/probes/timeline.py --config cfg2 --out $OUT/${TAG}_timeline_untraced_cfg2.txt > /dev/null 2>&1; rm -f $OUT/${TAG}_timeline_untraced_cfg2.txt.raw
# the strong-scaling regime (small shards), the acting path, hipGraph replay against the eager schedule, the projection GEMM alone
for b in 4 8 16; do python bench.py --batch $b --no-cpu-baseline --no-profile --no-traffic 2>/dev/null | cut -c1-330; done > $OUT/${TAG}_small_batches.txt
(python tools/acting_latency.py; python tools/acting_latency.py --envs 1) > $OUT/${TAG}_acting.txt 2>&1
python tools/probes/graph_capture.py 30 cfg2 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_hipgraph_cfg2.txt
python tools/probes/wres_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_wres_bench.txt
REFIL_EARLY=0 python bench.py --no-cpu-baseline --no-profile --no-traffic 2>/dev/null | cut -c1-330 > $OUT/${TAG}_bench_cfgT_no_early_prologue.txt
REFIL_EARLY_TARGET=0 python bench.py --no-cpu-baseline --no-profile --no-traffic 2>/dev/null | cut -c1-330 > $OUT/${TAG}_bench_cfgT_no_early_target.txt
# the replay buffer in pinned HOST memory (buffer_cpu_only): the PCIe-inclusive rate
python bench.py --fresh-batches 8 --host-buffer --no-cpu-baseline --no-traffic > $OUT/${TAG}_bench_cfgT_host_buffer.json 2> /dev/null
# what the fp32 matrix pipes sustain on real operand data (DVFS): bare MFMA streams, and the dominant projection on random / zero data
hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_dvfs_probe.hip -o /tmp/mfma_dvfs_probe 2>/dev/null && /tmp/mfma_dvfs_probe > $OUT/${TAG}_mfma_dvfs_probe.txt 2>&1
python tools/gemm_bench.py 50 dvfs 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_gemm_dvfs.txt
# round 5: the fused in_trans + attention launch against the launches it replaces (microbench), and the step with it switched off
(python tools/qkv_bench.py; python tools/qkv_bench.py --store --nvar 3; python tools/qkv_bench.py --dense; python tools/qkv_bench.py --cfg2) 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_qkv_bench.txt
REFIL_ATTN_QKV=0 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_bench_cfgT_qkv0.json 2> /dev/null
# round 5: the bf16 x 6 weight gradient alone on the GPU for a sweep of workgroup counts (profiles/r05_dws_target.txt part 1), the host's
# cost of one call on an idle queue
python tools/dws_bench.py --splits 16,32,64,128 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_dws_bench.txt
for a in "cfg2" "cfgT 4" "cfgT"; do python tools/probes/host_cost.py $a 2>&1 | grep -v amdgpu.ids; done > $OUT/${TAG}_host_cost.txt
ls -la $OUT | grep ${TAG}_
