#!/bin/bash
# The FIRST gpurun call after a round without a GPU (round 6 was one): verification of the committed HEAD before anything else, then
# exactly the measurements the round's opt-in work is waiting for. Everything lands under gpurun_out/ (copy what is kept to profiles/).
#
#   git rev-parse HEAD > tools/.head            # (gpurun snapshots the tree without .git)
#   gpurun --timeout 3300 -- 'bash tools/verify_head.sh [quick]'
#
# 1  full `pytest -m gpu` + smoke() + the driver's bench command, log with the commit hash at its top        
# 2  cfg5: the three attention launches against the fused launch of round 6 (attn_qkv_wide), interleaved on this box            (item 3)
# 3  the two probes round 5 left compiled-only and round 6 verified on the emulator: their TIMING loops              (items 2c, 5)
# 4  the fused launch's ISA after the zero-row accounting (one scratch reload per job): cfg-T A/B against the previous build is not
#    possible in one call -- the driver's bench command in (1) against profiles/r05_bench_cfgT_driver_cmd.json is the comparison
TAG=${TAG:-r06}
OUT=gpurun_out
mkdir -p $OUT
{
echo "HEAD $(cat tools/.head 2>/dev/null)   $(date -u +%FT%TZ)"
echo "=== 1a  pytest -m gpu"
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
echo "=== 1b  smoke"
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -5
echo "=== 1c  bench (the driver's command)"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | tail -1 | tee $OUT/${TAG}_bench_cfgT_driver_cmd.json | cut -c1-600
} > $OUT/${TAG}_verify_head.txt 2>&1
tail -25 $OUT/${TAG}_verify_head.txt
[ "$1" = quick ] && exit 0
{
echo "=== 2  cfg5 (48 entities, 24 agents): separate launches vs attn_qkv_wide, interleaved (median ms per step)"
BENCH_ARGS="--config cfg5 --steps 50 --warmup 10 --no-traffic --no-dense-region" bash tools/sweep.sh 3 "REFIL_ATTN_QKV_WIDE=0" "REFIL_ATTN_QKV_WIDE=1"
echo "=== 2b  cfg5 bench line with the fused launch (kernels[], roofline)"
REFIL_ATTN_QKV_WIDE=1 timeout 600 python bench.py --config cfg5 --steps 50 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/${TAG}_bench_cfg5_qkvwide.json | cut -c1-400
} > $OUT/${TAG}_cfg5_qkvwide_ab.txt 2>&1
{
echo "=== 3  probes (hipcc, gfx950)"
for p in gru_bf16_probe attn_core_bf16_probe mfma4_probe; do
  echo "--- $p"
  hipcc --offload-arch=gfx950 -O3 -I refil_amd/csrc tools/probes/$p.hip -o /tmp/$p 2>&1 | grep -E "error" ; timeout 120 /tmp/$p 2>&1 | grep -v amdgpu.ids
done
} > $OUT/${TAG}_probes_gpu.txt 2>&1
tail -40 $OUT/${TAG}_cfg5_qkvwide_ab.txt $OUT/${TAG}_probes_gpu.txt
