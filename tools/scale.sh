#!/bin/bash
# Scaling sweep on ONE node with N GPUs (run from the repo root): weak scaling (B=32 episodes per GPU) and strong scaling
# (global batch 64 and 256) at 1/2/4/8 GPUs, one rank per GPU over RCCL. Prints one bench JSON line per run into
# gpurun_out/scale_<mode>_n<N>.json and a summary table; NCCL_DEBUG=VERSION records the RCCL build in the .err files.
#   tools/scale.sh [max_gpus] [extra bench args...]
# Expected: the step's one all-reduce moves 1.74 MB (latency-bound, ~30-60 us on xGMI) behind a ~1.9 ms
# step => weak scaling >= 7.5x at 8 GPUs; strong scaling of B=32..64 ends on the per-step latency floor (~0.9 ms) => ~2-3x.
MAXN=${1:-8}; shift
OUT=gpurun_out; mkdir -p $OUT
export NCCL_DEBUG=VERSION HSA_ENABLE_IPC_MODE_LEGACY=0
run() {   # mode n args...
  local mode=$1 n=$2; shift 2
  local f=$OUT/scale_${mode}_n${n}.json
  if [ "$n" = 1 ]; then python bench.py --gpus 1 --no-cpu-baseline --no-traffic "$@" > $f 2> ${f%.json}.err
  else python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) \
         bench.py --gpus $n --no-cpu-baseline --no-traffic "$@" > $f 2> ${f%.json}.err; fi
}
for n in 1 2 4 8; do
  [ $n -le $MAXN ] || continue
  run weak $n --scaling weak "$@"
  run strong64 $n --scaling strong --global-batch 64 "$@"
  run strong256 $n --scaling strong --global-batch 256 "$@"
done
python - <<'PY'
import glob, json, re
rows = {}
for f in sorted(glob.glob("gpurun_out/scale_*_n*.json")):
    m = re.search(r"scale_(\w+)_n(\d+)\.json", f)
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception:
        continue
    rows.setdefault(m[1], {})[int(m[2])] = (j["value"], j["ms_per_step"])
for mode, d in rows.items():
    base = d.get(1, (None,))[0]
    print(mode)
    for n in sorted(d):
        v, ms = d[n]
        print(f"  n={n}: {v:12.0f} transitions/s  {ms:7.3f} ms/step" + (f"  x{v / base:.2f}" if base else ""))
PY
