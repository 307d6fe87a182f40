#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python tools/probes/wres_bench.py
python tools/probes/wres_bench.py 185856 128 128 5 > /dev/null
OUT=$PWD/gpurun_out/wres_pmc; mkdir -p $OUT
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  g=$(echo $grp | tr ' ' '_' | cut -c1-40)
  (cd /tmp && rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$g -o p -- python $GRAFT_REPO_ROOT/tools/probes/wres_bench.py 185856 128 128 3 > $OUT/$g.log 2>&1)
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/wres_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_wres" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:32s} n={len(v)} mean={sum(v)/len(v):.4g}")
PY
find gpurun_out/wres_pmc -name "*.csv" -size +200k -delete
