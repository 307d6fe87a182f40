"""Per-kernel SQ / GRBM counters from a rocprofv3 --pmc run (counter_collection CSV): averages per symbol.
usage: python tools/pmc_sq.py <dir with *counter_collection.csv> [name filter ...]"""
import csv
import glob
import sys
from collections import defaultdict

files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in files:
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("refil::", "").replace("void ", "").split("(")[0]
        acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[n][r["Counter_Name"]] += 1
flt = sys.argv[2:]
for n in sorted(acc):
    if flt and not any(x in n for x in flt):
        continue
    c = {k: acc[n][k] / cnt[n][k] for k in acc[n]}
    print(n[:70], " launches", max(cnt[n].values()))
    print("   ", "  ".join(f"{k}={v:.3g}" for k, v in sorted(c.items())))
    if "GRBM_GUI_ACTIVE" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        # MFMA busy cycles summed over SIMDs (cycles); GUI_ACTIVE = shader-clock cycles of the dispatch
        print(f"    mfma_busy / (gui_active * 1024 SIMDs) = {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] * 1024):.3f}")
