"""Calibrated MFMA-busy fractions per kernel from the rocprofv3 PMC passes of tools/pmc_mfma.sh.

SQ_VALU_MFMA_BUSY_CYCLES is reported by rocprofv3 summed over whatever SQ instances it samples, in units that
ROCm 7.2 does not document for gfx950 (derived metrics fall back to gfx94x formulas). Instead of guessing the
normalisation it is MEASURED: tools/probes/mfma_probe issues back-to-back fp32 MFMAs on every SIMD (its own timer
reports ~99 % of the 157.3 TFLOP/s peak), so for its kernels
        k = SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE          (per dispatch)
is the counter ratio of a fully busy chip, and a kernel's MFMA-busy fraction is its own ratio / (k / probe_util).
usage: python tools/pmc_mfma.py gpurun_out/<tag>_mfma   -> <dir>/mfma.json + a table on stdout"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

PEAK = 157.3


def fold(d):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = re.sub(r"^void\s+", "", r["Kernel_Name"]).replace("refil::", "")
            n = re.sub(r"\(.*\)$", "", n)
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    d = sys.argv[1]
    probe_tf = {}
    for line in open(os.path.join(d, "probe_timer.txt")):
        m = re.match(r"(.+?)\s+blocks=\s*(\d+)\s+([\d.]+) ms\s+([\d.]+) TFLOP/s", line)
        if m:
            probe_tf.setdefault(m[1].strip(), []).append((int(m[2]), float(m[4])))
    out = {"probe_timer_tflops": probe_tf}
    groups = [g for g in glob.glob(os.path.join(d, "probe_*")) if os.path.isdir(g)]
    P, B = defaultdict(lambda: defaultdict(list)), defaultdict(lambda: defaultdict(list))
    for g in groups:
        for n, cs in fold(g).items():
            for c, v in cs.items():
                P[n][c] += v
    for g in [g for g in glob.glob(os.path.join(d, "bench_*")) if os.path.isdir(g)]:
        for n, cs in fold(g).items():
            for c, v in cs.items():
                B[n][c] += v
    # calibration: the probe kernels at >= 2 blocks per CU run at ~99 % of peak (their own timer); every probe dispatch counts
    ks = []
    print("probe dispatches (busy / gui_active per dispatch):")
    for n, cs in sorted(P.items()):
        if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "GRBM_GUI_ACTIVE" in cs:
            r = [b / g for b, g in zip(cs["SQ_VALU_MFMA_BUSY_CYCLES"], cs["GRBM_GUI_ACTIVE"]) if g > 0]
            print(f"  {n[:60]:60s} " + " ".join(f"{x:8.1f}" for x in r))
            ks += r
    ks.sort()
    best = max((tf for v in probe_tf.values() for _, tf in v), default=PEAK * 0.99)
    k_full = ks[-1] / (best / PEAK) if ks else None         # the fastest probe dispatch <-> its utilisation by the timer
    out["calibration"] = {"busy_over_gui_active_at_full_rate": k_full, "probe_best_tflops": best, "probe_ratios": ks,
                          "note": "MFMA-busy fraction of a kernel = (SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE) / busy_over_gui_active_at_full_rate"}
    rows = {}
    for n, cs in sorted(B.items()):
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in cs or "GRBM_GUI_ACTIVE" not in cs or not k_full:
            continue
        busy = sum(cs["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(cs["SQ_VALU_MFMA_BUSY_CYCLES"])
        gui = sum(cs["GRBM_GUI_ACTIVE"]) / len(cs["GRBM_GUI_ACTIVE"])
        e = {"launches": len(cs["GRBM_GUI_ACTIVE"]), "mfma_busy_frac": round(busy / gui / k_full, 4) if gui > 0 else None,
             "busy_cycles_per_launch": round(busy), "gui_active_per_launch": round(gui)}
        for extra in ("SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F32"):
            if extra in cs:
                e[extra] = round(sum(cs[extra]) / len(cs[extra]))
        rows[n] = e
    out["kernels"] = rows
    json.dump(out, open(os.path.join(d, "mfma.json"), "w"), indent=1)
    print(f"calibration: busy/gui_active at full rate = {k_full}")
    for n, e in sorted(rows.items(), key=lambda kv: -(kv[1]["mfma_busy_frac"] or 0) * kv[1]["gui_active_per_launch"] * kv[1]["launches"]):
        print(f"  {e['mfma_busy_frac']:7.3f}  x{e['launches']:4d}  {n[:100]}")


if __name__ == "__main__":
    main()
