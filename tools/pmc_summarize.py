"""Fold the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only) into
profiles/rNN_pmc_traffic.json, keyed by the kernel names the in-library profiler (bench.py) reports.

    python tools/pmc_summarize.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE profiles/r01_pmc_traffic.json

Counter values are KiB. Per /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section) FETCH_SIZE
under-reports wide coalesced reads by 2x on gfx950, so hbm_bytes_per_launch = 2*fetch + write.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    """'void refil::gemm_kernel<4, 2, ...>(refil::GemmArgs)' -> 'gemm_kernel<4,2,...>' (profile.hip naming)."""
    n = re.sub(r"^void\s+", "", name)
    n = re.sub(r"\(.*\)$", "", n) if n.endswith(")") else n
    n = n.replace("refil::", "").replace("(anonymous namespace)::", "")
    n = re.sub(r"\(refil_mask_code\)", "", n)
    return n.replace(", ", ",").replace("true", "1").replace("false", "0")


def profiler_name(sym: str) -> str:
    """Name under which refil_profile_collect() reports this symbol (EPI/VEC instantiations lumped)."""
    m = re.match(r"gemm_kernel<(\d),(\d),(\d),(\d),([01]),([01]),\d,\d>$", sym)
    if m:
        b = lambda x: "true" if x == "1" else "false"
        return "gemm_kernel<%s,%s,%s,%s,%s,%s>" % (m[1], m[2], m[3], m[4], b(m[5]), b(m[6]))
    if sym.startswith("attn_fwd_mfma") or sym.startswith("attn_bwd_mfma"):
        return sym.split("<")[0]
    # kernel symbols whose launches the library profiler reports under another (rounds 1-2) name, all instantiations lumped
    alias = {"attn_fwd_pipe": "attn_fwd_mfma", "attn_bwd_pipe": "attn_bwd_mfma", "attn_qkv_fwd": "attn_qkv_fwd", "gru_fwd4_kernel": "gru_fwd_kernel<true>",
             "gru_fwd16_kernel": "gru_fwd_kernel<true>", "gru_bwd4_kernel": "gru_bwd_kernel", "gru_bwd16_kernel": "gru_bwd_kernel"}
    base = sym.split("<")[0]
    if base in alias:
        return alias[base]
    m = re.match(r"gru_fwd_kernel<([01])>$", sym)
    if m:
        return "gru_fwd_kernel<%s>" % ("true" if m[1] == "1" else "false")
    return sym


def fold(d):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            for key in {short(r["Kernel_Name"]), profiler_name(short(r["Kernel_Name"]))}:
                a = acc[key]
                a[0] += float(r["Counter_Value"]) * 1024.0
                a[1] += 1
    return acc


def main():
    fd, wd, out = sys.argv[1:4]
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 5           # learner steps in the profiled run (2 warm-up + 3)
    F, W = fold(fd), fold(wd)
    ks = {}
    total = 0.0
    seen = set()
    for f in glob.glob(os.path.join(fd, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            total += 2 * float(r["Counter_Value"]) * 1024.0
    for f in glob.glob(os.path.join(wd, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            total += float(r["Counter_Value"]) * 1024.0
    for k in sorted(set(F) | set(W)):
        f = F[k][0] / max(F[k][1], 1)
        w = W[k][0] / max(W[k][1], 1)
        ks[k] = {"fetch_bytes_per_launch_raw": round(f), "write_bytes_per_launch_raw": round(w),
                 "hbm_bytes_per_launch": round(2 * f + w), "launches_sampled": max(F[k][1], W[k][1])}
    json.dump({"note": __doc__.strip().split("\n\n")[-1].replace("\n", " "), "command": "python bench.py --steps 3 --warmup 2 "
               "--no-cpu-baseline --no-profile --serial", "steps_profiled": steps,
               "hbm_bytes_per_step_all_kernels": round(total / steps), "kernels": ks}, open(out, "w"), indent=1)
    print(f"{len(ks)} kernels -> {out}")


if __name__ == "__main__":
    main()
