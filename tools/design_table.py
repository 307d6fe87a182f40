#!/usr/bin/env python3
"""Markdown table of DESIGN.md section 8 from profiles/<tag>_bench_<cfg>.json.  usage: python tools/design_table.py [r03]"""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
names = {"cfgT": "cfg-T north-star (B=32, T=80, ne=32, d=h=128, refil) — **the bench line**",
         "cfg2": "cfg2 = configs[1] (B=32, T=80, ne=16, d=h=64, refil)",
         "cfg3": "cfg3 = configs[2] (B=64, T=80, ne=32, d=h=128, refil, \"roofline run\")",
         "cfg4": "cfg4 = configs[3] shape (B=32, T=150, ne=16, d=h=128, qmix_atten)",
         "cfg5": "cfg5 = configs[4] shape (B=32, T=80, ne=48, d=h=128, refil)"}
print("| config (`bench.py --config`) | transitions/s | ms/step mean (median) | host enqueue ms/step | executed GFLOP/step → `step_frac_executed` | dense-equivalent fraction | "
      "dominant kernel: bound, fraction of roof (isolated) | isolated kernel time ms/step | CPU oracle tr/s (threads) | live steps / hypernet entity rows / agent-net entity rows |")
print("|---|---|---|---|---|---|---|---|---|---|")
for c, nm in names.items():
    j = json.load(open(f"profiles/{tag}_bench_{c}.json"))
    r, cb, rows = j["roofline"], j.get("cpu_baseline") or {}, j["rows"]
    print(f"| {nm} | **{j['value'] / 1e6:.3f} M** | {j['ms_per_step']:.3f} ({j['median_ms_per_step']:.3f}) | {j['host_enqueue_ms_per_step']:.2f} | "
          f"{r['executed_gflop_per_step']:.1f} → {r['step_frac_executed']:.2f} | {r['step_frac_dense_equivalent']:.2f} | "
          f"`{r['kernel'].split('<')[0]}<{r['kernel'].split('<')[1][:6]}…>` {r['bound'].upper()} {r['frac']:.2f} | {r['gpu_ms_per_step_all_kernels']:.2f} | "
          f"{cb.get('value', 0):.0f} ({cb.get('cores', '-')}) | {rows['live_step_frac']:.2f} / {rows['entity_rows_frac_hypernets']:.2f} / {rows['entity_rows_frac_agent_nets']:.2f} |")
