#!/usr/bin/env python3
"""One-line digest of bench.py JSON lines on stdin (value, ms/step, median, host enqueue, dominant kernel)."""
import json
import sys

for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    j = json.loads(line)
    r = j.get("roofline") or {}
    print(f"{j['config']['workload'].split(':')[0]:6s} {j['value']:12.1f} {j['unit']}  {j['ms_per_step']:.3f} ms (median {j.get('median_ms_per_step')}) "
          f"enqueue {j.get('host_enqueue_ms_per_step')} ms  kernels isolated {r.get('gpu_ms_per_step_all_kernels')} ms  "
          f"dominant {r.get('kernel')} frac {r.get('frac')}")
