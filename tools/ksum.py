#!/usr/bin/env python3
"""Per-kernel table of a bench.py JSON line on stdin (use bench.py --top 100)."""
import json
import sys

j = json.loads([l for l in sys.stdin if l.startswith("{")][-1])
print(f"{j['ms_per_step']} ms/step")
for k in j["kernels"]:
    print(f"{k.get('ms_per_step_isolated', 0) * 1e3:8.1f} us/step isolated  {k['ms_per_step'] * 1e3:8.1f} in situ  x{k['launches_per_step']:3d}  "
          f"avg {k.get('avg_us_isolated', 0):7.1f}  {k['name']}")
