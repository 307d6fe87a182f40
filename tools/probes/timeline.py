"""Untraced timeline of ONE learner step: HIP events around every launch (the library's profiler), dumped with
REFIL_PROFILE_TIMELINE. Usage: python tools/probes/timeline.py [--config cfgT] [--out gpurun_out/timeline.txt]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="cfgT")
ap.add_argument("--out", default="gpurun_out/timeline.txt")
ap.add_argument("--batch", type=int, default=0, help="episodes instead of the config's (small shards)")
a = ap.parse_args()
raw = a.out + ".raw"
os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
os.environ["REFIL_PROFILE_TIMELINE"] = raw

import torch  # noqa: E402

import bench  # noqa: E402
from refil_amd import _lib  # noqa: E402

W = dict(bench.CONFIGS[a.config])
if a.batch:
    W["B"] = a.batch
dims = bench.workload_dims(W)
args, batch, learner, data, _ = bench.build(dims, W["imagine"], W["B"], W["T"], seed=100, device=torch.device("cuda:0"))
for i in range(8):
    learner.train(batch, t_env=0, episode_num=i)
torch.cuda.synchronize()
_lib.profile_enable(True)
for i in range(3):
    learner.train(batch, t_env=0, episode_num=i)
_lib.profile_collect()
_lib.profile_enable(False)
rows = [l.rstrip("\n").split("\t") for l in open(raw)]
streams = {}
recs = []
for st, t0, dt, name in rows:
    streams.setdefault(st, "q%d" % len(streams))
    recs.append((float(t0), float(dt), streams[st], name))
n = len(recs) // 3
last = sorted(recs[2 * n:])
base = last[0][0]
end = max(t + d for t, d, _, _ in last)
with open(a.out, "w") as f:
    f.write("step wall %.0f us, %d profiled launches, %d streams\n" % (end - base, len(last), len(streams)))
    for t, d, q, name in last:
        f.write("%8.1f %7.1f  %s  %s\n" % (t - base, d, q, name))
    busy = {}
    for t, d, q, _ in last:
        busy[q] = busy.get(q, 0.0) + d
    f.write("busy per stream (us): %s\n" % busy)
print(open(a.out).read())
