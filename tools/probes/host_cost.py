"""Host cost of ONE train() call with an idle GPU queue (no back-pressure): synchronize, call, time until the call returns; then the time
until the GPU is done. python tools/probes/host_cost.py [config] [B]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
W = dict(bench.CONFIGS[cfg])
if len(sys.argv) > 2:
    W["B"] = int(sys.argv[2])
dims = bench.workload_dims(W)
_, batch, learner, _, _ = bench.build(dims, W["imagine"], W["B"], W["T"], seed=100, device=torch.device("cuda:0"))
for i in range(20):
    learner.train(batch, t_env=0, episode_num=i)
torch.cuda.synchronize()
host, total = [], []
for i in range(100):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    learner.train(batch, t_env=0, episode_num=i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append(t1 - t0); total.append(t2 - t0)
host.sort(); total.sort()
N = 200
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(N):
    learner.train(batch, t_env=0, episode_num=i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{cfg} B={W['B']}: one call on an idle queue: host {1e3 * host[50]:.3f} ms (min {1e3 * host[0]:.3f}), until the GPU is done {1e3 * total[50]:.3f} ms; "
      f"back to back: calls return after {1e3 * (t1 - t0) / N:.3f} ms per step, wall {1e3 * (t2 - t0) / N:.3f} ms per step")
