"""Where the host time of a train() call goes (cProfile over N steps, GPU queue drained first so that nothing blocks).
python tools/probes/host_profile.py [config] [N]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
W = dict(bench.CONFIGS[cfg])
if len(sys.argv) > 3:
    W["B"] = int(sys.argv[3])            # episodes per step (the strong-scaling regime: B = 4 / 8)
dims = bench.workload_dims(W)
_, batch, learner, _, _ = bench.build(dims, W["imagine"], W["B"], W["T"], seed=100, device=torch.device("cuda:0"))
for i in range(10):
    learner.train(batch, t_env=0, episode_num=i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(N):
    learner.train(batch, t_env=0, episode_num=i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{cfg}: host enqueue {1e3 * (t1 - t0) / N:.3f} ms per step, wall {1e3 * (t2 - t0) / N:.3f} ms per step")
pr = cProfile.Profile()
pr.enable()
for i in range(N):
    learner.train(batch, t_env=0, episode_num=i)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
