"""Race check: two identically seeded learners take the same N steps on the same batch; parameters must be bit-identical
after every step (any cross-stream race in the schedule shows up as a divergence). python tools/probes/determinism.py [N] [config]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfgT"
W = dict(bench.CONFIGS[cfg])
dims = bench.workload_dims(W)
dev = torch.device("cuda", 0)
_, batch, la, _, _ = bench.build(dims, W["imagine"], W["B"], W["T"], seed=100, device=dev)
_, _, lb, _, _ = bench.build(dims, W["imagine"], W["B"], W["T"], seed=100, device=dev)
la._check_flat(); lb._check_flat()
assert torch.equal(la.flat_live, lb.flat_live)
for i in range(N):
    la.train(batch, t_env=0, episode_num=i)
    lb.train(batch, t_env=0, episode_num=i)
    torch.cuda.synchronize()
    assert torch.equal(la.flat_live, lb.flat_live), f"step {i}: replicas diverged (max |d| = {(la.flat_live - lb.flat_live).abs().max().item():.3e})"
print(f"{N} steps at {cfg}: replicas bit-identical after every step")
