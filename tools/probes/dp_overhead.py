"""What the data-parallel collective costs ONE rank's step, measured on one GPU: a one-rank RCCL group, the step's
all-reduce forced (REFIL_DP_FORCE=1) -- the backend runs it on a stream of its own, a fifth stream beside the step's four.
python tools/probes/dp_overhead.py [config]      (run once per GPU_MAX_HW_QUEUES setting: read at HIP initialisation)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfgT"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
W = dict(bench.CONFIGS[cfg]); dims = bench.workload_dims(W)
_, batch, learner, _, _ = bench.build(dims, W["imagine"], W["B"], W["T"], seed=100, device=dev)


def run(force, n=60):
    if force:
        os.environ["REFIL_DP_FORCE"] = "1"
    else:
        os.environ.pop("REFIL_DP_FORCE", None)
    for i in range(10):
        learner.train(batch, t_env=0, episode_num=i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        learner.train(batch, t_env=0, episode_num=i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


res = []
for rnd in range(3):
    res.append((run(False), run(True)))
print(f"{cfg} GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', 'default')}: " +
      "  ".join(f"no collective {a:.3f} ms / forced all-reduce {b:.3f} ms" for a, b in res))
dist.destroy_process_group()
