"""hipGraph capture of the learner step against the eager schedule: two identically seeded learners take the same N steps,
one eagerly, one through REFIL_HIPGRAPH=1 (eager first step, capture on the second, replay afterwards); parameters must
stay bit-identical, and the wall time per step of both is printed.
python tools/probes/graph_capture.py [N] [config]      (REFIL_GRADSTREAM / REFIL_MW_SIDE / ... select the schedule)"""
import faulthandler
import os
import sys
import time

faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
W = dict(bench.CONFIGS[cfg])
dims = bench.workload_dims(W)
dev = torch.device("cuda", 0)
_, batch, la, _, _ = bench.build(dims, W["imagine"], W["B"], W["T"], seed=100, device=dev)
_, _, lb, _, _ = bench.build(dims, W["imagine"], W["B"], W["T"], seed=100, device=dev)
la._check_flat(); lb._check_flat()


def run(learner, graphed, n, ep0):
    if graphed:
        os.environ["REFIL_HIPGRAPH"] = "1"
    else:
        os.environ.pop("REFIL_HIPGRAPH", None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        learner.train(batch, t_env=0, episode_num=ep0 + i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t2 - t0) / n * 1e3, (t1 - t0) / n * 1e3


run(la, False, 3, 0)
print("eager warm-up done", flush=True)
run(lb, True, 1, 0)
print("graphed learner: eager first step done", flush=True)
run(lb, True, 1, 1)
print("graphed learner: capture + first replay done", flush=True)
run(lb, True, 1, 2)
torch.cuda.synchronize()
print("graphed learner: replay done", flush=True)
assert torch.equal(la.flat_live, lb.flat_live), f"after 3 steps: max |d| = {(la.flat_live - lb.flat_live).abs().max().item():.3e}"
for rnd in range(3):
    e_ms, e_host = run(la, False, N, 3 + rnd * N)
    g_ms, g_host = run(lb, True, N, 3 + rnd * N)
    same = torch.equal(la.flat_live, lb.flat_live)
    print(f"{cfg} round {rnd}: eager {e_ms:.3f} ms/step (host enqueue {e_host:.3f}), graph replay {g_ms:.3f} ms/step (host {g_host:.3f}), "
          f"bit-identical: {same}", flush=True)
    assert same
