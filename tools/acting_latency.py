"""Acting-path latency on the GPU: EntityMAC.select_actions for one environment step of a
batch of parallel runners (reference: BasicMAC.select_actions / forward(t=int), src/controllers/basic_controller.py:19-45,
called once per env step by the episode runners with batch_size_run envs). Reports the per-step wall time of
init_hidden + T select_actions calls, host-synchronised after each step like a runner that needs the actions.

    python tools/acting_latency.py [--config cfgT] [--envs 8] [--steps 80]
"""
import argparse
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfgT", choices=list(bench.CONFIGS))
    ap.add_argument("--envs", type=int, default=8)          # batch_size_run of the parallel runner (default.yaml)
    ap.add_argument("--steps", type=int, default=80)
    ap.add_argument("--episodes", type=int, default=6)
    a = ap.parse_args()
    W = bench.CONFIGS[a.config]
    dims = bench.workload_dims(W)
    dev = torch.device("cuda", 0)
    args, batch, learner, _, _ = bench.build(dims, W["imagine"], a.envs, a.steps, seed=5, device=dev)
    mac = learner.mac
    per_step = []
    for ep in range(a.episodes):
        mac.init_hidden(a.envs)
        for t in range(a.steps):
            t0 = time.perf_counter()
            actions = mac.select_actions(batch, t_ep=t, t_env=ep * a.steps + t, test_mode=False)
            actions.cpu()                                   # the runner steps the envs with these actions
            if ep > 0:
                per_step.append((time.perf_counter() - t0) * 1e6)
    out = {"metric": "acting-path latency per environment step (select_actions + D2H of the actions)", "unit": "us",
           "median": round(statistics.median(per_step), 1), "p90": round(sorted(per_step)[int(0.9 * len(per_step))], 1),
           "envs": a.envs, "config": a.config, "n_entities": dims["ne"], "n_agents": dims["na"],
           "env_steps_per_s": round(a.envs * 1e6 / statistics.median(per_step), 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
