"""Long run of the whole act -> buffer -> sample -> train cycle (tests/test_gpu_run_loop.py's toy task): device memory in use, the
torch allocator's reserve and the process's resident set, sampled every 500 iterations. A leak of events / staging batches /
workspaces per step would show as growth. usage: python tools/probes/leak_check.py [iterations]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import test_gpu_run_loop as R  # noqa: E402


def rss_mb():
    for line in open("/proc/self/status"):
        if line.startswith("VmRSS"):
            return int(line.split()[1]) / 1024.0


def main():
    from refil_amd.components.episode_buffer import EpisodeBatch, ReplayBuffer
    from refil_amd.controllers import REGISTRY as mac_REGISTRY
    from refil_amd.learners import REGISTRY as le_REGISTRY
    from plugin_util import RecLogger
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    dev = torch.device("cuda", 0)
    torch.manual_seed(5); np.random.seed(5)
    rng = np.random.default_rng(5)
    n_envs, batch_size, buffer_size = 16, 32, 96
    scheme, groups, preprocess = R._scheme()
    args = R._args(True, anneal=20000, lr=0.005)
    buffer = ReplayBuffer(scheme, groups, buffer_size, R.T_LIMIT + 1, preprocess=preprocess, device=dev)
    mac = mac_REGISTRY[args.mac](buffer.scheme, groups, args)
    learner = le_REGISTRY[args.learner](mac, buffer.scheme, RecLogger(), args)
    learner.cuda()
    envs = R.MatchEnvs(n_envs, rng)
    new_batch = lambda: EpisodeBatch(scheme, groups, n_envs, R.T_LIMIT + 1, preprocess=preprocess, device=dev)  # noqa: E731
    t_env = episode = 0
    for it in range(iters + 1):
        ep, t_env, ret = R.run_episodes(envs, mac, new_batch, t_env, False)
        buffer.insert_episode_batch(ep)
        episode += n_envs
        if buffer.can_sample(batch_size):
            s = buffer.sample(batch_size)
            s = s[:, :s.max_t_filled()]
            learner.train(s, t_env, episode)
        if it % 500 == 0:
            torch.cuda.synchronize()
            free, total = torch.cuda.mem_get_info()
            print(f"iter {it:6d}: device in use {(total - free) / 2**20:9.1f} MiB, torch reserved {torch.cuda.memory_reserved() / 2**20:7.1f} MiB, "
                  f"host RSS {rss_mb():8.1f} MiB, train-mode return {ret.mean():.3f}", flush=True)


if __name__ == "__main__":
    main()
