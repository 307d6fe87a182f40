"""REFIL_HIPGRAPH=1: a QLearner.train step captured into a hipGraph (second call) and replayed (later calls) leaves the
same parameters as the eager schedule. Runs in child processes: the library reads its stream switches once per process."""
import multiprocessing as mp
import os

import pytest

pytestmark = pytest.mark.gpu


def _train(env, q):
    import sys
    for k, v in env.items():
        os.environ[k] = v
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch as th
    from golden_util import load
    from plugin_util import RecLogger, make_args, make_episode_batch
    from refil_amd.controllers import REGISTRY as mac_REGISTRY
    from refil_amd.learners import REGISTRY as le_REGISTRY
    g = load("refil_mid")
    cfg = g["cfg"]
    args = make_args(cfg)
    batch, groups = make_episode_batch(cfg, g["batch"])
    th.manual_seed(3)
    mac = mac_REGISTRY[args.mac](batch.scheme, groups, args)
    learner = le_REGISTRY[args.learner](mac, batch.scheme, RecLogger(), args)
    learner.cuda()
    batch.to("cuda")
    learner.generator = th.Generator().manual_seed(11)
    for i in range(5):                              # eager, capture + replay, replay x3 (fresh partition bits every step)
        learner.train(batch, t_env=i, episode_num=i)
    th.cuda.synchronize()
    q.put((learner.flat_live.cpu().numpy(), learner.square_avg.cpu().numpy(), dict(learner.logger.stats),
           len(learner._graphs)))


def _run(env):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_train, args=(env, q))
    p.start()
    out = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0
    return out


def test_graph_replay_equals_eager_steps():
    import numpy as np
    base = {}                                   # the default four-stream schedule
    p0, s0, st0, n0 = _run(base)
    p1, s1, st1, n1 = _run(dict(base, REFIL_HIPGRAPH="1"))
    assert n0 == 0 and n1 == 1
    assert np.isfinite(p1).all()
    assert np.abs(p1 - p0).max() < 1e-6 and np.abs(s1 - s0).max() < 1e-6
    for k in ("loss", "grad_norm", "td_error_abs"):
        assert abs(st1[k] - st0[k]) < 1e-5 * max(abs(st0[k]), 1e-3), k
