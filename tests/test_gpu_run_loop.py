"""The whole cycle a user of the reference runs, on a toy cooperative task: act (mac.select_actions on the HIP acting path) ->
EpisodeBatch.update -> ReplayBuffer.insert_episode_batch -> sample -> the max_t_filled() trim -> QLearner.train, in the order
of the reference's driver (src/run.py:228-275 and the batched runner's episode loop, src/runners/parallel_runner.py:84-186).
Every other GPU test feeds the learner synthetic or fixture batches; this one feeds it what its own policy generates --
episodes of different lengths and entity counts, epsilon-greedy actions restricted by avail_actions, a ring buffer that wraps,
target-network syncs -- and checks that the policy LEARNS the task (test_mode return from chance to near-optimal).

The environment is this test's own (the reference's environments are control plane, out of scope: SURVEY.md section 2): `na`
agents, each paired with one landmark entity only it can see; a landmark shows one of C colours, re-drawn every step; the team
reward of a step is the fraction of active agents whose action names their landmark's colour. Pairs are switched off at random per
episode (entity_mask / no-op-only avail_actions: REFIL's varying entity count), episodes end early at random (terminated = 1) or
at the time limit (terminated = 0, as the runners do for `episode_limit`)."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"        # (tests/test_emu_run_loop.py runs a private copy of this module with DEV = "cpu" on the CPU wavefront emulator)

NA, C, T_LIMIT = 4, 4, 10
NE, A, ED = 2 * NA, 1 + C, 2 + C + NA


class MatchEnvs:
    """`n` environments stepped together (numpy on the host, like the runners' worker processes)."""

    def __init__(self, n, rng):
        self.n, self.rng = n, rng

    def reset(self):
        n = self.n
        self.t = np.zeros(n, dtype=np.int64)
        self.active = np.zeros((n, NA), dtype=bool)
        for e in range(n):
            k = self.rng.integers(2, NA + 1)
            self.active[e, self.rng.permutation(NA)[:k]] = True
        self.horizon = 2 + self.rng.geometric(0.4, size=n)        # (a long tail: the longest episode of a minibatch varies; > T_LIMIT: the time limit ends it)
        self._draw()

    def _draw(self):
        self.colour = self.rng.integers(0, C, size=(self.n, NA))

    def observe(self, envs):
        """pre-transition data of the listed environments (the keys parallel_runner.py:102-108 collects)"""
        m = len(envs)
        ent = np.zeros((m, NE, ED), dtype=np.float32)
        emask = np.ones((m, NE), dtype=np.uint8)                   # 1 = entity absent
        omask = np.ones((m, NE, NE), dtype=np.uint8)               # 1 = row entity cannot see column entity
        avail = np.zeros((m, NA, A), dtype=np.int32)
        for i, e in enumerate(envs):
            for a in range(NA):
                ent[i, a, 0] = 1.0
                ent[i, a, 2 + C + a] = 1.0
                ent[i, NA + a, 1] = 1.0
                ent[i, NA + a, 2 + self.colour[e, a]] = 1.0
                ent[i, NA + a, 2 + C + a] = 1.0
                if self.active[e, a]:
                    emask[i, a] = emask[i, NA + a] = 0
                    omask[i, a, a] = omask[i, a, NA + a] = 0
                    omask[i, NA + a, NA + a] = 0
                    avail[i, a, 1:] = 1
                else:
                    avail[i, a, 0] = 1
            act = np.flatnonzero(self.active[e])
            omask[i][np.ix_(act, act)] = 0                         # agents see each other
        return {"entities": ent, "obs_mask": omask, "entity_mask": emask, "avail_actions": avail}

    def step(self, envs, actions):
        """-> reward, terminated (as stored: False when the time limit ended the episode), done"""
        rew = np.zeros(len(envs), dtype=np.float32)
        term = np.zeros(len(envs), dtype=np.uint8)
        done = np.zeros(len(envs), dtype=bool)
        for i, e in enumerate(envs):
            act = self.active[e]
            rew[i] = float(((actions[i] - 1 == self.colour[e]) & act).sum()) / act.sum()
            self.t[e] += 1
            if self.t[e] >= self.horizon[e]:
                done[i], term[i] = True, 1
            elif self.t[e] >= T_LIMIT:
                done[i] = True
        self._draw()
        return rew, term, done


def _scheme():
    from refil_amd.components.transforms import OneHot
    scheme = {
        "entities": {"vshape": ED, "group": "entities"},
        "obs_mask": {"vshape": NE, "group": "entities", "dtype": torch.uint8},
        "entity_mask": {"vshape": NE, "dtype": torch.uint8},
        "actions": {"vshape": (1,), "group": "agents", "dtype": torch.long},
        "avail_actions": {"vshape": (A,), "group": "agents", "dtype": torch.int},
        "reward": {"vshape": (1,)},
        "terminated": {"vshape": (1,), "dtype": torch.uint8},
    }
    return scheme, {"agents": NA, "entities": NE}, {"actions": ("actions_onehot", [OneHot(out_dim=A)])}


def _args(imagine, anneal, lr):
    return types.SimpleNamespace(
        agent="imagine_entity_attend_rnn" if imagine else "entity_attend_rnn", mac="entity_mac", learner="q_learner", mixer="flex_qmix",
        agent_output_type="q", action_selector="epsilon_greedy", epsilon_start=1.0, epsilon_finish=0.05, epsilon_anneal_time=anneal,
        n_agents=NA, n_actions=A, n_entities=NE, entity_shape=ED, entity_scheme=True, entity_last_action=True, gt_mask_avail=False,
        attn_embed_dim=32, attn_n_heads=2, rnn_hidden_dim=32, hypernet_embed=32, mixing_embed_dim=32, softmax_mixing_weights=True,
        pooling_type=None, double_q=True, gamma=0.9, lmbda=0.5, lr=lr, optim_alpha=0.99, optim_eps=0.00001, weight_decay=0,
        grad_norm_clip=10, target_update_interval=40, learner_log_interval=10 ** 9, device="cuda", use_cuda=True)


def run_episodes(envs, mac, new_batch, t_env, test_mode):
    """One batch of episodes (parallel_runner.py:84-186: the same update / select / update order and the same index lists)."""
    n = envs.n
    batch = new_batch()
    envs.reset()
    mac.init_hidden(batch_size=n)
    running = list(range(n))
    batch.update(envs.observe(running), bs=running, ts=0, mark_filled=True)
    ret = np.zeros(n)
    t = 0
    while running:
        actions = mac.select_actions(batch, t_ep=t, t_env=t_env, bs=running, test_mode=test_mode)
        batch.update({"actions": actions.unsqueeze(1)}, bs=running, ts=t, mark_filled=False)
        rew, term, done = envs.step(running, actions.cpu().numpy())
        for i, e in enumerate(running):
            ret[e] += rew[i]
        batch.update({"reward": rew[:, None], "terminated": term[:, None]}, bs=running, ts=t, mark_filled=False)
        # the observation after the last action is stored as well (the target network evaluates it), then the environment drops out
        batch.update(envs.observe(running), bs=running, ts=t + 1, mark_filled=True)
        if not test_mode:
            t_env += len(running)
        running = [e for i, e in enumerate(running) if not done[i]]
        t += 1
    return batch, t_env, ret / np.maximum(envs.t, 1)


@pytest.mark.parametrize("imagine,host_buffer", [(True, False), (False, True)])
def test_policy_learns_through_the_whole_cycle(imagine, host_buffer):
    before, after, lengths, trains, iters, wrapped, finite = _cycle(imagine, host_buffer)
    assert finite
    assert trains >= iters - 2 and wrapped                        # the ring wrapped many times
    assert len(lengths) >= 3, lengths                             # the max_t_filled() trim saw several lengths
    assert before < 0.45, before                                  # chance = 1 / C
    assert after > 0.85, (before, after, sorted(lengths))


def test_cycle_mechanics_short():
    """40 iterations of the same cycle (what the CPU tier runs on the emulator, where 800 take too long): every train() call happens,
    the ring buffer wraps, the max_t_filled() trim presents several lengths -- each trained through its untrimmed parent
    (refil_batch.t_limit) --, and the parameters stay finite. Learning itself is the 800-iteration test above."""
    before, after, lengths, trains, iters, wrapped, finite = _cycle(True, False, iters=40)
    assert finite and trains >= iters - 2 and wrapped and len(lengths) >= 2, (trains, wrapped, lengths)


def _cycle(imagine, host_buffer, iters=800, lr=0.005):
    from refil_amd.components.episode_buffer import EpisodeBatch, ReplayBuffer
    from refil_amd.controllers import REGISTRY as mac_REGISTRY
    from refil_amd.learners import REGISTRY as le_REGISTRY
    dev = torch.device("cuda", 0) if DEV == "cuda" else torch.device("cpu")
    torch.manual_seed(5)
    np.random.seed(5)
    rng = np.random.default_rng(5)
    n_envs, batch_size, buffer_size = 16, 32, 96
    scheme, groups, preprocess = _scheme()
    args = _args(imagine, anneal=int(0.5 * iters * n_envs * 4.5), lr=lr)
    buffer = ReplayBuffer(scheme, groups, buffer_size, T_LIMIT + 1, preprocess=preprocess,
                          device="cpu" if host_buffer else dev, sample_device=dev if (host_buffer and DEV == "cuda") else None)
    mac = mac_REGISTRY[args.mac](buffer.scheme, groups, args)
    from plugin_util import RecLogger
    logger = RecLogger()
    learner = le_REGISTRY[args.learner](mac, buffer.scheme, logger, args)
    if DEV == "cuda":
        learner.cuda()
    envs = MatchEnvs(n_envs, rng)

    def new_batch():
        return EpisodeBatch(scheme, groups, n_envs, T_LIMIT + 1, preprocess=preprocess, device=dev)

    def evaluate():
        return float(np.mean([run_episodes(envs, mac, new_batch, 0, True)[2].mean() for _ in range(4)]))

    before = evaluate()
    t_env, episode, lengths, trains = 0, 0, set(), 0
    for _ in range(iters):
        ep_batch, t_env, _ = run_episodes(envs, mac, new_batch, t_env, False)
        buffer.insert_episode_batch(ep_batch)
        episode += n_envs
        if buffer.can_sample(batch_size):
            sample = buffer.sample(batch_size)
            max_ep_t = sample.max_t_filled()                    # src/run.py:269-270
            sample = sample[:, :max_ep_t]
            lengths.add(int(max_ep_t))
            if sample.device != dev:
                sample.to(dev)
            learner.train(sample, t_env, episode)
            trains += 1
    torch.cuda.synchronize()
    after = evaluate()
    return (before, after, lengths, trains, iters, buffer.episodes_in_buffer == buffer_size and episode > 4 * buffer_size,
            bool(torch.isfinite(learner.flat_live).all()))


if __name__ == "__main__":
    import os, sys, time
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    for it, lr in [(400, 0.005), (800, 0.005), (800, 0.002)]:
        for im, hb in [(True, False), (False, True)]:
            t0 = time.time()
            r = _cycle(im, hb, iters=it, lr=lr)
            print(it, lr, im, hb, "before %.3f after %.3f" % r[:2], sorted(r[2]), "%.1fs" % (time.time() - t0), flush=True)
