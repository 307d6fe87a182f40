"""Builds the PyMARL-style objects (args namespace, EpisodeBatch, MAC, learner) of refil_amd the way
src/run.py:173-212 builds the reference's."""
import types

import torch as th


class RecLogger:
    def __init__(self):
        self.stats = {}
        self.infos = []
        self.console_logger = types.SimpleNamespace(info=lambda *a, **k: self.infos.append(a))

    def log_stat(self, key, value, t):
        self.stats[key] = float(value)


def make_args(cfg, imagine=None, **over):
    imagine = cfg.imagine if imagine is None else imagine
    kind = "ff" if cfg.agent_ff else "rnn"
    a = types.SimpleNamespace(
        agent=("imagine_entity_attend_" if imagine else "entity_attend_") + kind,
        mac="entity_mac", learner="q_learner", mixer=None if cfg.mixer_none else ("vdn" if cfg.mixer_vdn else ("lin_flex_qmix" if cfg.mixer_lin else "flex_qmix")),
        agent_output_type="q",
        train_gt_factors=False, train_rand_gt_factors=False, test_gt_factors=False, gt_obs_mask=cfg.gt_obs_mask,
        action_selector="epsilon_greedy", epsilon_start=1.0, epsilon_finish=0.05, epsilon_anneal_time=500000,
        n_agents=cfg.n_agents, n_actions=cfg.n_actions, n_entities=cfg.n_entities, entity_shape=cfg.entity_shape,
        entity_scheme=True, entity_last_action=cfg.entity_last_action, gt_mask_avail=False,
        attn_embed_dim=cfg.attn_embed_dim, attn_n_heads=cfg.attn_n_heads, rnn_hidden_dim=cfg.rnn_hidden_dim,
        hypernet_embed=cfg.hypernet_embed, mixing_embed_dim=cfg.mixing_embed_dim,
        softmax_mixing_weights=cfg.softmax_mixing_weights, pooling_type=cfg.pooling_type, double_q=cfg.double_q, gamma=cfg.gamma,
        lmbda=cfg.lmbda, lr=cfg.lr, optim_alpha=cfg.optim_alpha, optim_eps=cfg.optim_eps, weight_decay=cfg.weight_decay,
        grad_norm_clip=cfg.grad_norm_clip, target_update_interval=200, learner_log_interval=1, device="cuda",
        use_cuda=True)
    if cfg.mixer_non_lin != "elu":
        a.mixer_non_lin = cfg.mixer_non_lin
    for k, v in over.items():
        setattr(a, k, v)
    return a


def make_episode_batch(cfg, data, device="cpu"):
    from refil_amd.components.episode_buffer import EpisodeBatch
    from refil_amd.components.transforms import OneHot
    B, T1 = data["entities"].shape[:2]
    scheme = {
        "entities": {"vshape": cfg.entity_shape, "group": "entities"},
        "obs_mask": {"vshape": cfg.n_entities, "group": "entities", "dtype": th.uint8},
        "entity_mask": {"vshape": cfg.n_entities, "dtype": th.uint8},
        "actions": {"vshape": (1,), "group": "agents", "dtype": th.long},
        "avail_actions": {"vshape": (cfg.n_actions,), "group": "agents", "dtype": th.int},
        "reward": {"vshape": (1,)},
        "terminated": {"vshape": (1,), "dtype": th.uint8},
    }
    if "gt_mask" in data:                                  # src/run.py:187-188
        scheme["gt_mask"] = {"vshape": cfg.n_entities, "group": "agents", "dtype": th.uint8}
    groups = {"agents": cfg.n_agents, "entities": cfg.n_entities}
    preprocess = {"actions": ("actions_onehot", [OneHot(out_dim=cfg.n_actions)])}
    batch = EpisodeBatch(scheme, groups, B, T1, preprocess=preprocess, device=device)
    batch.update({k: v for k, v in data.items() if k != "filled"}, mark_filled=False)
    batch.data.transition_data["filled"].copy_(data["filled"])
    return batch, groups
