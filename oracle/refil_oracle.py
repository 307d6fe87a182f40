"""CPU oracle for the REFIL learner hot path -- TEST INFRASTRUCTURE, NOT THE PRODUCT.

This file is a pure-PyTorch fp32 *restatement* of the algorithm the reference runs in
``QLearner.train`` (reference: src/learners/q_learner.py:66-201) and everything below it.
It exists only so that

  * ``tests/``                      can check the HIP path against it,
  * ``__graft_entry__.smoke()``     can check one tiny invocation against it,
  * ``bench.py``'s ``cpu_baseline`` can time it on the GPU box's host cores ("kind": "port").

Nothing under ``refil_amd/`` imports this module; the product path fails loudly when the HIP
library is missing rather than falling back to this code.

Parity pinning: the reference ships no tests / golden vectors for this path,
so this oracle is pinned against the reference *itself*, imported in the build container by
``tools/make_golden.py`` (which writes ``tests/golden/*.npz``); ``tests/test_oracle_golden.py``
re-checks the oracle against those committed vectors on every run.

The restatement is functional (weights are a flat ``dict[str, Tensor]`` keyed by the reference's
state_dict names) and removes the reference's redundant work (fc1/K/V are shared by the three
"imagine" copies, the b1/w_final/V hypernets are evaluated once) -- results are identical.

Conventions: B episodes, T1 = T+1 stored steps, ne entities, na agents (= first na entities),
ed raw entity features, A actions, E = ed (+A if entity_last_action), masks are 1 = masked.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
NEG_UNAVAIL = -9999999.0  # q_learner.py:118,124


@dataclass
class Cfg:
    """The hot-path hyper-parameters (reference: src/config/default.yaml:35-58, algs/refil.yaml)."""
    n_agents: int
    n_entities: int
    n_actions: int
    entity_shape: int
    attn_embed_dim: int = 128
    attn_n_heads: int = 4
    rnn_hidden_dim: int = 64
    hypernet_embed: int = 128
    mixing_embed_dim: int = 32
    entity_last_action: bool = True
    softmax_mixing_weights: bool = True
    mixer_non_lin: str = "elu"
    imagine: bool = True           # 'imagine' in args.agent  (q_learner.py:86)
    agent_ff: bool = False         # entity_attend_ff agents (entity_ff_agent.py) instead of the recurrent ones
    mixer_lin: bool = False        # lin_flex_qmix (flex_qmix.py:124-172) instead of flex_qmix
    mixer_vdn: bool = False        # VDNMixer (modules/mixers/vdn.py:9-10): q_tot = sum of the agents' Qs, no parameters
    mixer_none: bool = False       # args.mixer = None (q_learner.py:19-21,131): per-agent TD loss, no mixing network
    gt_obs_mask: bool = False      # entity_ff_agent.py:34-35
    pooling_type: Optional[str] = None   # 'mean' / 'max': EntityPoolingLayer instead of attention (default.yaml:43)
    train_gt_factors: bool = False       # q_learner.py:88: imagined groups = ground-truth factors (batch["gt_mask"])
    train_rand_gt_factors: bool = False  # q_learner.py:89: random split OR-ed with the ground-truth factors
    double_q: bool = True
    gamma: float = 0.99
    lmbda: float = 0.5
    lr: float = 0.0005
    optim_alpha: float = 0.99
    optim_eps: float = 0.00001
    weight_decay: float = 0.0
    grad_norm_clip: float = 10.0

    @property
    def in_dim(self) -> int:
        return self.entity_shape + (self.n_actions if self.entity_last_action else 0)


HYPERNETS = ("hyper_w_1", "hyper_w_final", "hyper_b_1", "V")  # flex_qmix.py:69-73
LIN_HYPERNETS = ("hyper_w_1", "V")                           # flex_qmix.py:133-134


# --------------------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------------------
def _in_trans_shapes(cfg: Cfg, prefix: str, w: int) -> Dict[str, Tuple[int, ...]]:
    """EntityAttentionLayer: in_trans [3w,w] without bias (attention.py:21); EntityPoolingLayer: [w,w] + bias (:93)."""
    if cfg.pooling_type is None:
        return {prefix + "attn.in_trans.weight": (3 * w, w)}
    return {prefix + "attn.in_trans.weight": (w, w), prefix + "attn.in_trans.bias": (w,)}


def agent_param_shapes(cfg: Cfg) -> Dict[str, Tuple[int, ...]]:
    """state_dict layout of EntityAttentionRNNAgent (entity_rnn_agent.py:8-25)."""
    d, H, A, E = cfg.attn_embed_dim, cfg.rnn_hidden_dim, cfg.n_actions, cfg.in_dim
    if cfg.agent_ff:                                  # EntityAttentionFFAgent (entity_ff_agent.py:8-23)
        return {"fc1.weight": (d, E), "fc1.bias": (d,), **_in_trans_shapes(cfg, "", d),
                "attn.out_trans.weight": (d, d), "attn.out_trans.bias": (d,),
                "fc2.weight": (A, d), "fc2.bias": (A,)}
    return {
        "fc1.weight": (d, E), "fc1.bias": (d,),
        **_in_trans_shapes(cfg, "", d),
        "attn.out_trans.weight": (d, d), "attn.out_trans.bias": (d,),
        "fc2.weight": (H, d), "fc2.bias": (H,),
        "rnn.weight_ih": (3 * H, H), "rnn.weight_hh": (3 * H, H),
        "rnn.bias_ih": (3 * H,), "rnn.bias_hh": (3 * H,),
        "fc3.weight": (A, H), "fc3.bias": (A,),
    }


def mixer_param_shapes(cfg: Cfg) -> Dict[str, Tuple[int, ...]]:
    """state_dict layout of FlexQMixer (flex_qmix.py:28-38,69-73)."""
    h, M, E = cfg.hypernet_embed, cfg.mixing_embed_dim, cfg.in_dim
    out = {}
    if cfg.mixer_vdn or cfg.mixer_none:
        return out
    for net in (LIN_HYPERNETS if cfg.mixer_lin else HYPERNETS):
        out[f"{net}.fc1.weight"] = (h, E)
        out[f"{net}.fc1.bias"] = (h,)
        out.update(_in_trans_shapes(cfg, net + ".", h))
        out[f"{net}.attn.out_trans.weight"] = (h, h)
        out[f"{net}.attn.out_trans.bias"] = (h,)
        out[f"{net}.fc2.weight"] = (M, h)
        out[f"{net}.fc2.bias"] = (M,)
    return out


def init_params(shapes: Dict[str, Tuple[int, ...]], seed: int, scale: float = 1.0) -> Dict[str, Tensor]:
    """Deterministic numpy-seeded weights (uniform +-scale/sqrt(fan_in)); used by tests/bench so that
    fixtures need not store weight tensors."""
    import numpy as np
    rng = np.random.default_rng(seed)
    out = {}
    for k, shp in shapes.items():
        fan_in = shp[-1] if len(shp) > 1 else shp[0]
        bound = scale / math.sqrt(fan_in)
        out[k] = torch.from_numpy(rng.uniform(-bound, bound, size=shp).astype("float32"))
    return out


# --------------------------------------------------------------------------------------
# input assembly
# --------------------------------------------------------------------------------------
def build_entity_inputs(cfg: Cfg, entities: Tensor, actions: Tensor) -> Tensor:
    """entities || one-hot(previous action) -> [B,T1,ne,E].

    Restates EntityMAC._build_inputs (entity_controller.py:13-27) and QLearner._get_mixer_ins
    (q_learner.py:50-60), which build the same tensor: zeros at t=0 and for non-agent entities.
    """
    if not cfg.entity_last_action:
        return entities
    B, T1, ne, _ = entities.shape
    la = torch.zeros(B, T1, ne, cfg.n_actions, dtype=entities.dtype)
    onehot = F.one_hot(actions[..., 0].long(), cfg.n_actions).to(entities.dtype)  # transforms.py:15-19
    la[:, 1:, :cfg.n_agents] = onehot[:, :-1]
    return torch.cat([entities, la], dim=3)


def draw_partition_bits(B: int, ne: int, generator: Optional[torch.Generator] = None) -> Tensor:
    """The per-episode random 2-way split, drawn with exactly the reference's two RNG calls
    (entity_rnn_agent.py:94-96): p = rand(B,1,1); groupA = bernoulli(p repeated over ne)."""
    p = torch.rand(B, 1, 1, generator=generator).repeat(1, 1, ne)
    return torch.bernoulli(p, generator=generator).to(torch.uint8).reshape(B, ne)


def imagine_masks(group_bits: Tensor, entity_mask0: Tensor) -> Tuple[Tensor, Tensor]:
    """Closed form of entity_rnn_agent.py:97-114.

    same(i,j) = both active at t=0 and in the same random group.
    returns (Wmask_noobs, Imask_noobs) as bool [B,ne,ne]:
        W = not same                       (attend only within one's group)
        I = same or inactive_i or inactive_j  (attend only across groups)
    """
    g = group_bits.bool()
    inact = entity_mask0.bool()
    act_pair = (~inact)[:, :, None] & (~inact)[:, None, :]
    same = act_pair & (g[:, :, None] == g[:, None, :])
    return ~same, same | ~act_pair


def group_masks(cfg: "Cfg", entity_mask: Tensor, group_bits: Optional[Tensor] = None,
                gt_mask: Optional[Tensor] = None, rand_gt: bool = False) -> Tuple[Tensor, Tensor, Tensor]:
    """(within, interact, active) bool [B,Tg,na,ne] for the query rows (agents):
    random split (Tg = 1; entity_rnn_agent.py:97-108 == entity_ff_agent.py:98-109): within = not same,
    interact = same; ground-truth factors (Tg = T1; entity_ff_agent.py:93-95): within = gt_mask,
    interact = not gt_mask; randomised ground-truth factors (rand_gt, Tg = T1; entity_ff_agent.py:111-114):
    within = (not same) | gt_mask, interact = not within. active[i,j] = inactive0_i | inactive0_j (":91 activeattnmask")."""
    na = cfg.n_agents
    inact = entity_mask[:, 0].bool()
    active = (inact[:, :na, None] | inact[:, None, :])[:, None]
    if rand_gt:
        Wm, _ = imagine_masks(group_bits, entity_mask[:, 0])
        W = Wm[:, None, :na, :] | gt_mask.bool()
        return W, ~W, active
    if gt_mask is not None:
        W = gt_mask.bool()
        return W, ~W, active
    Wm, _ = imagine_masks(group_bits, entity_mask[:, 0])
    W = Wm[:, None, :na, :]
    # NOTE: for the random split the reference's "interact" (before OR-ing active) is exactly `same`
    return W, ~W, active


# --------------------------------------------------------------------------------------
# EntityPoolingLayer (attention.py:82-132) with several pre-masks sharing the in_trans projection
# --------------------------------------------------------------------------------------
def pooling_variants(x1: Tensor, w_in: Tensor, b_in: Tensor, w_out: Tensor, b_out: Tensor, pooling_type: str,
                     pre_masks: List[Tensor], post_mask: Tensor) -> List[Tensor]:
    """x1 [R,ne,w]; pre_masks: bool [R,na,ne]; post_mask bool [R,na]. Masked entities enter the pool as ZEROS (:117-118)
    and the mean divides by ne, masked or not (:122-123)."""
    ents = x1 @ w_in.t() + b_in                                              # :110
    outs = []
    for pm in pre_masks:
        rep = ents[:, None, :, :].expand(-1, pm.shape[1], -1, -1).masked_fill(pm[:, :, :, None], 0.0)   # :114-118
        pooled = rep.max(dim=2)[0] if pooling_type == "max" else rep.mean(dim=2)                      # :120-123
        o = pooled @ w_out.t() + b_out                                       # :125
        outs.append(o.masked_fill(post_mask[:, :, None], 0.0))               # :127-128
    return outs


def entity_layer(cfg: "Cfg", p: Dict[str, Tensor], prefix: str, x1: Tensor, n_heads: int, pre_masks: List[Tensor],
                 post_mask: Tensor) -> List[Tensor]:
    """The `attn` sub-module of agents and hypernets: attention, or pooling when cfg.pooling_type is set."""
    if cfg.pooling_type is None:
        return attention_variants(x1, p[prefix + "attn.in_trans.weight"], p[prefix + "attn.out_trans.weight"],
                                  p[prefix + "attn.out_trans.bias"], n_heads, pre_masks, post_mask)
    return pooling_variants(x1, p[prefix + "attn.in_trans.weight"], p[prefix + "attn.in_trans.bias"],
                            p[prefix + "attn.out_trans.weight"], p[prefix + "attn.out_trans.bias"], cfg.pooling_type,
                            pre_masks, post_mask)


# --------------------------------------------------------------------------------------
# EntityAttentionLayer (attention.py:24-79) with several pre-masks sharing Q/K/V
# --------------------------------------------------------------------------------------
def attention_variants(x1: Tensor, w_in: Tensor, w_out: Tensor, b_out: Tensor, n_heads: int,
                       pre_masks: List[Tensor], post_mask: Tensor) -> List[Tensor]:
    """x1 [R,ne,w]; pre_masks: list of bool [R,na,ne] (True = masked); post_mask bool [R,na].
    Returns one [R,na,w] tensor per pre-mask."""
    R, ne, w = x1.shape
    na = post_mask.shape[1]
    hd = w // n_heads
    scale = torch.tensor(float(hd)).sqrt()          # attention.py:18-19 (fp32 buffer)
    qkv = x1 @ w_in.t()                             # attention.py:46 (no bias)
    q = qkv[:, :na, :w].reshape(R, na, n_heads, hd).permute(0, 2, 1, 3)       # queries: agents only (:48)
    k = qkv[:, :, w:2 * w].reshape(R, ne, n_heads, hd).permute(0, 2, 3, 1)
    v = qkv[:, :, 2 * w:].reshape(R, ne, n_heads, hd).permute(0, 2, 1, 3)
    logits = (q @ k) / scale                        # [R,heads,na,ne]        (:54)
    outs = []
    for pm in pre_masks:
        ml = logits.masked_fill(pm[:, None, :, :], float("-inf"))            # :55-57
        wts = torch.softmax(ml, dim=3)                                        # :58
        wts = torch.where(torch.isnan(wts), torch.zeros_like(wts), wts)       # :60 fully-masked rows -> 0
        o = (wts @ v).permute(0, 2, 1, 3).reshape(R, na, w)                  # :61-64
        o = o @ w_out.t() + b_out                                             # :65
        outs.append(o.masked_fill(post_mask[:, :, None], 0.0))                # :66-67
    return outs


# --------------------------------------------------------------------------------------
# agent (entity_rnn_agent.py:31-64, 87-126)
# --------------------------------------------------------------------------------------
def gru_cell(x: Tensor, h: Tensor, w_ih: Tensor, w_hh: Tensor, b_ih: Tensor, b_hh: Tensor) -> Tensor:
    """torch.nn.GRUCell semantics (gate order r,z,n) as used at entity_rnn_agent.py:53."""
    gi = x @ w_ih.t() + b_ih
    gh = h @ w_hh.t() + b_hh
    H = h.shape[1]
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    return (1.0 - z) * n + z * h


def agent_forward(cfg: Cfg, p: Dict[str, Tensor], xe: Tensor, obs_mask: Tensor, entity_mask: Tensor,
                  h0: Optional[Tensor] = None, group_bits: Optional[Tensor] = None,
                  gt_mask: Optional[Tensor] = None, use_gt_factors: bool = False, use_rand_gt_factors: bool = False):
    """Returns (q [G,B,T1,na,A], hs, groups) where G = 3 when imagining (group_bits or use_gt_factors) else 1.

    Copy 0 = real obs mask, copy 1 = within-group, copy 2 = between-group (entity_rnn_agent.py:116-124,
    entity_ff_agent.py:122-129). ``groups`` = (Wmask_noobs, Imask_noobs) bool [B,Tg,na,ne] or None.
    Recurrent agent: fc1 -> attn -> relu(fc2) -> GRU -> fc3; feed-forward agent (cfg.agent_ff,
    entity_ff_agent.py:29-57): fc1 -> relu(attn) -> fc2."""
    B, T1, ne, E = xe.shape
    na, H = cfg.n_agents, cfg.rnn_hidden_dim
    R = B * T1
    om = (gt_mask if cfg.gt_obs_mask else obs_mask).bool()[:, :, :na, :]      # attention.py:44 slices to the queries
    pre = [om]
    groups = None
    if group_bits is not None or use_gt_factors:
        assert not (use_gt_factors and use_rand_gt_factors)                                # entity_ff_agent.py:112
        W, I, active = group_masks(cfg, entity_mask, group_bits, gt_mask if (use_gt_factors or use_rand_gt_factors) else None,
                                   rand_gt=use_rand_gt_factors)
        groups = (W | active, I | active)
        pre.append(W | om)
        pre.append(I | om)
    agent_mask = entity_mask.bool()[:, :, :na]
    x1 = torch.relu(xe.reshape(R, ne, E) @ p["fc1.weight"].t() + p["fc1.bias"])           # :38
    x2s = entity_layer(cfg, p, "", x1, cfg.attn_n_heads,
                       [m.expand(B, T1, na, ne).reshape(R, na, ne) for m in pre], agent_mask.reshape(R, na))
    G = len(x2s)
    x2 = torch.stack(x2s, 0)                                                               # [G,R,na,d]
    if cfg.agent_ff:
        hs = torch.relu(x2).reshape(G, B, T1, na, -1)                                      # entity_ff_agent.py:44
        q = hs @ p["fc2.weight"].t() + p["fc2.bias"]                                       # :49
        return q.masked_fill(agent_mask[None, :, :, :, None], 0.0), hs, groups            # :52
    x3 = torch.relu(x2 @ p["fc2.weight"].t() + p["fc2.bias"]).reshape(G, B, T1, na, H)     # :46-47
    h = torch.zeros(G * B * na, H) if h0 is None else h0.reshape(G * B * na, H)
    hs = []
    for t in range(T1):                                                                    # :50-55
        h = gru_cell(x3[:, :, t].reshape(G * B * na, H), h, p["rnn.weight_ih"], p["rnn.weight_hh"],
                     p["rnn.bias_ih"], p["rnn.bias_hh"])
        hs.append(h.reshape(G, B, na, H))
    hs = torch.stack(hs, 2)                                                                # [G,B,T1,na,H]
    q = hs @ p["fc3.weight"].t() + p["fc3.bias"]                                           # :57
    q = q.masked_fill(agent_mask[None, :, :, :, None], 0.0)                                # :60
    return q, hs, groups


# --------------------------------------------------------------------------------------
# mixer (flex_qmix.py:40-57, 79-121)
# --------------------------------------------------------------------------------------
def hypernet_x3(cfg: Cfg, p: Dict[str, Tensor], net: str, xe: Tensor, entity_mask: Tensor,
                attn_masks: Optional[List[Tensor]] = None) -> List[Tensor]:
    """AttentionHyperNet up to the masked fc2 output [R,na,M] (flex_qmix.py:41-50), one per attn mask.
    attn_masks: bool [R,ne,ne] (only the first na rows are used) or None for the default mask."""
    R, ne, _ = xe.shape
    na = cfg.n_agents
    em = entity_mask.bool()
    am = em[:, :na]
    if attn_masks is None:
        pre = [am[:, :, None] | em[:, None, :]]                      # flex_qmix.py:43-46
    else:
        pre = [m[:, :na, :] for m in attn_masks]   # [R,na,ne] (or [R,ne,ne]: only the query rows are used)
    x1 = torch.relu(xe @ p[f"{net}.fc1.weight"].t() + p[f"{net}.fc1.bias"])
    x2s = entity_layer(cfg, p, net + ".", x1, cfg.attn_n_heads, pre, am)
    outs = []
    for x2 in x2s:
        x3 = x2 @ p[f"{net}.fc2.weight"].t() + p[f"{net}.fc2.bias"]
        outs.append(x3.masked_fill(am[:, :, None], 0.0))
    return outs


def _mix_w(cfg: Cfg, x: Tensor) -> Tensor:
    return torch.softmax(x, dim=-1) if cfg.softmax_mixing_weights else x.abs()   # flex_qmix.py:102-105


def _non_lin(cfg: Cfg, x: Tensor) -> Tensor:
    return torch.tanh(x) if cfg.mixer_non_lin == "tanh" else F.elu(x)            # flex_qmix.py:75-77


def mixer_forward(cfg: Cfg, p: Dict[str, Tensor], agent_qs: Tensor, xe: Tensor, entity_mask: Tensor,
                  agent_qs_imagine: Optional[Tensor] = None,
                  groups: Optional[Tuple[Tensor, Tensor]] = None, ret_ingroup_prop: bool = False):
    """FlexQMixer.forward / LinearFlexQMixer.forward for the real call and (optionally) the imagined call.

    agent_qs [B,T,na]; xe [B,T,ne,E]; entity_mask [B,T,ne]; agent_qs_imagine [B,T,2na] = cat(caqW, caqI)
    (q_learner.py:96); groups = (Wmask, Imask) bool [B,Tg,na,ne], Tg in {1,T}.
    Returns q_tot [B,T,1] (and q_tot_imagine (and ingroup_prop))."""
    B, T, ne, E = xe.shape
    na, M = cfg.n_agents, cfg.mixing_embed_dim
    R = B * T
    if cfg.mixer_none:                                                                     # q_learner.py:131 not taken
        assert agent_qs_imagine is None, "mixer=None: caq_imagine [B,T,2na] does not broadcast against targets [B,T,na]"
        return agent_qs
    if cfg.mixer_vdn:                                                                      # vdn.py:9-10
        q_tot = agent_qs.sum(dim=2, keepdim=True)
        return q_tot if agent_qs_imagine is None else (q_tot, agent_qs_imagine.sum(dim=2, keepdim=True))
    xr = xe.reshape(R, ne, E)
    em = entity_mask.reshape(R, ne).bool()
    v = hypernet_x3(cfg, p, "V", xr, em)[0].mean(dim=(1, 2))                               # mode 'scalar' :55-56
    em_d = (em[:, :na, None] | em[:, None, :])
    masks = None
    if agent_qs_imagine is not None:
        masks = [em_d] + [m.expand(B, T, na, ne).reshape(R, na, ne) for m in groups]
    w1s = hypernet_x3(cfg, p, "hyper_w_1", xr, em, masks)

    if cfg.mixer_lin:                                                                      # flex_qmix.py:136-172
        def lin(qs, w1):                      # w1: alt_vector mode = mean over the embedding dim (:53-54)
            w = _mix_w(cfg, w1)
            return ((qs * w).sum(dim=1) + v).reshape(B, T, 1), w
        q_tot, _ = lin(agent_qs.reshape(R, na), w1s[0].mean(dim=2))
        if agent_qs_imagine is None:
            return q_tot
        q_im, w = lin(agent_qs_imagine.reshape(R, 2 * na), torch.cat([w1s[1].mean(dim=2), w1s[2].mean(dim=2)], dim=1))
        if ret_ingroup_prop:
            return q_tot, q_im, w[:, :na].sum(dim=1).mean()                               # :166-170
        return q_tot, q_im

    b1 = hypernet_x3(cfg, p, "hyper_b_1", xr, em)[0].mean(dim=1).reshape(R, 1, M)          # mode 'vector' :51-52
    w_final = _mix_w(cfg, hypernet_x3(cfg, p, "hyper_w_final", xr, em)[0].mean(dim=1)).reshape(R, M, 1)

    def mix(qs, w1):
        hidden = _non_lin(cfg, torch.bmm(qs, _mix_w(cfg, w1)) + b1)                        # :107
        return (torch.bmm(hidden, w_final) + v.reshape(R, 1, 1)).reshape(B, T, 1)         # :118-120

    q_tot = mix(agent_qs.reshape(R, 1, na), w1s[0])
    if agent_qs_imagine is None:
        return q_tot
    q_tot_im = mix(agent_qs_imagine.reshape(R, 1, 2 * na), torch.cat([w1s[1], w1s[2]], dim=1))  # :85-94
    return q_tot, q_tot_im


# --------------------------------------------------------------------------------------
# one learner step (q_learner.py:66-201)
# --------------------------------------------------------------------------------------
@dataclass
class StepOut:
    loss: Tensor = None            # the lambda-blended loss (what the reference logs as "loss", :172,185)
    q_loss: Tensor = None          # the plain TD loss before blending (:165)
    im_loss: Optional[Tensor] = None
    q: Tensor = None               # [G,B,T1,na,A]
    chosen_q: Tensor = None        # [G,B,T,na]
    target_max_q: Tensor = None    # [B,T,na]
    q_tot: Tensor = None           # [B,T,1]
    q_tot_imagine: Optional[Tensor] = None
    target_q_tot: Tensor = None
    targets: Tensor = None
    mask: Tensor = None
    stats: Dict[str, float] = field(default_factory=dict)


def learner_forward(cfg: Cfg, agent_p, mixer_p, tgt_agent_p, tgt_mixer_p, batch: Dict[str, Tensor],
                    group_bits: Optional[Tensor]) -> StepOut:
    na = cfg.n_agents
    rewards = batch["reward"][:, :-1]
    actions = batch["actions"][:, :-1].long()
    terminated = batch["terminated"][:, :-1].float()
    mask = batch["filled"][:, :-1].float().clone()
    mask[:, 1:] = mask[:, 1:] * (1 - terminated[:, :-1])                               # :68-72
    avail = batch["avail_actions"]
    xe = build_entity_inputs(cfg, batch["entities"], batch["actions"])
    out = StepOut()

    gt = batch.get("gt_mask")
    q, _, groups = agent_forward(cfg, agent_p, xe, batch["obs_mask"], batch["entity_mask"],
                                 group_bits=group_bits if cfg.imagine else None, gt_mask=gt,
                                 use_gt_factors=cfg.imagine and cfg.train_gt_factors,
                                 use_rand_gt_factors=cfg.imagine and cfg.train_rand_gt_factors)          # q_learner.py:87-89
    G = q.shape[0]
    chosen = torch.gather(q[:, :, :-1], 4, actions[None].expand(G, -1, -1, -1, -1)).squeeze(4)   # :91,109
    with torch.no_grad():
        tq, _, _ = agent_forward(cfg, tgt_agent_p, xe, batch["obs_mask"], batch["entity_mask"], gt_mask=gt)
        tq = tq[0, :, 1:].clone()
        tq[avail[:, 1:] == 0] = NEG_UNAVAIL                                             # :118
        if cfg.double_q:
            live = q[0].detach().clone()
            live[avail == 0] = NEG_UNAVAIL
            amax = live[:, 1:].max(dim=3, keepdim=True)[1]                             # :121-126
            tmax = torch.gather(tq, 3, amax).squeeze(3)
        else:
            tmax = tq.max(dim=3)[0]
        tq_tot = mixer_forward(cfg, tgt_mixer_p, tmax, xe[:, 1:], batch["entity_mask"][:, 1:])   # :154
    if cfg.imagine:
        caq_im = torch.cat([chosen[1], chosen[2]], dim=2)                              # :96
        groups = tuple(g if g.shape[1] == 1 else g[:, :-1] for g in groups)            # :137 "don't need last timestep"
        q_tot, q_tot_im = mixer_forward(cfg, mixer_p, chosen[0], xe[:, :-1], batch["entity_mask"][:, :-1],
                                        caq_im, groups)
    else:
        q_tot = mixer_forward(cfg, mixer_p, chosen[0], xe[:, :-1], batch["entity_mask"][:, :-1])
        q_tot_im = None
    targets = rewards + cfg.gamma * (1 - terminated) * tq_tot                           # :157
    mask = mask.expand_as(q_tot)                                                        # :161 (a no-op with a mixer: [B,T,1])
    td = (q_tot - targets.detach()) * mask
    msum = mask.sum()
    q_loss = (td ** 2).sum() / msum                                                     # :160-165
    loss = q_loss
    if cfg.imagine:
        im_td = (q_tot_im - targets.detach()) * mask
        im_loss = (im_td ** 2).sum() / msum                                             # :167-171
        loss = (1 - cfg.lmbda) * q_loss + cfg.lmbda * im_loss                           # :172
        out.im_loss = im_loss
    out.loss, out.q_loss = loss, q_loss
    out.q, out.chosen_q, out.target_max_q = q, chosen, tmax
    out.q_tot, out.q_tot_imagine, out.target_q_tot, out.targets, out.mask = q_tot, q_tot_im, tq_tot, targets, mask
    me = msum.item()
    out.stats = {                                                                       # :185-195 (incl. the /n_agents quirk)
        "td_error_abs": td.abs().sum().item() / me,
        "q_taken_mean": (q_tot * mask).sum().item() / (me * na),
        "target_mean": (targets * mask).sum().item() / (me * na),
    }
    return out


def clip_and_rmsprop(cfg: Cfg, params: List[Tensor], grads: List[Tensor], square_avg: List[Tensor]) -> float:
    """clip_grad_norm_ + torch.optim.RMSprop (no momentum, not centered) as configured at
    q_learner.py:37-38,177-178. In place. Returns the pre-clip global grad norm."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = min(1.0, cfg.grad_norm_clip / (total.item() + 1e-6))
    for p, g, sq in zip(params, grads, square_avg):
        g = g * coef
        if cfg.weight_decay != 0:
            g = g + cfg.weight_decay * p
        sq.mul_(cfg.optim_alpha).addcmul_(g, g, value=1 - cfg.optim_alpha)
        p.addcdiv_(g, sq.sqrt().add_(cfg.optim_eps), value=-cfg.lr)
    return total.item()


def train_step(cfg: Cfg, agent_p, mixer_p, tgt_agent_p, tgt_mixer_p, batch, group_bits,
               square_avg: Optional[Dict[str, Tensor]] = None):
    """Full step: forward, autograd backward (oracle only -- the product has hand-written backward
    kernels), clip, RMSprop. Mutates agent_p / mixer_p / square_avg. Returns (StepOut, grads, grad_norm)."""
    names = [("agent", k) for k in agent_p] + [("mixer", k) for k in mixer_p]     # q_learner.py:16,34 order
    leaves = []
    for which, k in names:
        d = agent_p if which == "agent" else mixer_p
        d[k] = d[k].detach().clone().requires_grad_(True)
        leaves.append(d[k])
    out = learner_forward(cfg, agent_p, mixer_p, tgt_agent_p, tgt_mixer_p, batch, group_bits)
    grads = torch.autograd.grad(out.loss, leaves, allow_unused=True)
    grads = [torch.zeros_like(p) if g is None else g for p, g in zip(leaves, grads)]
    if square_avg is None:
        square_avg = {}
    sq_list = []
    for (which, k), p in zip(names, leaves):
        key = f"{which}.{k}"
        if key not in square_avg:
            square_avg[key] = torch.zeros_like(p)
        sq_list.append(square_avg[key])
    with torch.no_grad():
        new = [p.detach().clone() for p in leaves]
        gnorm = clip_and_rmsprop(cfg, new, [g.clone() for g in grads], sq_list)
    for (which, k), p in zip(names, new):
        (agent_p if which == "agent" else mixer_p)[k] = p
    gdict = {f"{which}.{k}": g for (which, k), g in zip(names, grads)}
    return out, gdict, gnorm
