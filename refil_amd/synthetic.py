"""Seeded synthetic replay batches with the shapes / value distributions of the reference's
variable-army StarCraft env. Used by bench.py, tests and tools/make_golden.py.

Shape law (reference: src/envs/starcraft2/starcraft2custom.py:370-376,1137-1150):
    na = ne/2,  A = 6 + ne/2,  entity_shape = ne + (A-2) + 2 + 2 + 2 + 4
Field dtypes follow the entity scheme of src/run.py:178-192 (uint8 masks, int64 actions,
int32 avail_actions, int64 filled).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch


def sc2_shape_law(ne: int) -> Dict[str, int]:
    na = ne // 2
    A = 6 + ne // 2
    return {"n_entities": ne, "n_agents": na, "n_actions": A, "entity_shape": ne + (A - 2) + 10}


def make_batch(B: int, T: int, ne: int, seed: int = 0, na: Optional[int] = None, A: Optional[int] = None,
               ed: Optional[int] = None, min_active: int = 3, death_p: float = 0.02,
               full_length: bool = False) -> Dict[str, torch.Tensor]:
    """Returns a dict of CPU tensors shaped [B, T+1, ...] like EpisodeBatch.data.transition_data."""
    law = sc2_shape_law(ne)
    na = law["n_agents"] if na is None else na
    A = law["n_actions"] if A is None else A
    ed = law["entity_shape"] if ed is None else ed
    assert A >= 7 or A >= 2, "need at least no-op + stop"
    rng = np.random.default_rng(seed)
    T1 = T + 1
    n_en = ne - na
    ents = np.zeros((B, T1, ne, ed), np.float32)
    obs_mask = np.ones((B, T1, ne, ne), np.uint8)
    ent_mask = np.ones((B, T1, ne), np.uint8)
    avail = np.zeros((B, T1, na, A), np.int32)
    actions = np.zeros((B, T1, na, 1), np.int64)
    reward = rng.uniform(0.0, 0.5, size=(B, T1, 1)).astype(np.float32)
    terminated = np.zeros((B, T1, 1), np.uint8)
    filled = np.zeros((B, T1, 1), np.int64)
    n_move = min(4, max(A - 2, 0))
    for b in range(B):
        lo = min(min_active, na)
        n_ag = int(rng.integers(lo, na + 1))
        n_e = int(rng.integers(min(lo, n_en), n_en + 1)) if n_en > 0 else 0
        active = np.zeros(ne, bool)
        active[:n_ag] = True
        active[na:na + n_e] = True
        ent_mask[b, :, :] = (~active).astype(np.uint8)[None]
        death_t = np.where(active, rng.geometric(death_p, size=ne), 0)       # first step at which the unit is dead
        L = T if (b == 0 or full_length) else int(rng.integers(max(T // 2, 1), T + 1))
        filled[b, :L + 1] = 1
        if rng.random() < 0.8:
            terminated[b, L - 1] = 1
        tags = rng.integers(0, min(ne, ed), size=ne)
        for t in range(T1):
            alive = active & (t < death_t)
            # entity features
            for e in np.nonzero(alive)[0]:
                row = ents[b, t, e]
                row[tags[e]] = 1.0
                k = min(ne, ed)
                if e < na and k + (A - 2) <= ed:
                    row[k:k + A - 2] = (rng.random(A - 2) < 0.5)
                tail = ed - 10 if ed >= 10 else 0
                rest = ed - tail
                if rest > 0:
                    vals = np.concatenate([np.eye(2)[rng.integers(0, 2)], rng.uniform(0, 1, 4),
                                           rng.uniform(-1, 1, 4)])[:rest]
                    row[tail:tail + len(vals)] = vals
            # observability: 1 = cannot see
            vis = (rng.random((ne, ne)) < 0.3)
            blk = np.triu(vis[:na, :na], 1)
            vis[:na, :na] = blk | blk.T
            m = vis | ~alive[:, None] | ~alive[None, :]
            # dead-but-present units keep a masked row/column except the diagonal; padded ones are fully masked
            np.fill_diagonal(m, ~active)
            obs_mask[b, t] = m.astype(np.uint8)
            # available actions + chosen action
            for i in range(na):
                if not alive[i]:
                    avail[b, t, i, 0] = 1
                    actions[b, t, i, 0] = 0
                    continue
                av = np.zeros(A, np.int32)
                av[1] = 1
                av[2:2 + n_move] = rng.random(n_move) < 0.9
                if A > 2 + n_move:
                    av[2 + n_move:] = rng.random(A - 2 - n_move) < 0.3
                avail[b, t, i] = av
                actions[b, t, i, 0] = rng.choice(np.nonzero(av)[0])
    return {
        "entities": torch.from_numpy(ents),
        "obs_mask": torch.from_numpy(obs_mask),
        "entity_mask": torch.from_numpy(ent_mask),
        "actions": torch.from_numpy(actions),
        "avail_actions": torch.from_numpy(avail),
        "reward": torch.from_numpy(reward),
        "terminated": torch.from_numpy(terminated),
        "filled": torch.from_numpy(filled),
    }


def make_batch_fast(B: int, T: int, ne: int, seed: int = 0, device="cpu", na: Optional[int] = None,
                    A: Optional[int] = None) -> Dict[str, torch.Tensor]:
    """Vectorised generator with the same field distributions for the full bench sizes (the loop
    version above is O(B*T*ne) Python and is meant for fixture-sized batches). na / A override the shape law's agent and
    action counts (the real 3-8MMM scenario: ne=16, na=8, A=22 -- medivacs add heal targets, starcraft2custom.py:372-374);
    the entity width follows the feature layout: ne + (A-2) + 10."""
    law = sc2_shape_law(ne)
    na = law["n_agents"] if na is None else na
    A = law["n_actions"] if A is None else A
    ed = ne + (A - 2) + 10
    assert 1 <= na <= ne and A >= 7
    rng = np.random.default_rng(seed)
    T1 = T + 1
    n_ag = rng.integers(min(3, na), na + 1, size=B)
    n_en = rng.integers(min(3, ne - na), ne - na + 1, size=B)
    idx = np.arange(ne)[None, :]
    active = (idx < n_ag[:, None]) | ((idx >= na) & (idx < na + n_en[:, None]))                 # [B,ne]
    death_t = np.where(active, rng.geometric(0.02, size=(B, ne)), 0)
    tt = np.arange(T1)[None, :, None]
    alive = active[:, None, :] & (tt < death_t[:, None, :])                                       # [B,T1,ne]
    ent_mask = np.broadcast_to((~active)[:, None, :], (B, T1, ne)).astype(np.uint8).copy()
    ents = np.zeros((B, T1, ne, ed), np.float32)
    tags = rng.integers(0, ne, size=(B, ne))
    bi, ei = np.meshgrid(np.arange(B), np.arange(ne), indexing="ij")
    ents[bi, :, ei, tags] = 1.0
    ents[:, :, :na, ne:ne + A - 2] = rng.random((B, T1, na, A - 2)) < 0.5
    ents[..., ne + A - 2:ne + A] = np.eye(2, dtype=np.float32)[rng.integers(0, 2, size=(B, T1, ne))]
    ents[..., ne + A:ne + A + 4] = rng.uniform(0, 1, size=(B, T1, ne, 4))
    ents[..., ne + A + 4:] = rng.uniform(-1, 1, size=(B, T1, ne, 4))
    ents *= alive[..., None]
    vis = rng.random((B, T1, ne, ne)) < 0.3
    blk = np.triu(vis[:, :, :na, :na], 1)
    vis[:, :, :na, :na] = blk | np.swapaxes(blk, 2, 3)
    m = vis | ~alive[:, :, :, None] | ~alive[:, :, None, :]
    di = np.arange(ne)
    m[:, :, di, di] = np.broadcast_to((~active)[:, None, :], (B, T1, ne))
    avail = np.zeros((B, T1, na, A), np.int32)
    avail[..., 1] = 1
    avail[..., 2:6] = rng.random((B, T1, na, 4)) < 0.9
    avail[..., 6:] = rng.random((B, T1, na, A - 6)) < 0.3
    ag_alive = alive[:, :, :na]
    avail *= ag_alive[..., None]
    avail[..., 0] = ~ag_alive
    # uniform choice among available actions
    score = rng.random((B, T1, na, A)) * avail
    actions = score.argmax(-1)[..., None].astype(np.int64)
    L = rng.integers(max(T // 2, 1), T + 1, size=B)
    L[0] = T
    filled = (np.arange(T1)[None, :] <= L[:, None]).astype(np.int64)[..., None]
    terminated = np.zeros((B, T1, 1), np.uint8)
    term = rng.random(B) < 0.8
    terminated[np.arange(B)[term], L[term] - 1, 0] = 1
    out = {
        "entities": torch.from_numpy(ents),
        "obs_mask": torch.from_numpy(m.astype(np.uint8)),
        "entity_mask": torch.from_numpy(ent_mask),
        "actions": torch.from_numpy(actions),
        "avail_actions": torch.from_numpy(avail),
        "reward": torch.from_numpy(rng.uniform(0, 0.5, size=(B, T1, 1)).astype(np.float32)),
        "terminated": torch.from_numpy(terminated),
        "filled": torch.from_numpy(filled),
    }
    return {k: v.to(device) for k, v in out.items()}


def make_batch_group_matching(B: int, T: int, seed: int = 0, ne: int = 8, ed: int = 16, A: int = 3,
                              n_groups: int = 3) -> Dict[str, torch.Tensor]:
    """Synthetic replay with the shapes of the reference's GroupMatching env (BASELINE.json configs[0];
    src/envs/group_matching/group_matching.py: every entity is an agent, one-hot entity features, ground-truth group
    masks, 3 actions, fixed-length episodes) at the replay sizes of algs/refil_group_matching.yaml (B=8, T=50)."""
    rng = np.random.default_rng(seed)
    T1 = T + 1
    n_act = rng.integers(max(ne // 2, 2), ne + 1, size=B)
    active = np.arange(ne)[None, :] < n_act[:, None]                                       # [B,ne]
    ents = np.zeros((B, T1, ne, ed), np.float32)
    ids = rng.integers(0, ed, size=(B, T1, ne))
    np.put_along_axis(ents, ids[..., None], 1.0, axis=3)
    ents *= active[:, None, :, None]
    grp = rng.integers(0, n_groups, size=(B, ne))
    same = grp[:, :, None] == grp[:, None, :]
    inact = ~(active[:, :, None] & active[:, None, :])
    gt = (~same | inact).astype(np.uint8)                                                    # 1 = different group (or padded)
    vis = rng.random((B, T1, ne, ne)) < 0.2
    vis = vis | np.swapaxes(vis, 2, 3)
    om = vis | inact[:, None]
    di = np.arange(ne)
    om[:, :, di, di] = np.broadcast_to((~active)[:, None, :], (B, T1, ne))
    avail = np.ones((B, T1, ne, A), np.int32)
    actions = rng.integers(0, A, size=(B, T1, ne, 1)).astype(np.int64)
    filled = np.ones((B, T1, 1), np.int64)
    terminated = np.zeros((B, T1, 1), np.uint8)
    terminated[:, T - 1, 0] = 1
    return {
        "entities": torch.from_numpy(ents),
        "obs_mask": torch.from_numpy(om.astype(np.uint8)),
        "entity_mask": torch.from_numpy(np.broadcast_to((~active)[:, None, :], (B, T1, ne)).astype(np.uint8).copy()),
        "gt_mask": torch.from_numpy(np.broadcast_to(gt[:, None], (B, T1, ne, ne)).copy()),
        "actions": torch.from_numpy(actions),
        "avail_actions": torch.from_numpy(avail),
        "reward": torch.from_numpy(rng.uniform(0, 1.0, size=(B, T1, 1)).astype(np.float32)),
        "terminated": torch.from_numpy(terminated),
        "filled": torch.from_numpy(filled),
    }
