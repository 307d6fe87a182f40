"""Exploration parity of the acting path's action selectors against the reference's own draws (fixture written by
tools/make_golden.py:run_selector_case from src/components/action_selectors.py under a seeded global CPU generator):
epsilon-greedy at three points of the schedule and in test mode, multinomial sampling and its greedy test mode."""
import os
import types

import numpy as np
import torch

from refil_amd.components.action_selectors import REGISTRY

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "action_selectors.npz")


def test_action_selectors_match_reference_draws():
    z = np.load(GOLDEN, allow_pickle=False)
    args = types.SimpleNamespace(epsilon_start=1.0, epsilon_finish=0.05, epsilon_anneal_time=1000, test_greedy=True)
    q, avail, pol = torch.from_numpy(z["q"]), torch.from_numpy(z["avail"]), torch.from_numpy(z["policy"])
    seen = {"epsilon_greedy": 0, "multinomial": 0}
    explored = 0
    for key in z.files:
        if not key.endswith(".meta"):
            continue
        kind, i = key.split(".")[0], key.split(".")[1]
        seed, t_env, test_mode = (int(v) for v in z[key])
        sel = REGISTRY[kind](args)
        torch.manual_seed(seed)
        got = sel.select_action(q if kind == "epsilon_greedy" else pol, avail, t_env, test_mode=bool(test_mode))
        ref = torch.from_numpy(z[f"{kind}.{i}.actions"])
        assert got.dtype == torch.int64 and torch.equal(got, ref), (kind, t_env, test_mode)
        assert abs(float(sel.epsilon) - float(z[f"{kind}.{i}.epsilon"])) < 1e-12
        assert (avail.gather(2, got[..., None]) == 1).all()          # never an unavailable action
        seen[kind] += 1
        if kind == "epsilon_greedy" and not test_mode:
            greedy = q.masked_fill(avail == 0, -float("inf")).max(dim=2)[1]
            explored += int((got != greedy).sum())
    assert seen == {"epsilon_greedy": 4, "multinomial": 4}
    assert explored > 0, "the fixture must exercise the exploration branch"
