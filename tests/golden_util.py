"""Helpers to load tests/golden/*.npz (written by tools/make_golden.py from the real reference)."""
import ast
import os

import numpy as np
import torch

from oracle import refil_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["refil_tiny", "qmix_atten_tiny", "refil_abs_masked", "refil_odd", "refil_mid", "refil_vdn_tiny",
         "refil_tanh", "refil_tanh_abs", "refil_d128", "refil_rnn32", "refil_rnn128"]
TRAJ_CASES = ["refil_traj5"]      # consecutive train() calls: RMSprop state, weight decay, target syncs, checkpoint
POOL_CASES = ["refil_pool_mean", "refil_pool_max"]     # pooling_type = mean / max (EntityPoolingLayer)
GM_CASES = ["gm_refil_ff_lin"]       # BASELINE.json configs[0]: group_matching + FF agent + lin_flex_qmix
GM_TRAIN_CASES = ["gm_refil_train_gt", "gm_refil_train_randgt"]   # same alg with train_gt_factors / train_rand_gt_factors


def cfg_of(case):
    return orc.Cfg(
        n_agents=case["na"], n_entities=case["ne"], n_actions=case["A"], entity_shape=case["ed"],
        attn_embed_dim=case["d"], attn_n_heads=case["heads"], rnn_hidden_dim=case["H"],
        hypernet_embed=case["h"], mixing_embed_dim=case["M"],
        entity_last_action=case.get("entity_last_action", True),
        softmax_mixing_weights=case.get("softmax_mixing_weights", True),
        mixer_non_lin=case.get("mixer_non_lin", "elu"), imagine=case["imagine"],
        double_q=case.get("double_q", True), lmbda=case.get("lmbda", 0.5),
        grad_norm_clip=case.get("grad_norm_clip", 10),
        agent_ff=case.get("kind") == "gm", mixer_lin=case.get("kind") == "gm", mixer_vdn=case.get("mixer") == "vdn",
        mixer_none="mixer" in case and case["mixer"] is None,
        pooling_type=case.get("pooling_type"),
        train_gt_factors=bool(case.get("train_gt_factors", False)),
        train_rand_gt_factors=bool(case.get("train_rand_gt_factors", False)),
        weight_decay=case.get("weight_decay", 0.0),
    )


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    case = ast.literal_eval(str(z["case"]))
    cfg = cfg_of(case)

    def group(prefix):
        return {k[len(prefix):]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith(prefix)}

    return {
        "z": z, "case": case, "cfg": cfg,
        "batch": group("in."),
        "agent": {k: v for k, v in group("agent0.").items() if "scale_factor" not in k},
        "mixer": {k: v for k, v in group("mixer0.").items() if "scale_factor" not in k},
        "tagent": {k: v for k, v in group("tagent.").items() if "scale_factor" not in k},
        "tmixer": {k: v for k, v in group("tmixer.").items() if "scale_factor" not in k},
        "bits": torch.from_numpy(z["group_bits"].copy()),
    }


def load_traj(name):
    """Trajectory fixture (tools/make_golden.py:run_traj_case): state s0 before the first train() call, s{k} after
    call k-1; per call its batch `in{k}.*`, partition bits `bits{k}` and logged stats `stat{k}.*`."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    case = ast.literal_eval(str(z["case"]))

    def group(prefix):
        return {k[len(prefix):]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith(prefix)}

    def state(k):
        return {"agent": group(f"s{k}.agent."), "mixer": group(f"s{k}.mixer."), "tagent": group(f"s{k}.tagent."),
                "tmixer": group(f"s{k}.tmixer."), "sq": group(f"s{k}.sq.")}

    n = case["n_steps"]
    return {"z": z, "case": case, "cfg": cfg_of(case), "n_steps": n,
            "states": [state(k) for k in range(n + 1)],
            "batches": [group(f"in{k}.") for k in range(n)],
            "bits": [torch.from_numpy(z[f"bits{k}"].copy()) for k in range(n)],
            "stats": [{kk[len(f"stat{k}."):]: float(z[kk]) for kk in z.files if kk.startswith(f"stat{k}.")} for k in range(n)]}


def t_last_of(batch):
    """[B] last step any loss term of the episode depends on: 1 + the last t < T with mask[b,t] != 0
    (mask = filled * (1 - terminated[t-1]), q_learner.py:68-72); -1 if none. Outputs of later steps cannot influence
    the training step and are unspecified in the HIP path (it skips them)."""
    filled = batch["filled"][:, :-1, 0].float()
    term = batch["terminated"][:, :-1, 0].float()
    mask = filled.clone()
    mask[:, 1:] = mask[:, 1:] * (1 - term[:, :-1])
    T = mask.shape[1]
    idx = torch.arange(1, T + 1)[None].expand_as(mask)
    return torch.where(mask != 0, idx, torch.zeros_like(idx) - 1).max(dim=1)[0]


def live_steps(batch):
    """bool [B, T1]: steps t <= t_last[b]"""
    T1 = batch["filled"].shape[1]
    return torch.arange(T1)[None] <= t_last_of(batch)[:, None]


def rel_err(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


def assert_allclose(a, b, rtol, atol_frac, what=""):
    """Elementwise |a - b| <= atol + rtol |b| with atol = atol_frac * max|b| (rel_err above is max-abs over max-abs: one
    large element can hide many small wrong ones; here every element is held to its own scale, the absolute floor only
    covers cancellation around zero)."""
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    atol = atol_frac * b.abs().max().clamp_min(1e-30)
    viol = (a - b).abs() - (atol + rtol * b.abs())
    n_bad = int((viol > 0).sum().item())
    assert n_bad == 0, f"{what}: {n_bad} of {viol.numel()} elements outside atol {atol.item():.2e} + rtol {rtol:.0e} (worst excess {viol.max().item():.2e})"
