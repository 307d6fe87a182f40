"""Pins oracle/refil_oracle.py against golden vectors produced by the REAL reference learner
(tools/make_golden.py). CPU only."""
import numpy as np
import pytest
import torch

from oracle import refil_oracle as orc
from golden_util import CASES, GM_CASES, GM_TRAIN_CASES, POOL_CASES, TRAJ_CASES, load, load_traj, rel_err

TOL = 2e-5   # fp32, different op order than the reference (shared fc1/K/V, fused masks)


@pytest.mark.parametrize("name", CASES + GM_CASES + GM_TRAIN_CASES + POOL_CASES)
def test_oracle_matches_reference(name):
    g = load(name)
    z, cfg, case = g["z"], g["cfg"], g["case"]
    torch.manual_seed(0)
    agent, mixer = dict(g["agent"]), dict(g["mixer"])
    out, grads, gnorm = orc.train_step(cfg, agent, mixer, g["tagent"], g["tmixer"], g["batch"], g["bits"])
    B, T = case["B"], case["T"]
    assert rel_err(out.q.detach(), z["q"]) < TOL
    assert rel_err(out.chosen_q[0].detach(), z["chosen_q_real"]) < TOL
    assert rel_err(out.target_max_q, z["target_max_q"]) < TOL
    assert rel_err(out.q_tot.detach(), z["q_tot"]) < TOL
    assert rel_err(out.target_q_tot, z["target_q_tot"]) < TOL
    assert abs(out.loss.item() - float(z["stat.loss"])) < TOL * abs(float(z["stat.loss"]))
    if cfg.imagine:
        caq_im = torch.cat([out.chosen_q[1], out.chosen_q[2]], dim=2).detach()
        assert rel_err(caq_im, z["chosen_q_imagine"]) < TOL
        assert rel_err(out.q_tot_imagine.detach(), z["q_tot_imagine"]) < TOL
        assert abs(out.im_loss.item() - float(z["stat.im_loss"])) < TOL * abs(float(z["stat.im_loss"]))
        if "Wmask_noobs_t" in z.files:                            # (randomised) ground-truth factors: time-dependent groups
            W, I, act = orc.group_masks(cfg, g["batch"]["entity_mask"], g["bits"], g["batch"]["gt_mask"],
                                        rand_gt=cfg.train_rand_gt_factors)
            assert np.array_equal((W | act).numpy().astype(np.uint8), z["Wmask_noobs_t"])
            assert np.array_equal((I | act).numpy().astype(np.uint8), z["Imask_noobs_t"])
            W = None
        elif z["Wmask_noobs"].shape[-2] == cfg.n_entities:        # recurrent agent: full [ne,ne] masks
            Wm, Im = orc.imagine_masks(g["bits"], g["batch"]["entity_mask"][:, 0])
            assert np.array_equal(Wm.numpy().astype(np.uint8), z["Wmask_noobs"])
            assert np.array_equal(Im.numpy().astype(np.uint8), z["Imask_noobs"])
        if "Wmask_noobs" in z.files:
            W, I, act = orc.group_masks(cfg, g["batch"]["entity_mask"], g["bits"])
            assert np.array_equal((W | act)[:, 0].numpy().astype(np.uint8), z["Wmask_noobs"][:, :cfg.n_agents])
            assert np.array_equal((I | act)[:, 0].numpy().astype(np.uint8), z["Imask_noobs"][:, :cfg.n_agents])
    assert abs(gnorm - float(z["stat.grad_norm"])) < 1e-4 * float(z["stat.grad_norm"])
    for k in ("td_error_abs", "q_taken_mean", "target_mean"):
        assert abs(out.stats[k] - float(z["stat." + k])) < 1e-4 * max(abs(float(z["stat." + k])), 1e-3)
    gmax = max(v.abs().max().item() for v in grads.values())
    for k, gv in grads.items():
        if ("grad." + k) in z.files:
            ref = torch.from_numpy(z["grad." + k])
            assert (gv - ref).abs().max().item() < 1e-4 * gmax + 1e-9, k
            which, nm = k.split(".", 1)
            post = (agent if which == "agent" else mixer)[nm]
            assert (post - torch.from_numpy(z["post." + k])).abs().max().item() < 2e-6, k
        else:
            ref = float(z["gradnorm." + k])
            assert abs(gv.double().norm().item() - ref) < 1e-4 * max(ref, 1e-6), k
            which, nm = k.split(".", 1)
            post = (agent if which == "agent" else mixer)[nm].double()
            assert abs(post.sum().item() - float(z["postsum." + k])) < 2e-6 * post.numel() ** 0.5 + 1e-6, k


@pytest.mark.parametrize("name", TRAJ_CASES)
def test_oracle_trajectory_matches_reference(name):
    """Consecutive train() calls (q_learner.py:66-207): RMSprop with carried square_avg, weight decay
    (torch.optim.RMSprop: g += wd * p after the clip), hard target syncs every target_update_interval episodes."""
    g = load_traj(name)
    cfg, case = g["cfg"], g["case"]
    st = g["states"][0]
    agent, mixer, tagent, tmixer = dict(st["agent"]), dict(st["mixer"]), dict(st["tagent"]), dict(st["tmixer"])
    sq = {k: v.clone() for k, v in st["sq"].items()}
    last_sync = 0
    for s in range(g["n_steps"]):
        out, grads, gnorm = orc.train_step(cfg, agent, mixer, tagent, tmixer, g["batches"][s], g["bits"][s], square_avg=sq)
        if (s - last_sync) / case["target_update_interval"] >= 1.0:                   # q_learner.py:180-182, episode_num = s
            tagent, tmixer, last_sync = {k: v.clone() for k, v in agent.items()}, {k: v.clone() for k, v in mixer.items()}, s
        ref = g["stats"][s]
        assert abs(out.loss.item() - ref["loss"]) < TOL * abs(ref["loss"]), s
        assert abs(gnorm - ref["grad_norm"]) < 1e-4 * ref["grad_norm"], s
        nxt = g["states"][s + 1]
        for which, cur in (("agent", agent), ("mixer", mixer), ("tagent", tagent), ("tmixer", tmixer)):
            for k, v in cur.items():
                assert (v - nxt[which][k]).abs().max().item() < 3e-6, (s, which, k)
        for k, v in sq.items():
            assert rel_err(v, nxt["sq"][k]) < 1e-3 or nxt["sq"][k].abs().max() < 1e-12, (s, k)
    assert not torch.equal(g["states"][-2]["tagent"]["fc1.weight"], g["states"][-2]["agent"]["fc1.weight"])   # targets lag at the checkpoint


def test_partition_draw_matches_reference_rng_calls():
    g = load("refil_tiny")
    torch.manual_seed(g["case"]["seed"] + 7)
    bits = orc.draw_partition_bits(g["case"]["B"], g["case"]["ne"])
    assert torch.equal(bits, g["bits"])


@pytest.mark.parametrize("name", GM_CASES)
def test_oracle_gt_factor_diagnostics(name):
    """The log-step-only passes of cfg 1 (q_learner.py:98-105,138-147): imagine with ground-truth factors and
    LinearFlexQMixer's ingroup_prop."""
    g = load(name)
    z, cfg = g["z"], g["cfg"]
    b = g["batch"]
    xe = orc.build_entity_inputs(cfg, b["entities"], b["actions"])
    q_gt, _, groups = orc.agent_forward(cfg, g["agent"], xe, b["obs_mask"], b["entity_mask"], gt_mask=b["gt_mask"], use_gt_factors=True)
    assert rel_err(q_gt, z["q_gt"]) < TOL
    em = b["entity_mask"][:, :-1]
    _, q_im, prop = orc.mixer_forward(cfg, g["mixer"], torch.from_numpy(z["chosen_q_real"]), xe[:, :-1], em,
                                      torch.from_numpy(z["chosen_q_imagine"]),
                                      tuple(m for m in orc.group_masks(cfg, b["entity_mask"], g["bits"])[:2]) and
                                      tuple((m | orc.group_masks(cfg, b["entity_mask"], g["bits"])[2]) for m in orc.group_masks(cfg, b["entity_mask"], g["bits"])[:2]),
                                      ret_ingroup_prop=True)
    assert rel_err(q_im, z["q_tot_imagine"]) < TOL
    assert abs(prop.item() - float(z["stat.ingroup_prop"])) < 1e-5
    gtg = tuple(m[:, :-1] for m in groups)
    _, q_im_gt, prop_gt = orc.mixer_forward(cfg, g["mixer"], torch.from_numpy(z["chosen_q_real"]), xe[:, :-1], em,
                                            torch.from_numpy(z["chosen_q_imagine_gt"]), gtg, ret_ingroup_prop=True)
    assert rel_err(q_im_gt, z["q_tot_imagine_gt"]) < TOL
    assert abs(prop_gt.item() - float(z["stat.gt_ingroup_prop"])) < 1e-5
