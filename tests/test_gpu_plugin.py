"""The reference's plugin surface (REGISTRY dicts, EntityMAC, QLearner.train, save/load_models) on top
of the HIP path, checked against the golden vectors of the real reference."""
import os

import pytest
import torch as th

pytestmark = pytest.mark.gpu

DEV = "cuda"        # (tests/test_emu_plugin.py runs a private copy of this module with DEV = "cpu" on the CPU wavefront emulator)


def _place(learner):
    """learner.cuda() (src/run.py:211-212) where there is a GPU"""
    if DEV == "cuda":
        learner.cuda()

from golden_util import load, load_traj, rel_err
from oracle import refil_oracle as orc
from plugin_util import RecLogger, make_args, make_episode_batch


def _build(name, **over):
    from refil_amd.controllers import REGISTRY as mac_REGISTRY
    from refil_amd.learners import REGISTRY as le_REGISTRY
    g = load(name)
    cfg = g["cfg"]
    args = make_args(cfg, device=DEV, use_cuda=DEV == "cuda", **over)
    batch, groups = make_episode_batch(cfg, g["batch"])
    mac = mac_REGISTRY[args.mac](batch.scheme, groups, args)
    logger = RecLogger()
    learner = le_REGISTRY[args.learner](mac, batch.scheme, logger, args)
    _place(learner)
    batch.to(DEV)
    # load the reference's initial weights by state_dict name
    mac.agent.load_state_dict({k[len("agent0."):]: th.from_numpy(g["z"][k]) for k in g["z"].files if k.startswith("agent0.")})
    learner.mixer.load_state_dict({k[len("mixer0."):]: th.from_numpy(g["z"][k]) for k in g["z"].files if k.startswith("mixer0.")})
    learner.target_mac.agent.load_state_dict({k[len("tagent."):]: th.from_numpy(g["z"][k]) for k in g["z"].files if k.startswith("tagent.")})
    learner.target_mixer.load_state_dict({k[len("tmixer."):]: th.from_numpy(g["z"][k]) for k in g["z"].files if k.startswith("tmixer.")})
    return g, args, batch, mac, learner, logger


@pytest.mark.parametrize("name", ["refil_tiny", "qmix_atten_tiny", "refil_abs_masked", "refil_odd", "refil_vdn_tiny",
                                  "refil_pool_mean", "refil_pool_max", "refil_rnn32", "refil_rnn128"])
def test_qlearner_train_matches_reference(name):
    g, args, batch, mac, learner, logger = _build(name)
    z, case = g["z"], g["case"]
    th.manual_seed(case["seed"] + 7)                 # same seed => same partition as the reference run
    learner.train(batch, t_env=0, episode_num=0)
    th.cuda.synchronize()
    for k in ("loss", "grad_norm", "td_error_abs", "q_taken_mean", "target_mean") + (("im_loss",) if g["cfg"].imagine else ()):
        ref = float(z["stat." + k])
        assert abs(logger.stats[k] - ref) < 2e-4 * max(abs(ref), 1e-3), (k, logger.stats[k], ref)
    sd = {**{"agent." + k: v for k, v in mac.agent.state_dict().items()},
          **{"mixer." + k: v for k, v in learner.mixer.state_dict().items()}}
    for k in z.files:
        if k.startswith("post."):
            assert (sd[k[5:]].cpu() - th.from_numpy(z[k])).abs().max().item() < 5e-6, k


def test_registry_keys_and_state_dict_names():
    from refil_amd.controllers import REGISTRY as mac_REGISTRY
    from refil_amd.learners import REGISTRY as le_REGISTRY
    from refil_amd.modules.agents import REGISTRY as agent_REGISTRY
    assert "q_learner" in le_REGISTRY and "entity_mac" in mac_REGISTRY
    assert {"entity_attend_rnn", "imagine_entity_attend_rnn"} <= set(agent_REGISTRY)
    g, args, batch, mac, learner, logger = _build("refil_tiny")
    assert set(mac.agent.state_dict()) == {k[len("agent0."):] for k in g["z"].files if k.startswith("agent0.")}
    assert set(learner.mixer.state_dict()) == {k[len("mixer0."):] for k in g["z"].files if k.startswith("mixer0.")}
    assert len(learner.params) == 41


def test_acting_path_step_by_step_equals_full_sequence():
    """mac.forward(batch, t=int) with carried hidden state (runners: parallel_runner.py:121) reproduces
    mac.forward(batch, t=None); select_actions respects avail_actions."""
    g, args, batch, mac, learner, logger = _build("refil_tiny")
    B, T1 = batch.batch_size, batch.max_seq_length
    mac.init_hidden(B)
    q_all = mac.forward(batch, t=None)
    assert rel_err(q_all.cpu(), g["z"]["q"][0]) < 1e-4
    mac.init_hidden(B)
    for t in range(T1):
        q_t = mac.forward(batch, t=t)
        assert rel_err(q_t.cpu(), q_all[:, t].cpu()) < 1e-5
    mac.init_hidden(B)
    acts = mac.select_actions(batch, t_ep=0, t_env=0, test_mode=True)
    avail = batch["avail_actions"][:, 0]
    assert (avail.gather(2, acts.unsqueeze(2)) == 1).all()


def test_imagine_forward_returns_groups_like_reference():
    g, args, batch, mac, learner, logger = _build("refil_tiny")
    z = g["z"]
    mac.init_hidden(batch.batch_size)
    q, groups = mac.forward(batch, t=None, imagine=True, group_bits=g["bits"])
    B = batch.batch_size
    assert rel_err(q.reshape(3, B, *q.shape[1:]).cpu(), z["q"]) < 1e-4
    assert th.equal(groups[0][:, 0].cpu(), th.from_numpy(z["Wmask_noobs"]))
    assert th.equal(groups[1][:, 0].cpu(), th.from_numpy(z["Imask_noobs"]))


def test_mixer_module_forward():
    g, args, batch, mac, learner, logger = _build("refil_tiny")
    z, cfg = g["z"], g["cfg"]
    xe = orc.build_entity_inputs(cfg, g["batch"]["entities"], g["batch"]["actions"]).to(DEV)
    em = batch["entity_mask"]
    q = learner.mixer(th.from_numpy(z["chosen_q_real"]).to(DEV), (xe[:, :-1], em[:, :-1]))
    assert rel_err(q.cpu(), z["q_tot"]) < 1e-4
    qi = learner.mixer(th.from_numpy(z["chosen_q_imagine"]).to(DEV), (xe[:, :-1], em[:, :-1]), imagine_groups=g["bits"].to(DEV))
    assert rel_err(qi.cpu(), z["q_tot_imagine"]) < 1e-4
    # the reference's own calling convention: imagine_groups = (Wmask, Imask) mask tensors (flex_qmix.py:85-94), as the
    # imagine agent returns them (entity_rnn_agent.py:130) and QLearner slices them (q_learner.py:137)
    T = xe.shape[1] - 1
    Wm = th.from_numpy(z["Wmask_noobs"]).to(DEV)[:, None].repeat(1, T, 1, 1)
    Im = th.from_numpy(z["Imask_noobs"]).to(DEV)[:, None].repeat(1, T, 1, 1)
    qi2 = learner.mixer(th.from_numpy(z["chosen_q_imagine"]).to(DEV), (xe[:, :-1], em[:, :-1]), imagine_groups=(Wm, Im))
    assert rel_err(qi2.cpu(), z["q_tot_imagine"]) < 1e-4
    assert th.equal(qi2, qi)                       # same kernels, same mask words: bit-identical
    qi3 = learner.mixer(th.from_numpy(z["chosen_q_imagine"]).to(DEV), (xe[:, :-1], em[:, :-1]), imagine_groups=[Wm[:, :1], Im[:, :1]])
    assert th.equal(qi3, qi)
    # arbitrary masks (not derivable from a 2-way split) against the oracle's mixer
    gen = th.Generator().manual_seed(5)
    Wr = (th.rand(Wm.shape, generator=gen) < 0.4)
    Ir = (th.rand(Wm.shape, generator=gen) < 0.4)
    mixer_p = {k[len("mixer0."):]: th.from_numpy(z[k]) for k in z.files if k.startswith("mixer0.")}
    caq = th.from_numpy(z["chosen_q_imagine"])
    _, ref_im = orc.mixer_forward(cfg, mixer_p, th.from_numpy(z["chosen_q_real"]), xe[:, :-1].cpu(), g["batch"]["entity_mask"][:, :-1],
                                  caq, (Wr[:, :, :cfg.n_agents], Ir[:, :, :cfg.n_agents]))     # (the oracle takes the agents' rows)
    qi4 = learner.mixer(caq.to(DEV), (xe[:, :-1], em[:, :-1]), imagine_groups=(Wr.to(DEV), Ir.to(DEV)))
    assert rel_err(qi4.cpu(), ref_im) < 1e-4
    tq = learner.target_mixer(th.from_numpy(z["target_max_q"]).to(DEV), (xe[:, 1:], em[:, 1:]))
    assert rel_err(tq.cpu(), z["target_q_tot"]) < 1e-4


def test_checkpoint_roundtrip_in_reference_format(tmp_path):
    g, args, batch, mac, learner, logger = _build("refil_tiny")
    th.manual_seed(3)
    learner.train(batch, 0, 0)
    learner.save_models(str(tmp_path))
    assert {"agent.th", "mixer.th", "opt.th"} <= set(os.listdir(tmp_path))
    opt = th.load(str(tmp_path / "opt.th"))
    assert len(opt["state"]) == 41 and set(opt["state"][0]) == {"step", "square_avg"}
    g2, args2, batch2, mac2, learner2, logger2 = _build("refil_tiny")
    learner2.load_models(str(tmp_path))
    th.cuda.synchronize()
    # what a reload defines: live agent + mixer + RMSprop state; the target agent is loaded from agent.th as well and the
    # target mixer is not checkpointed (q_learner.py:222-229)
    for (k, a), (_, b) in zip(mac.agent.state_dict().items(), mac2.agent.state_dict().items()):
        assert th.equal(a, b), k
    for (k, a), (_, b) in zip(learner.mixer.state_dict().items(), learner2.mixer.state_dict().items()):
        assert th.equal(a, b), k
    for (k, a), (_, b) in zip(mac2.agent.state_dict().items(), learner2.target_mac.agent.state_dict().items()):
        assert th.equal(a, b), k
    assert th.equal(learner.square_avg, learner2.square_avg) and learner.square_avg.abs().max().item() > 0
    # ... and the next step of the reloaded learner equals the next step of the one that kept running (same targets)
    learner._update_targets(); learner2._update_targets()
    th.manual_seed(4); learner.train(batch, 1, 1)
    th.manual_seed(4); learner2.train(batch2, 1, 1)
    th.cuda.synchronize()
    assert th.equal(learner.flat_live, learner2.flat_live)
    assert th.equal(learner.square_avg, learner2.square_avg)
    assert abs(logger.stats["loss"] - logger2.stats["loss"]) == 0.0


def _build_traj(name, state, **over):
    from refil_amd.controllers import REGISTRY as mac_REGISTRY
    from refil_amd.learners import REGISTRY as le_REGISTRY
    g = load_traj(name)
    cfg, case = g["cfg"], g["case"]
    args = make_args(cfg, device=DEV, use_cuda=DEV == "cuda", target_update_interval=case["target_update_interval"], **over)
    batches = [make_episode_batch(cfg, b, device=DEV)[0] for b in g["batches"]]
    groups = {"agents": cfg.n_agents, "entities": cfg.n_entities}
    mac = mac_REGISTRY[args.mac](batches[0].scheme, groups, args)
    logger = RecLogger()
    learner = le_REGISTRY[args.learner](mac, batches[0].scheme, logger, args)
    _place(learner)
    if state is not None:
        st = g["states"][state]
        mac.agent.load_state_dict(st["agent"], strict=False)
        learner.mixer.load_state_dict(st["mixer"], strict=False)
        learner.target_mac.agent.load_state_dict(st["tagent"], strict=False)
        learner.target_mixer.load_state_dict(st["tmixer"], strict=False)
    return g, args, batches, mac, learner, logger


def _assert_state(g, k, mac, learner, tol):
    nxt = g["states"][k]
    for which, sd in (("agent", mac.agent.state_dict()), ("mixer", learner.mixer.state_dict()),
                      ("tagent", learner.target_mac.agent.state_dict()), ("tmixer", learner.target_mixer.state_dict())):
        for name, ref in nxt[which].items():
            assert (sd[name].cpu() - ref).abs().max().item() < tol, (k, which, name)
    opt = learner._opt_state_dict()
    names = ["agent." + n for n, _ in mac.agent.named_parameters()] + ["mixer." + n for n, _ in learner.mixer.named_parameters()]
    for i, name in enumerate(names):
        ref = nxt["sq"][name]
        assert rel_err(opt["state"][i]["square_avg"].cpu(), ref) < 2e-3 or ref.abs().max() < 1e-12, (k, name)


def test_qlearner_trajectory_matches_reference():
    """Five consecutive QLearner.train calls through the plugin surface against the reference's own run: non-zero RMSprop
    state, weight_decay = 1e-4, target syncs after episode 2 and 4 (q_learner.py:175-182,203-207)."""
    g, args, batches, mac, learner, logger = _build_traj("refil_traj5", 0)
    assert args.weight_decay == 1e-4
    for s in range(g["n_steps"]):
        th.manual_seed(g["case"]["seed"] + 7 + s)
        learner.train(batches[s], t_env=s, episode_num=s)
        th.cuda.synchronize()
        for k in ("loss", "im_loss", "grad_norm", "td_error_abs", "q_taken_mean", "target_mean"):
            ref = g["stats"][s][k]
            assert abs(logger.stats[k] - ref) < 2e-4 * max(abs(ref), 1e-3), (s, k, logger.stats[k], ref)
        _assert_state(g, s + 1, mac, learner, 5e-6 * (s + 1))
    assert sum("Updated target network" in str(i) for i in logger.infos) == 2


def test_reference_written_checkpoint_loads_and_continues(tmp_path):
    """f3: agent.th / mixer.th / opt.th laid out exactly as the REFERENCE writes them (state_dict names incl. the
    attn.scale_factor buffers, torch.optim.RMSprop.state_dict() layout; q_learner.py:216-220, basic_controller.py:87-88),
    filled with the reference's own values after four train() calls, load into a fresh learner; its next train() call
    reproduces the reference's fifth call."""
    g = load_traj("refil_traj5")
    cfg, ck = g["cfg"], g["case"]["checkpoint_after"]
    st = g["states"][ck]
    hd_a, hd_m = cfg.attn_embed_dim // cfg.attn_n_heads, cfg.hypernet_embed // cfg.attn_n_heads
    agent_sd = dict(st["agent"]); agent_sd["attn.scale_factor"] = th.sqrt(th.scalar_tensor(hd_a))       # attention.py:18-19
    mixer_sd = dict(st["mixer"])
    for net in ("hyper_w_1", "hyper_w_final", "hyper_b_1", "V"):
        mixer_sd[net + ".attn.scale_factor"] = th.sqrt(th.scalar_tensor(hd_m))
    g0, args, batches, mac, learner, logger = _build_traj("refil_traj5", None)          # fresh, randomly initialised
    names = ["agent." + n for n, _ in mac.agent.named_parameters()] + ["mixer." + n for n, _ in learner.mixer.named_parameters()]
    opt_sd = {"state": {i: {"step": ck, "square_avg": st["sq"][n].clone()} for i, n in enumerate(names)},
              "param_groups": [{"lr": cfg.lr, "momentum": 0, "alpha": cfg.optim_alpha, "eps": cfg.optim_eps, "centered": False,
                                "weight_decay": cfg.weight_decay, "params": list(range(len(names)))}]}
    th.save(agent_sd, str(tmp_path / "agent.th")); th.save(mixer_sd, str(tmp_path / "mixer.th")); th.save(opt_sd, str(tmp_path / "opt.th"))
    learner.load_models(str(tmp_path))
    for name, ref in st["agent"].items():                     # the reference loads agent.th into the target MAC too (:224-225)
        assert th.equal(mac.agent.state_dict()[name].cpu(), ref), name
        assert th.equal(learner.target_mac.agent.state_dict()[name].cpu(), ref), name
    # the reference's uninterrupted run still had lagging targets at this point: put them in place, then step
    learner.target_mac.agent.load_state_dict(st["tagent"], strict=False)
    learner.target_mixer.load_state_dict(st["tmixer"], strict=False)
    learner.last_target_update_episode = 2
    th.manual_seed(g["case"]["seed"] + 7 + ck)
    learner.train(batches[ck], t_env=ck, episode_num=ck)
    th.cuda.synchronize()
    for k in ("loss", "grad_norm"):
        assert abs(logger.stats[k] - g["stats"][ck][k]) < 2e-4 * abs(g["stats"][ck][k]), k
    _assert_state(g0, ck + 1, mac, learner, 5e-6)


def test_qlearner_without_mixer(tmp_path):
    """args.mixer = None (q_learner.py:19-21): `learner.mixer is None`, cuda() / save_models() / load_models() skip the mixer
    like the reference's guards (:212-214,218-219,226-227); train() takes the TD loss per agent, as the code below :131 does
    when no mixer is configured (mask expanded to [B,T,n_agents], :161). The reference's own train() raises at :81
    (`self.mixer.train()`), so this mode is pinned by the oracle restatement only."""
    import dataclasses
    from refil_amd.controllers import REGISTRY as mac_REGISTRY
    from refil_amd.learners import REGISTRY as le_REGISTRY
    g = load("qmix_atten_tiny")
    cfg = dataclasses.replace(g["cfg"], mixer_none=True)
    args = make_args(cfg, device=DEV, use_cuda=DEV == "cuda")
    assert args.mixer is None
    batch, groups = make_episode_batch(cfg, g["batch"])
    mac = mac_REGISTRY[args.mac](batch.scheme, groups, args)
    logger = RecLogger()
    learner = le_REGISTRY[args.learner](mac, batch.scheme, logger, args)
    _place(learner)
    batch.to(DEV)
    assert learner.mixer is None and not hasattr(learner, "target_mixer")
    z = g["z"]
    agent = {k[len("agent0."):]: th.from_numpy(z[k]) for k in z.files if k.startswith("agent0.")}
    tagent = {k[len("tagent."):]: th.from_numpy(z[k]) for k in z.files if k.startswith("tagent.")}
    mac.agent.load_state_dict(agent)
    learner.target_mac.agent.load_state_dict(tagent)
    learner.train(batch, t_env=0, episode_num=0)
    th.cuda.synchronize()
    a2 = {k: v.clone() for k, v in agent.items()}
    out, grads, gnorm = orc.train_step(cfg, a2, {}, tagent, {}, g["batch"], None)
    assert abs(logger.stats["loss"] - out.loss.item()) < 2e-4 * out.loss.item()
    assert abs(logger.stats["grad_norm"] - gnorm) < 2e-4 * gnorm
    for k in ("td_error_abs", "q_taken_mean", "target_mean"):
        assert abs(logger.stats[k] - out.stats[k]) < 2e-4 * max(abs(out.stats[k]), 1e-3), k
    for k, v in mac.agent.state_dict().items():
        assert (v.cpu() - a2[k]).abs().max().item() < 5e-6, k
    learner.save_models(str(tmp_path))
    assert sorted(os.listdir(tmp_path)) == ["agent.th", "opt.th"]        # no mixer.th (q_learner.py:218-219)
    learner.load_models(str(tmp_path))


def test_long_trajectory_tracks_oracle():
    """40 consecutive train() calls on one batch (RMSprop state carried, hard target syncs every 7 episodes, a fresh
    partition every call): the HIP learner stays on the oracle's trajectory -- errors of single steps must not
    accumulate into a different optimisation path (tools/soak.py runs the long unattended version)."""
    g, args, batch, mac, learner, logger = _build("refil_abs_masked", target_update_interval=7)
    cfg, z = g["cfg"], g["z"]
    agent = {k[len("agent0."):]: th.from_numpy(z[k]).clone() for k in z.files if k.startswith("agent0.")}
    mixer = {k[len("mixer0."):]: th.from_numpy(z[k]).clone() for k in z.files if k.startswith("mixer0.")}
    tagent = {k[len("tagent."):]: th.from_numpy(z[k]).clone() for k in z.files if k.startswith("tagent.")}
    tmixer = {k[len("tmixer."):]: th.from_numpy(z[k]).clone() for k in z.files if k.startswith("tmixer.")}
    sq = {}
    B, ne = g["batch"]["entities"].shape[0], cfg.n_entities
    gen = th.Generator().manual_seed(77)
    last_sync = 0
    for ep in range(40):
        bits = orc.draw_partition_bits(B, ne, generator=gen)
        out, _, gnorm = orc.train_step(cfg, agent, mixer, tagent, tmixer, g["batch"], bits, square_avg=sq)
        learner.train(batch, t_env=ep, episode_num=ep, group_bits=bits)
        if (ep - last_sync) / 7 >= 1.0:                    # q_learner.py:180-182
            tagent = {k: v.detach().clone() for k, v in agent.items()}
            tmixer = {k: v.detach().clone() for k, v in mixer.items()}
            last_sync = ep
        if ep % 10 == 9 or ep == 39:
            th.cuda.synchronize()
            ref = out.loss.item()
            assert abs(logger.stats["loss"] - ref) < 2e-3 * max(abs(ref), 1e-3), (ep, logger.stats["loss"], ref)
            assert abs(logger.stats["grad_norm"] - gnorm) < 5e-3 * gnorm, (ep, logger.stats["grad_norm"], gnorm)
    sd = mac.agent.state_dict()
    worst = max((sd[k].cpu() - agent[k]).abs().max().item() for k in agent)
    assert worst < 2e-4, worst


def test_target_update_copies_flat_buffer():
    g, args, batch, mac, learner, logger = _build("refil_tiny", target_update_interval=1)
    th.manual_seed(1)
    learner.train(batch, 0, episode_num=1)
    th.cuda.synchronize()
    assert th.equal(learner.flat_target, learner.flat_live)
    assert any("Updated target network" in str(i) for i in logger.infos)
    for (k, a), (_, b) in zip(mac.agent.state_dict().items(), learner.target_mac.agent.state_dict().items()):
        assert th.equal(a, b), k


def test_cfg1_group_matching_ff_agent_linear_mixer():
    """BASELINE.json configs[0]: refil_group_matching (imagine_entity_attend_ff + lin_flex_qmix, test_gt_factors)
    on real GroupMatching episodes, through the plugin surface, against the reference's own run."""
    g, args, batch, mac, learner, logger = _build("gm_refil_ff_lin", gt_mask_avail=True, test_gt_factors=True)
    z, case = g["z"], g["case"]
    assert type(mac.agent).__name__ == "ImagineEntityAttentionFFAgent" and type(learner.mixer).__name__ == "LinearFlexQMixer"
    th.manual_seed(case["seed"] + 7)
    learner.train(batch, t_env=0, episode_num=0)
    th.cuda.synchronize()
    for k in ("loss", "im_loss", "grad_norm", "td_error_abs", "q_taken_mean", "target_mean", "ingroup_prop", "gt_ingroup_prop"):
        ref = float(z["stat." + k])
        assert abs(logger.stats[k] - ref) < 2e-4 * max(abs(ref), 1e-3), (k, logger.stats[k], ref)
    sd = {**{"agent." + k: v for k, v in mac.agent.state_dict().items()},
          **{"mixer." + k: v for k, v in learner.mixer.state_dict().items()}}
    for k in z.files:
        if k.startswith("post."):
            assert (sd[k[5:]].cpu() - th.from_numpy(z[k])).abs().max().item() < 5e-6, k
    # imagine forward with ground-truth factors through the MAC (q_learner.py:99)
    g2, args2, batch2, mac2, learner2, _ = _build("gm_refil_ff_lin", gt_mask_avail=True)
    mac2.init_hidden(batch2.batch_size)
    q, groups = mac2.forward(batch2, t=None, imagine=True, use_gt_factors=True)
    assert rel_err(q.reshape(3, batch2.batch_size, *q.shape[1:]).cpu(), z["q_gt"]) < 1e-4
    mac2.init_hidden(batch2.batch_size)
    acts = mac2.select_actions(batch2, t_ep=0, t_env=0, test_mode=True)
    assert acts.shape == (batch2.batch_size, g["cfg"].n_agents)


@pytest.mark.parametrize("name,flag", [("gm_refil_train_gt", "train_gt_factors"), ("gm_refil_train_randgt", "train_rand_gt_factors")])
def test_cfg1_training_with_ground_truth_factor_options(name, flag):
    """default.yaml:52-53 / q_learner.py:87-89: the imagined groups of the TRAINING pass are the ground-truth
    factors, or the random split OR-ed with them (entity_ff_agent.py:93-95,111-114) -- against the reference's run."""
    g, args, batch, mac, learner, logger = _build(name, gt_mask_avail=True, test_gt_factors=True, **{flag: True})
    z, case = g["z"], g["case"]
    th.manual_seed(case["seed"] + 7)
    learner.train(batch, t_env=0, episode_num=0)
    th.cuda.synchronize()
    for k in ("loss", "im_loss", "grad_norm", "td_error_abs", "q_taken_mean", "target_mean", "ingroup_prop", "gt_ingroup_prop"):
        ref = float(z["stat." + k])
        assert abs(logger.stats[k] - ref) < 2e-4 * max(abs(ref), 1e-3), (k, logger.stats[k], ref)
    sd = {**{"agent." + k: v for k, v in mac.agent.state_dict().items()},
          **{"mixer." + k: v for k, v in learner.mixer.state_dict().items()}}
    for k in z.files:
        if k.startswith("post."):
            assert (sd[k[5:]].cpu() - th.from_numpy(z[k])).abs().max().item() < 5e-6, k
    # the groups the agent hands to the mixer are the reference's time-dependent masks
    g2, args2, batch2, mac2, _, _ = _build(name, gt_mask_avail=True)
    mac2.init_hidden(batch2.batch_size)
    _, groups = mac2.forward(batch2, t=None, imagine=True, group_bits=g["bits"].to(DEV), **{flag.replace("train_", "use_"): True})
    assert th.equal(groups[0].cpu(), th.from_numpy(z["Wmask_noobs_t"]))
    assert th.equal(groups[1].cpu(), th.from_numpy(z["Imask_noobs_t"]))
