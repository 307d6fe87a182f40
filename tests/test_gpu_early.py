"""Early prologue (refil_batch.ready_event, QLearner.train with batch.ready_event): the input assembly and row lists of
step k+1 run on a side stream beside the end of step k, in alternating workspace slots. Two identically seeded
learners take the same steps on two alternating batches, one with the events and one without: bit-identical parameters."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"        # (tests/test_emu_early.py runs a private copy of this module with DEV = "cpu" on the CPU wavefront emulator)


def _dev():
    return torch.device("cuda", 0) if DEV == "cuda" else torch.device("cpu")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


@pytest.mark.parametrize("et", ["1", "3", "0"])           # REFIL_EARLY_TARGET: target hypernets early (default) / + target agent / off
@pytest.mark.parametrize("cfg", ["cfg2", "cfg4"])
def test_early_prologue_is_bit_identical(cfg, et, monkeypatch):
    import bench
    from refil_amd import _lib
    monkeypatch.delenv("REFIL_EARLY", raising=False)
    monkeypatch.setenv("REFIL_EARLY_TARGET", et)
    W = dict(bench.CONFIGS[cfg])
    dims = bench.workload_dims(W)
    dev = _dev()
    B = 8
    _, b1, la, _, _ = bench.build(dims, W["imagine"], B, W["T"], seed=100, device=dev)
    _, b2, lb, _, _ = bench.build(dims, W["imagine"], B, W["T"], seed=200, device=dev)
    assert b1.ready_event is not None and b2.ready_event is not None
    la._check_flat(); lb._check_flat()
    assert torch.equal(la.flat_live, lb.flat_live)
    la.args.target_update_interval = lb.args.target_update_interval = 4      # target syncs after steps 4 and 8
    st0 = {k: _lib.get_stat(k) for k in ("learner_steps", "early_prologue_steps", "early_target_hypernet_steps", "early_target_agent_steps")}

    class Plain:                                   # the same batch without the event: the prologue stays on the caller's stream
        def __init__(self, b):
            self._b = b
            self.ready_event = None

        def __getattr__(self, k):
            return getattr(self._b, k)

        def __getitem__(self, k):
            return self._b[k]

    _lib.profile_enable(True)
    for i in range(12):
        b = (b1, b2, b2)[i % 3]                    # (b2 twice in a row: the slot alternates even when the batch does not)
        la.train(b, t_env=0, episode_num=i)
        lb.train(Plain(b), t_env=0, episode_num=i)
        torch.cuda.synchronize()
        assert torch.equal(la.flat_live, lb.flat_live), f"step {i}: max |d| = {(la.flat_live - lb.flat_live).abs().max().item():.3e}"
        assert torch.equal(la.flat_target, lb.flat_target)
    _lib.profile_enable(False)
    _lib.profile_collect()
    # the early paths were taken: 24 learner steps, 11 of la's with the early prologue (not its first: new workspace), and the
    # target nets' early forward (refil_batch.target_version) on every one of those whose target parameters had not just been
    # rewritten -- the first early step (no version on record: the first call ran on a new arena) and the two steps behind the target
    # syncs stay in order
    st = {k: _lib.get_stat(k) - v for k, v in st0.items()}
    assert st["learner_steps"] == 24 and st["early_prologue_steps"] == 11, st
    assert st["early_target_hypernet_steps"] == (8 if et != "0" else 0) and st["early_target_agent_steps"] == (8 if et == "3" else 0), st
    assert float((la.flat_live - la.flat_target).abs().max()) > 0


def test_sampled_batches_early_gather_is_bit_identical(monkeypatch):
    """ReplayBuffer.sample() with the early path: the gather runs on the library's hypernet-chain stream into two alternating
    staging minibatches, train() starts its prologue from the batch's ready event. A second learner / buffer pair takes the same
    samples with REFIL_EARLY=0 (gather and prologue in order on the caller's stream, one staging minibatch): bit-identical
    parameters after every step. (The pairs alternate step by step so that both see the same row-count hints.)"""
    import numpy as np

    import bench
    W = dict(bench.CONFIGS["cfg2"])
    dims = bench.workload_dims(W)
    dev = _dev()
    B = 8
    monkeypatch.delenv("REFIL_EARLY", raising=False)
    _, _, la, _, bufa = bench.build(dims, W["imagine"], B, W["T"], seed=300, device=dev, fresh=4)
    _, _, lb, _, bufb = bench.build(dims, W["imagine"], B, W["T"], seed=300, device=dev, fresh=4)
    la._check_flat(); lb._check_flat()
    for i in range(8):
        monkeypatch.delenv("REFIL_EARLY", raising=False)
        np.random.seed(100 + i)
        b = bufa.sample(B)
        assert b.ready_event is not None
        la.train(b, t_env=0, episode_num=i)
        monkeypatch.setenv("REFIL_EARLY", "0")
        np.random.seed(100 + i)
        b = bufb.sample(B)
        assert b.ready_event is None
        lb.train(b, t_env=0, episode_num=i)
        torch.cuda.synchronize()
        assert torch.equal(la.flat_live, lb.flat_live), f"step {i}: max |d| = {(la.flat_live - lb.flat_live).abs().max().item():.3e}"


def test_side_stream_is_stable_and_usable():
    """refil_side_stream: the library's hypernet-chain stream -- the same handle on every call, usable as a torch stream."""
    import ctypes as C

    from refil_amd import _lib
    a, b = C.c_void_p(), C.c_void_p()
    _lib.check(_lib.lib().refil_side_stream(C.byref(a)), "refil_side_stream")
    _lib.check(_lib.lib().refil_side_stream(C.byref(b)), "refil_side_stream")
    assert a.value and a.value == b.value
    s = torch.cuda.ExternalStream(a.value, device=torch.device("cuda", 0))
    x = torch.zeros(1024, device="cuda")
    ev = torch.cuda.Event()
    ev.record()
    s.wait_event(ev)
    with torch.cuda.stream(s):
        x += 1
    s.synchronize()
    assert float(x.sum()) == 1024.0


def test_first_call_autotune(monkeypatch):
    """QLearner's first-call autotuner (opt-in, REFIL_AUTOTUNE=1): measures the launch-size knobs in situ on a shape large enough
    for the row lists, keeps what wins -- only values of tuning.PARITY_TESTED --, touches neither the parameters nor the RNG; the
    step's results agree with the untuned schedule to rounding (the knobs move the summation order of the split weight-gradient
    reductions). The key ignores B and T1 (a run loop that trims to max_t_filled() tunes once per row bucket, not per length);
    without the switch nothing is measured and the schedule is the deterministic built-in one."""
    import bench
    from refil_amd import tuning
    from refil_amd.learners.q_learner import QLearner
    W = dict(bench.CONFIGS["cfg2"])
    dims = bench.workload_dims(W)
    dev = _dev()
    saved, measured = dict(QLearner._TUNED), QLearner._MEASURED[0]
    try:
        QLearner._TUNED.clear()
        QLearner._MEASURED[0] = 0
        monkeypatch.setenv("REFIL_AUTOTUNE", "1")
        monkeypatch.delenv("REFIL_AUTOTUNE_CACHE", raising=False)
        _, batch, la, _, _ = bench.build(dims, W["imagine"], W["B"], W["T"], seed=100, device=dev)
        la._check_flat()
        p0 = la.flat_live.clone()
        st = la.generator.get_state().clone()
        la._last_dims = None
        la.train(batch, t_env=0, episode_num=0)
        assert len(QLearner._TUNED) == 1 and len(la._autotune_log) == 6          # measured: one entry per candidate
        assert QLearner._MEASURED[0] == 1
        tuning.check(la.tuning_chosen())
        # a shorter view of the same batch (the reference's max_t_filled() trim) lands in the same bucket or the next one down:
        # no second measurement for a batch one step shorter
        key_full = tuning.bucket_key(la._last_dims)
        from refil_amd.engine import clone_dims
        assert tuning.bucket_key(clone_dims(la._last_dims, T1=la._last_dims.T1 - 1)) == key_full
        assert tuning.bucket_key(clone_dims(la._last_dims, T1=la._last_dims.T1 // 4)) != key_full
        monkeypatch.delenv("REFIL_AUTOTUNE")               # the default: no measuring, built-in schedule
        QLearner._TUNED.clear()
        _, _, lb, _, _ = bench.build(dims, W["imagine"], W["B"], W["T"], seed=100, device=dev)
        lb._check_flat()
        assert torch.equal(p0, lb.flat_live)
        lb.generator.set_state(st)
        lb.train(batch, t_env=0, episode_num=0)
        torch.cuda.synchronize()
        assert not hasattr(lb, "_autotune_log") and QLearner._TUNED == {tuning.bucket_key(lb._last_dims): {}}
        assert lb.tuning_chosen() == {}
        ga, gb = la.grads[:la._n], lb.grads[:lb._n]
        assert (ga - gb).abs().max().item() <= 2e-6 * gb.abs().max().item()
        assert torch.equal(la.generator.get_state(), lb.generator.get_state())      # the tuner drew nothing
        # a given setting outside the parity-tested set is refused
        monkeypatch.setenv("REFIL_AUTOTUNE", "dw4_target=77")
        QLearner._TUNED.clear()
        with pytest.raises(ValueError):
            lb.train(batch, t_env=0, episode_num=0)
    finally:
        QLearner._TUNED.clear()
        QLearner._TUNED.update(saved)
        QLearner._MEASURED[0] = measured
        QLearner._apply_tuning({})


@pytest.mark.parametrize("cfg,B", [("cfgT", 32), ("cfg2", 32), ("cfg4", 16)])
def test_unsynchronised_steps_equal_in_order_steps(cfg, B, monkeypatch):
    """The race check the early paths need: WITHOUT a host synchronisation between the steps the early prologue and the early target
    forward of step k+1 really run beside the tail of step k (the other tests synchronise after every step, where nothing is left to
    overlap with). 40 back-to-back steps on two alternating batches with target syncs every 7 steps, against the same steps in order
    (no ready event): bit-identical parameters, optimiser state and target parameters."""
    import bench
    monkeypatch.delenv("REFIL_EARLY", raising=False)
    monkeypatch.delenv("REFIL_EARLY_TARGET", raising=False)
    W = dict(bench.CONFIGS[cfg])
    dims = bench.workload_dims(W)
    dev = _dev()
    _, b1, la, _, _ = bench.build(dims, W["imagine"], B, W["T"], seed=100, device=dev)
    _, b2, lb, _, _ = bench.build(dims, W["imagine"], B, W["T"], seed=200, device=dev)
    la._check_flat(); lb._check_flat()
    la.args.target_update_interval = lb.args.target_update_interval = 7

    class Plain:
        def __init__(self, b):
            self._b = b
            self.ready_event = None

        def __getattr__(self, k):
            return getattr(self._b, k)

        def __getitem__(self, k):
            return self._b[k]

    from refil_amd import _lib
    e0 = _lib.get_stat("early_target_hypernet_steps")
    for rnd in range(2):
        for i in range(20):
            la.train((b1, b2)[i % 2], t_env=0, episode_num=20 * rnd + i)        # no synchronisation: 20 steps in flight
        torch.cuda.synchronize()
        for i in range(20):
            lb.train(Plain((b1, b2)[i % 2]), t_env=0, episode_num=20 * rnd + i)
        torch.cuda.synchronize()
        assert torch.equal(la.flat_live, lb.flat_live), f"round {rnd}: max |d| = {(la.flat_live - lb.flat_live).abs().max().item():.3e}"
        assert torch.equal(la.flat_target, lb.flat_target) and torch.equal(la.square_avg, lb.square_avg)
    assert _lib.get_stat("early_target_hypernet_steps") - e0 >= 28


def test_max_t_filled_trim_trains_through_the_parent(monkeypatch):
    """The reference's run loop hands train() batch[:, :batch.max_t_filled()] -- a different length almost every step. QLearner
    trains through the untrimmed parent (the cut-off steps carry no loss weight and are skipped on the device anyway): the results
    equal training on the view as given (REFIL_UNTRIM=0) up to the summation order of the split reductions, and the batch shape --
    hence the early prologue and the early target forward -- survives from step to step."""
    import bench
    from refil_amd import _lib
    W = dict(bench.CONFIGS["cfg2"])
    dims = bench.workload_dims(W)
    dev = _dev()
    B = 8
    learners, batches = [], []
    for _ in range(2):
        _, b1, l, _, _ = bench.build(dims, W["imagine"], B, W["T"], seed=100, device=dev)
        _, b2, _, _, _ = bench.build(dims, W["imagine"], B, W["T"], seed=200, device=dev)
        _, b3, _, _, _ = bench.build(dims, W["imagine"], B, W["T"], seed=300, device=dev)
        for b, L in ((b1, 33), (b2, 57), (b3, 46)):              # longest episode of each batch; its last step is filled and NOT
            b.data.transition_data["filled"][:, L:] = 0            # terminated, so the transition the trim cuts off (L-1 -> L)
            b.data.transition_data["terminated"][:, L - 1:] = 0    # would carry loss weight in the parent without refil_batch.t_limit
        l._check_flat()
        learners.append(l); batches.append((b1, b2, b3))
    la, lb = learners
    assert torch.equal(la.flat_live, lb.flat_live)
    e0, s0 = _lib.get_stat("early_prologue_steps"), _lib.get_stat("learner_steps")
    for i in range(6):
        monkeypatch.delenv("REFIL_UNTRIM", raising=False)
        b = batches[0][i % 3]
        v = b[:, :b.max_t_filled()]
        assert v.max_seq_length in (33, 57, 46) and v._untrimmed is b
        la.train(v, t_env=0, episode_num=i)
        assert la._last_dims.T1 == W["T"] + 1                    # trained on the parent's length
        monkeypatch.setenv("REFIL_UNTRIM", "0")
        b = batches[1][i % 3]
        lb.train(b[:, :b.max_t_filled()], t_env=0, episode_num=i)
        assert lb._last_dims.T1 in (33, 57, 46)
        torch.cuda.synchronize()
        # same transitions, same loss, same gradients (up to the summation order of the split reductions: the grids differ)
        ga, gb = la.grads[:la._n], lb.grads[:lb._n]
        assert (ga - gb).abs().max().item() <= 2e-6 * gb.abs().max().item(), i
        sa, sb = la.grads[la._n:].tolist(), lb.grads[lb._n:].tolist()
        assert sa[_lib.STAT_MASK_SUM] == sb[_lib.STAT_MASK_SUM] and abs(sa[_lib.STAT_TD_SQ] - sb[_lib.STAT_TD_SQ]) <= 1e-5 * abs(sb[_lib.STAT_TD_SQ])
        # (RMSprop's first steps are steep where |g| ~ eps: re-align the replicas so that every step is compared from one state)
        lb.flat_live.copy_(la.flat_live); lb.square_avg.copy_(la.square_avg); lb.flat_target.copy_(la.flat_target)
    # la kept its layout: every step after the first took the early prologue; lb's layout changed every step: none did
    assert _lib.get_stat("learner_steps") - s0 == 12 and _lib.get_stat("early_prologue_steps") - e0 == 5


def test_out_of_band_target_writes_are_seen(monkeypatch):
    """target_mixer.load_state_dict / target_mac.load_state between two train() calls (no _update_targets, no load_models): the target
    parameters' own version counters change, so the next step neither runs the target forward early nor reuses the composed target maps
    of the previous call. Learner A relies on that detection, learner B bumps the explicit epoch: bit-identical parameters."""
    import bench
    monkeypatch.delenv("REFIL_EARLY", raising=False)
    monkeypatch.setenv("REFIL_EARLY_TARGET", "1")
    W = dict(bench.CONFIGS["cfg2"])
    dims = bench.workload_dims(W)
    dev = _dev()
    _, b1, la, _, _ = bench.build(dims, W["imagine"], 8, W["T"], seed=100, device=dev)
    _, b2, lb, _, _ = bench.build(dims, W["imagine"], 8, W["T"], seed=100, device=dev)
    la._check_flat(); lb._check_flat()
    torch.manual_seed(5)
    new_mix = {k: torch.randn_like(v) * 0.1 for k, v in la.target_mixer.state_dict().items()}
    new_agent = {k: torch.randn_like(v) * 0.1 for k, v in la.target_mac.agent.state_dict().items()}
    for i in range(8):
        if i == 4:                                  # (steps 1-3 ran with the early target forward: the cached state is warm)
            v0 = la._tv()
            for l in (la, lb):
                l.target_mixer.load_state_dict(new_mix)
                l.target_mac.agent.load_state_dict(new_agent)
            assert la._tv() != v0, "the target parameters were rewritten: the version must change"
            lb._target_epoch += 1
        la.train(b1, t_env=0, episode_num=i)
        lb.train(b2, t_env=0, episode_num=i)
        torch.cuda.synchronize()
        assert torch.equal(la.flat_target, lb.flat_target)
        assert torch.equal(la.flat_live, lb.flat_live), f"step {i}: max |d| = {(la.flat_live - lb.flat_live).abs().max().item():.3e}"
