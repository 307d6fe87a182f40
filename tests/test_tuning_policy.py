"""Policy around the launch-size knobs (refil_amd/tuning.py): what a timed run may use is a closed, parity-tested set; the autotuner
is opt-in; its key ignores the batch length and size (the reference's run loop trims every sampled batch to max_t_filled(),
src/run.py:269-270: a key with T1 in it would re-tune per distinct episode length)."""
import pytest

from refil_amd import _lib, tuning


def _dims(B, T1, ne=32):
    return _lib.make_dims(B=B, T1=T1, ne=ne, na=ne // 2, ed=62, A=22, d=128, heads=4, H=64, hyp=128, M=32, entity_last_action=1,
                          imagine=1, softmax_mixing_weights=1, double_q=1, gamma=0.99, lmbda=0.5)


def test_bucket_key_ignores_batch_length_and_size_within_a_row_bucket():
    k = tuning.bucket_key(_dims(32, 81))
    assert k == tuning.bucket_key(_dims(32, 80)) == tuning.bucket_key(_dims(32, 70))      # the same power-of-two row bucket
    assert k == tuning.bucket_key(_dims(16, 160))                                          # B and T1 are not in the key, their product is
    assert k != tuning.bucket_key(_dims(32, 40)) and k != tuning.bucket_key(_dims(64, 81))
    assert k != tuning.bucket_key(_dims(32, 81, ne=16))                                    # network / environment dims are
    lengths = {tuning.bucket_key(_dims(32, t)) for t in range(2, 152)}                     # every length a 150-step env can produce
    assert len(lengths) <= 8                                                               # = at most MAX_TUNES measurements


def test_only_parity_tested_values_pass():
    assert tuning.check({"dw4_target": 96, "gru_pd": 2, "dw4_min_out": 2000, "dw_target": 384, "compose_early": 1})
    for bad in ({"dw4_target": 112}, {"gru_pd": 3}, {"nonsense": 1}, {"dw_target": 1024}):
        with pytest.raises(ValueError):
            tuning.check(bad)
    assert tuning.parse_env("dw4_target=96,gru_pd=2") == {"dw4_target": 96, "gru_pd": 2}
    with pytest.raises(ValueError):
        tuning.parse_env("dw4_target=77")
    assert tuning.check({"dws_target": 128}) and tuning.check({"dws_target": 64, "attn_qkv": 3})
    with pytest.raises(ValueError):
        tuning.check({"dws_target": 32})                     # (measured, slower, and without a parity case)
    for knob, values in tuning.CANDIDATES:                   # what the tuner tries is inside what the tests cover
        assert all(v in tuning.PARITY_TESTED[knob] for v in values)


def test_autotune_is_opt_in(monkeypatch):
    monkeypatch.delenv("REFIL_AUTOTUNE", raising=False)
    assert tuning.mode() == "off"
    monkeypatch.setenv("REFIL_AUTOTUNE", "0")
    assert tuning.mode() == "off"
    monkeypatch.setenv("REFIL_AUTOTUNE", "1")
    assert tuning.mode() == "measure"
    monkeypatch.setenv("REFIL_AUTOTUNE", "gru_pd=2")
    assert tuning.mode() == {"gru_pd": 2}


def test_apply_tuning_takes_back_the_non_candidate_knobs_it_set(monkeypatch):
    """A setting that names a non-candidate knob (attn_qkv, wres_split ...: REFIL_AUTOTUNE strings / cache entries may) must not leak
    it into later settings that do not; a value the CALLER set through refil_set_tuning is never touched."""
    from refil_amd.learners.q_learner import QLearner
    state, calls = {}, []

    class FakeLib:
        def refil_set_tuning(self, name, value):
            calls.append((name.decode(), int(value)))
            state[name.decode()] = int(value)
            return 0

    monkeypatch.setattr(_lib, "lib", lambda: FakeLib())
    monkeypatch.setattr(QLearner, "_APPLIED", [None])
    monkeypatch.setattr(QLearner, "_NAMED", [frozenset()])
    state["wres_split"] = 0                                   # the caller's own choice, made before the first train()
    QLearner._apply_tuning({"gru_pd": 2, "attn_qkv": 0})
    assert state["attn_qkv"] == 0 and state["gru_pd"] == 2 and state["wres_split"] == 0
    assert "wres_split" not in [k for k, _ in calls]
    QLearner._apply_tuning({"dw4_target": 96})                # another bucket's setting: it does not name attn_qkv
    assert state["attn_qkv"] == -1 and state["gru_pd"] == -1 and state["dw4_target"] == 96 and state["wres_split"] == 0
    assert QLearner._APPLIED[0] == {"dw4_target": 96}
    calls.clear()
    QLearner._apply_tuning({"dw4_target": 96})                # unchanged: no calls
    assert calls == []
    QLearner._apply_tuning({})
    assert "attn_qkv" not in [k for k, _ in calls] and state["dw4_target"] == -1


def test_a_dropped_step_is_not_logged_as_training():
    """grad_norm NaN with a finite loss = the optimiser kernel dropped the step (row-list time-out): the log step raises; a NaN loss is
    a diverged run and is logged like the reference logs it."""
    from refil_amd.learners.q_learner import QLearner
    st = [100.0, 3.0, 2.0, 5.0, 1.0, 1.0, float("nan"), 0.0]
    with pytest.raises(RuntimeError, match="REFIL_LISTS_FUSED=0"):
        QLearner._check_step_not_dropped(st)
    st[_lib.STAT_GRAD_NORM] = 0.7
    QLearner._check_step_not_dropped(st)
    QLearner._check_step_not_dropped([100.0, float("nan"), 2.0, 5.0, 1.0, 1.0, float("nan"), 0.0])


def test_bench_useful_fraction_of_the_fused_attention_launch():
    """bench.py prints, beside the FLOPs credited to attn_qkv_fwd (whole 16-entity tiles), the share on entity rows that can influence
    the loss: 1 on dense data, ~0.57 at the round-5 cfg-T row counts (profiles/r05_qkv_bench.txt: 34 162 of 60 416 live entity slots)."""
    import bench
    ne, na, steps = 32, 16, 2592
    dense = dict(live_steps=steps, entity_rows_agent=steps * ne, entity_rows_hyper=steps * ne, agent_rows=steps * na)
    assert bench.qkv_useful_fraction(dense, ne, na) == pytest.approx(1.0)
    live = 1888
    r05 = dict(live_steps=live, entity_rows_agent=int(0.38 * steps * ne), entity_rows_hyper=int(0.49 * steps * ne), agent_rows=int(0.48 * steps * na))
    u = bench.qkv_useful_fraction(r05, ne, na)
    assert 0.5 < u < 0.75
    assert bench.qkv_useful_fraction(dict(live_steps=live, entity_rows_agent=0, entity_rows_hyper=0, agent_rows=0), ne, na) == 0.0


def test_autotune_cache_round_trip_and_rejection(monkeypatch, tmp_path):
    """REFIL_AUTOTUNE_CACHE: a measured setting is kept per (device, library version, shape bucket) and reused by later processes; an entry with a
    value outside the parity-tested set (an old file, a hand edit) is ignored, as is an unreadable file -- the caller then measures again."""
    import json
    path = tmp_path / "tune.json"
    monkeypatch.setenv("REFIL_AUTOTUNE_CACHE", str(path))
    k1, k2 = tuning.bucket_key(_dims(32, 81)), tuning.bucket_key(_dims(32, 81, ne=16))
    assert tuning.cache_get(k1) is None                               # no file yet
    tuning.cache_put(k1, {"dw4_target": 96, "gru_pd": 2})
    tuning.cache_put(k2, {})
    assert tuning.cache_get(k1) == {"dw4_target": 96, "gru_pd": 2} and tuning.cache_get(k2) == {}
    assert tuning.cache_get(tuning.bucket_key(_dims(64, 81))) is None    # another bucket: not measured yet
    d = json.load(open(path))
    d[next(k for k in d if d[k])]["dw4_target"] = 112                 # a value without parity coverage
    json.dump(d, open(path, "w"))
    assert tuning.cache_get(k1) is None
    path.write_text("{ not json")
    assert tuning.cache_get(k1) is None
    tuning.cache_put(k1, {"gru_pd": 2})                               # an unreadable file is replaced, not fatal
    assert tuning.cache_get(k1) == {"gru_pd": 2}
    monkeypatch.delenv("REFIL_AUTOTUNE_CACHE")
    tuning.cache_put(k1, {"gru_pd": 4})                               # no cache configured: nothing written, nothing read
    assert tuning.cache_get(k1) is None
