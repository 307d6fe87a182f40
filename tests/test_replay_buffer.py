"""ReplayBuffer (reference: src/components/episode_buffer.py:206-246): ring insertion with wrap-around, uniform sampling
without replacement -- host logic on CPU; the device-side gather (refil_replay_gather) on the GPU."""
import numpy as np
import pytest
import torch as th

from refil_amd.components.episode_buffer import EpisodeBatch, ReplayBuffer
from refil_amd.components.transforms import OneHot


def _scheme(ne=6, na=3, ed=5, A=4):
    scheme = {
        "entities": {"vshape": ed, "group": "entities"},
        "obs_mask": {"vshape": ne, "group": "entities", "dtype": th.uint8},
        "entity_mask": {"vshape": ne, "dtype": th.uint8},
        "actions": {"vshape": (1,), "group": "agents", "dtype": th.long},
        "avail_actions": {"vshape": (A,), "group": "agents", "dtype": th.int},
        "reward": {"vshape": (1,)},
        "terminated": {"vshape": (1,), "dtype": th.uint8},
        "epsilon": {"vshape": (1,), "episode_const": True},
    }
    return scheme, {"agents": na, "entities": ne}, {"actions": ("actions_onehot", [OneHot(out_dim=A)])}


def _episodes(n, T1, tag0, device="cpu", ne=6, na=3, ed=5, A=4):
    scheme, groups, pre = _scheme(ne, na, ed, A)
    b = EpisodeBatch(scheme, groups, n, T1, preprocess=pre, device=device)
    g = th.Generator().manual_seed(tag0)
    data = {
        "entities": th.randn(n, T1, ne, ed, generator=g), "obs_mask": (th.rand(n, T1, ne, ne, generator=g) < 0.3).to(th.uint8),
        "entity_mask": (th.rand(n, T1, ne, generator=g) < 0.3).to(th.uint8), "actions": th.randint(0, A, (n, T1, na, 1), generator=g),
        "avail_actions": th.ones(n, T1, na, A, dtype=th.int), "terminated": th.zeros(n, T1, 1, dtype=th.uint8),
        "reward": th.arange(n, dtype=th.float32).view(n, 1, 1).expand(n, T1, 1) + tag0,          # tags the episode
    }
    b.update(data)
    b.update({"epsilon": th.full((n, 1), float(tag0))})
    return b


def test_ring_insert_wraps_around_like_reference():
    scheme, groups, pre = _scheme()
    buf = ReplayBuffer(scheme, groups, 5, 4, preprocess=pre)
    assert not buf.can_sample(1)
    buf.insert_episode_batch(_episodes(3, 4, 100))
    assert buf.episodes_in_buffer == 3 and buf.buffer_index == 3 and buf.can_sample(3) and not buf.can_sample(4)
    buf.insert_episode_batch(_episodes(4, 4, 200))                     # 2 fit, 2 wrap to slots 0, 1 (episode_buffer.py:225-228)
    assert buf.episodes_in_buffer == 5 and buf.buffer_index == 2
    tags = buf["reward"][:, 0, 0].tolist()
    assert tags == [202.0, 203.0, 102.0, 200.0, 201.0]
    assert buf["epsilon"][:, 0].tolist() == [200.0, 200.0, 100.0, 200.0, 200.0]
    assert (buf["filled"] == 1).all()
    oh = buf["actions_onehot"]
    assert th.equal(oh.argmax(-1, keepdim=True), buf["actions"]) and (oh.sum(-1) == 1).all()   # preprocess ran on insert


def test_sample_is_uniform_without_replacement():
    scheme, groups, pre = _scheme()
    buf = ReplayBuffer(scheme, groups, 8, 3, preprocess=pre)
    buf.insert_episode_batch(_episodes(8, 3, 0))
    np.random.seed(3)
    s = buf.sample(5)
    assert s.batch_size == 5 and s.max_seq_length == 3
    tags = s["reward"][:, 0, 0].tolist()
    assert len(set(tags)) == 5 and all(0 <= t < 8 for t in tags)
    np.random.seed(3)
    ids = np.random.choice(8, 5, replace=False)                          # the reference's draw (episode_buffer.py:238)
    assert tags == [float(i) for i in ids]
    full = buf.sample(8)                                                 # batch == buffer: the first batch_size episodes, in order (:235-236)
    assert full["reward"][:, 0, 0].tolist() == [float(i) for i in range(8)]


@pytest.mark.gpu
def test_device_gather_equals_fancy_indexing():
    scheme, groups, pre = _scheme(ne=8, na=4, ed=7, A=5)
    buf = ReplayBuffer(scheme, groups, 12, 9, preprocess=pre, device="cuda")
    buf.insert_episode_batch(_episodes(7, 9, 10, device="cuda", ne=8, na=4, ed=7, A=5))
    buf.insert_episode_batch(_episodes(9, 9, 50, device="cuda", ne=8, na=4, ed=7, A=5))      # wraps
    assert buf.episodes_in_buffer == 12 and buf.buffer_index == 4
    for seed in (1, 2, 3):
        np.random.seed(seed)
        got = buf.sample(6)                                   # refil_replay_gather
        np.random.seed(seed)
        ids = np.random.choice(12, 6, replace=False)
        ref = buf[ids]                                        # the reference's path
        th.cuda.synchronize()
        assert "obs_mask" in buf._packed and "obs_mask" not in buf.data.transition_data        # stored as words, served as bytes
        assert set(got.data.transition_data) == set(ref.data.transition_data)
        for k, v in ref.data.transition_data.items():
            assert th.equal(got.data.transition_data[k], v), k
        for k, v in ref.data.episode_data.items():
            assert th.equal(got.data.episode_data[k], v), k
        staging = {sl["batch"]["entities"].data_ptr() for sl in buf._staging[6]["slots"]}
        assert got.batch_size == 6 and got.max_seq_length == 9 and got["entities"].data_ptr() in staging
    # the two staging minibatches alternate (the previous sample may still be training: tests/test_gpu_early.py): fixed addresses
    a, b, c = (buf.sample(6)["entities"].data_ptr() for _ in range(3))
    assert a == c and a != b and {a, b} == staging


@pytest.mark.gpu
def test_gather_rejects_bad_arguments():
    import ctypes as C
    from refil_amd import _lib
    f = (_lib.GatherField * 1)(_lib.GatherField(0, 0, 16, 16, 16, 0, 0))
    ids = th.zeros(2, dtype=th.int64, device="cuda")
    assert _lib.lib().refil_replay_gather(f, 1, _lib.ptr(ids), 2, 4, None) != 0
    assert b"field 0" in _lib.lib().refil_last_error()


@pytest.mark.gpu
@pytest.mark.parametrize("ne", [8, 33, 64])
def test_bit_packed_mask_storage_equals_byte_storage(ne):
    """obs_mask kept as one int64 word per row in the device buffer against the byte storage
    (pack_masks=False): insertion with ring wrap-around, buffer["obs_mask"], fancy indexing and the gather launch's expansion
    into the staging minibatch are bit-identical; the packed buffer holds 8 bytes per mask row instead of ne."""
    scheme, groups, pre = _scheme(ne=ne, na=4, ed=7, A=5)
    bufs = [ReplayBuffer(scheme, groups, 10, 6, preprocess=pre, device="cuda", pack_masks=pm) for pm in (True, False)]
    for tag, n in ((10, 7), (50, 6)):                       # the second insertion wraps
        eps = _episodes(n, 6, tag, device="cuda", ne=ne, na=4, ed=7, A=5)
        for b in bufs:
            b.insert_episode_batch(eps)
    packed, plain = bufs
    assert "obs_mask" in packed._packed and packed._packed["obs_mask"].shape == (10, 6, ne) and packed._packed["obs_mask"].dtype == th.int64
    assert not plain._packed and plain["obs_mask"].shape == (10, 6, ne, ne)
    assert th.equal(packed["obs_mask"], plain["obs_mask"])
    ids = [7, 0, 3]
    assert th.equal(packed[ids]["obs_mask"], plain[ids]["obs_mask"])
    for seed in (1, 2):
        np.random.seed(seed)
        a = packed.sample(5)
        np.random.seed(seed)
        b = plain.sample(5)
        th.cuda.synchronize()
        assert a["obs_mask"].dtype == th.uint8 and a["obs_mask"].shape == (5, 6, ne, ne)
        for k in b.data.transition_data:
            assert th.equal(a[k], b[k]), k
    s1 = packed.sample(5, copy=True)
    keep = s1["entities"].clone()
    packed.sample(5)                                          # overwrites the staging minibatch, not the copy
    assert th.equal(s1["entities"], keep)


def test_slices_are_views_that_share_events():
    """batch[:, :max_t] (the reference's max_t_filled() trim, run.py:269-270) is a view of the batch's memory: it carries the
    parent's ready_event and hands a consumer's consumed_event up the chain (EpisodeBatch.__getitem__)."""
    b = _episodes(4, 6, 1)
    b.ready_event = object()
    v = b[:, :3]
    assert v.ready_event is b.ready_event and v._parent is b and v.max_seq_length == 3
    assert v["entities"].data_ptr() == b["entities"].data_ptr()
    w = v[1:3]
    assert w._parent is v and w.batch_size == 2
    f = b[[0, 2]]                                            # fancy indexing copies: no parent, no event
    assert getattr(f, "_parent", None) is None and f.ready_event is None


@pytest.mark.gpu
def test_update_on_packed_keys_marks_filled_and_orders_the_gather():
    """ReplayBuffer.update is a legal write path (reference: episode_buffer.py:77-105): with ONLY bit-packed keys in `data` it
    still marks `filled`, checks the shape, and records the write event the early gather of sample() waits for."""
    scheme, groups, pre = _scheme(ne=8, na=4, ed=7, A=5)
    buf = ReplayBuffer(scheme, groups, 6, 5, preprocess=pre, device="cuda")
    om = (th.rand(2, 5, 8, 8) < 0.5).to(th.uint8)
    assert getattr(buf, "_write_event", None) is None
    buf.update({"obs_mask": om}, bs=slice(1, 3), ts=slice(0, 5))
    assert buf._write_event is not None
    assert th.equal(buf["obs_mask"][1:3].cpu(), om)
    assert (buf["filled"][1:3] == 1).all() and (buf["filled"][0] == 0).all() and (buf["filled"][3:] == 0).all()
    with pytest.raises(ValueError):
        buf.update({"obs_mask": om[:, :, :4]}, bs=slice(1, 3), ts=slice(0, 5))
    buf.update({"obs_mask": om}, bs=slice(3, 5), ts=slice(0, 5), mark_filled=False)
    assert (buf["filled"][3:5] == 0).all()
    buf.to("cuda")                                           # (moves the packed words as well)
    assert buf._packed["obs_mask"].is_cuda


@pytest.mark.gpu
def test_direct_update_is_ordered_before_the_early_gather():
    """A write through the public update() on the caller's stream, then sample(): the gather on the library's side stream must see
    it (it waits for the buffer's write event). Repeated with a slow kernel in front of the write to open the race window."""
    scheme, groups, pre = _scheme(ne=8, na=4, ed=7, A=5)
    buf = ReplayBuffer(scheme, groups, 8, 9, preprocess=pre, device="cuda")
    buf.insert_episode_batch(_episodes(8, 9, 10, device="cuda", ne=8, na=4, ed=7, A=5))
    big = th.randn(4096, 4096, device="cuda")
    for it in range(6):
        np.random.seed(it)
        for _ in range(3):
            big = big @ big * 1e-3                          # keeps the caller's stream busy in front of the write
        new = th.full((8, 9, 1), 1000.0 + it, device="cuda")
        buf.update({"reward": new}, mark_filled=False)
        s = buf.sample(5)
        th.cuda.synchronize()
        assert (s["reward"] == 1000.0 + it).all(), it


@pytest.mark.gpu
def test_host_resident_buffer_hands_out_device_minibatches():
    """buffer_cpu_only (run.py:199-200): storage in pinned host memory, sample() = the one-launch gather reading host memory
    from the GPU. Same episodes, same bytes as the reference's host-side fancy indexing; inserts between samples (ring wrap) are
    seen; the early path (side stream, alternating staging minibatches) and the in-order path agree."""
    scheme, groups, pre = _scheme(ne=8, na=4, ed=7, A=5)
    buf = ReplayBuffer(scheme, groups, 12, 9, preprocess=pre, device="cpu", sample_device="cuda")
    assert buf.sample_device is not None and all(v.is_pinned() and not v.is_cuda for v in buf.data.transition_data.values())
    buf.insert_episode_batch(_episodes(7, 9, 10, ne=8, na=4, ed=7, A=5))
    full = buf.sample(7)                                      # batch == buffer: the first episodes in order, on the device
    assert full["reward"].is_cuda and full["reward"][:, 0, 0].tolist() == [10.0 + i for i in range(7)]
    buf.insert_episode_batch(_episodes(9, 9, 50, device="cuda", ne=8, na=4, ed=7, A=5))      # from a device runner; wraps
    assert buf.episodes_in_buffer == 12 and buf.buffer_index == 4
    for seed in (1, 2, 3, 4):
        np.random.seed(seed)
        got = buf.sample(6)
        np.random.seed(seed)
        ids = np.random.choice(12, 6, replace=False)
        ref = buf[ids]                                        # host-side fancy indexing (the reference's path)
        th.cuda.synchronize()
        assert got.device == th.device("cuda") or str(got.device).startswith("cuda")
        for k, v in ref.data.transition_data.items():
            assert got[k].is_cuda and th.equal(got[k].cpu(), v), k
        for k, v in ref.data.episode_data.items():
            assert th.equal(got[k].cpu(), v), k
        if seed == 2:                                         # a host write between two samples waits for the gather in flight
            buf.update({"reward": th.full((12, 9, 1), 777.0)}, mark_filled=False)
            np.random.seed(99)
            assert (buf.sample(6)["reward"] == 777.0).all()



def test_max_t_filled_trim_is_recognised():
    """batch[:, :batch.max_t_filled()] (run.py:269-270) is marked as a trim that cuts off unfilled steps only; any other slice is not."""
    b = _episodes(4, 9, 1)
    b.data.transition_data["filled"][:, 6:] = 0
    m = b.max_t_filled()
    assert int(m) == 6
    v = b[:, :m]
    assert v._untrimmed is b and v.max_seq_length == 6
    assert getattr(b[:, :5], "_untrimmed", None) is None          # a shorter cut drops filled steps
    assert getattr(b[1:, :m], "_untrimmed", None) is None         # a batch slice is another batch
    assert getattr(_episodes(4, 9, 2)[:, :6], "_untrimmed", None) is None     # max_t_filled() was never asked


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("REFIL_FUZZ_REPLAY_N", "10"))))
def test_random_schemes_gather_equals_fancy_indexing(seed):
    """Replay fuzz: random entity / agent counts (mask rows of 1..64 bits, packed and byte storage), feature widths that are not
    multiples of 4 (unaligned field sizes: the gather's scalar tail), buffer sizes, inserts that wrap the ring more than once,
    device and pinned-host storage -- sample() through refil_replay_gather equals the reference's fancy indexing bit for bit."""
    import random
    rnd = random.Random(1000 + seed)
    ne = rnd.choice([1, 2, 5, 8, 13, 31, 32, 33, 47, 63, 64])
    na = rnd.randint(1, ne)
    ed, A, T1, cap = rnd.randint(1, 37), rnd.randint(2, 19), rnd.randint(1, 23), rnd.randint(3, 17)
    where = rnd.choice(["device", "device_bytes", "host"])
    scheme, groups, pre = _scheme(ne=ne, na=na, ed=ed, A=A)
    kw = dict(device="cuda") if where != "host" else dict(device="cpu", sample_device="cuda")
    if where == "device_bytes":
        kw["pack_masks"] = False
    buf = ReplayBuffer(scheme, groups, cap, T1, preprocess=pre, **kw)
    ref = ReplayBuffer(scheme, groups, cap, T1, preprocess=pre, device="cpu")          # the reference's path: host storage, fancy indexing
    tag = 0
    for _ in range(rnd.randint(2, 6)):
        n = rnd.randint(1, cap)
        eps = _episodes(n, T1, 100 * tag, ne=ne, na=na, ed=ed, A=A)
        tag += 1
        buf.insert_episode_batch(eps)
        ref.insert_episode_batch(eps)
        assert buf.episodes_in_buffer == ref.episodes_in_buffer and buf.buffer_index == ref.buffer_index
        for _ in range(2):
            bs = rnd.randint(1, buf.episodes_in_buffer)
            s0 = rnd.randint(0, 10 ** 6)
            np.random.seed(s0)
            got = buf.sample(bs)
            np.random.seed(s0)
            want = ref.sample(bs)
            th.cuda.synchronize()
            assert got.batch_size == want.batch_size == bs
            for k, v in want.data.transition_data.items():
                assert th.equal(got[k].cpu(), v), (k, where, ne, na, ed, A, T1, cap)
            for k, v in want.data.episode_data.items():
                assert th.equal(got[k].cpu(), v), k
