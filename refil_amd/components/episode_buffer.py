"""Scheme-driven episode storage with the reference's API (src/components/episode_buffer.py:6-246):
EpisodeBatch = dict of dense tensors [B, T, (group), *vshape] plus a `filled` mask; ReplayBuffer = ring
of episodes with uniform sampling. Kept device-resident so that a sampled, time-truncated view is
handed to the HIP learner without copies (the C ABI takes batch/time strides)."""
from types import SimpleNamespace

import os

import numpy as np
import torch as th


def _as_tuple(v):
    return (v,) if isinstance(v, int) else tuple(v)


class EpisodeBatch:
    def __init__(self, scheme, groups, batch_size, max_seq_length, data=None, preprocess=None, device="cpu"):
        self.scheme = dict(scheme)
        self.groups = groups
        self.batch_size = batch_size
        self.max_seq_length = max_seq_length
        self.preprocess = {} if preprocess is None else preprocess
        self.device = device
        # optional th.cuda.Event recorded behind the last write of the fields by whoever assembled the batch ahead of time: lets
        # QLearner.train start the step's input-only kernels before the previous step has finished (q_learner.py: early prologue)
        self.ready_event = None
        self.consumed_event = None       # recorded by QLearner.train behind its last read of the batch (ReplayBuffer's staging reuse)
        if data is not None:
            self.data = data
            return
        self.data = SimpleNamespace(transition_data={}, episode_data={})
        self._allocate(self.scheme, groups, batch_size, max_seq_length, self.preprocess)

    # -- construction -------------------------------------------------------------------------
    def _allocate(self, scheme, groups, batch_size, max_seq_length, preprocess):
        for key, (new_key, transforms) in (preprocess or {}).items():
            assert key in scheme, f"preprocess source {key} not in scheme"
            vshape, dtype = self.scheme[key]["vshape"], self.scheme[key].get("dtype", th.float32)
            for tr in transforms:
                vshape, dtype = tr.infer_output_info(vshape, dtype)
            entry = {"vshape": vshape, "dtype": dtype}
            for inherit in ("group", "episode_const"):
                if inherit in self.scheme[key]:
                    entry[inherit] = self.scheme[key][inherit]
            self.scheme[new_key] = entry
        assert "filled" not in scheme, '"filled" is reserved for the validity mask'
        scheme["filled"] = {"vshape": (1,), "dtype": th.long}
        self.scheme.setdefault("filled", scheme["filled"])
        for key, info in scheme.items():
            assert "vshape" in info, f"scheme entry {key} needs a vshape"
            shape = _as_tuple(info["vshape"])
            group = info.get("group")
            if group:
                assert group in groups, f"group {group} has no size"
                shape = (groups[group],) + shape
            dtype = info.get("dtype", th.float32)
            if info.get("episode_const", False):
                self.data.episode_data[key] = th.zeros((batch_size,) + shape, dtype=dtype, device=self.device)
            else:
                self.data.transition_data[key] = th.zeros((batch_size, max_seq_length) + shape, dtype=dtype, device=self.device)

    def extend(self, scheme, groups=None):
        self._allocate(scheme, self.groups if groups is None else groups, self.batch_size, self.max_seq_length, None)

    def to(self, device):
        for store in (self.data.transition_data, self.data.episode_data):
            for k in store:
                store[k] = store[k].to(device)
        self.device = device

    # -- writes -------------------------------------------------------------------------------
    def update(self, data, bs=slice(None), ts=slice(None), mark_filled=True):
        sl = self._parse_slices((bs, ts))
        for k, v in data.items():
            if k in self.data.transition_data:
                store, where = self.data.transition_data, sl
                if mark_filled:
                    store["filled"][sl] = 1
                    mark_filled = False
            elif k in self.data.episode_data:
                store, where = self.data.episode_data, sl[0]
            else:
                raise KeyError(f"{k} not found in transition or episode data")
            dtype = self.scheme[k].get("dtype", th.float32)
            v = v.to(device=self.device, dtype=dtype) if isinstance(v, th.Tensor) else th.tensor(v, dtype=dtype, device=self.device)
            dest = store[k][where]
            self._check_safe_view(v, dest)
            store[k][where] = v.view_as(dest)
            if k in self.preprocess:
                new_k, transforms = self.preprocess[k]
                out = store[k][where]
                for tr in transforms:
                    out = tr.transform(out)
                store[new_k][where] = out.view_as(store[new_k][where])

    @staticmethod
    def _check_safe_view(v, dest):
        idx = v.dim() - 1
        for s in reversed(dest.shape):
            if idx >= 0 and v.shape[idx] == s:
                idx -= 1
            elif s != 1:
                raise ValueError(f"Unsafe reshape of {tuple(v.shape)} to {tuple(dest.shape)}")

    # -- reads --------------------------------------------------------------------------------
    def __getitem__(self, item):
        if isinstance(item, str):
            if item in self.data.episode_data:
                return self.data.episode_data[item]
            if item in self.data.transition_data:
                return self.data.transition_data[item]
            raise ValueError(item)
        if isinstance(item, tuple) and all(isinstance(it, str) for it in item):
            new = SimpleNamespace(transition_data={}, episode_data={})
            for key in item:
                if key in self.data.transition_data:
                    new.transition_data[key] = self.data.transition_data[key]
                elif key in self.data.episode_data:
                    new.episode_data[key] = self.data.episode_data[key]
                else:
                    raise KeyError(f"Unrecognised key {key}")
            scheme = {k: self.scheme[k] for k in item}
            groups = {self.scheme[k]["group"]: self.groups[self.scheme[k]["group"]] for k in item if "group" in self.scheme[k]}
            return EpisodeBatch(scheme, groups, self.batch_size, self.max_seq_length, data=new, device=self.device)
        sl = self._parse_slices(item)
        new = SimpleNamespace(transition_data={k: v[sl] for k, v in self.data.transition_data.items()},
                              episode_data={k: v[sl[0]] for k, v in self.data.episode_data.items()})
        out = EpisodeBatch(self.scheme, self.groups, self._count(sl[0], self.batch_size),
                           self._count(sl[1], self.max_seq_length), data=new, device=self.device)
        # a basic slice (e.g. the reference's batch[:, :max_t_filled()] trim, run.py:269-270) is a VIEW of this batch's memory: it
        # is ready when this batch is, and whoever consumes it has consumed (part of) this batch -- QLearner.train records
        # consumed_event on the whole chain of parents, so ReplayBuffer.sample's staging reuse stays ordered behind the reader
        if all(isinstance(i, slice) for i in sl):
            out.ready_event = self.ready_event
            out._parent = self
            # batch[:, :batch.max_t_filled()] -- the reference's trim -- cuts off nothing but steps no episode has filled: the
            # learner may train on the untrimmed parent instead (QLearner.train: same loss, and the batch shape stays the same from
            # step to step). Recognised by VALUE: the stop of the time slice equals what max_t_filled() returned for this batch.
            mtf = self.__dict__.get("_mtf")
            bsl, tsl = sl
            if (mtf is not None and bsl == slice(None) and tsl.start in (None, 0) and tsl.step in (None, 1) and tsl.stop is not None
                    and int(tsl.stop) == mtf[0]):
                out._untrimmed = self
        return out

    @staticmethod
    def _count(idx, size):
        if isinstance(idx, (list, np.ndarray)):
            return len(idx)
        if isinstance(idx, th.Tensor):
            return idx.numel()
        lo, hi, step = idx.indices(size)
        return 1 + (hi - lo - 1) // step

    @staticmethod
    def _parse_slices(items):
        if isinstance(items, (slice, int, list, np.ndarray, th.Tensor)):
            items = (items, slice(None))
        if isinstance(items[1], list):
            raise IndexError("Indexing across Time must be contiguous")
        return tuple(slice(it, it + 1) if isinstance(it, int) else it for it in items)

    def max_t_filled(self):
        m = th.sum(self.data.transition_data["filled"], 1).max(0)[0]
        # (remembered for __getitem__: recognising the reference's batch[:, :max_t_filled()] trim; the int() is the host
        # synchronisation the caller's slice would perform anyway)
        self.__dict__["_mtf"] = (int(m),)
        return m

    def __repr__(self):
        return (f"EpisodeBatch. Batch Size:{self.batch_size} Max_seq_len:{self.max_seq_length} "
                f"Keys:{self.scheme.keys()} Groups:{self.groups.keys()}")


class ReplayBuffer(EpisodeBatch):
    """Ring of episodes with the reference's API (episode_buffer.py:206-246).

    Device buffers keep byte masks whose rows are at most 64 wide (`obs_mask`, `gt_mask`: uint8 [.., ne, ne]) BIT-PACKED --
    one int64 word per row (bit j = mask[.., i, j]), 8 bytes instead of ne: `insert_episode_batch`
    packs (refil_pack_mask_bits), `sample` expands the sampled episodes into the staging minibatch inside the one gather
    launch, and `buffer["obs_mask"]` / `buffer[ids]` expand on demand with torch ops (API compatibility, not a hot path).
    pack_masks=False keeps the bytes.

    Host-resident buffers (`buffer_cpu_only: True`, src/run.py:199-200,272-273, default.yaml:17): with device="cpu" and
    sample_device=<the learner's GPU> the storage lives in PINNED host memory and `sample()` hands out a DEVICE minibatch:
    the same one-launch gather reads the sampled episodes straight out of host memory (the pinned pages are mapped into
    the GPU's address space: a zero-copy read over PCIe, no host-side fancy indexing, no staging copy), on the library's side
    stream, beside the previous train step -- the reference's `episode_sample.to(args.device)` (a synchronous fancy-index copy
    on the host followed by a blocking H2D) has nothing left to do. Inserts are host writes: they wait for gathers in flight."""

    PACKABLE = ("obs_mask", "gt_mask")

    def __init__(self, scheme, groups, buffer_size, max_seq_length, preprocess=None, device="cpu", pack_masks=None, sample_device=None):
        super().__init__(scheme, groups, buffer_size, max_seq_length, preprocess=preprocess, device=device)
        self.buffer_size = buffer_size
        self.buffer_index = 0
        self.episodes_in_buffer = 0
        self._packed = {}            # key -> int64 [N, T1, rows] words; the byte tensor of the key is dropped
        self.sample_device = None    # host-resident storage, device minibatches (see the class docstring)
        self._write_count = 0
        self._gather_done = None     # event behind the last zero-copy gather (host writes wait for it)
        if sample_device is not None and th.device(device).type == "cpu" and th.device(sample_device).type == "cuda":
            self.sample_device = th.device(sample_device)
            for store in (self.data.transition_data, self.data.episode_data):
                for k in store:
                    store[k] = store[k].pin_memory()
        on_gpu = th.device(device).type == "cuda"
        if pack_masks is None:
            pack_masks = on_gpu
        if pack_masks and on_gpu:
            for k in self.PACKABLE:
                v = self.data.transition_data.get(k)
                if v is not None and v.dtype == th.uint8 and v.dim() == 4 and v.shape[-1] <= 64:
                    self._packed[k] = th.zeros(v.shape[:-1], dtype=th.int64, device=self.device)
                    self._packed_width = getattr(self, "_packed_width", {})
                    self._packed_width[k] = v.shape[-1]
                    del self.data.transition_data[k]

    # -- packed byte masks ----------------------------------------------------------------------
    def _pack(self, key, v):
        """uint8 [..., rows, width] (device) -> int64 [..., rows] through refil_pack_mask_bits"""
        import ctypes as C
        from .. import _lib
        v = v.to(device=self.device, dtype=th.uint8).contiguous()
        out = th.empty(v.shape[:-1], dtype=th.int64, device=self.device)
        _lib.check(_lib.lib().refil_pack_mask_bits(_lib.ptr(v), _lib.ptr(out), C.c_int64(out.numel()), C.c_int32(v.shape[-1]),
                                                   _lib.current_stream_ptr()), "refil_pack_mask_bits")
        return out

    def _unpack(self, key, words):
        w = self._packed_width[key]
        sh = th.arange(w, device=words.device, dtype=th.int64)
        return ((words[..., None] >> sh) & 1).to(th.uint8)

    def to(self, device):
        super().to(device)
        for k in self._packed:
            self._packed[k] = self._packed[k].to(device)

    def update(self, data, bs=slice(None), ts=slice(None), mark_filled=True):
        if self._gather_done is not None:                   # host-resident storage: a gather launch may still be reading it
            self._gather_done.synchronize()
        self._write_count = getattr(self, "_write_count", 0) + 1
        packed = {k: v for k, v in data.items() if k in self._packed}
        if packed:
            sl = self._parse_slices((bs, ts))
            for k, v in packed.items():
                dest = self._packed[k][sl]
                v = v if isinstance(v, th.Tensor) else th.as_tensor(v)
                self._check_safe_view(v, th.empty(tuple(dest.shape) + (self._packed_width[k],), device="meta"))
                self._packed[k][sl] = self._pack(k, v).view_as(dest)
            if mark_filled:                                 # (packed keys are transition data: same rule as EpisodeBatch.update)
                self.data.transition_data["filled"][sl] = 1
                mark_filled = False
            data = {k: v for k, v in data.items() if k not in self._packed}
        if data:
            super().update(data, bs, ts, mark_filled)
        self._record_write()

    def _record_write(self):
        """Every write path of the buffer ends here: sample()'s early gather runs on the library's side stream and waits for
        this event, so a write through the public update() on the caller's stream is ordered before the gather that follows."""
        if th.device(self.device).type == "cuda":
            if getattr(self, "_write_event", None) is None:
                self._write_event = th.cuda.Event()
            self._write_event.record()

    def __getitem__(self, item):
        if th.device(self.device).type == "cuda":
            self._views_handed_out = True                  # (sample()'s early gather then also waits for the caller's stream)
        if not self._packed:
            return super().__getitem__(item)
        if isinstance(item, str):
            if item in self._packed:
                return self._unpack(item, self._packed[item])
            return super().__getitem__(item)
        if isinstance(item, tuple) and all(isinstance(it, str) for it in item):
            out = super().__getitem__(tuple(k for k in item if k not in self._packed))
            for k in item:
                if k in self._packed:
                    out.data.transition_data[k] = self._unpack(k, self._packed[k])
                    out.scheme[k] = self.scheme[k]
            return out
        out = super().__getitem__(item)
        sl = self._parse_slices(item)
        for k, words in self._packed.items():
            out.data.transition_data[k] = self._unpack(k, words[sl])
        return out

    def insert_episode_batch(self, ep_batch):
        n = ep_batch.batch_size
        if self.buffer_index + n > self.buffer_size:                       # wrap around the ring
            left = self.buffer_size - self.buffer_index
            self.insert_episode_batch(ep_batch[0:left, :])
            self.insert_episode_batch(ep_batch[left:, :])
            return
        where = slice(self.buffer_index, self.buffer_index + n)
        self.update(ep_batch.data.transition_data, where, slice(0, ep_batch.max_seq_length), mark_filled=False)
        self.update(ep_batch.data.episode_data, where)
        self.buffer_index += n
        self.episodes_in_buffer = max(self.episodes_in_buffer, self.buffer_index)
        self.buffer_index %= self.buffer_size               # (update() recorded the write event the early gather of sample() waits for)

    def can_sample(self, batch_size):
        return self.episodes_in_buffer >= batch_size

    def sample(self, batch_size, copy=False):
        """Uniform sample without replacement (episode_buffer.py:233-240). On a device buffer the result is one of the buffer's
        two STAGING minibatches for this batch size -- fixed addresses, alternating, overwritten by the sample(batch_size) after
        the next one (with REFIL_EARLY=0 / REFIL_HIPGRAPH=1: a single one, overwritten by the next call); pass copy=True to get
        an independent batch (the reference always returns a fresh copy) when more samples must be alive at once."""
        assert self.can_sample(batch_size)
        if self.sample_device is not None:
            out = self._sample_host(batch_size)
            if copy:
                out = EpisodeBatch(out.scheme, out.groups, out.batch_size, out.max_seq_length, device=out.device,
                                   data=SimpleNamespace(transition_data={k: v.clone() for k, v in out.data.transition_data.items()},
                                                        episode_data={k: v.clone() for k, v in out.data.episode_data.items()}))
            return out
        if self.episodes_in_buffer == batch_size:
            return self[:batch_size]
        ep_ids = np.random.choice(self.episodes_in_buffer, batch_size, replace=False)     # uniform, w/o replacement
        if th.device(self.device).type == "cuda":
            out = self._gather(ep_ids)
            if copy:
                out = EpisodeBatch(out.scheme, out.groups, out.batch_size, out.max_seq_length, device=out.device,
                                   data=SimpleNamespace(transition_data={k: v.clone() for k, v in out.data.transition_data.items()},
                                                        episode_data={k: v.clone() for k, v in out.data.episode_data.items()}))
            return out
        return self[ep_ids]

    # -- device-resident sampling -----------------------------------------
    def _gather(self, ep_ids):
        """The sampled episodes, copied by ONE HIP launch (refil_replay_gather) into a staging minibatch that is
        reused by later calls with the same batch size (fixed addresses: no allocation per step). Same contents as the
        reference's `self[ep_ids]` (episode_buffer.py:123-159); the reference's max_t_filled() trim (run.py:266-273) --
        a host synchronisation per step -- is not needed, the learner skips finished episodes' steps on the device."""
        import ctypes as C
        from .. import _lib
        n = len(ep_ids)
        ep_ids = np.ascontiguousarray(ep_ids, dtype=np.int64)
        if ep_ids.size and (ep_ids.min() < 0 or ep_ids.max() >= self.buffer_size):
            raise IndexError(f"episode ids must lie in [0, {self.buffer_size})")
        nfields = len(self.data.transition_data) + len(self.data.episode_data) + len(self._packed)
        if nfields > _lib.MAX_GATHER_FIELDS:            # (more scheme keys than one gather launch takes: the reference's path)
            return self[ep_ids]
        early = os.environ.get("REFIL_EARLY") != "0" and os.environ.get("REFIL_HIPGRAPH") != "1"
        return self._gather_on(ep_ids, n, early, self.device)

    # -- host-resident storage, device minibatches (buffer_cpu_only) -------------------------------
    def _draw(self, n):
        """episode_buffer.py:233-240: the first n episodes when the buffer holds exactly n, else uniform without replacement"""
        if self.episodes_in_buffer == n:
            return np.arange(n)
        return np.random.choice(self.episodes_in_buffer, n, replace=False)

    def _sample_host(self, n):
        """Host-resident storage: the one-launch gather reads the sampled episodes out of pinned host memory (zero-copy over PCIe)
        into a device staging minibatch, on the library's side stream like a device buffer's early gather. Measured and NOT
        adopted: a look-ahead gather of the NEXT call's sample on a stream of the buffer's own (exact: the generator was put back
        after the look-ahead draw and the prefetch dropped when anything had changed) -- a fifth busy stream is a cliff on this
        runtime (cfg-T 2.28 -> 3.15 ms per step, cfg2 0.93 -> 2.25; DESIGN.md lesson 23), and on one of the step's four streams
        the gather sits in front of that stream's chain whichever one it is."""
        import ctypes  # noqa: F401
        from .. import _lib
        dev = self.sample_device
        ep_ids = np.ascontiguousarray(self._draw(n), dtype=np.int64)
        if ep_ids.size and (ep_ids.min() < 0 or ep_ids.max() >= self.buffer_size):
            raise IndexError(f"episode ids must lie in [0, {self.buffer_size})")
        if len(self.data.transition_data) + len(self.data.episode_data) + len(self._packed) > _lib.MAX_GATHER_FIELDS:
            # more scheme keys than one gather launch takes: the reference's path (host-side indexing, then the copy to the device)
            out = self[ep_ids]
            out.to(dev)
            return out
        early = os.environ.get("REFIL_EARLY") != "0" and os.environ.get("REFIL_HIPGRAPH") != "1"
        with th.cuda.device(dev):
            out = self._gather_on(ep_ids, n, early, dev)
            if self._gather_done is None:
                self._gather_done = th.cuda.Event()
            side = self._staging[n].get("side")
            self._gather_done.record(side if (early and side is not None) else th.cuda.current_stream(dev))
        return out

    def _gather_on(self, ep_ids, n, early, dev):
        import ctypes as C
        from .. import _lib
        st = self._staging.get(n) if hasattr(self, "_staging") else None
        if st is None:
            if not hasattr(self, "_staging"):
                self._staging = {}
            slots = []
            for _ in range(2):                             # two staging minibatches: the previous sample may still be training
                tdata = {k: th.zeros((n,) + tuple(v.shape[1:]), dtype=v.dtype, device=dev) for k, v in self.data.transition_data.items()}
                for k, words in self._packed.items():      # the staging minibatch holds the BYTES the learner's C ABI takes
                    tdata[k] = th.zeros((n,) + tuple(words.shape[1:]) + (self._packed_width[k],), dtype=th.uint8, device=dev)
                batch = EpisodeBatch(self.scheme, self.groups, n, self.max_seq_length, device=dev,
                                     data=SimpleNamespace(
                                         transition_data=tdata,
                                         episode_data={k: th.zeros((n,) + tuple(v.shape[1:]), dtype=v.dtype, device=dev)
                                                       for k, v in self.data.episode_data.items()}))
                fields = (_lib.GatherField * _lib.MAX_GATHER_FIELDS)()
                nf = 0
                for store_src, store_dst in ((self.data.transition_data, batch.data.transition_data),
                                             (self.data.episode_data, batch.data.episode_data)):
                    for k, src in store_src.items():
                        dst = store_dst[k]
                        assert src.is_contiguous() and dst.is_contiguous()
                        eb = src[0].numel() * src.element_size()
                        fields[nf] = _lib.GatherField(src.data_ptr(), dst.data_ptr(), eb, eb, eb, 0, 0)
                        nf += 1
                for k, words in self._packed.items():
                    dst = batch.data.transition_data[k]
                    fields[nf] = _lib.GatherField(words.data_ptr(), dst.data_ptr(), words[0].numel() * 8, dst[0].numel(), dst[0].numel(),
                                                  self._packed_width[k], 0)
                    nf += 1
                slots.append({"batch": batch, "fields": fields, "nf": nf, "ids_dev": th.empty(n, dtype=th.int64, device=dev),
                              "handed_out": False})
            ids_host = [(th.empty(n, dtype=th.int64).pin_memory(), th.cuda.Event()) for _ in range(8)]
            st = self._staging[n] = {"slots": slots, "ids_host": ids_host, "slot": 0, "which": 0}
        host, ev = st["ids_host"][st["slot"]]
        st["slot"] = (st["slot"] + 1) % len(st["ids_host"])
        ev.synchronize()                                   # (the async upload issued 8 samples ago has long completed)
        host.copy_(th.from_numpy(ep_ids))
        if not early:
            # in order on the caller's stream, always into the same staging minibatch (fixed addresses: REFIL_HIPGRAPH)
            sl = st["slots"][0]
            sl["ids_dev"].copy_(host, non_blocking=True)
            ev.record()
            _lib.check(_lib.lib().refil_replay_gather(sl["fields"], C.c_int32(sl["nf"]), _lib.ptr(sl["ids_dev"]), C.c_int32(n),
                                                      C.c_int64(self.buffer_size), _lib.current_stream_ptr()), "refil_replay_gather")
            sl["batch"].ready_event = None
            return sl["batch"]
        # Early: the gather runs on the library's hypernet-chain stream (refil_side_stream: behind the previous train step's
        # hypernet backward, in front of the next step's early prologue -- NOT behind the whole previous step, and not on a fifth
        # stream), alternating between the two staging minibatches. It waits for the buffer's last insert and for the last reader
        # of the staging minibatch it overwrites (QLearner.train records batch.consumed_event; a consumer that does not is
        # covered by an event on the caller's stream at this point). The caller's stream waits for the gather, so whatever else
        # reads the batch there is ordered as before; train()'s early prologue waits for batch.ready_event only.
        st["which"] ^= 1
        sl = st["slots"][st["which"]]
        batch = sl["batch"]
        sp = C.c_void_p()
        _lib.check(_lib.lib().refil_side_stream(C.byref(sp)), "refil_side_stream")
        if st.get("side_ptr") != sp.value:                 # (the library re-creates its streams after refil_release_streams)
            st["side_ptr"], st["side"] = sp.value, th.cuda.ExternalStream(sp.value, device=dev)
        side = st["side"]
        cur = th.cuda.current_stream(dev)
        wev = getattr(self, "_write_event", None)
        if wev is not None:
            side.wait_event(wev)
        if getattr(self, "_views_handed_out", False):
            # buffer[...] handed out a tensor that aliases buffer storage: a write through it bypasses update() and its event,
            # so the gather is also ordered behind everything enqueued on the caller's stream so far
            aev = st.setdefault("alias_ev", th.cuda.Event())
            aev.record(cur)
            side.wait_event(aev)
        # the last reader of this staging minibatch -- or, for a consumer that records nothing and on first use (the zero fill
        # of a new staging tensor is enqueued on the caller's stream), everything enqueued on the caller's stream so far
        cev = batch.consumed_event if sl["handed_out"] else None
        if cev is None:
            cev = sl.setdefault("cur_ev", th.cuda.Event())
            cev.record(cur)
        side.wait_event(cev)
        with th.cuda.stream(side):
            sl["ids_dev"].copy_(host, non_blocking=True)
            ev.record()
            _lib.check(_lib.lib().refil_replay_gather(sl["fields"], C.c_int32(sl["nf"]), _lib.ptr(sl["ids_dev"]), C.c_int32(n),
                                                      C.c_int64(self.buffer_size), C.c_void_p(sp.value)), "refil_replay_gather")
            if batch.ready_event is None:
                batch.ready_event = th.cuda.Event()
            batch.ready_event.record()
        cur.wait_event(batch.ready_event)
        batch.consumed_event = None
        sl["handed_out"] = True
        return batch

    def __repr__(self):
        return (f"ReplayBuffer. {self.episodes_in_buffer}/{self.buffer_size} episodes. "
                f"Keys:{self.scheme.keys()} Groups:{self.groups.keys()}")
