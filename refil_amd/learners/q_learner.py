"""QLearner with the reference's plugin surface (reference: src/learners/q_learner.py:10-229) driving
the MI355X-native step: one C-ABI call for forward+backward, one flat RCCL all-reduce under data
parallelism, one C-ABI call for clip + RMSprop. No autograd, no torch.optim.
"""
from __future__ import annotations

import copy
import os

import torch as th
from .. import _lib, dp, tuning
from ..engine import LearnerEngine, dims_from_args
from ..modules.mixers.flex_qmix import FlexQMixer, LinearFlexQMixer
from ..modules.mixers.vdn import VDNMixer


def _require_gpu(dev):
    """The learner has no CPU path: its parameters must live on the GPU before the first train() / save / load. (A function of its own so
    that the CPU test tier can run this class on tests/emu -- the kernel sources on a CPU wavefront emulator, tests/emu_util.py -- by
    standing in for it there; nothing in the package does.)"""
    if dev.type != "cuda":
        raise RuntimeError("refil_amd.QLearner runs on the GPU only: call learner.cuda() first (no CPU fallback)")


class QLearner:
    def __init__(self, mac, scheme, logger, args):
        self.args = args
        self.mac = mac
        self.logger = logger
        self.last_target_update_episode = 0
        self.mixer = None
        if args.mixer is not None:
            if args.mixer == "flex_qmix":
                assert args.entity_scheme, "FlexQMixer only available with entity scheme"
                self.mixer = FlexQMixer(args)
            elif args.mixer == "lin_flex_qmix":
                assert args.entity_scheme, "FlexQMixer only available with entity scheme"
                self.mixer = LinearFlexQMixer(args)
            elif args.mixer == "vdn":
                self.mixer = VDNMixer()
            elif args.mixer == "qmix":
                raise NotImplementedError(f"mixer {args.mixer} is outside the REFIL hot path built so far (SURVEY.md 8f4)")
            else:
                raise ValueError("Mixer {} not recognised.".format(args.mixer))
        # args.mixer = None (q_learner.py:19-21): no mixing network, `learner.mixer is None` like the reference. The flat
        # parameter buffer then has an empty mixer section, carried by a parameter-less holder.
        self._mix = self.mixer if self.mixer is not None else VDNMixer()
        self.params = list(mac.parameters()) + list(self._mix.parameters())       # q_learner.py:16,34 order
        if self.mixer is not None:
            self.target_mixer = copy.deepcopy(self.mixer)
        self._tmix = self.target_mixer if self.mixer is not None else VDNMixer()
        self.target_mac = copy.deepcopy(mac)
        self.log_stats_t = -self.args.learner_log_interval - 1
        self._step_count = 0
        self._target_epoch = 0
        self._engine = None
        self._flat_ready = False
        self.generator = None      # optional torch.Generator for the partition draw (defaults to the global CPU RNG)

    # ------------------------------------------------------------------------------------------
    def _setup_flat(self):
        """[agent | mixer] flat buffers for live / target nets, RMSprop state and gradients."""
        dev = next(self.mac.agent.parameters()).device
        _require_gpu(dev)
        self._engine = LearnerEngine(dev)
        d0 = dims_from_args(self.args, 1, 2)
        L = _lib.param_layout(d0)
        self._L, self._n, self._na = L, int(L.total), int(L.agent_total)
        self.flat_live = th.zeros(self._n, dtype=th.float32, device=dev)
        self.flat_target = th.zeros(self._n, dtype=th.float32, device=dev)
        self.square_avg = th.zeros(self._n, dtype=th.float32, device=dev)
        self.grads = th.zeros(self._n + _lib.REFIL_NSTAT, dtype=th.float32, device=dev)
        self.mac.agent.adopt(self.flat_live[:self._na])
        self._mix.adopt(self.flat_live[self._na:])
        self.target_mac.agent.adopt(self.flat_target[:self._na])
        self._tmix.adopt(self.flat_target[self._na:])
        self.params = list(self.mac.parameters()) + list(self._mix.parameters())
        g_agent = self.grads[:self._na]
        g_mixer = self.grads[self._na:self._n]
        for mod, g in ((self.mac.agent, g_agent), (self._mix, g_mixer)):      # expose .grad as views of the flat grads
            named = dict(mod.named_parameters())
            for name, off, shape in mod._fields():
                n = 1
                for s in shape:
                    n *= s
                named[name].grad = g[off:off + n].view(*shape)
        self._bits_host = None
        self._bits_dev = None
        self._graphs = {}
        self._buckets = None
        self._flat_ready = True

    def _check_flat(self):
        if not self._flat_ready:
            self._setup_flat()
            return
        for mod, st in ((self.mac.agent, self.flat_live[:self._na]), (self._mix, self.flat_live[self._na:]),
                        (self.target_mac.agent, self.flat_target[:self._na]), (self._tmix, self.flat_target[self._na:])):
            if getattr(mod, "_flat", None) is None or mod._flat.data_ptr() != st.data_ptr() or not mod._is_flat():
                mod.adopt(st)

    def _draw_partition(self, B, ne, device, bernoulli=True):
        """entity_rnn_agent.py:94-96: p = rand(B,1,1); groupA = bernoulli(p.repeat(1,1,ne)) -- drawn with the
        CPU generator (same two calls, same stream as the reference's CPU run) and shipped to the device.
        With use_gt_factors the FF agent consumes only the rand() call (entity_ff_agent.py:91-95): bernoulli=False.

        train() never synchronises the host, so the async H2D copy of step k may still be queued when step k+1
        draws: the pinned staging buffers form a ring, and a slot is rewritten only after the event recorded
        behind its last copy has completed."""
        # data parallel: every rank draws the partition of the GLOBAL batch from an identically seeded generator and keeps
        # the rows of its shard (episodes [rank*B, (rank+1)*B) of the global batch, dp.shard_episodes), so that the union over
        # the ranks IS the single-process draw and the generators stay in step
        W, R = dp.world(), dp.rank()
        p = th.rand(B * W, 1, 1, generator=self.generator).repeat(1, 1, ne)
        if not bernoulli:
            return None
        bits = th.bernoulli(p, generator=self.generator).to(th.uint8).reshape(B * W, ne)
        if W > 1:
            bits = dp.shard_bits(bits, R, W)
        if self._bits_host is None or self._bits_host[0][0].shape != bits.shape:
            self._bits_host = [(th.empty_like(bits).pin_memory(), th.cuda.Event()) for _ in range(16)]
            self._bits_slot = 0
        host, ev = self._bits_host[self._bits_slot]
        self._bits_slot = (self._bits_slot + 1) % len(self._bits_host)
        ev.synchronize()                       # no-op unless the copy issued 16 steps ago is still pending
        host.copy_(bits)
        # one device buffer per shape, reused by every step (stream order protects it): a fixed address, so that a
        # captured step (REFIL_HIPGRAPH=1) finds the partition where it was at capture time
        if self._bits_dev is None or self._bits_dev.shape != bits.shape or self._bits_dev.device != th.device(device):
            self._bits_dev = th.empty(bits.shape, dtype=th.uint8, device=device)
        self._bits_dev.copy_(host, non_blocking=True)
        ev.record()
        return self._bits_dev

    # ------------------------------------------------------------------------------------------
    # First-call autotuner (opt-in: REFIL_AUTOTUNE=1; refil_amd/tuning.py holds the policy and the parity-tested value sets).
    # A few launch-size / launch-order knobs of the step (include/refil_hip.h: refil_set_tuning) have no best value that a rule
    # predicts: the step runs four streams deep, a kernel's time alone says little about the step's (DESIGN.md lessons 10, 22),
    # and the optimum moves with the shape (cfg-T: 96 workgroups per 4x4-tile weight-gradient launch -1.4 %, cfg4: +1.7 %). With
    # REFIL_AUTOTUNE=1 the first train() call on a shape BUCKET measures the candidates in situ -- forward_backward on the
    # caller's own batch, parameters untouched, no RNG consumed -- greedily, one knob at a time, interleaved A/B/A/B, and keeps a
    # candidate only if it wins by more than the run-to-run spread. ~1 s once per bucket, at most tuning.MAX_TUNES buckets per
    # process. The knobs move the summation order of the split weight-gradient reductions with them: results agree to
    # rounding, not bit for bit, between settings -- which is why the DEFAULT is the deterministic built-in schedule.
    _TUNED = {}          # tuning.bucket_key(dims) -> {knob: value}
    _APPLIED = [None]
    _MEASURED = [0]      # buckets measured by this process

    _NAMED = [frozenset()]   # non-candidate knobs the applied setting named: set by this module, so this module takes them back

    @staticmethod
    def _apply_tuning(setting):
        if QLearner._APPLIED[0] == setting:
            return
        # the autotuner's own knobs are reset to their defaults when the setting does not name them; the arithmetic-form / kernel-choice
        # knobs (wres_split, dw_split, attn_qkv, dws_target: never candidates) are left as the CALLER set them -- refil_set_tuning("wres_split", 0)
        # before the first train() is the documented way to compare with earlier builds -- unless an earlier setting
        # of THIS module named one (a REFIL_AUTOTUNE string or cache entry with attn_qkv=0): that value goes back to the default with
        # the setting that brought it, so tuning_chosen() is what the library runs
        own = {k for k, _ in tuning.CANDIDATES}
        for k in tuning.PARITY_TESTED:
            if k in own or k in setting or k in QLearner._NAMED[0]:
                _lib.check(_lib.lib().refil_set_tuning(k.encode(), int(setting.get(k, -1))), "refil_set_tuning")
        QLearner._NAMED[0] = frozenset(k for k in setting if k not in own)
        QLearner._APPLIED[0] = dict(setting)

    def tuning_chosen(self):
        """The setting the last train() call ran with ({} = built-in defaults)."""
        return dict(QLearner._APPLIED[0] or {})

    def _tune(self, dims, fields, bits, ready):
        key = tuning.bucket_key(dims)
        got = QLearner._TUNED.get(key)
        if got is None:
            mode = tuning.mode()
            rows = dims.B * dims.T1 * dims.ne
            if isinstance(mode, dict):                      # REFIL_AUTOTUNE="dw4_target=96,...": given, not measured
                got = mode
            elif (mode == "off" or rows < tuning.MIN_ROWS or os.environ.get("REFIL_HIPGRAPH") == "1" or
                  os.environ.get("REFIL_DP_BUCKETS") == "1" or os.environ.get("REFIL_DENSE") == "1"):
                got = {}
            else:
                got = tuning.cache_get(key)                 # REFIL_AUTOTUNE_CACHE=<file>: measured once, reused by later processes
                if got is None:
                    if QLearner._MEASURED[0] >= tuning.MAX_TUNES:
                        got = {}
                    else:
                        QLearner._MEASURED[0] += 1
                        got = self._autotune(dims, fields, bits, ready)
                        tuning.cache_put(key, got)
            QLearner._TUNED[key] = got
        QLearner._apply_tuning(got)

    def _autotune(self, dims, fields, bits, ready):
        e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)

        def timed(setting, n=10):
            QLearner._apply_tuning(setting)
            for _ in range(3):
                self._engine.forward_backward(dims, fields, bits, self.flat_live, self.flat_target, self.grads, ready_event=ready, target_version=self._tv())
            e0.record()
            for _ in range(n):
                self._engine.forward_backward(dims, fields, bits, self.flat_live, self.flat_target, self.grads, ready_event=ready, target_version=self._tv())
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) / n

        best = {}
        timed(best)                                        # (first touch: workspace, lazy initialisation)
        n_steps = max(10, min(40, int(25.0 / max(timed(best), 1e-3))))      # >= 25 ms per measurement: short steps need more of them
        log = []
        for knob, values in tuning.CANDIDATES:
            for v in values:
                cand = tuning.check(dict(best, **{knob: v}), "autotune candidate")
                ta = tb = 0.0
                for _ in range(3):                         # interleaved: drift hits both alike
                    ta += timed(best, n_steps)
                    tb += timed(cand, n_steps)
                log.append((knob, v, round(ta / 3, 4), round(tb / 3, 4)))
                if tb < 0.993 * ta:
                    best = cand
        self._autotune_log = log
        if os.environ.get("REFIL_AUTOTUNE_LOG") == "1":
            print(f"refil autotune B={dims.B} T1={dims.T1} ne={dims.ne}: {best or 'defaults'}  (knob, value, ms default, ms candidate): {log}",
                  flush=True)
        return best

    def _fields(self, batch):
        names = ["entities", "obs_mask", "entity_mask", "actions", "avail_actions", "reward", "terminated", "filled"]
        if getattr(self.args, "gt_mask_avail", False):
            names.append("gt_mask")
        out = {}
        for k in names:
            v = batch[k]
            if not v[0, 0].is_contiguous():
                v = v.contiguous()
            out[k] = v
        return out

    # ------------------------------------------------------------------------------------------
    def train(self, batch, t_env: int, episode_num: int, group_bits=None):
        self._check_flat()
        args = self.args
        # The reference's run loop trims every sampled batch to its longest episode (run.py:269-270). The library skips the steps no
        # episode has filled by itself, so a batch[:, :max_t_filled()] view is trained through its untrimmed parent: same loss and
        # gradients (the cut-off steps carry no loss weight and are never computed), and the batch shape -- hence the workspace
        # layout, the early prologue / early target forward and the tuner's bucket -- stays the same from step to step.
        # REFIL_UNTRIM=0: train on the view as given.
        t_limit = 0
        # (REFIL_HIPGRAPH=1: the captured step does not carry t_limit -- the view is trained as given)
        while (getattr(batch, "_untrimmed", None) is not None and os.environ.get("REFIL_UNTRIM") != "0" and
               os.environ.get("REFIL_HIPGRAPH") != "1" and
               args.mixer != "lin_flex_qmix"):         # (lin_flex_qmix logs ingroup_prop as a mean over ALL (b,t) of the batch it was given)
            t_limit = t_limit or batch.max_seq_length  # (refil_batch.t_limit: the transitions the view cut off stay cut off)
            batch = batch._untrimmed
        B, T1 = batch.batch_size, batch.max_seq_length
        dims = dims_from_args(args, B, T1)
        tgt, trgt = bool(getattr(args, "train_gt_factors", False)), bool(getattr(args, "train_rand_gt_factors", False))
        if not dims.agent_ff:       # ImagineEntityAttentionRNNAgent.forward swallows both flags (**kwargs, entity_rnn_agent.py:87)
            tgt = trgt = False
        assert not (tgt and trgt), "Can only select one of use_rand_gt_factors and use_gt_factors"   # entity_ff_agent.py:112
        dims.gt_factors = 2 if trgt else int(tgt)                                   # q_learner.py:88-89
        fields = self._fields(batch)
        dev = self.flat_live.device
        will_log = t_env - self.log_stats_t >= args.learner_log_interval
        gt_ingroup = None
        if will_log and dims.imagine and getattr(args, "test_gt_factors", False):
            gt_ingroup = self._gt_ingroup_prop(dims, fields, B, T1)                     # with the pre-update weights
        bits = None
        if dims.imagine:        # (rand() is drawn even when train_gt_factors ignores it: the reference consumes the RNG too)
            bits = group_bits.to(dev).to(th.uint8).contiguous() if group_bits is not None else \
                self._draw_partition(B, args.n_entities, dev, bernoulli=dims.gt_factors != 1)
        self._last_dims = dims
        # Early prologue: a batch that carries `ready_event` (ReplayBuffer.sample / whoever assembled it records it behind the
        # last write of its fields) lets the step's input assembly and row lists run on a side stream beside the END of the
        # previous step (refil_batch.ready_event, include/refil_hip.h). Same results; REFIL_EARLY=0 switches it off.
        ready = getattr(batch, "ready_event", None)
        if ready is not None and (os.environ.get("REFIL_EARLY") == "0" or os.environ.get("REFIL_HIPGRAPH") == "1" or
                                  any(fields[k].data_ptr() != batch[k].data_ptr() for k in fields)):      # (a field was copied just now)
            ready = None
        self._tune(dims, fields, bits, ready)
        # data parallel over episodes: all-reduce(SUM) of [grads | stat sums]; the global sum(mask) normaliser is
        # applied afterwards by the optimiser kernel (q_learner.py:165). One collective after the step, or
        # (REFIL_DP_BUCKETS=1) the mixer bucket underneath the agent's BPTT + the agent bucket after the step.
        if dp.world() > 1 and os.environ.get("REFIL_DP_BUCKETS") == "1":
            if self._buckets is None:
                self._buckets = dp.BucketedAllReduce(self.grads, self._na)
            with self._buckets:
                self._engine.forward_backward(dims, fields, bits, self.flat_live, self.flat_target, self.grads, ready_event=ready, target_version=self._tv(), t_limit=t_limit)
            self._buckets.finish()
            self._optimiser_step()
        elif os.environ.get("REFIL_HIPGRAPH") == "1" and group_bits is None:
            self._graphed_step(dims, fields, bits)
        elif dp.world() == 1 and os.environ.get("REFIL_DP_FORCE") != "1":
            a = self.args                              # one C call for the whole step (refil_learner_step)
            self._engine.step(dims, fields, bits, self.flat_live, self.flat_target, self.grads, self.square_avg, a.lr, a.optim_alpha,
                              a.optim_eps, a.weight_decay, a.grad_norm_clip, ready_event=ready, target_version=self._tv(), t_limit=t_limit)
        else:
            self._engine.forward_backward(dims, fields, bits, self.flat_live, self.flat_target, self.grads, ready_event=ready, target_version=self._tv(), t_limit=t_limit)
            dp.allreduce_sum_(self.grads)
            self._optimiser_step()
        if ready is not None:                          # (a producer that reuses the batch's memory waits for this: ReplayBuffer.sample)
            ce = batch.__dict__.get("_consumed_ev")
            if ce is None:
                ce = batch.__dict__["_consumed_ev"] = th.cuda.Event()
            ce.record()
            b = batch
            while b is not None:                       # (a time-trimmed view consumes its parents' memory: EpisodeBatch.__getitem__)
                b.consumed_event = ce
                b = getattr(b, "_parent", None)
        self._step_count += 1

        if (episode_num - self.last_target_update_episode) / args.target_update_interval >= 1.0:
            self._update_targets()
            self.last_target_update_episode = episode_num

        if t_env - self.log_stats_t >= args.learner_log_interval:
            st = self.grads[self._n:].tolist()                         # the only host sync, on log steps
            msum = st[_lib.STAT_MASK_SUM]
            self._check_step_not_dropped(st)
            q_loss = st[_lib.STAT_TD_SQ] / msum
            if dims.imagine:
                im_loss = st[_lib.STAT_IM_TD_SQ] / msum
                self.logger.log_stat("loss", (1 - args.lmbda) * q_loss + args.lmbda * im_loss, t_env)   # :172,185
                self.logger.log_stat("im_loss", im_loss, t_env)
            else:
                self.logger.log_stat("loss", q_loss, t_env)
            if getattr(args, "test_gt_factors", False) and dims.imagine:                # q_learner.py:188-190
                # (the stat sums are all-reduced over the data-parallel ranks; B is the per-rank shard)
                self.logger.log_stat("ingroup_prop", st[_lib.STAT_INGROUP_SUM] / (dp.world() * B * (T1 - 1)), t_env)
                self.logger.log_stat("gt_ingroup_prop", float(dp.mean_scalar(gt_ingroup)), t_env)
            self.logger.log_stat("grad_norm", st[_lib.STAT_GRAD_NORM], t_env)
            self.logger.log_stat("td_error_abs", st[_lib.STAT_TD_ABS] / msum, t_env)
            self.logger.log_stat("q_taken_mean", st[_lib.STAT_QTOT_SUM] / (msum * args.n_agents), t_env)    # :194 quirk kept
            self.logger.log_stat("target_mean", st[_lib.STAT_TARGET_SUM] / (msum * args.n_agents), t_env)
            self.log_stats_t = t_env

    @staticmethod
    def _check_step_not_dropped(st):
        """The optimiser kernel drops a step whose one-launch row lists timed out (mixer.hip: rmsprop_kernel): parameters untouched,
        grad_norm = NaN while the loss statistics of the forward pass are finite. Training on with frozen parameters would be silent
        -- the reference cannot get there --, so the log step that sees that signature raises. (A NaN loss is a diverged run, which the
        reference logs as it is: not this.)"""
        import math
        if math.isnan(st[_lib.STAT_GRAD_NORM]) and math.isfinite(st[_lib.STAT_TD_SQ]) and math.isfinite(st[_lib.STAT_MASK_SUM]):
            raise RuntimeError("refil: the optimiser dropped this step (grad_norm is NaN, the loss is finite): the one-launch row-list "
                               "kernel timed out waiting for its grid on this device (partitioned / CU-masked GPU, or another process "
                               "holding the compute units). Set REFIL_LISTS_FUSED=0; the next library call reports the same.")

    def _tv(self):
        """refil_batch.target_version: changes whenever the target parameters may have been rewritten -- the explicit counter of
        _update_targets / load_models plus torch's in-place version counters of the flat buffer AND of every target parameter.
        (FlatParamModule.adopt binds a parameter with `p.data = view`, which gives the Parameter a version counter of its own:
        target_mixer.load_state_dict(...) / target_mac.load_state(...) bump the parameters' counters, not the flat buffer's.)
        Unchanged since the previous call = the library may run the target nets' forward early, beside the end of the
        previous step, and keep their composed maps."""
        key = (id(self.target_mac), id(self._tmix))
        got = self.__dict__.get("_target_params")
        if got is None or got[0] != key:
            got = (key, list(self.target_mac.parameters()) + list(self._tmix.parameters()))
            self._target_params = got
        ps = got[1]
        v = self.flat_target._version
        for p in ps:
            v += p._version
        return ((self._target_epoch + 1) << 32) | (v & 0xFFFFFFFF)

    def _optimiser_step(self):
        a = self.args
        self._engine.clip_rmsprop(self.flat_live, self.grads, self.square_avg, self._n, a.lr, a.optim_alpha, a.optim_eps,
                                  a.weight_decay, a.grad_norm_clip)

    def _graphed_step(self, dims, fields, bits):
        """REFIL_HIPGRAPH=1 (opt-in): the ~80 launches of a step (four streams, fork / join by events, no allocation, no host
        sync) are captured into a hipGraph the second time a (shapes, buffer addresses) combination is seen and replayed
        afterwards -- one launch per step. Measured on ROCm 7.2 / MI355X (tools/probes/graph_capture.py, cfg2): the replay
        is bit-identical to the eager schedule but SLOWER (0.99 ms against 0.84 ms per step; the graph launch itself costs
        0.44 ms of host time), so it stays off by default. Single process: forward + backward + optimiser in one graph; data parallel:
        forward + backward only (the all-reduce and the optimiser stay eager). The batch tensors must keep their
        addresses (ReplayBuffer's device staging batch does); the kernels' grid sizes are frozen at the row counts of the
        captured step, which is safe (they loop over the device-side counts) but tuned for that batch."""
        # the graph holds raw addresses: the batch fields, the partition bits, the flat parameter / gradient buffers AND the
        # engine's workspace arena (grow-only: a larger request re-allocates it, and a graph captured against the old arena
        # would replay into freed memory). The arena is sized for this step BEFORE the key is taken.
        import ctypes as _C
        nbytes = _lib.lib().refil_learner_workspace_bytes(_C.byref(dims))
        self._engine.ws.get(nbytes)
        key = (bytes(dims), tuple((k, v.data_ptr(), v.stride(0), v.stride(1)) for k, v in sorted(fields.items())),
               0 if bits is None else bits.data_ptr(), dp.world(), self._engine.ws.buf.data_ptr(), self.flat_live.data_ptr(),
               self.flat_target.data_ptr(), self.grads.data_ptr(), self.square_avg.data_ptr())
        ent = self._graphs.get(key)
        fused = dp.world() == 1

        def body():
            self._engine.forward_backward(dims, fields, bits, self.flat_live, self.flat_target, self.grads)
            if fused:
                self._optimiser_step()

        if ent is None:                         # first sight: eager (lazy initialisation inside the library happens here)
            self._graphs[key] = ent = {"graph": None}
            body()
        elif ent["graph"] is None:
            g = th.cuda.CUDAGraph()
            with th.cuda.graph(g):
                body()
            ent["graph"] = g
            g.replay()
        else:
            ent["graph"].replay()
        if not fused:
            dp.allreduce_sum_(self.grads)
            self._optimiser_step()

    def _gt_ingroup_prop(self, dims, fields, B, T1):
        """Log-step-only diagnostic of refil_group_matching (q_learner.py:98-105,143-147): imagine with the
        ground-truth factors and report the in-group mixing-weight mass of LinearFlexQMixer."""
        from ..engine import clone_dims
        if not dims.mixer_lin:
            raise TypeError("test_gt_factors needs lin_flex_qmix (FlexQMixer.forward has no ret_ingroup_prop, flex_qmix.py:79)")
        T = T1 - 1
        dg = clone_dims(dims, gt_factors=1)
        q, _ = self._engine.agent_forward(dg, fields, None, self.flat_live, None, first_step_zero=True)
        acts = fields["actions"][:, :-1].long()
        chosen = th.gather(q[:, :, :-1], 4, acts[None].expand(3, -1, -1, -1, -1)).squeeze(4)      # glue: [3,B,T,na]
        caq_im = th.cat([chosen[1], chosen[2]], dim=2).contiguous()
        mfields = {k: v[:, :-1] for k, v in fields.items()}
        _, _, ing = self._engine.mixer_forward(clone_dims(dg, T1=T), mfields, None, self.flat_live, chosen[0].contiguous(),
                                               caq_im, 0, T, want_ingroup=True)
        return ing[0] / (B * T)

    def _update_targets(self):
        self._check_flat()
        self.flat_target.copy_(self.flat_live)          # one flat D2D copy (q_learner.py:203-207)
        self._target_epoch += 1
        self.logger.console_logger.info("Updated target network")

    def cuda(self):
        self.mac.cuda()
        self.target_mac.cuda()
        if self.mixer is not None:                      # q_learner.py:212-214
            self.mixer.cuda()
            self.target_mixer.cuda()
        self._flat_ready = False

    # -- checkpoints in the reference's format (agent.th / mixer.th / opt.th) -------------------
    def _opt_state_dict(self):
        self._check_flat()
        state = {}
        idx = 0
        for mod, base in ((self.mac.agent, 0), (self._mix, self._na)):
            offs = {name: (off, shape) for name, off, shape in mod._fields()}
            for name, _ in mod.named_parameters():
                off, shape = offs[name]
                n = 1
                for s in shape:
                    n *= s
                state[idx] = {"step": self._step_count,
                              "square_avg": self.square_avg[base + off:base + off + n].view(*shape).clone()}
                idx += 1
        group = {"lr": self.args.lr, "momentum": 0, "alpha": self.args.optim_alpha, "eps": self.args.optim_eps,
                 "centered": False, "weight_decay": self.args.weight_decay, "params": list(range(idx))}
        return {"state": state, "param_groups": [group]}

    def _load_opt_state_dict(self, sd):
        self._check_flat()
        idx = 0
        for mod, base in ((self.mac.agent, 0), (self._mix, self._na)):
            offs = {name: (off, shape) for name, off, shape in mod._fields()}
            for name, _ in mod.named_parameters():
                off, shape = offs[name]
                n = 1
                for s in shape:
                    n *= s
                st = sd["state"].get(idx)
                if st is not None:
                    self.square_avg[base + off:base + off + n].view(*shape).copy_(st["square_avg"])
                    step = st.get("step", 0)
                    self._step_count = int(step.item() if hasattr(step, "item") else step)
                idx += 1

    def save_models(self, path):
        self.mac.save_models(path)
        if self.mixer is not None:                      # q_learner.py:218-219
            th.save(self.mixer.state_dict(), "{}/mixer.th".format(path))
        th.save(self._opt_state_dict(), "{}/opt.th".format(path))

    def load_models(self, path, evaluate=False):
        self.mac.load_models(path)
        self.target_mac.load_models(path)       # like the reference: targets are not checkpointed (:224-225)
        self._target_epoch += 1
        if not evaluate:
            if self.mixer is not None:                  # q_learner.py:226-227
                self.mixer.load_state_dict(th.load("{}/mixer.th".format(path), map_location=lambda storage, loc: storage))
            self._load_opt_state_dict(th.load("{}/opt.th".format(path), map_location=lambda storage, loc: storage))
