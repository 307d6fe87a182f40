"""Data-parallel plumbing of the learner step (one process per GPU, torch.distributed; backend "nccl" is
RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The path shards over EPISODES: every op of the step is independent across
batch rows, the only cross-rank quantity is the loss normaliser sum(mask) (q_learner.py:165,171).
Recipe: each rank back-propagates the UN-normalised loss of its shard; ONE all-reduce(SUM) of the flat
buffer [gradients | stat sums] then yields both the global gradient sum and the global sum(mask); the
optimiser kernel divides by it. Replicas stay bit-identical (same reduced buffer, same update), so no
parameter broadcast is needed after step 0. The message is small (1.74 MB at the north-star shape), i.e.
latency-bound on xGMI: one collective per step, never one per tensor.
"""
from __future__ import annotations

import os
from typing import Dict

import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


_oneshot = {}        # (data_ptr, numel) -> OneShotAllReduce


def allreduce_sum_(flat: torch.Tensor) -> torch.Tensor:
    """In-place SUM over ranks of the flat [grads | stats] buffer (no-op for a single process).
    REFIL_ALLREDUCE=oneshot: the library's one-hop peer-memory all-reduce instead of the backend's collective."""
    if world() > 1 or (os.environ.get("REFIL_DP_FORCE") == "1" and dist.is_available() and dist.is_initialized()):
        # (REFIL_DP_FORCE=1: the collective also in a one-rank group -- tools/probes/dp_overhead.py measures what the backend's
        # extra stream costs a rank's step on ONE GPU)
        if os.environ.get("REFIL_ALLREDUCE") == "oneshot" and flat.is_cuda and world() > 1:
            key = (flat.data_ptr(), flat.numel())
            if key not in _oneshot:
                _oneshot[key] = OneShotAllReduce(flat.numel(), flat.device)
            _oneshot[key](flat)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


class OneShotAllReduce:
    """refil_oneshot_* (include/refil_hip.h): every rank stages its buffer in IPC-exported device memory and sums all
    ranks' staged buffers itself, in rank order -- one hop over xGMI's point-to-point links instead of a ring's
    2 (N-1), bit-identical results on all ranks. The IPC handles are exchanged once through torch.distributed's
    object all-gather (any backend). Validated with two processes on one GPU (tests/test_gpu_dp.py); opt-in, and gated by
    a start-up equality check against the backend's all-reduce. A peer that does not arrive within
    REFIL_ONESHOT_TIMEOUT_S (default 120 s) is FATAL: that call's buffer is overwritten with NaN and every later call
    raises (the replicas can no longer be trusted to be identical)."""

    def __init__(self, n_floats: int, device):
        import ctypes as C
        from . import _lib
        self._lib, self._C = _lib, C
        self.n = int(n_floats)
        self.ctx = C.c_void_p()
        mine = (C.c_uint8 * (3 * _lib.IPC_HANDLE_BYTES))()
        with torch.cuda.device(device):
            _lib.check(_lib.lib().refil_oneshot_create(C.c_int32(world()), C.c_int32(rank()), C.c_int64(self.n), mine,
                                                       C.byref(self.ctx)), "refil_oneshot_create")
            gathered = [None] * world()
            dist.all_gather_object(gathered, bytes(mine))
            blob = b"".join(gathered)
            buf = (C.c_uint8 * len(blob)).from_buffer_copy(blob)
            _lib.check(_lib.lib().refil_oneshot_connect(self.ctx, buf), "refil_oneshot_connect")
        dist.barrier()                           # every rank has mapped every peer before the first reduction
        self._self_check(device)

    def _self_check(self, device):
        """Gate: one reduction of a known vector through the peer-memory path must equal the backend's all-reduce before
        the path is trusted with gradients (cross-device visibility of the staged data is a property of the platform)."""
        g = torch.Generator().manual_seed(4321 + rank())
        x = torch.randn(self.n, generator=g).to(device)
        ref = x.clone()
        dist.all_reduce(ref, op=dist.ReduceOp.SUM)
        self(x)
        torch.cuda.synchronize(device)
        err = (x - ref).abs().max().item()
        if not (err <= 1e-5 * max(ref.abs().max().item(), 1.0)) or self.timed_out():
            raise RuntimeError(f"one-shot peer all-reduce self-check failed on rank {rank()} (max abs diff {err:.3e} against "
                               "torch.distributed.all_reduce): unset REFIL_ALLREDUCE to use the backend's collective")

    def __call__(self, flat: torch.Tensor) -> torch.Tensor:
        assert flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous() and flat.numel() == self.n
        self._lib.check(self._lib.lib().refil_oneshot_allreduce(self.ctx, self._lib.ptr(flat), self._lib.current_stream_ptr()),
                        "refil_oneshot_allreduce")
        return flat

    def timed_out(self) -> bool:
        t = self._C.c_int32(0)
        self._lib.check(self._lib.lib().refil_oneshot_status(self.ctx, self._C.byref(t)), "refil_oneshot_status")
        return bool(t.value)

    def close(self):
        if self.ctx:
            self._lib.lib().refil_oneshot_destroy(self.ctx)
            self.ctx = self._C.c_void_p()


class BucketedAllReduce:
    """Two-bucket version of allreduce_sum_ (REFIL_DP_BUCKETS=1): [mixer grads | stats] is reduced as soon as the
    hypernets' backward has been enqueued -- on the stream those gradients complete on, i.e. underneath the agent's
    BPTT -- and [agent grads] after the step (SURVEY.md section 8e topology note). Same sums, same result.

        ar = BucketedAllReduce(flat, n_agent)        # flat = [agent | mixer | stats]
        with ar:                                     # installs the library hook for this step
            engine.forward_backward(...)
        ar.finish()                                  # reduces the agent bucket, waits for the mixer bucket
    """

    def __init__(self, flat: torch.Tensor, n_agent: int):
        from . import _lib
        self.flat, self.n_agent = flat, n_agent
        self.work = None
        self._cb = _lib.GRADS_HOOK(self._hook)       # keep the ctypes thunk alive
        self._lib = _lib

    def _hook(self, user, stream):
        if world() == 1:
            return
        ext = torch.cuda.ExternalStream(stream, device=self.flat.device) if stream else torch.cuda.current_stream()
        with torch.cuda.stream(ext):                 # the collective is ordered behind the mixer gradients' stream
            self.work = dist.all_reduce(self.flat[self.n_agent:], op=dist.ReduceOp.SUM, async_op=True)

    def __enter__(self):
        self.work = None
        self._lib.check(self._lib.lib().refil_set_mixer_grads_hook(self._cb, None), "refil_set_mixer_grads_hook")
        return self

    def __exit__(self, *exc):
        self._lib.lib().refil_set_mixer_grads_hook(self._lib.GRADS_HOOK(0), None)
        return False

    def finish(self):
        if world() == 1:
            return self.flat
        dist.all_reduce(self.flat[:self.n_agent], op=dist.ReduceOp.SUM)
        if self.work is not None:
            self.work.wait()                         # (the current stream waits for the mixer bucket)
            self.work = None
        else:                                        # the hook did not fire (no mixer): reduce the rest now
            dist.all_reduce(self.flat[self.n_agent:], op=dist.ReduceOp.SUM)
        return self.flat


def mean_scalar(x) -> float:
    """Mean over ranks of a per-shard scalar diagnostic (log steps only; equal shard sizes)."""
    if world() == 1:
        return float(x)
    t = x.detach().reshape(1).clone() if isinstance(x, torch.Tensor) else torch.tensor([float(x)])
    if dist.get_backend() == "nccl" and not t.is_cuda:
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.item() / world()


def shard_episodes(fields: Dict[str, torch.Tensor], r: int, n: int) -> Dict[str, torch.Tensor]:
    """Rank r of n takes episodes [r*B/n, (r+1)*B/n) of the sampled minibatch (views, no copies)."""
    B = next(iter(fields.values())).shape[0]
    assert B % n == 0, f"batch of {B} episodes does not shard over {n} ranks"
    per = B // n
    return {k: v[r * per:(r + 1) * per] for k, v in fields.items()}


def shard_bits(bits: torch.Tensor, r: int, n: int) -> torch.Tensor:
    """The partition bits are drawn ONCE for the global batch (shared seed) and sliced per rank, so the
    union over ranks equals the single-GPU draw exactly."""
    per = bits.shape[0] // n
    return bits[r * per:(r + 1) * per].contiguous()
