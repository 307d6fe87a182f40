"""Host-side driver of the HIP learner step: owns the workspace arena and marshals tensors to the
C ABI (include/refil_hip.h). No arithmetic happens here -- every number is produced by the kernels in
refil_amd/csrc. Raises if the library is missing (no fallback path).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import _lib
from ._lib import REFIL_NSTAT, Batch, DebugOut, Dims, check, lib


def dims_from_args(args, B: int, T1: int) -> Dims:
    """refil_dims from the PyMARL `args` namespace (same flag names as src/config/*.yaml)."""
    return _lib.make_dims(
        B=B, T1=T1, ne=args.n_entities, na=args.n_agents, ed=args.entity_shape, A=args.n_actions,
        d=args.attn_embed_dim, heads=args.attn_n_heads, H=args.rnn_hidden_dim, hyp=args.hypernet_embed,
        M=args.mixing_embed_dim, entity_last_action=int(bool(args.entity_last_action)),
        imagine=int("imagine" in args.agent), softmax_mixing_weights=int(bool(args.softmax_mixing_weights)),
        mixer_tanh=int(getattr(args, "mixer_non_lin", "elu") == "tanh"), double_q=int(bool(args.double_q)),
        agent_ff=int(args.agent.endswith("_ff")), mixer_lin=int(getattr(args, "mixer", None) == "lin_flex_qmix"),
        mixer_vdn=int(getattr(args, "mixer", None) == "vdn"), gt_factors=0, gt_obs_mask=int(bool(getattr(args, "gt_obs_mask", False))),
        pooling={None: 0, "mean": 1, "max": 2}[getattr(args, "pooling_type", None)],
        mixer_none=int(getattr(args, "mixer", None) is None),
        gamma=float(args.gamma), lmbda=float(getattr(args, "lmbda", 0.0)))


def clone_dims(d: Dims, **kw) -> Dims:
    out = Dims()
    C.memmove(C.byref(out), C.byref(d), C.sizeof(Dims))
    for k, v in kw.items():
        setattr(out, k, v)
    return out


class Workspace:
    """Grow-only device arena handed to the library."""

    def __init__(self, device):
        self.device = device
        self.buf: Optional[torch.Tensor] = None
        self.fresh = False      # the zero fill of a new arena is enqueued on the current stream: the first call on it stays in order

    def get(self, nbytes: int) -> torch.Tensor:
        if self.buf is None or self.buf.numel() < nbytes:
            self.buf = None
            self.fresh = True
            # zeroed once: rows the step skips (padded entities, steps after an episode's end) are never written, and
            # whatever they hold is multiplied by exact zeros downstream -- it has to be finite
            self.buf = torch.zeros(nbytes + 256, dtype=torch.uint8, device=self.device)
        return self.buf

    def ptr_size(self, nbytes: int):
        b = self.get(nbytes)
        p = (b.data_ptr() + 255) & ~255
        return C.c_void_p(p), C.c_size_t(b.numel() - (p - b.data_ptr()))


class LearnerEngine:
    def __init__(self, device="cuda"):
        self.device = torch.device(device)
        self.ws = Workspace(self.device)
        self.scratch = torch.zeros(4096, dtype=torch.uint8, device=self.device)
        lib()   # fail early and loudly if the HIP library is not built

    # -- q_learner.py:66-176 -------------------------------------------------------------------
    def forward_backward(self, dims: Dims, fields: Dict[str, torch.Tensor], group_bits: Optional[torch.Tensor],
                         params_live: torch.Tensor, params_target: torch.Tensor, grads: torch.Tensor,
                         debug: bool = False, ready_event: Optional[torch.cuda.Event] = None, target_version: int = 0, t_limit: int = 0):
        """grads: flat fp32 [layout.total + REFIL_NSTAT]. Returns dict of debug tensors if debug.
        ready_event: recorded behind the last write of the batch fields (refil_batch.ready_event): the step's input
        assembly and row lists then run on a side stream, beside the end of the previous step on the current stream."""
        nbytes = lib().refil_learner_workspace_bytes(C.byref(dims))
        if nbytes == 0:
            raise RuntimeError("refil_learner_workspace_bytes: " + lib().refil_last_error().decode())
        wp, wsz = self.ws.ptr_size(nbytes)
        b = _lib.make_batch(fields, group_bits)
        # (not on a new arena: its zero fill sits on the current stream, the early prologue would write the arena on another one --
        # and the allocator may hand a new arena the address of an old one the library remembers the layout of)
        if ready_event is not None and not b._converted and not self.ws.fresh:
            b.ready_event = ready_event.cuda_event
        # (refil_batch.target_version: early target forward, kept composed maps -- never on a new arena: it is zero-filled, and the
        # allocator may have given it the address of an old one the library still remembers)
        b.target_version = 0 if self.ws.fresh else int(target_version) & 0xFFFFFFFFFFFFFFFF
        b.t_limit = int(t_limit)
        self.ws.fresh = False
        dbg = None
        out = {}
        if debug:
            G = 3 if dims.imagine else 1
            B, T1, T, na, A = dims.B, dims.T1, dims.T1 - 1, dims.na, dims.A
            mk = lambda *s: torch.zeros(*s, dtype=torch.float32, device=self.device)
            out = {"q": mk(G, B, T1, na, A), "chosen_q": mk(G, B, T, na), "target_max_q": mk(B, T, na),
                   "q_tot": mk(B, T), "q_tot_imagine": mk(B, T), "target_q_tot": mk(B, T), "targets": mk(B, T)}
            dbg = DebugOut()
            for k, v in out.items():
                setattr(dbg, k, v.data_ptr())
        check(lib().refil_learner_forward_backward(
            C.byref(dims), C.byref(b), _lib.ptr(params_live), _lib.ptr(params_target), _lib.ptr(grads), wp, wsz,
            C.byref(dbg) if dbg is not None else None, _lib.current_stream_ptr()), "refil_learner_forward_backward")
        return out

    def step(self, dims: Dims, fields, group_bits, params_live, params_target, grads, square_avg, lr, alpha, eps, weight_decay, clip,
             ready_event=None, target_version: int = 0, t_limit: int = 0, comm=None):
        """refil_learner_step: forward + backward + clip + RMSprop in ONE C call. comm: an ncclComm_t (ctypes.c_void_p) for the
        data-parallel all-reduce of [grads | stats] between backward and optimiser; None = single process."""
        nbytes = lib().refil_learner_workspace_bytes(C.byref(dims))
        if nbytes == 0:
            raise RuntimeError("refil_learner_workspace_bytes: " + lib().refil_last_error().decode())
        wp, wsz = self.ws.ptr_size(nbytes)
        b = _lib.make_batch(fields, group_bits)
        if ready_event is not None and not b._converted and not self.ws.fresh:
            b.ready_event = ready_event.cuda_event
        b.target_version = 0 if self.ws.fresh else int(target_version) & 0xFFFFFFFFFFFFFFFF
        b.t_limit = int(t_limit)
        self.ws.fresh = False
        hy = _lib.OptHyper(lr, alpha, eps, weight_decay, clip)
        check(lib().refil_learner_step(C.byref(dims), C.byref(b), _lib.ptr(params_live), _lib.ptr(params_target), _lib.ptr(grads),
                                       _lib.ptr(square_avg), C.byref(hy), comm, wp, wsz, _lib.ptr(self.scratch), _lib.current_stream_ptr()),
              "refil_learner_step")

    def row_counts(self, dims: Dims):
        """Row-list diagnostics of the last forward_backward with these dims (host sync; benchmarks / tests only)."""
        nbytes = lib().refil_learner_workspace_bytes(C.byref(dims))
        wp, wsz = self.ws.ptr_size(nbytes)
        out = (C.c_int32 * 6)()
        check(lib().refil_learner_row_counts(C.byref(dims), wp, wsz, out, _lib.current_stream_ptr()), "refil_learner_row_counts")
        keys = ("lists", "entity_rows_agent", "entity_rows_hyper", "agent_rows", "live_steps", "steps")
        d = dict(zip(keys, list(out)))
        d["entity_rows"], d["all_agent_rows"] = dims.B * dims.T1 * dims.ne, dims.B * dims.T1 * dims.na
        return d

    # -- q_learner.py:177-178 ------------------------------------------------------------------
    def clip_rmsprop(self, params: torch.Tensor, grads: torch.Tensor, square_avg: torch.Tensor, n: int, lr: float,
                     alpha: float, eps: float, weight_decay: float, clip: float):
        stats = grads[n:n + REFIL_NSTAT]
        check(lib().refil_clip_rmsprop_step(
            _lib.ptr(params), _lib.ptr(grads), _lib.ptr(square_avg), C.c_int64(n), C.c_float(lr), C.c_float(alpha),
            C.c_float(eps), C.c_float(weight_decay), C.c_float(clip), _lib.ptr(stats), _lib.ptr(self.scratch),
            _lib.current_stream_ptr()), "refil_clip_rmsprop_step")

    # -- basic_controller.py:28-67 -------------------------------------------------------------
    def agent_forward(self, dims: Dims, fields: Dict[str, torch.Tensor], group_bits, params: torch.Tensor,
                      h0: Optional[torch.Tensor], first_step_zero: bool):
        G = 3 if dims.imagine else 1
        nbytes = lib().refil_agent_workspace_bytes(C.byref(dims))
        wp, wsz = self.ws.ptr_size(nbytes)
        b = _lib.make_batch(fields, group_bits)
        q = torch.empty(G, dims.B, dims.T1, dims.na, dims.A, dtype=torch.float32, device=self.device)
        h = torch.empty(G, dims.B, dims.na, dims.H, dtype=torch.float32, device=self.device)
        check(lib().refil_agent_forward(C.byref(dims), C.byref(b), C.c_int32(int(first_step_zero)), _lib.ptr(params),
                                        _lib.ptr(h0), _lib.ptr(h), _lib.ptr(q), wp, wsz, _lib.current_stream_ptr()),
              "refil_agent_forward")
        return q, h

    # -- flex_qmix.py:79-121 -------------------------------------------------------------------
    def mixer_forward(self, dims: Dims, fields: Dict[str, torch.Tensor], group_bits, params: torch.Tensor,
                      agent_qs: torch.Tensor, agent_qs_imagine: Optional[torch.Tensor], t0: int, T: int,
                      params_ptr: Optional[int] = None, want_ingroup: bool = False, mask_words=None, mask_row_bits=None):
        nbytes = lib().refil_mixer_workspace_bytes(C.byref(dims))
        wp, wsz = self.ws.ptr_size(nbytes)
        b = _lib.make_batch(fields, group_bits, mask_words=mask_words, mask_row_bits=mask_row_bits)
        q_tot = torch.empty(dims.B, T, dtype=torch.float32, device=self.device)
        q_im = torch.empty(dims.B, T, dtype=torch.float32, device=self.device) if agent_qs_imagine is not None else None
        ing = torch.zeros(1, dtype=torch.float32, device=self.device) if want_ingroup else None
        pp = C.c_void_p(params_ptr) if params_ptr is not None else _lib.ptr(params)
        check(lib().refil_mixer_forward(C.byref(dims), C.byref(b), C.c_int32(t0), C.c_int32(T), pp,
                                        _lib.ptr(agent_qs), _lib.ptr(agent_qs_imagine), _lib.ptr(q_tot), _lib.ptr(q_im),
                                        _lib.ptr(ing), wp, wsz, _lib.current_stream_ptr()), "refil_mixer_forward")
        if want_ingroup:
            return q_tot, q_im, ing
        return q_tot, q_im
