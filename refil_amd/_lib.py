"""ctypes binding of librefil_hip.so (C ABI declared in include/refil_hip.h).

The product path has NO fallback: if the library is missing or a call fails, a RuntimeError is
raised. PyTorch is used by the callers only to own device memory and streams; everything crossing
this boundary is a raw pointer, a size or a plain C struct.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# REFIL_LIB_PATH: load another build of the library (A/B runs of two builds on one GPU box, tools/ab.sh)
LIB_PATH = os.environ.get("REFIL_LIB_PATH") or os.path.join(_HERE, "librefil_hip.so")

REFIL_NSTAT = 8
STAT_MASK_SUM, STAT_TD_SQ, STAT_IM_TD_SQ, STAT_TD_ABS, STAT_QTOT_SUM, STAT_TARGET_SUM, STAT_GRAD_NORM, STAT_INGROUP_SUM = range(8)

GEMM_RELU, GEMM_RELU_BWD, GEMM_ACCUM, GEMM_A_OUTC, GEMM_B_OUTC, GEMM_COLSUM_A = 1, 2, 4, 8, 16, 32
MASK_OBS, MASK_OBS_WITHIN, MASK_OBS_INTERACT, MASK_ENTITY, MASK_WITHIN, MASK_INTERACT = range(6)
MASK_OBS_GTW, MASK_OBS_GTI, MASK_GTW, MASK_GTI = 6, 7, 8, 9
MASK_OBS_RGTW, MASK_OBS_RGTI, MASK_RGTW, MASK_RGTI = 10, 11, 12, 13


class Dims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "B", "T1", "ne", "na", "ed", "A", "d", "heads", "H", "hyp", "M", "entity_last_action", "imagine",
        "softmax_mixing_weights", "mixer_tanh", "double_q", "agent_ff", "mixer_lin", "mixer_vdn", "gt_factors", "gt_obs_mask", "pooling", "mixer_none")] + [("gamma", C.c_float), ("lmbda", C.c_float)]


class ParamLayout(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "total", "agent_total",
        "ag_fc1_w", "ag_fc1_b", "ag_in_w", "ag_out_w", "ag_out_b", "ag_fc2_w", "ag_fc2_b",
        "ag_w_ih", "ag_w_hh", "ag_b_ih", "ag_b_hh", "ag_fc3_w", "ag_fc3_b",
        "mix_fc1_w", "mix_fc1_b", "mix_in_w", "mix_out_w", "mix_out_b", "mix_fc2_w", "mix_fc2_b",
        "mix_fc1_w_stride", "mix_fc1_b_stride", "mix_in_w_stride", "mix_out_w_stride", "mix_out_b_stride",
        "mix_fc2_w_stride", "mix_fc2_b_stride")]


class Batch(C.Structure):
    _fields_ = [
        ("entities", C.c_void_p), ("ent_sB", C.c_int64), ("ent_sT", C.c_int64),
        ("obs_mask", C.c_void_p), ("om_sB", C.c_int64), ("om_sT", C.c_int64),
        ("entity_mask", C.c_void_p), ("em_sB", C.c_int64), ("em_sT", C.c_int64),
        ("actions", C.c_void_p), ("ac_sB", C.c_int64), ("ac_sT", C.c_int64),
        ("avail_actions", C.c_void_p), ("av_sB", C.c_int64), ("av_sT", C.c_int64),
        ("reward", C.c_void_p), ("rw_sB", C.c_int64), ("rw_sT", C.c_int64),
        ("terminated", C.c_void_p), ("tm_sB", C.c_int64), ("tm_sT", C.c_int64),
        ("filled", C.c_void_p), ("fl_sB", C.c_int64), ("fl_sT", C.c_int64),
        ("gt_mask", C.c_void_p), ("gt_sB", C.c_int64), ("gt_sT", C.c_int64),
        ("group_bits", C.c_void_p),
        ("mask_words", C.c_void_p), ("mask_row_bits", C.c_void_p),
        ("ready_event", C.c_void_p),
        ("target_version", C.c_uint64),
        ("t_limit", C.c_int32),
    ]


class OptHyper(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("lr", "alpha", "eps", "weight_decay", "grad_norm_clip")]


class DebugOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "q", "chosen_q", "target_max_q", "q_tot", "q_tot_imagine", "target_q_tot", "targets")]


class RowMap(C.Structure):
    _fields_ = [("grp", C.c_int32), ("gstride", C.c_int32), ("off", C.c_int32)]


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("bias", C.c_void_p), ("aux", C.c_void_p),
        ("rowmask", C.c_void_p), ("colsum", C.c_void_p), ("partial", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("lda", C.c_int32), ("ldb", C.c_int32), ("ldc", C.c_int32),
        ("sA", C.c_int64), ("sB", C.c_int64), ("sC", C.c_int64), ("sBias", C.c_int64), ("sColsum", C.c_int64),
        ("a_map", RowMap), ("b_map", RowMap), ("c_map", RowMap),
        ("rowmask_mod", C.c_int32), ("batch", C.c_int32), ("splits", C.c_int32), ("flags", C.c_int32),
        ("bias2", C.c_void_p), ("rowscale", C.c_void_p), ("rowscale_mod", C.c_int32),
        ("row_index", C.c_void_p), ("row_count", C.c_void_p), ("row_count_hint", C.c_int32),
    ]


class AttnDesc(C.Structure):
    _fields_ = [
        ("Q", C.c_void_p), ("ldq", C.c_int32),
        ("K", C.c_void_p), ("V", C.c_void_p), ("ldkv", C.c_int32),
        ("O", C.c_void_p), ("sO", C.c_int64), ("ldo", C.c_int32),
        ("dO", C.c_void_p), ("dQ", C.c_void_p), ("dK", C.c_void_p), ("dV", C.c_void_p),
        ("R", C.c_int32), ("T1", C.c_int32), ("ne", C.c_int32), ("na", C.c_int32), ("heads", C.c_int32), ("hd", C.c_int32),
        ("nvar", C.c_int32), ("var", C.c_int32 * 3),
        ("obs_mask", C.c_void_p), ("om_sB", C.c_int64), ("om_sT", C.c_int64),
        ("ent_mask", C.c_void_p), ("ent_mask0", C.c_void_p), ("group_bits", C.c_void_p),
        ("gt_mask", C.c_void_p), ("gt_sB", C.c_int64), ("gt_sT", C.c_int64),
        ("t_last", C.c_void_p), ("kv_dead", C.c_void_p), ("q_dead", C.c_void_p),
        ("mask_words", C.c_void_p), ("row_bits", C.c_void_p), ("mask_words_nvar", C.c_int32),
    ]


class AttnQkvDesc(C.Structure):
    _fields_ = [("attn", AttnDesc), ("X", C.c_void_p), ("ldx", C.c_int32), ("W_in", C.c_void_p),
                ("q_out", C.c_void_p), ("k_out", C.c_void_p), ("v_out", C.c_void_p)]


class GruDesc(C.Structure):
    _fields_ = [
        ("gi", C.c_void_p), ("hsx", C.c_void_p), ("w_hh", C.c_void_p), ("b_hh", C.c_void_p),
        ("save_r", C.c_void_p), ("save_z", C.c_void_p), ("save_n", C.c_void_p), ("save_ghn", C.c_void_p),
        ("dhs", C.c_void_p), ("dgi", C.c_void_p), ("dgh", C.c_void_p),
        ("NR", C.c_int32), ("T1", C.c_int32), ("na", C.c_int32), ("H", C.c_int32),
        ("t_last", C.c_void_p), ("B", C.c_int32), ("zero_h0", C.c_int32), ("ever", C.c_void_p),
    ]


class GatherField(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("src_episode_bytes", C.c_int64), ("dst_episode_bytes", C.c_int64),
                ("copy_bytes", C.c_int64), ("unpack_width", C.c_int32), ("reserved", C.c_int32)]


MAX_GATHER_FIELDS = 24      # refil_amd/csrc/replay.hip


GRADS_HOOK = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)


class ProfileEntry(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("launches", C.c_int64), ("total_ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double), ("flops_bf16x6", C.c_double)]


# every symbol include/refil_hip.h declares (tests/test_abi.py checks the library exports them all)
EXPORTS = [
    "refil_get_param_layout", "refil_learner_workspace_bytes", "refil_learner_forward_backward", "refil_learner_step",
    "refil_clip_rmsprop_step", "refil_agent_workspace_bytes", "refil_agent_forward",
    "refil_mixer_workspace_bytes", "refil_mixer_forward", "refil_gemm", "refil_attn_forward",
    "refil_attn_backward", "refil_pool_forward", "refil_pool_backward", "refil_gru_forward", "refil_gru_backward", "refil_last_error", "refil_version",
    "refil_profile_enable", "refil_profile_collect", "refil_set_overlap", "refil_release_streams", "refil_replay_gather",
    "refil_learner_row_counts", "refil_attn_mask_words", "refil_set_mixer_grads_hook",
    "refil_oneshot_create", "refil_oneshot_connect", "refil_oneshot_allreduce", "refil_oneshot_status", "refil_oneshot_destroy",
    "refil_allreduce_flat", "refil_pack_mask_bits", "refil_side_stream", "refil_set_tuning", "refil_get_stat",
    "refil_attn_qkv_forward",
]
IPC_HANDLE_BYTES = 64

_lib = None


def lib():
    """Load (once) and return the ctypes handle. Raises if the HIP library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -m refil_amd.build` (hipcc, gfx950). "
            "refil_amd has no CPU/PyTorch fallback for the learner hot path.")
    # PyTorch first: its wheel ships its own HIP runtime (torch/lib/libamdhip64.so). Loaded before it, this library would bind to the
    # system's copy instead and the process would hold two runtimes -- ours then sees no device ("no ROCm-capable device is detected")
    # although torch does. With torch's runtime already mapped the loader resolves ours to the same one.
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    L.refil_last_error.restype = C.c_char_p
    L.refil_version.restype = C.c_int
    for name in ("refil_learner_workspace_bytes", "refil_agent_workspace_bytes", "refil_mixer_workspace_bytes"):
        getattr(L, name).restype = C.c_size_t
        getattr(L, name).argtypes = [C.POINTER(Dims)]
    L.refil_get_param_layout.argtypes = [C.POINTER(Dims), C.POINTER(ParamLayout)]
    L.refil_learner_forward_backward.argtypes = [
        C.POINTER(Dims), C.POINTER(Batch), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
        C.POINTER(DebugOut), C.c_void_p]
    L.refil_learner_step.argtypes = [
        C.POINTER(Dims), C.POINTER(Batch), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(OptHyper), C.c_void_p,
        C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.refil_clip_rmsprop_step.argtypes = [
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
        C.c_void_p, C.c_void_p, C.c_void_p]
    L.refil_agent_forward.argtypes = [
        C.POINTER(Dims), C.POINTER(Batch), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
        C.c_void_p, C.c_size_t, C.c_void_p]
    L.refil_mixer_forward.argtypes = [
        C.POINTER(Dims), C.POINTER(Batch), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.refil_gemm.argtypes = [C.POINTER(GemmDesc), C.c_void_p]
    L.refil_attn_forward.argtypes = [C.POINTER(AttnDesc), C.c_void_p]
    L.refil_attn_backward.argtypes = [C.POINTER(AttnDesc), C.c_void_p]
    L.refil_attn_qkv_forward.argtypes = [C.POINTER(AttnQkvDesc), C.c_void_p]
    L.refil_attn_mask_words.argtypes = [C.POINTER(AttnDesc), C.c_void_p, C.c_void_p, C.c_void_p]
    L.refil_pool_forward.argtypes = [C.POINTER(AttnDesc), C.c_int32, C.c_void_p]
    L.refil_pool_backward.argtypes = [C.POINTER(AttnDesc), C.c_int32, C.c_void_p]
    L.refil_gru_forward.argtypes = [C.POINTER(GruDesc), C.c_void_p]
    L.refil_gru_backward.argtypes = [C.POINTER(GruDesc), C.c_void_p]
    L.refil_set_overlap.argtypes = [C.c_int]
    L.refil_side_stream.argtypes = [C.POINTER(C.c_void_p)]
    L.refil_set_tuning.argtypes = [C.c_char_p, C.c_int64]
    L.refil_get_stat.argtypes = [C.c_char_p]
    L.refil_get_stat.restype = C.c_int64
    L.refil_set_mixer_grads_hook.argtypes = [GRADS_HOOK, C.c_void_p]
    L.refil_oneshot_create.argtypes = [C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.POINTER(C.c_void_p)]
    L.refil_oneshot_connect.argtypes = [C.c_void_p, C.c_void_p]
    L.refil_oneshot_allreduce.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.refil_oneshot_status.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    L.refil_oneshot_destroy.argtypes = [C.c_void_p]
    L.refil_learner_row_counts.argtypes = [C.POINTER(Dims), C.c_void_p, C.c_size_t, C.POINTER(C.c_int32), C.c_void_p]
    L.refil_replay_gather.argtypes = [C.POINTER(GatherField), C.c_int32, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p]
    L.refil_pack_mask_bits.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
    L.refil_profile_enable.argtypes = [C.c_int]
    L.refil_profile_collect.argtypes = [C.POINTER(ProfileEntry), C.c_int]
    _lib = L
    return L


def get_stat(name: str) -> int:
    """refil_get_stat: schedule counters of this thread (learner_steps, early_prologue_steps, early_target_*_steps)."""
    return int(lib().refil_get_stat(name.encode()))


def profile_enable(on: bool):
    check(lib().refil_profile_enable(int(on)), "refil_profile_enable")


def profile_collect():
    """[{name, launches, total_ms, flops, bytes, flops_bf16x6}] aggregated per kernel symbol since profile_enable(True)."""
    buf = (ProfileEntry * 64)()
    n = lib().refil_profile_collect(buf, 64)
    if n > 0:
        check(n, "refil_profile_collect")
    return [dict(name=buf[i].name.decode(), launches=buf[i].launches, total_ms=buf[i].total_ms, flops=buf[i].flops,
                 bytes=buf[i].bytes, flops_bf16x6=buf[i].flops_bf16x6) for i in range(-n)]


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed (rc={rc}): {lib().refil_last_error().decode()}")


def ptr(t):
    """Device (or host) address of a torch tensor, None -> NULL."""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream_ptr():
    import torch
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)      # (10x cheaper than building a Stream object, four times per step)
    if raw is not None:
        return C.c_void_p(raw(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def make_dims(**kw) -> Dims:
    d = Dims()
    for k, v in kw.items():
        setattr(d, k, v)
    return d


def param_layout(dims: Dims) -> ParamLayout:
    out = ParamLayout()
    check(lib().refil_get_param_layout(C.byref(dims), C.byref(out)), "refil_get_param_layout")
    return out


_FIELD_DTYPES = None


def _field_dtypes():
    """Element types the kernels read (include/refil_hip.h, refil_batch): the reference's scheme dtypes
    (src/run.py:178-192, episode_buffer.py:51-54)."""
    global _FIELD_DTYPES
    if _FIELD_DTYPES is None:
        import torch
        _FIELD_DTYPES = {"entities": torch.float32, "obs_mask": torch.uint8, "entity_mask": torch.uint8,
                         "actions": torch.int64, "avail_actions": torch.int32, "reward": torch.float32,
                         "terminated": torch.uint8, "filled": torch.int64, "gt_mask": torch.uint8}
    return _FIELD_DTYPES


def make_batch(fields: dict, group_bits=None, device=None, mask_words=None, mask_row_bits=None) -> Batch:
    """fields: name -> DEVICE tensor [B,T1,...] (any batch/time strides, inner dims contiguous).

    The kernels reinterpret raw pointers, so every field is checked here: a host tensor is an error (the library has no
    CPU path), a scheme dtype other than the kernel's (th.bool masks, int64 avail_actions, float terminated from a
    custom env -- all of which the reference's torch ops accept) is converted (bool -> uint8 is a zero-copy view).
    Converted tensors are kept alive on the returned struct until it is dropped."""
    import torch
    b = Batch()
    b._converted = False        # a field was converted HERE, on the current stream: no earlier event covers it
    keep = []
    names = {"entities": "ent", "obs_mask": "om", "entity_mask": "em", "actions": "ac", "avail_actions": "av",
             "reward": "rw", "terminated": "tm", "filled": "fl", "gt_mask": "gt"}
    want = _field_dtypes()
    for name, short in names.items():
        t = fields.get(name)
        if t is None:
            continue
        if not t.is_cuda:
            raise ValueError(f"batch field {name} lives on {t.device}: move the batch to the GPU first "
                             "(EpisodeBatch.to(device)); refil_amd has no host path")
        if device is not None and t.device != torch.device(device):
            raise ValueError(f"batch field {name} is on {t.device}, the learner on {device}")
        if t.dim() < 2:
            raise ValueError(f"batch field {name}: expected [B, T1, ...], got shape {tuple(t.shape)}")
        if t.dtype != want[name]:
            if t.dtype == torch.bool and want[name] == torch.uint8:
                t = t.view(torch.uint8)
            elif want[name].is_floating_point or not t.dtype.is_floating_point or name in ("terminated", "filled"):
                t = t.to(want[name])
            else:
                raise TypeError(f"batch field {name}: dtype {t.dtype} cannot stand in for {want[name]}")
            keep.append(t)
            b._converted = True
        inner = t[0, 0]
        if not inner.is_contiguous():
            raise ValueError(f"batch field {name}: inner dims must be contiguous")
        setattr(b, name, t.data_ptr())
        setattr(b, short + "_sB", t.stride(0))
        setattr(b, short + "_sT", t.stride(1))
    if group_bits is not None:
        if not group_bits.is_cuda:
            raise ValueError("group_bits must be a device tensor [B, n_entities]")
        if group_bits.dtype != torch.uint8 or not group_bits.is_contiguous():
            group_bits = group_bits.to(torch.uint8).contiguous()
        keep.append(group_bits)
        b.group_bits = group_bits.data_ptr()
    if mask_words is not None:
        assert mask_words.is_cuda and mask_words.dtype == torch.int64 and mask_words.is_contiguous()
        assert mask_row_bits is not None and mask_row_bits.dtype == torch.int64 and mask_row_bits.is_contiguous()
        keep += [mask_words, mask_row_bits]
        b.mask_words, b.mask_row_bits = mask_words.data_ptr(), mask_row_bits.data_ptr()
    b._keep = keep
    return b
