"""Launch-size / launch-order knobs of the learner step (include/refil_hip.h: refil_set_tuning) and the policy around them.

The knobs have no arithmetic meaning: they move workgroup counts, split counts (and with them the summation order of the
split weight-gradient reductions) and the recurrences' prefetch distance, which selects OTHER template instantiations
of the recurrence kernels. Every value a timed run may use is therefore listed here and compared with the torch / oracle
references by tests/ (test_gpu_ops.py::test_gru_forward_backward[pd], test_gpu_learner.py::PRODUCTION["cfgT_tuned*"]):

  * PARITY_TESTED[knob] is the closed set of values the first-call autotuner may pick and bench.py may time;
  * bench.py refuses to time a setting outside it (check()), so a bench line never rests on an untested kernel variant.

Policy (the reference's learner is deterministic for a seed; so is this one by default):
  * REFIL_AUTOTUNE unset / "0": the built-in defaults, no measuring -- two runs with one seed produce identical parameters;
  * REFIL_AUTOTUNE=1: the first train() call of a process on a shape BUCKET measures the candidates in situ (q_learner.py).
    The bucket ignores the batch length and the batch size: key = network dims + the power-of-two bucket of B*T1*ne entity
    rows, so a caller that keeps the reference's max_t_filled() trim (run.py:269-270) tunes once per bucket, not once per
    distinct episode length; at most MAX_TUNES buckets per process are ever measured (later ones take the defaults);
  * REFIL_AUTOTUNE="dw4_target=96,gru_pd=2": the given setting (checked against PARITY_TESTED), no measuring;
  * REFIL_AUTOTUNE_CACHE=<file>: measured settings kept across processes, keyed by device name + library version + bucket.
"""
from __future__ import annotations

import json
import os

# knob -> values with parity coverage (-1 / absent = the built-in default, always covered)
PARITY_TESTED = {
    "dw4_target": (96, 128),
    # workgroups a bf16 x 6 weight-gradient launch (gemm_dws_kernel) aims for: 64 = the built-in default (inside the step the narrow launch wins:
    # profiles/r05_dws_target.txt), 128 / 256 are faster alone on the GPU. Not an autotuner candidate.
    "dws_target": (64, 128, 256),
    "gru_pd": (2, 4),
    "dw4_min_out": (2000,),
    "dw_target": (384, 512),
    "compose_early": (0, 1),
    # 6 (the built-in default): the weight-resident GEMMs compute their fp32 products as six bf16 matrix-pipe products of a 3-way operand
    # split (gemm_wres.hip: wr_split; fp32-accurate -- tests/test_gpu_ops.py::test_wres_split_accuracy); 0: the v_mfma_f32_32x32x2_f32 form.
    # Not an autotuner candidate (the choice changes the arithmetic, not just a launch size).
    "wres_split": (0, 6),
    # the same choice for the weight gradients with 128-column outputs (gemm_dw4.hip: gemm_dws_kernel; tests/test_gpu_ops.py::test_gemm_dws_accuracy)
    "dw_split": (0, 6),
    # in_trans + attention core as ONE launch (attention_qkv.hip): bit 0 target hypernets, 1 target agent, 2 live hypernets, 3 live agent;
    # 15 = the built-in default where the shape is instantiated, 0 = the separate launches of rounds 1-4 (same bf16 x 6 arithmetic of the
    # projections, another summation order: not an autotuner candidate)
    "attn_qkv": (0, 3, 15),
    # the same launch for more than 32 entities / 16 agents (three key tiles / two agent tiles, one wave per SIMD): 1 = take it. Parity-tested
    # (tests/test_gpu_learner.py::test_wide_fused_attention_step_matches_oracle, PRODUCTION["cfg5_ne48*_qkvwide"]) but built in a round without
    # a GPU: off by default until it has been timed against the three launches it replaces. Not an autotuner candidate.
    "attn_qkv_wide": (0, 1),
}
# what the first-call autotuner tries, in this order (greedy, one knob at a time)
CANDIDATES = (("dw4_target", (96,)), ("gru_pd", (2,)), ("dw4_min_out", (2000,)), ("dw_target", (384,)), ("compose_early", (0, 1)))
MAX_TUNES = 8           # shape buckets a process may measure; the rest run the defaults
MIN_ROWS = 20000        # below this many entity rows the step is launch-bound: nothing to tune


def check(setting: dict, what: str = "tuning") -> dict:
    """Raises unless every knob / value of `setting` is in the parity-tested set."""
    for k, v in setting.items():
        if k not in PARITY_TESTED:
            raise ValueError(f"{what}: unknown knob '{k}' (known: {sorted(PARITY_TESTED)})")
        if int(v) not in PARITY_TESTED[k]:
            raise ValueError(f"{what}: {k}={v} has no parity coverage (tested values: {PARITY_TESTED[k]}; "
                             "add it to refil_amd/tuning.py AND to the parametrised tests first)")
    return setting


def parse_env(env: str) -> dict:
    return check({k: int(v) for k, v in (kv.split("=") for kv in env.split(",") if kv)}, "REFIL_AUTOTUNE")


def bucket_key(dims) -> bytes:
    """Network / environment dims with B and T1 blanked + the power-of-two bucket of the entity-row count."""
    import ctypes as C
    from ._lib import Dims
    d = Dims()
    C.memmove(C.byref(d), C.byref(dims), C.sizeof(Dims))
    rows = int(dims.B) * int(dims.T1) * int(dims.ne)
    d.B, d.T1 = 0, 0
    return bytes(d) + max(rows - 1, 1).bit_length().to_bytes(2, "little")


def mode():
    """'off' | 'measure' | dict (a given setting)"""
    env = os.environ.get("REFIL_AUTOTUNE", "")
    if "=" in env:
        return parse_env(env)
    return "measure" if env == "1" else "off"


def _cache_id(key: bytes) -> str:
    import torch
    from . import _lib
    dev = torch.cuda.get_device_name(torch.cuda.current_device()) if torch.cuda.is_available() else "cpu"
    return f"{dev}|lib{_lib.lib().refil_version()}|{key.hex()}"


def cache_get(key: bytes):
    path = os.environ.get("REFIL_AUTOTUNE_CACHE")
    if not path or not os.path.exists(path):
        return None
    try:
        got = json.load(open(path)).get(_cache_id(key))
        return None if got is None else check({k: int(v) for k, v in got.items()}, "REFIL_AUTOTUNE_CACHE")
    except (OSError, ValueError):
        return None


def cache_put(key: bytes, setting: dict):
    path = os.environ.get("REFIL_AUTOTUNE_CACHE")
    if not path:
        return
    try:
        d = json.load(open(path)) if os.path.exists(path) else {}
    except (OSError, ValueError):
        d = {}
    d[_cache_id(key)] = setting
    with open(path, "w") as f:
        json.dump(d, f)
