"""CPU tier: the learner-step parity tests of tests/test_gpu_learner.py executed on the CPU wavefront emulator (tests/emu; see
test_emu_ops.py for what that is and is not): one refil_learner_forward_backward + refil_clip_rmsprop_step through the C ABI of the
emulator build -- every kernel launch of the step -- against the golden vectors generated from the reference itself and against the oracle.

Default selection: every golden fixture, the 5-call reference trajectory, the oracle / degenerate-episode / config-matrix cases and a few
fuzz shapes (~3 min). REFIL_EMU_FULL=1 adds everything of the gpu tier below production size -- the bit-identity tests of the schedule variants,
the early prologue / early target forward at engine level, all fuzz draws (~20 min); the BASELINE-size cases run offline (tools/emu_suite.sh:
two minutes per north-star step)."""
import os
import shutil

import pytest

import emu_util

pytestmark = pytest.mark.skipif(not (shutil.which("clang++") or os.path.exists("/opt/rocm/lib/llvm/bin/clang++")),
                                reason="the emulator build needs a host clang++ (vector extensions, __bf16)")

_G = emu_util.load_copy("test_gpu_learner", DEV="cpu")
FULL = os.environ.get("REFIL_EMU_FULL") == "1"


@pytest.fixture(autouse=True)
def _emulated_library():
    with emu_util.active():
        yield


_ALWAYS = ("test_learner_step_matches_reference_golden", "test_trajectory_matches_reference_golden", "test_learner_step_matches_oracle",
           "test_gt_factor_diagnostics_match_reference", "test_config_matrix_matches_oracle", "test_time_truncated_strided_batch_equals_contiguous",
           "test_wide_fused_attention_step_matches_oracle", "test_fused_attention_lds_fallback_is_bit_identical",
           "test_single_call_step_equals_the_three_calls")
_NEVER = ("test_production_size_step_matches_oracle", "test_full_size_properties")       # BASELINE sizes: the GPU tier


def _first(fn, n):
    """the first n parametrisations of a gpu-tier test (the fuzzers' seeds are fixed: the same shapes every run)"""
    marks = [m for m in getattr(fn, "pytestmark", []) if m.name == "parametrize"]
    if FULL or not marks:
        return fn
    import functools
    import types
    g = types.FunctionType(fn.__code__, fn.__globals__, fn.__name__, fn.__defaults__, fn.__closure__)
    g = functools.update_wrapper(g, fn)
    del g.__wrapped__
    m = marks[-1]
    names, values = m.args[0], list(m.args[1])[:n]
    kw = dict(m.kwargs)
    if "ids" in kw and not callable(kw["ids"]):
        kw["ids"] = list(kw["ids"])[:n]
    g.pytestmark = [x for x in fn.pytestmark if x is not m] + [pytest.mark.parametrize(names, values, **kw).mark]
    return g


for _k, _v in list(vars(_G).items()):
    if not _k.startswith("test_") or _k in _NEVER:
        continue
    if _k in _ALWAYS or FULL:
        globals()[_k] = _v
    elif _k in ("test_degenerate_episodes_match_oracle", "test_random_shapes_match_oracle", "test_random_row_list_shapes_match_oracle",
                "test_random_variants_match_oracle", "test_random_shapes_acting_path_matches_oracle", "test_deferred_split_reductions_are_bit_identical",
                "test_row_counts_match_the_batch", "test_mixer_grads_hook_fires_when_the_mixer_bucket_is_final"):
        globals()[_k] = _first(_v, 3 if "random" in _k else 1)
del _k, _v
