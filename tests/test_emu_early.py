"""CPU tier (REFIL_EMU_FULL=1: minutes each): the QLearner-level schedule tests of tests/test_gpu_early.py that do not need real streams, on the CPU
wavefront emulator (tests/emu; see test_emu_plugin.py for how the plugin layer runs there): the early prologue / early target forward through
train() against the same steps without batch.ready_event; a target network rewritten behind the learner's back (load_state_dict) is seen
through the parameters' version counters; the reference's batch[:, :max_t_filled()] view is trained through its untrimmed parent
(refil_batch.t_limit) with the results of training on the view as given."""
import os
import shutil

import pytest

import emu_util

pytestmark = [pytest.mark.skipif(not (shutil.which("clang++") or os.path.exists("/opt/rocm/lib/llvm/bin/clang++")),
                                 reason="the emulator build needs a host clang++ (vector extensions, __bf16)"),
              pytest.mark.skipif(os.environ.get("REFIL_EMU_FULL") != "1", reason="minutes each on the emulator: REFIL_EMU_FULL=1")]

_G = emu_util.load_copy("test_gpu_early", DEV="cpu")


@pytest.fixture(autouse=True)
def _emulated_library():
    with emu_util.active():
        yield


test_out_of_band_target_writes_are_seen = _G.test_out_of_band_target_writes_are_seen
test_max_t_filled_trim_trains_through_the_parent = _G.test_max_t_filled_trim_trains_through_the_parent


@pytest.mark.parametrize("cfg,et", [("cfg2", "1")])
def test_early_prologue_is_bit_identical(cfg, et, monkeypatch):
    _G.test_early_prologue_is_bit_identical(cfg, et, monkeypatch)
