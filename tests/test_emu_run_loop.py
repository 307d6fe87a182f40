"""CPU tier: the whole cycle of tests/test_gpu_run_loop.py -- mac.select_actions on the acting path, EpisodeBatch.update, ring-buffer insert,
sample, the max_t_filled() trim, QLearner.train on data the policy itself generates -- for 40 iterations on the CPU wavefront emulator
(tests/emu; see test_emu_ops.py and test_emu_plugin.py): the mechanics, not the learning curve (800 iterations: the gpu tier)."""
import os
import shutil

import pytest

import emu_util

pytestmark = pytest.mark.skipif(not (shutil.which("clang++") or os.path.exists("/opt/rocm/lib/llvm/bin/clang++")),
                                reason="the emulator build needs a host clang++ (vector extensions, __bf16)")

_G = emu_util.load_copy("test_gpu_run_loop", DEV="cpu")


@pytest.fixture(autouse=True)
def _emulated_library():
    with emu_util.active():
        yield


@pytest.mark.skipif(os.environ.get("REFIL_EMU_FULL") != "1", reason="minutes on the emulator: REFIL_EMU_FULL=1")
def test_cycle_mechanics_short():
    _G.test_cycle_mechanics_short()
