"""CPU tier: the reference's PLUGIN surface -- QLearner.train, EntityMAC, the mixer modules, save_models / load_models, the registries -- of
tests/test_gpu_plugin.py (same bodies, a private copy of the module with DEV = "cpu") executed on the CPU wavefront emulator (tests/emu; see
test_emu_ops.py): QLearner.train against the reference's golden vectors, the five-call trajectory with carried RMSprop state and target
syncs, checkpoints in the reference's format (round trip, and one the reference itself wrote), args.mixer = None, group matching with
ground-truth factors. The package's code runs as it is; tests/emu_util.active() stands in for what only a GPU has (events, pinned memory,
the learner's GPU-only guard) and hands the engine host batches."""
import os
import shutil

import pytest

import emu_util

pytestmark = pytest.mark.skipif(not (shutil.which("clang++") or os.path.exists("/opt/rocm/lib/llvm/bin/clang++")),
                                reason="the emulator build needs a host clang++ (vector extensions, __bf16)")

_G = emu_util.load_copy("test_gpu_plugin", DEV="cpu")


@pytest.fixture(autouse=True)
def _emulated_library():
    with emu_util.active():
        yield


for _k, _v in list(vars(_G).items()):
    # (test_long_trajectory_tracks_oracle: 40 steps against the oracle, ~2 min here: REFIL_EMU_FULL=1)
    if _k.startswith("test_") and (_k != "test_long_trajectory_tracks_oracle" or os.environ.get("REFIL_EMU_FULL") == "1"):
        globals()[_k] = _v
del _k, _v
