"""CPU tier: the op-level parity tests of tests/test_gpu_ops.py (same bodies, same shapes, same tolerances against fp32 torch) executed on
the CPU wavefront emulator of tests/emu -- the kernel SOURCES of refil_amd/csrc compiled for the host, 64-lane waves emulated lane by lane
(MFMA lane layouts, DPP, permutes, buffer range checks, LDS, barriers). It checks what the kernels COMPUTE where no GPU exists; it says
nothing about timing, waitcnt hazards or register pressure, and it is not the parity gate: `pytest -m gpu` on the MI355X is.

The emulator is test infrastructure: refil_amd never loads it (tests/test_abi.py::test_missing_library_fails_loudly still holds)."""
import os
import shutil

import pytest

import emu_util

pytestmark = pytest.mark.skipif(not (shutil.which("clang++") or os.path.exists("/opt/rocm/lib/llvm/bin/clang++")),
                                reason="the emulator build needs a host clang++ (vector extensions, __bf16)")

_G = emu_util.load_copy("test_gpu_ops", DEV="cpu")


@pytest.fixture(autouse=True)
def _emulated_library():
    with emu_util.active():
        yield


# the gpu tier's tests and their fixtures, collected here under this module's marks (none: they run in the CPU tier)
for _k, _v in list(vars(_G).items()):
    if _k.startswith("test_") or _k in ("wres_mode", "dw_mode", "set_tuning"):
        globals()[_k] = _v
del _k, _v
