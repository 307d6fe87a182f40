"""The CPU wavefront emulator checks itself (tests/emu/selftest.hip): the five matrix-instruction layouts against a host matmul, DPP
controls / permutes / shuffles / readlane / ballot against their ISA definitions, collectives in divergent control flow, the raw-buffer
range check, static + dynamic LDS with early-exit lanes and a partial last wave, atomics, co-resident workgroups behind a spin barrier."""
import ctypes
import os
import shutil
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CLANG = "/opt/rocm/lib/llvm/bin/clang++" if os.path.exists("/opt/rocm/lib/llvm/bin/clang++") else shutil.which("clang++")

pytestmark = pytest.mark.skipif(CLANG is None, reason="the emulator build needs a host clang++ (vector extensions, __bf16)")


def test_emulator_selftest(tmp_path):
    sys.path.insert(0, os.path.join(HERE, "emu"))
    import build_emu
    src = tmp_path / "selftest.cpp"
    with open(os.path.join(HERE, "emu", "selftest.hip")) as fh:
        src.write_text(build_emu.transform(fh.read(), "selftest.hip"))
    lib = tmp_path / "libemu_selftest.so"
    subprocess.check_call([CLANG, "-x", "c++", "-std=c++17", "-O1", "-fPIC", "-shared", "-pthread", "-fno-strict-aliasing", "-ffp-contract=off",
                           "-I", os.path.join(HERE, "emu", "include"), "-w", str(src), os.path.join(HERE, "emu", "emu_rt.cpp"), "-o", str(lib), "-ldl"])
    L = ctypes.CDLL(str(lib))
    buf = ctypes.create_string_buffer(8192)
    fails = L.emu_selftest(buf, 8192)
    assert fails == 0, f"{fails} mismatches:\n{buf.value.decode()}"
