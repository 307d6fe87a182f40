"""Cycle counters of the 4-row forward recurrence (debug build: REFIL_EXTRA_FLAGS=-DREFIL_GRU_TIMING python -m refil_amd.build --force, or a
variant through tools/build_variant.sh + REFIL_LIB_PATH). Shape of the bench: 96 episode copies x 16 agents, T1 = 81, H = 64.
    python tools/probes/gru_timing.py [H]"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import hip_ops as ho  # noqa: E402
from refil_amd import _lib  # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 64
GB, T1, na = 96, 81, 16
NR = GB * na
dev = "cuda"
torch.manual_seed(0)
gi = torch.randn(GB * T1 * na, 3 * H, device=dev)
hsx = torch.zeros(GB, T1 + 1, na, H, device=dev)
saves = [torch.zeros(GB * T1 * na, H, device=dev) for _ in range(4)]
whh, bhh = (torch.randn(3 * H, H) / 8).to(dev), (torch.randn(3 * H) / 8).to(dev)
d = ho.gru_desc(gi, hsx, whh, bhh, NR, T1, na, H=H, saves=saves)
for _ in range(3):
    ho.gru_forward(d)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    ho.gru_forward(d)
torch.cuda.synchronize()
us = (time.perf_counter() - t0) / 20 * 1e6
print(f"gru_fwd4 H={H}: {us:.1f} us per launch = {us / T1 * 1e3:.0f} ns per dependent step")
lib = _lib.lib()
if hasattr(lib, "refil_debug_gru_timing"):
    out = (C.c_ulonglong * 8)()
    lib.refil_debug_gru_timing(out)
    steps = max(int(out[5]), 1)
    names = ["wait for the step's inputs", "LDS reads + recurrent product", "k-slice sums", "gates", "stores + barrier"]
    tot = sum(int(out[i]) for i in range(5))
    for i, n in enumerate(names):
        print(f"  {n:32s} {int(out[i]) / steps:8.0f} cycles per step")
    print(f"  {'total':32s} {tot / steps:8.0f} cycles per step ({steps} steps; implied clock {tot / (us * 1e-6) / 1e9:.2f} GHz)")
# the backward recurrence at the same shape
dhs = torch.randn(GB, T1, na, H, device=dev)
dgi = torch.zeros(GB * T1 * na, 3 * H, device=dev)
dgh = torch.zeros(GB * T1 * na, H, device=dev)
db = ho.gru_desc(gi, hsx, whh, bhh, NR, T1, na, H=H, saves=saves, dhs=dhs, dgi=dgi, dgh=dgh)
for _ in range(3):
    ho.gru_backward(db)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    ho.gru_backward(db)
torch.cuda.synchronize()
us = (time.perf_counter() - t0) / 20 * 1e6
print(f"gru_bwd4 H={H}: {us:.1f} us per launch = {us / T1 * 1e3:.0f} ns per dependent step")
