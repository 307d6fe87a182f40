"""Per-phase cycle counts inside gemm_wres_kernel (debug build: REFIL_EXTRA_FLAGS=-DREFIL_WR_TIMING python -m refil_amd.build --force).
Runs the dominant shape (hypernets' K/V projection, dense rows) and prints mean cycles per row tile: top-of-tile (index / epilogue
operand requests), MFMA section, epilogue; s_memtime / readcyclecounter ticks are shader cycles."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import hip_ops
from refil_amd import _lib

dev = "cuda"
M, N, K, batch = 82944, 256, 128, 4
x = torch.randn(M, K * batch, device=dev)
W = torch.randn(batch, N, K, device=dev) / K ** 0.5
y = torch.empty(batch, M, N, device=dev)
fn = lambda: hip_ops.gemm(x, W, y, M, N, K, K * batch, K, N, batch=batch, sA=K, sB=N * K, sC=M * N, sBias=N)
for _ in range(3):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); fn(); e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
nw = 2048
buf = (C.c_ulonglong * (4 * nw))()
rc = _lib.lib().refil_debug_wres_timing(buf, nw)
assert rc == 0
import numpy as np
a = np.array(list(buf), dtype=np.float64).reshape(nw, 4)
a = a[a[:, 3] > 0]
tiles = a[:, 3]
print(f"launch {us:.1f} us; {len(a)} waves, tiles per wave {tiles.min():.0f}..{tiles.max():.0f}")
tot = a[:, :3].sum(1)
print(f"cycles per wave total: mean {tot.mean():.0f} max {tot.max():.0f}  -> implied clock {tot.max() / us / 1e3:.2f} GHz (max wave / launch time)")
for i, nm in enumerate(("top of tile", "MFMA section", "epilogue")):
    per = a[:, i] / tiles
    print(f"  {nm:14s} {per.mean():9.0f} cycles per tile (min {per.min():.0f} max {per.max():.0f})")
print(f"  MFMA floor: {4 * 16 * 4 * 64} cycles per tile (256 MFMAs x 64)")
