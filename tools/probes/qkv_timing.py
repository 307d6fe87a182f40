"""Per-phase cycle counts inside attn_qkv_fwd (debug build: REFIL_EXTRA_FLAGS=-DREFIL_QKV_TIMING python -m refil_amd.build --force).
One net of the hypernet shape on SC2-law rows; mean cycles per job: top (row words, key permutation), projection (incl. the waits for
the x rows), projection stores, core."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import hip_ops as ho
from refil_amd import _lib

dense = "--dense" in sys.argv
store = "--store" in sys.argv
nvar = 3 if "--nvar3" in sys.argv else 1
B, T1, ne, na, heads, hd = 32, 81, 32, 16, 4, 32
w, R, dev = heads * hd, B * T1, "cuda"
g = torch.Generator(device="cpu").manual_seed(0)
em = torch.ones(B, T1, ne, dtype=torch.uint8)
t_last = torch.zeros(B, dtype=torch.int32)
for b in range(B):
    n = int(torch.randint(3, na + 1, (1,), generator=g)) if not dense else na
    em[b, :, :n] = 0; em[b, :, na:na + n] = 0
    t_last[b] = (T1 - 1) if (dense or b == 0) else int(torch.randint(T1 // 2, T1, (1,), generator=g))
x = torch.randn(R * ne + 8, 4 * w, generator=g).to(dev)
W = (torch.randn(3 * w, w, generator=g) / w ** 0.5).to(dev)
emd = em.reshape(R, ne).to(dev)
bits = (torch.rand(B, ne, generator=g) < 0.5).to(torch.uint8).to(dev)
dummy = torch.zeros(4, device=dev)
d = ho.attn_desc(dummy, dummy, dummy, w, 2 * w, R, T1, ne, na, heads, hd, [_lib.MASK_ENTITY, _lib.MASK_WITHIN, _lib.MASK_INTERACT][:nvar], ent_mask=emd,
                 ent_mask0=emd.view(B, T1, ne)[:, 0].contiguous(), group_bits=bits)
ho.attn_skip(d, t_last.to(dev), em.reshape(R * ne).to(dev), em[:, :, :na].reshape(R * na).contiguous().to(dev))
ho.attn_mask_words(d, na)
O = torch.zeros(nvar, R * na, w, device=dev)
kv = torch.zeros(R * ne, 2 * w, device=dev); qb = torch.zeros(R * na, w, device=dev)
fn = lambda: ho.attn_qkv_forward(d, x, 4 * w, W, O, w, R * na * w, q_out=qb if store else None, k_out=kv if store else None, v_out=kv[:, w:] if store else None)
for _ in range(3):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); fn(); e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
nw = 2048
buf = (C.c_ulonglong * (8 * nw))()
assert _lib.lib().refil_debug_qkv_timing(buf, nw) == 0
a = np.array(list(buf), dtype=np.float64).reshape(nw, 8)
a = a[a[:, 4] > 0]
jobs = a[:, 4]
print(f"launch {us:.1f} us; {len(a)} waves with work, jobs per wave {jobs.min():.0f}..{jobs.max():.0f} (mean {jobs.mean():.2f}); live rows {int((torch.arange(T1)[None] <= t_last[:, None]).sum())}")
print(f"loop cycles per wave: mean {a[:, 5].mean():.0f} max {a[:, 5].max():.0f}; prologue (W staging, row table, first fetch) mean {a[:, 6].mean():.0f} max {a[:, 6].max():.0f}"
      f"  -> implied clock {(a[:, 5] + a[:, 6]).max() / us / 1e3:.2f} GHz")
for i, nm in enumerate(("top (words, key map)", "projection (+ x waits)", "projection stores", "core")):
    per = a[:, i] / jobs
    print(f"  {nm:24s} {per.mean():9.0f} cycles per job (min {per.min():.0f} max {per.max():.0f})")
print("  matrix-pipe floor per job: 240 x 16 (two key tiles) / 144 x 16 (one) bf16 + 32 x 32 fp32 cycles, two waves share a SIMD")
