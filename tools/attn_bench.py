"""Attention-core microbenchmark at the bench shape (hypernet hyper_w_1 call: 3 mask variants, R = B*T rows).
    python tools/attn_bench.py [--iters 20] [--bwd-only]
Used under rocprofv3 --pmc to see what bounds attn_fwd_mfma / attn_bwd_mfma."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import hip_ops as ho  # noqa: E402
from refil_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--bwd-only", action="store_true")
    ap.add_argument("--nvar", type=int, default=3)
    a = ap.parse_args()
    B, T, ne, na, heads, hd = 32, 80, 32, 16, 4, 32
    d = heads * hd
    R = B * T
    dev = "cuda"
    g = torch.Generator(device="cpu").manual_seed(0)
    Q = torch.randn(R * na, d, generator=g).to(dev)
    KV = torch.randn(R * ne, 2 * d, generator=g).to(dev)
    em = (torch.rand(B, T, ne, generator=g) < 0.3).to(torch.uint8).to(dev)
    em0 = em[:, 0].contiguous()
    bits = (torch.rand(B, ne, generator=g) < 0.5).to(torch.uint8).to(dev)
    variants = [_lib.MASK_ENTITY, _lib.MASK_WITHIN, _lib.MASK_INTERACT][:a.nvar]
    desc = ho.attn_desc(Q, KV, KV[:, d:], d, 2 * d, R, T, ne, na, heads, hd, variants, ent_mask=em.view(R, ne), ent_mask0=em0, group_bits=bits)
    O = torch.empty(a.nvar, R * na, d, device=dev)
    dO = torch.randn(a.nvar, R * na, d, device=dev)
    dQ = torch.empty_like(Q)
    dKV = torch.empty_like(KV)

    def fwd():
        ho.attn_forward(desc, O, d, R * na * d)

    def bwd():
        ho.attn_backward(desc, dO, d, R * na * d, dQ, dKV, dKV[:, d:])

    for name, f in (("fwd", fwd), ("bwd", bwd)):
        if name == "fwd" and a.bwd_only:
            continue
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            f()
        torch.cuda.synchronize()
        print(f"{name}: {(time.perf_counter() - t0) / a.iters * 1e6:.1f} us/launch (R={R}, nvar={a.nvar})")


if __name__ == "__main__":
    main()
