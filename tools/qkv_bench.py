"""in_trans + attention forward: the fused launch (attention_qkv.hip) against the three launches it replaces (K/V projection, Q
projection, attention core) at the bench shape, on SC2-law rows (padded entities dead, ragged episode ends).
    python tools/qkv_bench.py [--iters 20] [--nets 4] [--nvar 1] [--store] [--cfg2]
Prints us per forward of `nets` attention blocks sharing rows and masks (a mixer's hypernets)."""
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
    ap.add_argument("--nets", type=int, default=4)
    ap.add_argument("--nvar", type=int, default=1)
    ap.add_argument("--store", action="store_true")
    ap.add_argument("--cfg2", action="store_true")
    ap.add_argument("--dense", action="store_true")
    a = ap.parse_args()
    B, T1, ne, na, heads, hd = (32, 81, 16, 8, 4, 16) if a.cfg2 else (32, 81, 32, 16, 4, 32)
    w, R, dev = heads * hd, B * T1, "cuda"
    nets = a.nets
    g = torch.Generator(device="cpu").manual_seed(0)
    # SC2 law: n active agents = n active enemies ~ U{3..na}; episode length ~ U{T/2..T}
    em = torch.ones(B, T1, ne, dtype=torch.uint8)
    t_last = torch.zeros(B, dtype=torch.int32)
    for b in range(B):
        n = int(torch.randint(3, na + 1, (1,), generator=g)) if not a.dense else na
        em[b, :, :n] = 0
        em[b, :, na:na + n] = 0
        t_last[b] = (T1 - 1) if (a.dense or b == 0) else int(torch.randint(T1 // 2, T1, (1,), generator=g))
    x = torch.randn(R * ne + 8, nets * w, generator=g).to(dev)      # (+8: the scratch row behind the row lists' padding)
    W = (torch.randn(nets, 3 * w, w, generator=g) / w ** 0.5).to(dev)
    emd = em.reshape(R, ne).to(dev)
    variants = [_lib.MASK_ENTITY, _lib.MASK_WITHIN, _lib.MASK_INTERACT][:a.nvar]
    bits = (torch.rand(B, ne, generator=g) < 0.5).to(torch.uint8).to(dev)
    kv = torch.zeros(nets, R * ne + 8, 2 * w, device=dev)
    qb = torch.zeros(nets, R * na + 8, w, device=dev)
    O = torch.zeros(nets, a.nvar, R * na, w, device=dev)
    kdead = em.reshape(R * ne).to(dev)
    qdead = em[:, :, :na].reshape(R * na).contiguous().to(dev)
    tl = t_last.to(dev)
    live = (torch.arange(T1)[None, :] <= t_last[:, None])
    rows_e = torch.nonzero((live[:, :, None] & ~em.bool()).reshape(-1)).flatten().to(torch.int32)
    rows_a = torch.nonzero((live[:, :, None] & ~em[:, :, :na].bool()).reshape(-1)).flatten().to(torch.int32)

    def pad(lst, trash):
        n = lst.numel()
        padded = torch.full(((n + 63) // 64 * 64 + 128,), trash, dtype=torch.int32)
        padded[:n] = lst
        return padded.to(dev), torch.tensor([n], dtype=torch.int32, device=dev)
    le, ce = pad(rows_e, R * ne)
    la, ca = pad(rows_a, R * na)
    print(f"R={R} live rows={int(live.sum())} listed entity rows={rows_e.numel()} ({rows_e.numel() / (R * ne):.2f}) agent rows={rows_a.numel()}")

    descs = []
    for n in range(nets):
        d = ho.attn_desc(qb[n], kv[n], kv[n][:, w:], w, 2 * w, R, T1, ne, na, heads, hd, variants, ent_mask=emd, ent_mask0=emd.view(B, T1, ne)[:, 0].contiguous(), group_bits=bits)
        ho.attn_skip(d, tl, kdead, qdead)
        ho.attn_mask_words(d, na)
        descs.append(d)

    def unfused():
        ho.gemm(x, W[:, w:], kv, R * ne, 2 * w, w, nets * w, w, 2 * w, batch=nets, sA=w, sB=3 * w * w, sC=(R * ne + 8) * 2 * w, row_index=le, row_count=ce)
        ho.gemm(x, W, qb, R * na, w, w, nets * w, w, w, batch=nets, sA=w, sB=3 * w * w, sC=(R * na + 8) * w, a_map=(na, ne, 0), row_index=la, row_count=ca)
        for n in range(nets):
            ho.attn_forward(descs[n], O[n], w, R * na * w)

    def fused():
        for n in range(nets):
            ho.attn_qkv_forward(descs[n], x[:, n * w:], nets * w, W[n], O[n], w, R * na * w,
                                q_out=qb[n] if a.store else None, k_out=kv[n] if a.store else None, v_out=kv[n][:, w:] if a.store else None)

    res = {}
    for name, f in (("unfused (K/V GEMM + Q GEMM + attention, one launch per net for the core)", unfused), ("fused, one launch per net", fused)):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        res[name] = O.clone()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            f()
        torch.cuda.synchronize()
        print(f"{name}: {(time.perf_counter() - t0) / a.iters * 1e6:.1f} us per forward of {nets} nets (nvar={a.nvar}, store={a.store})")
    a_, b_ = list(res.values())
    lq = (live[:, :, None] & ~em[:, :, :na].bool()).reshape(-1).to(dev)
    err = (a_ - b_)[:, :, lq].abs().max().item()
    print(f"max |fused - unfused| on live rows: {err:.3e} (max |O| {a_[:, :, lq].abs().max().item():.3e})")


if __name__ == "__main__":
    main()
