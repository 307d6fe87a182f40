"""Weight gradients of in_trans at the bench shape, alone on the GPU: dW[N,K] = dy[rows,N]^T x[rows,K] over a device row list
(gemm_dw4.hip: gemm_dws_kernel, the bf16 x 6 form), for a sweep of reduction splits (= workgroups).
    python tools/dws_bench.py [--iters 20] [--splits 16,32,64,128]
Prints us per launch, the HBM rate of the operands read once and the matrix rate of the fp32-equivalent work."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import hip_ops as ho  # noqa: E402
from refil_amd import _lib  # noqa: E402

A_OUTC, B_OUTC, COLSUM_A = _lib.GEMM_A_OUTC, _lib.GEMM_B_OUTC, _lib.GEMM_COLSUM_A


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--splits", default="8,16,32,64,128")
    ap.add_argument("--frac", type=float, default=0.46)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--hot", type=int, default=0, help="list entries fold onto this many rows (cache-resident operands: what is left is the kernel's own pace)")
    ap.add_argument("--only", default="", help="substring of the shape names to run")
    a = ap.parse_args()
    dev = "cuda"
    B, T1, ne, na = 32, 81, 32, 16
    R = B * T1 * ne
    g = torch.Generator().manual_seed(0)
    keep = torch.nonzero(torch.rand(R, generator=g) < a.frac).flatten().to(torch.int32)
    n = keep.numel()
    lst = torch.full(((n + 63) // 64 * 64 + 128,), R, dtype=torch.int32)
    lst[:n] = keep if not a.hot else keep % a.hot
    lst, cnt = lst.to(dev), torch.tensor([n], dtype=torch.int32, device=dev)
    # agent rows of the listed (b,t) rows: the Q projection's gradient reads x through the agent-row map
    ra = B * T1 * na
    keep_a = torch.nonzero(torch.rand(ra, generator=g) < a.frac).flatten().to(torch.int32)
    na_ = keep_a.numel()
    lsta = torch.full(((na_ + 63) // 64 * 64 + 128,), ra, dtype=torch.int32)
    lsta[:na_] = keep_a if not a.hot else keep_a % a.hot
    lsta, cnta = lsta.to(dev), torch.tensor([na_], dtype=torch.int32, device=dev)
    print(f"entity rows {R}, listed {n}; agent rows {ra}, listed {na_}")
    shapes = [("K/V, 4 hypernets", 256, 128, 4, R, lst, cnt, n, None), ("K/V, agent", 256, 128, 1, R, lst, cnt, n, None),
              ("Q, 4 hypernets", 128, 128, 4, ra, lsta, cnta, na_, (na, ne, 0)), ("Q, agent", 128, 128, 1, ra, lsta, cnta, na_, (na, ne, 0)),
              ("thin 128x64", 128, 64, 1, R, lst, cnt, n, None), ("thin 256x52", 256, 52, 1, R, lst, cnt, n, None), ("thin 128x64 x4", 128, 64, 4, R, lst, cnt, n, None)]
    for name, N, K, batch, rows, l, c, nl, bmap in shapes:
        if a.only not in name:
            continue
        dy = torch.randn(rows + 8, batch * N, generator=g).to(dev)
        x = torch.randn(R + 8, batch * K, generator=g).to(dev)
        for splits in [int(s) for s in a.splits.split(",")]:
            dW = torch.zeros(batch, N, K, device=dev)
            db = torch.zeros(batch, N, device=dev)
            partial = torch.zeros(batch * splits * (N * K + N), device=dev)

            def run():
                ho.gemm(dy, x, dW, N, K, rows, batch * N, batch * K, K, flags=A_OUTC | B_OUTC | COLSUM_A, colsum=db, partial=partial, splits=splits,
                        batch=batch, sA=N, sB=K, sC=N * K, sColsum=N, row_index=l, row_count=c, b_map=bmap or (0, 0, 0))
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.iters):
                run()
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / a.iters * 1e6
            gb = 4.0 * batch * nl * (N + K) / 1e9
            tf = 2.0 * batch * N * K * nl / 1e12
            msg = ""
            if a.check and not a.hot:
                kl = (keep if bmap is None else keep_a).long()
                xr = x[:R].cpu() if bmap is None else x[:R].cpu().view(B * T1, ne, -1)[:, :na].reshape(ra, -1)
                ref = dy[:rows].cpu()[kl, :N].double().t() @ xr[kl, :K].double()
                msg = f"  max err {(dW[0].cpu().double() - ref).abs().max().item():.2e} (|ref| max {ref.abs().max().item():.1f})"
            print(f"{name:18s} N={N} K={K} batch={batch} splits={splits:3d} ({batch * splits * (N // 256 if N >= 256 else 1):4d} workgroups): {us:7.1f} us  "
                  f"{gb / us * 1e6 / 1e3:5.2f} TB/s  {tf / us * 1e6:6.1f} TFLOP/s{msg}")


if __name__ == "__main__":
    main()
