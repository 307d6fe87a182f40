"""Micro-benchmark of the GEMM shapes of the cfg-T learner step through the C ABI (for rocprofv3 / PMC runs).
usage: python tools/gemm_bench.py [reps] [kv]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch

import hip_ops
from refil_amd._lib import GEMM_A_OUTC, GEMM_B_OUTC, GEMM_RELU

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20

dev = "cuda"
NE, NA = 82944, 41472


def timeit(name, fn, flops):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"{name:34s} {us:9.1f} us  {flops / us / 1e6:7.1f} TFLOP/s")


def nt(M, N, K, batch=1, relu=False, bias=True):
    x = torch.randn(M, K * batch, device=dev)
    W = torch.randn(batch, N, K, device=dev) / K ** 0.5
    b = torch.randn(batch, N, device=dev) if bias else None
    y = torch.empty(batch, M, N, device=dev)
    fn = lambda: hip_ops.gemm(x, W, y, M, N, K, K * batch, K, N, flags=(GEMM_RELU if relu else 0), bias=b, batch=batch, sA=K,
                              sB=N * K, sC=M * N, sBias=N)
    timeit(f"NT M={M} N={N} K={K} b={batch}", fn, 2.0 * M * N * K * batch)


def dx(M, N, K, batch=1):
    dy = torch.randn(batch, M, N, device=dev)
    W = torch.randn(batch, N, K, device=dev)
    o = torch.empty(M, K * batch, device=dev)
    fn = lambda: hip_ops.gemm(dy, W, o, M, K, N, N, K, K * batch, flags=GEMM_B_OUTC, batch=batch, sA=M * N, sB=N * K, sC=K)
    timeit(f"dX M={M} N={N} K={K} b={batch}", fn, 2.0 * M * N * K * batch)


def dw(R, N, K, batch=1, splits=256):
    dy = torch.randn(batch, R, N, device=dev)
    x = torch.randn(batch, R, K, device=dev)
    dW = torch.empty(batch, N, K, device=dev)
    part = torch.empty(batch * splits * (N * K + N) + 64, device=dev)
    fn = lambda: hip_ops.gemm(dy, x, dW, N, K, R, N, K, K, flags=GEMM_A_OUTC | GEMM_B_OUTC, partial=part, batch=batch,
                              splits=splits, sA=R * N, sB=R * K, sC=N * K)
    timeit(f"dW R={R} N={N} K={K} b={batch} s={splits}", fn, 2.0 * R * N * K * batch)


only = sys.argv[2] if len(sys.argv) > 2 else ""      # e.g. "kv": just the dominant shape (for PMC runs)
if only == "dvfs":
    # the same launch on random, small-magnitude and all-zero operands: the chip clocks to its power budget (MI355X_MICROARCH.md,
    # "DVFS give-back"), so the operand DATA moves the rate of a matrix-bound kernel -- how far is the 2.4 GHz peak reachable?
    M, N, K, batch = NE, 256, 128, 4
    for name, fill in (("random N(0,1)", lambda t: t.normal_()), ("zeros", lambda t: t.zero_()), ("ones", lambda t: t.fill_(1.0)),
                       ("random N(0,1) again", lambda t: t.normal_())):
        x = torch.empty(M, K * batch, device=dev); fill(x)
        W = torch.empty(batch, N, K, device=dev); fill(W)
        y = torch.empty(batch, M, N, device=dev)
        fn = lambda: hip_ops.gemm(x, W, y, M, N, K, K * batch, K, N, batch=batch, sA=K, sB=N * K, sC=M * N, sBias=N)
        timeit(f"NT kv {name}", fn, 2.0 * M * N * K * batch)
    sys.exit(0)
if only == "dw":
    dw(NE, 256, 128, batch=4, splits=128)
    dw(3 * NA, 128, 128, splits=486)
    sys.exit(0)
if only == "kv":
    nt(NE, 256, 128, batch=4, bias=False)
    sys.exit(0)
nt(NE, 128, 84, relu=True)
nt(NE, 512, 84, relu=True)
nt(NE, 256, 128, bias=False)
nt(NE, 256, 128, batch=4, bias=False)
nt(3 * NA, 128, 128)
nt(3 * NA, 192, 64)
nt(3 * NA, 64, 128, relu=True)
nt(4096 * 32, 128, 128)
nt(4096 * 32, 128, 1024)
dx(NE, 256, 128, batch=4)
dx(3 * NA, 128, 128)
dw(NE, 256, 128, batch=4, splits=128)
dw(NE, 512, 84, splits=256)
