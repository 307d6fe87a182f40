"""Stand-alone timing of the weight-resident projection GEMM at the north-star shape (rows x 128 -> 128, the launch bench.py
prices against the fp32-MFMA roof): python tools/probes/wres_bench.py [rows] [N] [K] [reps]   (rows a multiple of 32)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

import hip_ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 5808 * 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 128
K = int(sys.argv[3]) if len(sys.argv) > 3 else 128
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
dev = torch.device("cuda", 0)
torch.manual_seed(0)
x = torch.randn(M, K, device=dev)
W = torch.randn(N, K, device=dev) / K ** 0.5
b = torch.randn(N, device=dev)
y = torch.empty(M, N, device=dev)
for mode, flags in (("x W^T + b, relu", 1),):
    for _ in range(3):
        hip_ops.gemm(x, W, y, M, N, K, K, K, N, flags=flags, bias=b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        hip_ops.gemm(x, W, y, M, N, K, K, K, N, flags=flags, bias=b)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"{mode}: M={M} N={N} K={K}: {us:.1f} us per launch, {2.0 * M * N * K / us / 1e6:.1f} TFLOP/s, "
          f"{4.0 * (M * K + M * N) / us / 1e6:.2f} TB/s", flush=True)
ref = torch.relu(x[:4096] @ W.t() + b)
import hashlib  # noqa: E402
print("max |err| on the first 4096 rows:", (y[:4096] - ref).abs().max().item(), " sha1 of y:", hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:16])
