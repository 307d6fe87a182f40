"""When do the workgroups of a persistent attention-backward launch start? (debug build: -DREFIL_ATTN_TIMING, tools/_libs/librefil_attn_timing.so)
Runs learner steps of the bench configuration (four streams, in situ) and, for the LAST attn_bwd_pipe launch of a step, prints the
distribution of the workgroups' start delays (relative to the first one) and lifetimes -- with a static row assignment a workgroup that
becomes resident late still has its whole share of rows to do.  --serial: the same with the chains serialised (kernel alone on the GPU)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from refil_amd import _lib

cfg = "cfgT"
W = dict(bench.CONFIGS[cfg])
dims = bench.workload_dims(W)
args, batch, learner, data, _ = bench.build(dims, W["imagine"], W["B"], W["T"], seed=100, device=torch.device("cuda:0"))
if "--serial" in sys.argv:
    _lib.lib().refil_set_overlap(0)
for i in range(10):
    learner.train(batch, t_env=0, episode_num=i)
torch.cuda.synchronize()
for rep in range(3):
    learner.train(batch, t_env=0, episode_num=0)
    torch.cuda.synchronize()
    n = 1024
    buf = (C.c_ulonglong * (2 * n))()
    assert _lib.lib().refil_debug_attn_timing(buf, n) == 0
    a = np.array(list(buf), dtype=np.float64).reshape(n, 2)
    a = a[(a[:, 0] > 0) & (a[:, 1] > 0)]
    t0 = a[:, 0].min()
    start = (a[:, 0] - t0) / 100.0          # us (100 MHz)
    life = (a[:, 1] - a[:, 0]) / 100.0
    end = (a[:, 1] - t0) / 100.0
    q = lambda x: " ".join(f"{np.percentile(x, p):7.1f}" for p in (0, 25, 50, 75, 90, 100))
    print(f"rep {rep}: {len(a)} workgroups; kernel span {end.max():.1f} us")
    print(f"   start delay us (min 25% 50% 75% 90% max): {q(start)}")
    print(f"   lifetime us                             : {q(life)}")
    print(f"   end us                                  : {q(end)}")
