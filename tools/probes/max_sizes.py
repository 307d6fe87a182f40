"""Largest shapes the library takes (64 entities = one mask word; long episodes; big batches): runs, stays finite, how fast, how much
workspace. python tools/probes/max_sizes.py"""
import sys, os, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import bench
for (ne, B, T) in [(64, 32, 80), (64, 64, 150), (48, 64, 150), (32, 128, 150)]:
    W = dict(bench.CONFIGS["cfgT"], ne=ne, B=B, T=T)
    dims = bench.workload_dims(W)
    t0 = time.time()
    args, batch, learner, data, _ = bench.build(dims, True, B, T, seed=1, device=torch.device("cuda", 0))
    for i in range(3):
        learner.train(batch, t_env=0, episode_num=i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(5):
        learner.train(batch, t_env=0, episode_num=i)
    e1.record(); torch.cuda.synchronize()
    ok = torch.isfinite(learner.flat_live).all().item()
    print(f"ne={ne} B={B} T={T}: {e0.elapsed_time(e1)/5:.2f} ms/step, finite={ok}, workspace {learner._engine.ws.buf.numel()/2**30:.1f} GiB, mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB, {B*T/(e0.elapsed_time(e1)/5)*1e3/1e6:.2f} M tr/s", flush=True)
    del learner, batch
    torch.cuda.empty_cache()
