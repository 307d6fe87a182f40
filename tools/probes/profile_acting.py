import cProfile, pstats, sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
W = bench.CONFIGS["cfgT"]; dims = bench.workload_dims(W)
args, batch, learner, _, _ = bench.build(dims, W["imagine"], 8, 80, seed=5, device=torch.device("cuda", 0))
mac = learner.mac
def run(n):
    mac.init_hidden(8)
    for t in range(n):
        a = mac.select_actions(batch, t_ep=t, t_env=t, test_mode=False); a.cpu()
run(80)
pr = cProfile.Profile(); pr.enable(); run(80); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
