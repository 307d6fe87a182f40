import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
W = dict(bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfgT"])
dims = bench.workload_dims(W)
args, batch, learner, data, _ = bench.build(dims, W["imagine"], W["B"], W["T"], seed=100, device=torch.device("cuda:0"))
for i in range(3):
    learner.train(batch, t_env=0, episode_num=i)
torch.cuda.synchronize()
sys.stderr.write("=====STEP\n")
learner.train(batch, t_env=0, episode_num=5)
torch.cuda.synchronize()
