"""Soak test on the GPU: a few hundred QLearner.train steps on one synthetic batch (bench shapes, smaller B/T). Checks
that parameters and statistics stay finite and that the loss goes down (usage: python tools/soak.py [steps])."""
import sys, os, math
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
def main():
    import torch
    import bench
    W = bench.CONFIGS["cfgT"]
    dims = bench.workload_dims(W)
    args, batch, learner, data, _ = bench.build(dims, W["imagine"], 16, 40, seed=3, device=torch.device("cuda", 0))
    from plugin_util import RecLogger
    learner.logger = RecLogger()
    learner.args.learner_log_interval = 1
    learner.args.target_update_interval = 50
    losses = []
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    for i in range(N):
        learner.train(batch, t_env=i, episode_num=i)
        if i % 50 == 0 or i == N - 1:
            torch.cuda.synchronize()
            st = learner.logger.stats
            print(i, {k: round(v, 5) for k, v in st.items() if k in ("loss", "grad_norm", "td_error_abs", "q_taken_mean")})
            losses.append(st.get("loss"))
    flat = learner.flat_live
    assert torch.isfinite(flat).all(), "non-finite parameters"
    assert all(l is not None and math.isfinite(l) for l in losses)
    print("finite OK; loss first/last", losses[0], losses[-1])


if __name__ == "__main__":
    main()
