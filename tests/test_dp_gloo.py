"""world_size-2 gloo test (CPU) of the data-parallel recipe in refil_amd/dp.py: un-normalised shard
gradients + ONE all-reduce(SUM) of [grads | stats] + division by the global sum(mask) reproduce the
single-process gradients of the reference's global-mean loss (q_learner.py:165,171). The per-shard
compute is done by the CPU oracle here (no GPU in this test); the product computes the same quantities
with the HIP kernels (tests/test_gpu_learner.py::test_full_size_properties checks their additivity)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import refil_oracle as orc


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _case():
    from refil_amd.synthetic import make_batch
    cfg = orc.Cfg(n_agents=3, n_entities=6, n_actions=5, entity_shape=9, attn_embed_dim=16, attn_n_heads=4,
                  hypernet_embed=16, mixing_embed_dim=32)
    batch = make_batch(4, 5, 6, seed=21, na=3, A=5, ed=9, min_active=1, death_p=0.05)
    agent = orc.init_params(orc.agent_param_shapes(cfg), 1)
    mixer = orc.init_params(orc.mixer_param_shapes(cfg), 2)
    tagent = orc.init_params(orc.agent_param_shapes(cfg), 3)
    tmixer = orc.init_params(orc.mixer_param_shapes(cfg), 4)
    torch.manual_seed(5)
    bits = orc.draw_partition_bits(4, 6)
    return cfg, batch, bits, agent, mixer, tagent, tmixer


def _sum_loss_flat(cfg, batch, bits, agent, mixer, tagent, tmixer):
    """[d(SUM-loss)/d(params) | sum(mask), sum td^2, sum td_im^2] -- what the HIP step hands to the all-reduce."""
    names = [("a", k) for k in agent] + [("m", k) for k in mixer]
    leaves = []
    a, m = dict(agent), dict(mixer)
    for w, k in names:
        d = a if w == "a" else m
        d[k] = d[k].clone().requires_grad_(True)
        leaves.append(d[k])
    out = orc.learner_forward(cfg, a, m, tagent, tmixer, batch, bits)
    msum = out.mask.sum()
    sum_loss = ((1 - cfg.lmbda) * out.q_loss + cfg.lmbda * out.im_loss) * msum
    grads = torch.autograd.grad(sum_loss, leaves)
    stats = torch.stack([msum, out.q_loss.detach() * msum, out.im_loss.detach() * msum])
    return torch.cat([g.reshape(-1) for g in grads] + [stats]).detach()


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from refil_amd import dp
    cfg, batch, bits, agent, mixer, tagent, tmixer = _case()
    shard = {k: v.contiguous() for k, v in dp.shard_episodes(batch, rank, world).items()}
    flat = _sum_loss_flat(cfg, shard, dp.shard_bits(bits, rank, world), agent, mixer, tagent, tmixer)
    dp.allreduce_sum_(flat)
    q.put((rank, flat))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_allreduce_equals_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert torch.equal(got[0], got[1]), "replicas diverge after the all-reduce"
    cfg, batch, bits, agent, mixer, tagent, tmixer = _case()
    full = _sum_loss_flat(cfg, batch, bits, agent, mixer, tagent, tmixer)
    n = full.numel() - 3
    assert torch.allclose(got[0][n:], full[n:], rtol=1e-5), "global sum(mask) / loss sums differ"
    g_dp = got[0][:n] / got[0][n]                 # the optimiser kernel's 1/sum(mask)
    g_ref = full[:n] / full[n]
    assert (g_dp - g_ref).abs().max().item() < 2e-6 * max(g_ref.abs().max().item(), 1e-6)


def test_shard_helpers_cover_the_batch_exactly():
    from refil_amd import dp
    cfg, batch, bits, *_ = _case()
    parts = [dp.shard_episodes(batch, r, 2) for r in range(2)]
    for k, v in batch.items():
        assert torch.equal(torch.cat([p[k] for p in parts]), v)
    assert torch.equal(torch.cat([dp.shard_bits(bits, r, 2) for r in range(2)]), bits)


def _validate_worker(rank, world, port, q, diverge):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["GPU_MAX_HW_QUEUES"] = "8"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    torch.manual_seed(3)
    live, sq = torch.randn(1000), torch.rand(1000)
    if diverge and rank == 1:
        live.view(torch.int32)[17] ^= 1                      # one mantissa bit of one parameter on one rank
    cs = bench.replica_checksum_of(live, sq)
    identical, cs0, allcs = bench.validate_replicas(cs, world)
    rec = bench.comm_record(1008, 12.5, "gloo", None, identical, cs0, 0.25)
    q.put((rank, identical, allcs, rec))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("diverge", [False, True])
def test_bench_replica_validation_two_ranks(diverge):
    """bench.py's self-validation of an N > 1 line (the `comm` object: replicas_identical, replica_checksums, ranks, hw_queues) on two gloo
    ranks: identical replicas pass; ONE flipped mantissa bit on one rank is seen by every rank (bench.py then exits instead of printing)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_validate_worker, args=(r, 2, port, q, diverge)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, identical, allcs, rec in got:
        assert identical == (not diverge)
        assert len(allcs) == 2 and (allcs[0] == allcs[1]) == (not diverge)
        assert rec["replicas_identical"] == (not diverge) and rec["ranks"] == 2 and rec["hw_queues"] == "8"
        assert rec["replica_checksums"] == allcs[0] and rec["bytes"] == 4032 and rec["loss_step0"] == 0.25
