"""CPU tier, world_size 2 (gloo): the data-parallel recipe of SURVEY.md section 8e with the HIP KERNELS doing the per-shard work -- on the CPU
wavefront emulator (tests/emu) --, not the oracle (tests/test_dp_gloo.py does that): every rank runs refil_learner_forward_backward on its
episodes (partition drawn once for the global batch and sliced), ONE all-reduce(SUM) of [grads | stats] over the job's process group,
then refil_clip_rmsprop_step, which divides by the GLOBAL sum(mask). Checked: the replicas end bit-identical (no parameter broadcast ever
happens), and equal to the single-process step on the whole batch up to the summation order of the gradient reduction."""
import os
import shutil
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.skipif(not (shutil.which("clang++") or os.path.exists("/opt/rocm/lib/llvm/bin/clang++")),
                                reason="the emulator build needs a host clang++ (vector extensions, __bf16)")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _case():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import emu_util
    L = emu_util.load_copy("test_gpu_learner", DEV="cpu")
    return L, L._oracle_case(6, 7, 16, seed=55, imagine=True, d=64, h=64)


def _step(L, cfg, batch, bits, agent, mixer, tagent, tmixer, allreduce):
    """one train step through the C ABI of the emulator build; allreduce(grads) sits where QLearner.train calls dp.allreduce_sum_"""
    from refil_amd import _lib, flat
    from refil_amd.engine import LearnerEngine
    B, T1 = batch["entities"].shape[:2]
    dims = L._dims(cfg, B, T1)
    eng = LearnerEngine("cpu")
    live = flat.pack(dims, agent, mixer, "cpu")
    targ = flat.pack(dims, tagent, tmixer, "cpu")
    n = flat.total(dims)
    grads = torch.full((n + _lib.REFIL_NSTAT,), float("nan"))
    eng.forward_backward(dims, {k: v.contiguous() for k, v in batch.items()}, bits.contiguous(), live, targ, grads)
    allreduce(grads)
    sq = torch.zeros(n)
    eng.clip_rmsprop(live, grads, sq, n, cfg.lr, cfg.optim_alpha, cfg.optim_eps, cfg.weight_decay, cfg.grad_norm_clip)
    return live.clone(), sq.clone(), grads.clone()


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["EMU_THREADS"] = "16"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    import emu_util
    from refil_amd import dp
    L, (cfg, batch, bits, agent, mixer, tagent, tmixer) = _case()
    shard = dp.shard_episodes(batch, rank, world)
    with emu_util.active():
        live, sq, grads = _step(L, cfg, shard, dp.shard_bits(bits, rank, world), agent, mixer, tagent, tmixer, dp.allreduce_sum_)
    q.put((rank, live, sq, grads))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_rank_step_with_the_kernels_equals_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {r: (live, sq, grads) for r, live, sq, grads in (q.get(timeout=800) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for a, b in zip(got[0], got[1]):
        assert torch.equal(a, b), "the replicas diverge"
    import emu_util
    L, (cfg, batch, bits, agent, mixer, tagent, tmixer) = _case()
    with emu_util.active():
        live1, sq1, grads1 = _step(L, cfg, batch, bits, agent, mixer, tagent, tmixer, lambda g: None)
    live2, sq2, grads2 = got[0]
    from refil_amd import _lib
    n = live1.numel()
    # sum(mask), the loss sums and the TD statistics are sums over episodes: equal up to fp32 summation order
    st1, st2 = grads1[n:n + 6].double(), grads2[n:n + 6].double()
    assert torch.allclose(st1, st2, rtol=2e-6, atol=1e-6), (st1, st2)
    assert st1[_lib.STAT_MASK_SUM] == st2[_lib.STAT_MASK_SUM]
    gmax = grads1[:n].abs().max().item()
    assert (grads1[:n] - grads2[:n]).abs().max().item() < 5e-6 * gmax, "summed shard gradients vs the whole-batch gradients"
    assert abs(grads1[n + _lib.STAT_GRAD_NORM].item() - grads2[n + _lib.STAT_GRAD_NORM].item()) < 5e-6 * grads1[n + _lib.STAT_GRAD_NORM].item()
    # the post-step parameters: 5e-6 absolute plus the first-order effect of the gradient difference through RMSprop's first step, which is
    # steep where |g| ~ eps (tests/test_gpu_learner.py: assert_post_close)
    msum = st1[_lib.STAT_MASK_SUM].item()
    sa = (1.0 - cfg.optim_alpha) ** 0.5
    g = grads1[:n].double().abs() / msum
    dg = (grads1[:n].double() - grads2[:n].double()).abs() / msum
    tol = 5e-6 + 1.5 * cfg.lr * cfg.optim_eps / (sa * g + cfg.optim_eps) ** 2 * dg
    assert ((live1.double() - live2.double()).abs() <= tol).all()
    assert (live1 != flat_before(L, cfg, batch, agent, mixer)).float().mean() > 0.9, "the step moved the parameters"


def flat_before(L, cfg, batch, agent, mixer):
    from refil_amd import flat
    B, T1 = batch["entities"].shape[:2]
    return flat.pack(L._dims(cfg, B, T1), agent, mixer, "cpu")
