"""Learner-throughput bench of the MI355X-native REFIL hot path.

    python bench.py --gpus N --steps K --warmup W [--config cfgT|cfg2|cfg3|cfg4|cfg5] [--scaling weak|strong]
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

One "step" = one full QLearner.train() (reference: src/learners/q_learner.py:66-201) on a synthetic replay minibatch
resident in HBM: live + target agent forward, mixers, TD loss, hand-written backward, (all-reduce when N > 1),
clip + RMSprop, periodic target sync. Default workload = the north-star shape of BASELINE.json (B=32 episodes per GPU,
T=80 transitions, n_entities=32, attn/hypernet dim 128, REFIL); the other BASELINE.json configs are selected with
--config (SURVEY.md section 8d "Configs restated"). Weak scaling by default (per-GPU batch fixed, value = transitions/s
summed over all GPUs); --scaling strong fixes the GLOBAL batch (--global-batch, default 64) and shards it.

Prints ONE JSON line (rank 0). Extra objects:
  roofline     -- the kernel symbol with the largest isolated GPU time per step, timed with HIP events on the launch
                  stream by the library's profiler right after the timed region (events are kept out of the timed
                  region itself so they cannot perturb `value`). Every kernel is priced against BOTH roofs it can hit and
                  `bound` names the one that binds (the larger floor): the matrix pipe it ISSUES on -- fp32 products computed as
                  six bf16 matrix-pipe products of an exact 3-way operand split (gemm_wres / gemm_dws / the in_trans part of
                  attn_qkv_fwd; refil_profile_entry.flops_bf16x6) are priced at 2500 / 6 = 416.7 TFLOP/s of fp32 work, the rest at the
                  fp32 matrix instruction's 157.3 TFLOP/s -- and HBM at 8 TB/s with the bytes the launch really moved (rocprofv3 PMC:
                  2 * FETCH_SIZE + WRITE_SIZE, collected by bench.py itself or handed in with --traffic-json; without them the
                  algorithmic bytes). `frac` = that floor / measured time; `frac_vs_fp32_instruction` keeps the round-1..4 figure
                  (algorithmic fp32 FLOPs / time / 157.3 TFLOP/s: a comparison with the fp32 instruction, which the split form may
                  exceed). achieved / peak / unit are in the binding roof's unit (hbm: algorithmic bytes per launch / time). Every
                  kernels[] row carries bound, mfma_frac and hbm_frac.
  cpu_baseline -- oracle/refil_oracle.py (a fixture-pinned CPU port of the reference learner) timed on this box's host
                  cores on a bounded sample of the same workload (N=1, rank 0 only).
"""
import argparse
import json
import os
import statistics
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4 = the step's four streams). A process group on RCCL
# creates streams of its own: with four queues they share with the step's streams and EVERY step is 17 % (cfg-T) to 40 %
# (cfg2) slower, with or without a collective in it (tools/probes/dp_overhead.py, one-rank group on one GPU); with eight the
# step is unchanged and the all-reduce costs ~10 us. Single-process runs are unaffected (A/B +-0.2 %). Read at HIP initialisation.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch
import torch.distributed as dist

PEAK_FP32_MFMA_TFLOPS = 157.3
PEAK_BF16_MFMA_TFLOPS = 2500.0     # dense (MI355X_MICROARCH.md; 2495 measured)
PEAK_SPLIT_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0      # fp32 products computed as six bf16 matrix-pipe products: 416.7 TFLOP/s of fp32 work
PEAK_HBM_GBS = 8000.0
# BASELINE.json configs. B = episodes per GPU under weak scaling.
CONFIGS = {
    "cfgT": dict(B=32, T=80, ne=32, d=128, h=128, imagine=True, what="north-star target shape (BASELINE.json north_star / metric)"),
    "cfg2": dict(B=32, T=80, ne=16, d=64, h=64, imagine=True, what="BASELINE.json configs[1]"),
    "cfg3": dict(B=64, T=80, ne=32, d=128, h=128, imagine=True, what="BASELINE.json configs[2], roofline run"),
    "cfg4": dict(B=32, T=150, ne=16, d=128, h=128, imagine=False, what="BASELINE.json configs[3] shape: 3-8sz, qmix_atten (no imagination), T up to 150"),
    "cfg5": dict(B=32, T=80, ne=48, d=128, h=128, imagine=True, A=54,
                 what="BASELINE.json configs[4] scaled to 48 entities, MMM action law (medivac heal targets: A = 6 + 24 + 24 = 54, entity width 110, "
                      "E = 164: SURVEY.md section 8d 'scaled variant')"),
}
COMMON = dict(heads=4, H=64, M=32)


def algorithmic_flops(B, T, ne, na, E, A, d, h, H, M, G):
    """SURVEY.md section 8d closed form (2mnk per GEMM, backward 2x / first layers 1x, targets forward-only)."""
    T1 = T + 1

    def net(rows, w, V, tail):
        fc1 = 2 * ne * E * w
        qkv = 2 * na * w * w + 4 * ne * w * w
        core = 4 * na * ne * w
        out = 2 * na * w * w
        fwd = rows * (fc1 + qkv + V * (core + out + tail))
        bwd = rows * (fc1 + 2 * qkv + V * 2 * (core + out + tail))
        return fwd, bwd
    tail_a = 2 * na * d * H + 12 * na * H * H + 2 * na * H * A
    tail_h = 2 * na * h * M
    la_f, la_b = net(B * T1, d, G, tail_a)
    ta_f, _ = net(B * T1, d, 1, tail_a)
    lm = [net(B * T, h, G, tail_h)] + [net(B * T, h, 1, tail_h) for _ in range(3)]
    tm = [net(B * T, h, 1, tail_h) for _ in range(4)]
    return la_f + la_b + ta_f + sum(f + b for f, b in lm) + sum(f for f, _ in tm)


def qkv_useful_fraction(rows, ne, na):
    """Share of the FLOPs credited to the fused in_trans + attention launch (attention_qkv.hip) that falls on entity rows which can
    influence the loss. The launch is credited, per live (b,t) row, with the projections of ALL ne entities / na agents and the full
    na x ne core (SURVEY.md section 8d's per-row figure): it computes whole 16-entity tiles, where the row-list GEMMs it replaced ran
    exactly the listed rows. `rows` = LearnerEngine.row_counts(): listed entity rows of the agent nets / the hypernets, active agent
    rows, live steps. One symbol covers the agent launches (1 of 5 nets) and the hypernet launches (4 of 5): weighted like that.
    Projections: 2 na w^2 (Q) + 4 ne w^2 (K, V) per row; core: ~ na ne w per variant -- the projections dominate (w >= 64), the core's
    na ne product scales with both live fractions."""
    live_rows_e = max(rows["live_steps"] * ne, 1)
    live_rows_a = max(rows["live_steps"] * na, 1)
    f_a = min(1.0, rows["agent_rows"] / live_rows_a)
    out = 0.0
    for share, listed in ((0.2, rows["entity_rows_agent"]), (0.8, rows["entity_rows_hyper"])):
        f_e = min(1.0, listed / live_rows_e)
        out += share * (2.0 * na * f_a + 4.0 * ne * f_e) / (2.0 * na + 4.0 * ne)
    return out


def replica_checksum_of(flat_live, square_avg):
    """bit-level checksums of the live parameters and the RMSprop state (int32 views summed in int64)"""
    return torch.stack([flat_live.view(torch.int32).to(torch.int64).sum(), square_avg.view(torch.int32).to(torch.int64).sum()])


def validate_replicas(cs, world):
    """Self-validation of a multi-GPU run (after the timed region): every rank's checksums gathered over the job's own process group;
    the replicas must be bit-identical (one all-reduce(SUM) of [grads | stats] and the global sum(mask) in the optimiser kernel keep them
    so: no parameter broadcast ever happens). Returns (identical, checksums of rank 0, all ranks' checksums). Backend-agnostic: the 2-rank
    gloo test of the CPU tier runs this very function (tests/test_dp_gloo.py)."""
    gathered = [torch.zeros_like(cs) for _ in range(world)]
    dist.all_gather(gathered, cs)
    identical = all(torch.equal(g, gathered[0]) for g in gathered)
    return bool(identical), [int(x) for x in gathered[0].tolist()], [[int(x) for x in g.tolist()] for g in gathered]


def comm_record(n_floats, allreduce_us, backend, nccl_version, identical, checksums, loss_step0):
    """the `comm` object of an N > 1 bench line"""
    return {"bytes": n_floats * 4, "allreduce_us_per_step": allreduce_us, "backend": backend,
            "nccl_version": nccl_version, "algorithm": os.environ.get("REFIL_ALLREDUCE", "backend all_reduce(SUM), one collective per step"),
            "buckets": os.environ.get("REFIL_DP_BUCKETS") == "1",
            "ranks": dist.get_world_size(), "replicas_identical": bool(identical),
            "replica_checksums": checksums, "loss_step0": loss_step0,
            "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES")}


def workload_dims(W):
    """Network / environment sizes of a CONFIGS entry (the SC2 shape law of refil_amd.synthetic for its entity count)."""
    from refil_amd.synthetic import sc2_shape_law
    W = dict(W, **COMMON)
    law = sc2_shape_law(W["ne"])
    A = W.get("A", law["n_actions"])                       # (an action-count override: the entity width follows the feature layout)
    return dict(ne=W["ne"], na=law["n_agents"], A=A, ed=W["ne"] + (A - 2) + 10, d=W["d"], h=W["h"],
                heads=W["heads"], H=W["H"], M=W["M"])


def make_args(dims, imagine):
    return types.SimpleNamespace(
        agent="imagine_entity_attend_rnn" if imagine else "entity_attend_rnn", mac="entity_mac", learner="q_learner", mixer="flex_qmix",
        agent_output_type="q", action_selector="epsilon_greedy", epsilon_start=1.0, epsilon_finish=0.05, epsilon_anneal_time=500000,
        n_agents=dims["na"], n_actions=dims["A"], n_entities=dims["ne"], entity_shape=dims["ed"], entity_scheme=True,
        entity_last_action=True, gt_mask_avail=False, attn_embed_dim=dims["d"], attn_n_heads=dims["heads"],
        rnn_hidden_dim=dims["H"], hypernet_embed=dims["h"], mixing_embed_dim=dims["M"], softmax_mixing_weights=True,
        pooling_type=None, double_q=True, gamma=0.99, lmbda=0.5, lr=0.0005, optim_alpha=0.99, optim_eps=0.00001,
        weight_decay=0, grad_norm_clip=10, target_update_interval=200, learner_log_interval=10 ** 9, device="cuda",
        use_cuda=True)


class _Logger:
    def __init__(self):
        self.console_logger = types.SimpleNamespace(info=lambda *a, **k: None)

    def log_stat(self, *a):
        pass


def densify(data):
    """--dense-data: every entity active and observed for the whole (full-length, unterminated) episode."""
    d = {k: v.clone() for k, v in data.items()}
    d["entity_mask"].zero_()
    d["obs_mask"].zero_()
    d["filled"].fill_(1)
    d["terminated"].zero_()
    d["avail_actions"].fill_(1)
    return d


def build(dims, imagine, B, T, seed, device, shard=None, dense=False, fresh=0, host_buffer=False):
    """shard = (rank, world): strong scaling -- generate the GLOBAL batch of B episodes and keep this rank's slice.
    fresh = K > 0: additionally a device ReplayBuffer holding K*B episodes (K independently seeded batches)."""
    from refil_amd.components.episode_buffer import EpisodeBatch
    from refil_amd.components.transforms import OneHot
    from refil_amd.controllers import REGISTRY as mac_REGISTRY
    from refil_amd.learners import REGISTRY as le_REGISTRY
    from refil_amd.synthetic import make_batch_fast
    args = make_args(dims, imagine)
    data = make_batch_fast(B, T, dims["ne"], seed=seed, na=dims["na"], A=dims["A"])
    if dense:
        data = densify(data)
    if shard is not None:
        r, n = shard
        per = B // n
        data = {k: v[r * per:(r + 1) * per].contiguous() for k, v in data.items()}
        B = per
    scheme = {
        "entities": {"vshape": dims["ed"], "group": "entities"},
        "obs_mask": {"vshape": dims["ne"], "group": "entities", "dtype": torch.uint8},
        "entity_mask": {"vshape": dims["ne"], "dtype": torch.uint8},
        "actions": {"vshape": (1,), "group": "agents", "dtype": torch.long},
        "avail_actions": {"vshape": (dims["A"],), "group": "agents", "dtype": torch.int},
        "reward": {"vshape": (1,)},
        "terminated": {"vshape": (1,), "dtype": torch.uint8},
    }
    groups = {"agents": dims["na"], "entities": dims["ne"]}
    def episode_batch(dk):
        eb = EpisodeBatch(scheme, groups, B, T + 1, preprocess={"actions": ("actions_onehot", [OneHot(dims["A"])])}, device=device)
        eb.update({kk: v for kk, v in dk.items() if kk != "filled"}, mark_filled=False)
        eb.data.transition_data["filled"].copy_(dk["filled"])
        return eb
    batch = episode_batch(data)
    torch.manual_seed(0)                                   # identical replicas on every rank
    mac = mac_REGISTRY[args.mac](batch.scheme, groups, args)
    learner = le_REGISTRY[args.learner](mac, batch.scheme, _Logger(), args)
    if device.type == "cuda":                              # (a host device: the CPU tier's emulator runs of tests/test_gpu_early.py)
        learner.cuda()
    learner.generator = torch.Generator().manual_seed(1234)     # the SAME stream on every rank: train() draws the global partition and slices it
    buffer = None
    if fresh > 0:
        from refil_amd.components.episode_buffer import ReplayBuffer
        # host_buffer: the reference's buffer_cpu_only layout -- pinned host storage, device minibatches gathered over PCIe
        buffer = ReplayBuffer(scheme, groups, fresh * B, T + 1, preprocess={"actions": ("actions_onehot", [OneHot(dims["A"])])},
                              device="cpu" if host_buffer else device, sample_device=device if host_buffer else None)
        for k in range(fresh):
            dk = data if k == 0 else make_batch_fast(B, T, dims["ne"], seed=seed + 1000 * k, na=dims["na"], A=dims["A"])
            if dense and k:
                dk = densify(dk)
            buffer.insert_episode_batch(episode_batch(dk))
    if device.type == "cuda" or torch.cuda.Event.__name__ == "_HostEvent":      # (or tests/emu_util's stand-in event)
        batch.ready_event = torch.cuda.Event()             # the batch is complete here: lets train() run its prologue early
        batch.ready_event.record()
    learner._bench_episode_batch = episode_batch           # (bench.py's second, densified timed region builds its batch with it)
    return args, batch, learner, data, buffer


def gpu_state(index):
    """clock / power / temperature of the device right after the timed region (rocm-smi; box-to-box spread of the headline is ~5 %: the
    line carries the state it was measured in). None when rocm-smi is not available."""
    import shutil
    import subprocess
    if shutil.which("rocm-smi") is None:
        return None
    try:
        r = subprocess.run(["rocm-smi", "-d", str(index), "--showclocks", "--showpower", "--showtemp", "--json"], stdout=subprocess.PIPE,
                           stderr=subprocess.DEVNULL, timeout=20)
        card = next(iter(json.loads(r.stdout.decode()).values()))
        keep = {}
        for k, v in card.items():
            kl = k.lower()
            if any(t in kl for t in ("sclk", "mclk", "fclk", "power", "temperature (sensor junction)", "temperature (sensor edge)")):
                keep[k] = v
        return keep or None
    except Exception:
        return None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(dims, imagine, data_full, target_seconds=20.0):
    """Time the CPU oracle (port of the reference learner) on a bounded sample of the same workload."""
    from oracle import refil_oracle as orc
    cfg = orc.Cfg(n_agents=dims["na"], n_entities=dims["ne"], n_actions=dims["A"], entity_shape=dims["ed"],
                  attn_embed_dim=dims["d"], attn_n_heads=dims["heads"], hypernet_embed=dims["h"], imagine=imagine)
    agent = orc.init_params(orc.agent_param_shapes(cfg), 1)
    mixer = orc.init_params(orc.mixer_param_shapes(cfg), 2)
    tagent = orc.init_params(orc.agent_param_shapes(cfg), 3)
    tmixer = orc.init_params(orc.mixer_param_shapes(cfg), 4)
    T = data_full["entities"].shape[1] - 1

    def run(Bs, n):
        batch = {k: v[:Bs].contiguous() for k, v in data_full.items()}
        torch.manual_seed(0)
        bits = orc.draw_partition_bits(Bs, dims["ne"]) if imagine else None
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            orc.train_step(cfg, dict(agent), dict(mixer), tagent, tmixer, batch, bits)
            ts.append(time.perf_counter() - t0)
        return ts
    # pick the thread count that makes the CPU port fastest on this host (all cores is not always best for these
    # small GEMMs); `cores` reports the count actually used, `host_cores` what the box has
    ncpu = os.cpu_count() or 1
    cands = sorted({c for c in (8, 16, 32, 64) if 1 <= c <= ncpu}) or [1]
    threads, probe = cands[0], None
    for c in cands:
        torch.set_num_threads(c)
        t = run(4, 2)[-1]                                  # 2nd run: thread pool warm
        if probe is None or t < probe:
            threads, probe = c, t
    torch.set_num_threads(threads)
    per_ep = probe / 4
    Bfull = data_full["entities"].shape[0]
    Bs = Bfull
    while Bs > 4 and per_ep * Bs * 7 > target_seconds:
        Bs //= 2
    ts = run(Bs, 7)[2:]                                    # 2 warm-up + 5 timed
    best = statistics.median(ts)
    return {"value": round(Bs * T / best, 1), "unit": "transitions/s", "cores": threads, "host_cores": ncpu, "cpu_model": cpu_model(),
            "kind": "port",
            "sample": f"median of 5 oracle train steps (after 2 warm-ups) on B={Bs} of the bench batch (T={T}, ne={dims['ne']}, "
                      f"d={dims['d']}), fp32 torch CPU, {threads} threads (fastest of {cands})", "ms_per_step": round(best * 1e3, 1)}


def collect_traffic(argv_cfg, timeout_s=240, tuning=None):
    """HBM bytes per launch and per step from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, kernel-trace +
    pmc only) of a short serialised run of THIS bench configuration, folded exactly like tools/pmc_summarize.py:
    bytes = 2 * FETCH_SIZE + WRITE_SIZE (KiB -> bytes; the factor 2 is gfx950's FETCH_SIZE under-report for wide coalesced reads,
    MI355X_MICROARCH.md). Returns None when rocprofv3 is not on PATH or a pass fails (the bench line then says traffic: null)."""
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_summarize as ps
    steps_prof = 5
    tmp = tempfile.mkdtemp(prefix="refil_pmc_")
    # (the profiled runs take the schedule the timed region ran with -- given, not measured again: the autotuner's launches would
    # be counted into the five profiled steps)
    env = dict(os.environ, TMPDIR=tmp, REFIL_AUTOTUNE=",".join(f"{k}={v}" for k, v in (tuning or {}).items()) or "0")
    dirs = {}
    try:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, c)
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", c, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-profile", "--serial",
                   "--no-traffic", "--no-dense-region"] + argv_cfg
            r = subprocess.run(cmd, cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s)
            if r.returncode != 0:
                return None
            dirs[c] = d
        F, Wr = ps.fold(dirs["FETCH_SIZE"]), ps.fold(dirs["WRITE_SIZE"])
        total = 0.0
        import csv
        import glob
        skip = ("FillFunctor", "copyBuffer", "fillBuffer")        # the one-time zero fill of the arena and torch's set-up copies
        for c, mul in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
            for f in glob.glob(os.path.join(dirs[c], "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if not any(x in row["Kernel_Name"] for x in skip):
                        total += mul * float(row["Counter_Value"]) * 1024.0
        per = {}
        for k in set(F) | set(Wr):
            f = F[k][0] / max(F[k][1], 1) if k in F else 0.0
            w = Wr[k][0] / max(Wr[k][1], 1) if k in Wr else 0.0
            per[k] = round(2 * f + w)
        launches = {k: max(F[k][1] if k in F else 0, Wr[k][1] if k in Wr else 0) for k in per}
        return {"per_launch": per, "launches": launches, "hbm_bytes_per_step": total / steps_prof, "steps": steps_prof}
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="cfgT", choices=sorted(CONFIGS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--global-batch", type=int, default=64, help="episodes of the whole job under --scaling strong")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--top", type=int, default=10, help="kernels listed in the `kernels` array (by isolated time per step)")
    ap.add_argument("--batch", type=int, default=0, help="episodes per GPU instead of the config's")
    ap.add_argument("--serial", action="store_true", help="serialise the two chains on one stream for the whole run "
                    "(kernel-quality profiling: in-situ == isolated); the default overlaps them")
    ap.add_argument("--traffic-json", default=None, help="rocprofv3 PMC summary (tools/pmc_summarize.py) of THIS build to quote "
                    "roofline.traffic from; without it traffic is null (never a stale file)")
    ap.add_argument("--no-traffic", action="store_true", help="do not collect roofline.traffic / hbm_gb_per_step with rocprofv3 PMC passes")
    ap.add_argument("--fresh-batches", type=int, default=0, help="K > 0: a device replay buffer of K*B episodes is filled once and every step "
                    "trains on a FRESH ReplayBuffer.sample(B) drawn INSIDE the timed region (the gather launch and stale row-count hints "
                    "are part of the step); default: one resident batch")
    ap.add_argument("--host-buffer", action="store_true", help="with --fresh-batches K: the replay buffer lives in pinned HOST memory (the reference's "
                    "buffer_cpu_only) and every step's sample(B) is gathered over PCIe by the GPU inside the timed region: the PCIe-inclusive rate")
    ap.add_argument("--no-dense-region", action="store_true", help="skip the second short timed region on the densified batch (`dense_data` object)")
    ap.add_argument("--dense-data", action="store_true", help="synthetic data without padding: every entity alive, full-length episodes "
                    "(nothing for the row lists to skip: the dense-equivalent FLOPs are the executed FLOPs)")
    a = ap.parse_args()
    # The library's default schedule is deterministic (no measuring). The bench opts into the first-call autotuner
    # (refil_amd/tuning.py): it may only pick knob values of tuning.PARITY_TESTED -- every one of them is compared with the torch /
    # oracle references by tests/ -- and the line reports what it chose (config.schedule_autotune). REFIL_AUTOTUNE=0 times the defaults.
    os.environ.setdefault("REFIL_AUTOTUNE", "1")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # REFIL_BENCH_ONE_GPU=1 (tests only): all ranks share cuda:0 over gloo, to exercise the N>1 control flow
    # on a one-GPU box; the real launch is one rank per GPU over RCCL ("nccl")
    one_gpu = os.environ.get("REFIL_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "gloo" if one_gpu else "nccl"
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"

    from refil_amd import _lib
    W = dict(CONFIGS[a.config], **COMMON)
    if a.batch > 0:
        W["B"] = a.batch
    dims = workload_dims(W)
    T = W["T"]
    if a.scaling == "strong":
        assert a.global_batch % world == 0, "--global-batch must be a multiple of the number of GPUs"
        B = a.global_batch // world                        # per rank
        args, batch, learner, data, buffer = build(dims, W["imagine"], a.global_batch, T, seed=100, device=device, shard=(rank, world),
                                                    dense=a.dense_data, fresh=a.fresh_batches, host_buffer=a.host_buffer)
        global_B = a.global_batch
    else:
        B = W["B"]
        args, batch, learner, data, buffer = build(dims, W["imagine"], B, T, seed=100 + rank, device=device, dense=a.dense_data, fresh=a.fresh_batches, host_buffer=a.host_buffer)
        global_B = B * world

    if a.serial:
        _lib.lib().refil_set_overlap(0)

    import numpy as np
    np.random.seed(7 + rank)            # (ReplayBuffer.sample draws the episode ids with numpy's global generator, like the reference)

    def step(i):
        if buffer is not None:          # a fresh minibatch per step: one gather launch over every scheme field, inside the timed region
            learner.train(buffer.sample(B), t_env=0, episode_num=i)
        else:
            learner.train(batch, t_env=0, episode_num=i)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step_loss():
        """the loss of the step just taken from the (all-reduced) stat sums: (1 - lmbda) sum(td^2) / sum(mask) + lmbda sum(td_im^2) / sum(mask)"""
        st = learner.grads[learner._n:learner._n + _lib.REFIL_NSTAT].double().cpu()
        q = st[_lib.STAT_TD_SQ] / st[_lib.STAT_MASK_SUM]
        return float((1 - args.lmbda) * q + args.lmbda * st[_lib.STAT_IM_TD_SQ] / st[_lib.STAT_MASK_SUM]) if W["imagine"] else float(q)

    def replica_checksum():
        return replica_checksum_of(learner.flat_live, learner.square_avg)

    loss_step0 = None
    for i in range(a.warmup):
        step(i)
        if i == 0:              # (outside the timed region) the job's first loss: under strong scaling it equals the single-process loss of the
            torch.cuda.synchronize()     # same global batch up to summation order -- a first multi-GPU run can be checked against an N = 1 line
            loss_step0 = step_loss()
    barrier()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(a.steps):
        step(a.warmup + i)
        marks[i + 1].record()                              # (an event record costs ~1 us; the per-step medians come from these)
    host_enqueue = time.perf_counter() - t0                # host time to enqueue K steps (no sync inside)
    barrier()
    elapsed = time.perf_counter() - t0
    per_step = [marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps)]
    from refil_amd import dp, tuning
    tuned = tuning.check(learner.tuning_chosen(), "bench.py: the timed schedule")       # refuses a knob value without parity coverage
    # transitions that carry loss weight (sum of the TD mask, q_learner.py:68-72 -- all-reduced under data parallelism): the
    # headline counts B*T slots per step like BASELINE.json's metric, this is the rate of the slots that are not padding
    mask_sum = float(learner.grads[learner._n + _lib.STAT_MASK_SUM].item())
    rank_ms = [elapsed / a.steps * 1e3]
    comm = None
    if world > 1:
        tl = [None] * world
        dist.all_gather_object(tl, rank_ms[0])
        rank_ms = tl
        # the step's collective alone, on the stream the step issues it on (10 back-to-back all-reduces of the [grads | stats] buffer)
        ce0, ce1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        gb = learner.grads.clone()
        dp.allreduce_sum_(gb)
        barrier()
        ce0.record()
        for _ in range(10):
            dp.allreduce_sum_(gb)
        ce1.record()
        ce1.synchronize()
        try:
            ver = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:
            ver = None
        # Self-validation of a multi-GPU run: the replicas must still be bit-identical after the timed steps (one all-reduce(SUM) of
        # [grads | stats], the global sum(mask) applied by the optimiser kernel: no parameter broadcast ever happens) -- checked on bit-level
        # checksums of the live parameters and the RMSprop state, gathered over the job's own process group; a divergence is an error
        cs = replica_checksum()
        identical, cs0, _ = validate_replicas(cs, world)
        comm = comm_record(gb.numel(), round(ce0.elapsed_time(ce1) * 100.0, 1), backend, ver, identical, cs0, loss_step0)
        if not identical:
            raise SystemExit(f"bench.py: the data-parallel replicas diverged after {a.warmup + a.steps} steps (rank {rank}: {cs.tolist()} vs rank 0: "
                             f"{cs0}) -- the line would not be a valid measurement")
    # a second, short timed region on the DENSIFIED batch (no padding, full-length episodes: nothing for the row lists to skip),
    # so that the headline cannot be read as a dense rate
    dense = None
    if not a.dense_data and not a.no_dense_region and buffer is None:
        db = learner._bench_episode_batch(densify(data))        # (`data` is this rank's shard)
        db.ready_event = torch.cuda.Event()
        db.ready_event.record()
        nd = max(5, min(a.steps, 10))
        for i in range(3):
            learner.train(db, t_env=0, episode_num=0)
        barrier()
        td0 = time.perf_counter()
        for i in range(nd):
            learner.train(db, t_env=0, episode_num=0)
        barrier()
        td = time.perf_counter() - td0
        if world > 1:
            t = torch.tensor([td], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            td = t.item()
        dense = {"value": round(global_B * T * nd / td, 1), "ms_per_step": round(td / nd * 1e3, 3), "steps": nd}
        if not a.no_profile:
            _lib.profile_enable(True)
            for i in range(3):
                learner.train(db, t_env=0, episode_num=0)
            dfl = sum(e["flops"] for e in _lib.profile_collect()) / 3
            _lib.profile_enable(False)
            dense["executed_gflop_per_step"] = round(dfl / 1e9, 2)
            dense["step_frac_executed"] = round(dfl / (td / nd) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)
        for i in range(3):                                  # (back on the bench batch: row-count hints, early-prologue slots)
            step(a.warmup + a.steps)
        barrier()
    profiled = not a.no_profile       # every rank runs the extra passes (train() all-reduces); rank 0 reports
    ents, nprof = [], max(3, min(a.steps, 10))
    if profiled:
        # HIP events around every launch, both streams overlapping as in the timed region (in situ)
        _lib.profile_enable(True)
        for i in range(nprof):
            step(a.warmup + a.steps + i)
        ents = _lib.profile_collect()
        _lib.profile_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    ms_per_step = elapsed / a.steps * 1e3
    value = global_B * T * a.steps / elapsed
    rows = learner._engine.row_counts(learner._last_dims)

    E = dims["ed"] + dims["A"]
    G = 3 if W["imagine"] else 1
    flops_step = algorithmic_flops(B, T, dims["ne"], dims["na"], E, dims["A"], dims["d"], dims["h"], dims["H"], dims["M"], G)
    roofline = None
    kernels = None
    if profiled:
        tot = sum(e["total_ms"] for e in ents)
        # second pass with the two streams serialised: every kernel alone on the GPU (kernel quality). These
        # are the durations rocprofv3 --kernel-trace reports for the same steps (and for a whole `--serial` run)
        _lib.lib().refil_set_overlap(0)
        iso = {}
        for rep in range(2):          # two passes, the faster one per kernel: a one-off stall (seen once: 4 ms inside one launch
            _lib.profile_enable(True)   # of ten) must not decide which kernel is "dominant"
            for i in range(nprof):
                step(a.warmup + a.steps + (1 + rep) * nprof + i)
            for e in _lib.profile_collect():
                if e["name"] not in iso or e["total_ms"] < iso[e["name"]]["total_ms"]:
                    iso[e["name"]] = e
            _lib.profile_enable(False)
        _lib.lib().refil_set_overlap(0 if a.serial else -1)
        ents.sort(key=lambda e: -(iso.get(e["name"], e)["total_ms"]))       # by the kernel's own (isolated) cost
        kernels = []
        for e in ents[:max(a.top, 1)]:
            k = {"name": e["name"], "launches_per_step": e["launches"] // nprof, "ms_per_step": round(e["total_ms"] / nprof, 4),
                 "avg_us": round(1e3 * e["total_ms"] / e["launches"], 2),
                 "tflops": round(e["flops"] / (e["total_ms"] * 1e-3) / 1e12, 2) if e["flops"] > 0 else None}
            x = iso.get(e["name"])
            if x:
                k["ms_per_step_isolated"] = round(x["total_ms"] / nprof, 4)
                k["avg_us_isolated"] = round(1e3 * x["total_ms"] / x["launches"], 2)
                if x["flops"] > 0:
                    k["tflops_isolated"] = round(x["flops"] / (x["total_ms"] * 1e-3) / 1e12, 2)
            kernels.append(k)
        iso_total = sum(x["total_ms"] for x in iso.values()) / nprof
        dom = next((e for e in ents if e["flops"] > 0), ents[0])            # dominant kernel (largest isolated time per step)
        dom_iso = iso.get(dom["name"], dom)
        traffic, hbm_step, traffic_src, tr = None, None, "not collected in this run", None
        if a.traffic_json:
            tj = json.load(open(a.traffic_json))
            traffic = tj["kernels"].get(dom["name"], {}).get("hbm_bytes_per_launch")
            hbm_step, traffic_src = tj.get("hbm_bytes_per_step_all_kernels"), "--traffic-json"
        elif not a.no_traffic and world == 1 and rank == 0:
            cfg_argv = ["--config", a.config] + (["--batch", str(a.batch)] if a.batch else []) + (["--dense-data"] if a.dense_data else [])
            tr = collect_traffic(cfg_argv, tuning=tuned)
            if tr:
                traffic, hbm_step, traffic_src = tr["per_launch"].get(dom["name"]), tr["hbm_bytes_per_step"], "rocprofv3 PMC passes run by bench.py"
        # what the launches of one step execute: the GEMM scopes report the listed rows (device counts); the attention scopes
        # report every (b,t) row and skip the finished steps inside the kernel -- scaled by the live-step fraction here
        live_frac = rows["live_steps"] / max(rows["steps"], 1) if rows["lists"] else 1.0
        executed_flops = sum(x["flops"] * (live_frac if n.startswith("attn_") else 1.0) for n, x in iso.items()) / nprof
        # Both roofs per kernel (see the module docstring): the matrix pipe(s) the kernel issues on and HBM with the measured bytes.
        # Row-list GEMMs report the FLOPs / bytes of the rows they process; the attention launches report every (b,t) row and skip
        # the finished steps inside the kernel: scaled by the live-step fraction here.
        per_launch_pmc = tr["per_launch"] if tr else (
            {k: v.get("hbm_bytes_per_launch") for k, v in json.load(open(a.traffic_json))["kernels"].items()} if a.traffic_json else {})

        def attn_bytes_scale(name):
            # attention launches: the profiler's bytes are the dense (all rows) operand + result sizes. The kernels move data for the
            # K / V rows and the query rows that can influence the loss only; one symbol covers the agent launch (1 net) and the
            # hypernet launch (4 nets): K/V-side bytes scale with the listed entity rows, Q-side bytes with the active agent rows
            if not name.startswith("attn_"):
                return 1.0
            if not rows["lists"]:
                return rows["live_steps"] / max(rows["steps"], 1)
            f_e = (rows["entity_rows_agent"] + 4.0 * rows["entity_rows_hyper"]) / (5.0 * max(rows["entity_rows"], 1))
            f_a = rows["agent_rows"] / max(rows["all_agent_rows"], 1)
            w_e, w_a = (4.0 * dims["ne"], 3.0 * dims["na"]) if "bwd" in name else (2.0 * dims["ne"], 2.0 * dims["na"])
            return (w_e * f_e + w_a * f_a) / (w_e + w_a)

        def price(name, x):
            """roofs of kernel `name` from its isolated profile entry x: seconds per launch it cannot beat on the matrix pipes / on HBM"""
            t = 1e-3 * x["total_ms"] / x["launches"]
            fsc = live_frac if name.startswith("attn_") else 1.0
            fl, fs = fsc * x["flops"] / x["launches"], fsc * x.get("flops_bf16x6", 0.0) / x["launches"]
            t_mfma = fs / (PEAK_SPLIT_TFLOPS * 1e12) + (fl - fs) / (PEAK_FP32_MFMA_TFLOPS * 1e12)
            alg_bytes = attn_bytes_scale(name) * x["bytes"] / x["launches"]
            pmc = per_launch_pmc.get(name)
            t_hbm = (pmc if pmc else alg_bytes) / (PEAK_HBM_GBS * 1e9)
            return {"t": t, "flops": fl, "flops_bf16x6": fs, "t_mfma": t_mfma, "t_hbm": t_hbm, "alg_bytes": alg_bytes, "pmc_bytes": pmc,
                    "bound": "hbm" if t_hbm > t_mfma else "mfma"}

        for k in kernels:
            x = iso.get(k["name"])
            if x and x["launches"]:
                pr = price(k["name"], x)
                k["bound"] = pr["bound"]
                k["mfma_frac"] = round(pr["t_mfma"] / pr["t"], 4)
                k["hbm_frac"] = round(pr["t_hbm"] / pr["t"], 4)
                k["hbm_bytes_per_launch"] = pr["pmc_bytes"] if pr["pmc_bytes"] else None
                if k["name"].startswith("attn_qkv") and rows["lists"]:
                    # credited FLOPs include the dead entity slots of the 16-entity tiles: the fraction on rows that matter, and the roof
                    # fraction counted on those alone
                    u = qkv_useful_fraction(rows, dims["ne"], dims["na"])
                    k["useful_flops_frac"] = round(u, 4)
                    k["useful_flops_per_launch"] = round(pr["flops"] * u)
                    k["mfma_frac_useful"] = round(pr["t_mfma"] * u / pr["t"], 4)
        pd = price(dom["name"], dom_iso)
        t_situ = 1e-3 * dom["total_ms"] / dom["launches"]
        hbm_bound = pd["bound"] == "hbm"
        if hbm_bound:
            ach_iso, ach_situ = pd["alg_bytes"] / pd["t"] / 1e9, pd["alg_bytes"] / t_situ / 1e9
            peak, unit = PEAK_HBM_GBS, "GB/s"
        else:
            # the roof of the mix of matrix instructions this kernel issues, in TFLOP/s of algorithmic fp32 work
            peak = pd["flops"] / pd["t_mfma"] / 1e12 if pd["t_mfma"] > 0 else PEAK_FP32_MFMA_TFLOPS
            ach_iso, ach_situ = pd["flops"] / pd["t"] / 1e12, pd["flops"] / t_situ / 1e12
            unit = "TFLOP/s"
        live = attn_bytes_scale(dom["name"])
        # the projection kernel that executes the most FLOPs per step as well (matrix-core evidence when the dominant kernel is HBM-bound)
        gemm = max((e for e in ents if e["name"].startswith("gemm_") and e["flops"] > 0), key=lambda e: iso.get(e["name"], e)["flops"], default=None)
        gemm_iso = iso.get(gemm["name"], gemm) if gemm else None
        pg = price(gemm["name"], gemm_iso) if gemm else None
        # matrix-pipe floor of the whole step: every kernel's FLOPs priced on the pipe it issues them on
        step_mfma_floor = sum(price(n, x)["t_mfma"] * x["launches"] for n, x in iso.items() if x["launches"]) / nprof
        roofline = {"kernel": dom["name"], "bound": pd["bound"], "achieved": round(ach_iso, 2), "peak": round(peak, 1),
                    "unit": unit, "frac": round(max(pd["t_mfma"], pd["t_hbm"]) / pd["t"], 4), "traffic": traffic,
                    "mfma_frac": round(pd["t_mfma"] / pd["t"], 4), "hbm_frac": round(pd["t_hbm"] / pd["t"], 4),
                    "frac_vs_fp32_instruction": round(pd["flops"] / pd["t"] / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                    # the fused in_trans + attention launch is credited with whole 16-entity tiles; on the entity rows that can influence the loss:
                    "useful_flops_frac": round(qkv_useful_fraction(rows, dims["ne"], dims["na"]), 4) if dom["name"].startswith("attn_qkv") and rows["lists"] else None,
                    "frac_useful": round(max(pd["t_mfma"] * qkv_useful_fraction(rows, dims["ne"], dims["na"]), pd["t_hbm"]) / pd["t"], 4)
                    if dom["name"].startswith("attn_qkv") and rows["lists"] else None,
                    "roofs": {"fp32_mfma_tflops": PEAK_FP32_MFMA_TFLOPS, "bf16x6_tflops_of_fp32_work": round(PEAK_SPLIT_TFLOPS, 1), "hbm_gbs": PEAK_HBM_GBS,
                              "flops_per_launch": pd["flops"], "flops_per_launch_on_bf16x6": pd["flops_bf16x6"],
                              "mfma_floor_us": round(pd["t_mfma"] * 1e6, 2), "hbm_floor_us": round(pd["t_hbm"] * 1e6, 2),
                              "hbm_bytes_used": "rocprofv3 PMC" if pd["pmc_bytes"] else "algorithmic (no PMC pass in this run)",
                              "note": "bound = the larger floor; frac = that floor / the measured launch time. fp32 products issued as six bf16 "
                                      "matrix-pipe products of an exact 3-way operand split (fp32-accurate: tests/test_gpu_ops.py::test_wres_split_accuracy) "
                                      "are priced at the dense bf16 peak / 6, everything else at the fp32 matrix instruction's peak; "
                                      "frac_vs_fp32_instruction = the rounds 1-4 figure, which this form may exceed"},
                    "traffic_by_kernel_mb_per_step": None if not tr else {
                        k: round(v * tr["launches"].get(k, 0) / tr["steps"] / 1e6, 1) for k, v in sorted(
                            tr["per_launch"].items(), key=lambda kv: -kv[1] * tr["launches"].get(kv[0], 0))[:24]
                        if "<" not in k or k in {x["name"] for x in ents} or k.split("<")[0] not in {x["name"].split("<")[0] for x in ents}},
                    "traffic_unit": f"HBM bytes/launch (rocprofv3 PMC 2*FETCH_SIZE+WRITE_SIZE, {traffic_src})",
                    "hbm_gb_per_step": None if hbm_step is None else round(hbm_step / 1e9, 3),
                    "hbm_gb_per_s_over_step": None if hbm_step is None else round(hbm_step / (ms_per_step * 1e-3) / 1e9, 1),
                    "executed_gflop_per_step": round(executed_flops / 1e9, 2),
                    "step_frac_executed": round(executed_flops / (ms_per_step * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                    "mfma_floor_ms_executed": round(executed_flops / (PEAK_FP32_MFMA_TFLOPS * 1e12) * 1e3, 3),
                    "algorithmic_bytes_per_launch": round(live * dom_iso["bytes"] / dom_iso["launches"]),
                    "achieved_in_situ": round(ach_situ, 2), "frac_in_situ": round(ach_situ / peak, 4),
                    "heaviest_gemm": None if not gemm else {
                        "kernel": gemm["name"], "bound": pg["bound"], "launches_per_step": gemm["launches"] // nprof,
                        "achieved_tflops": round(pg["flops"] / pg["t"] / 1e12, 2), "frac": round(max(pg["t_mfma"], pg["t_hbm"]) / pg["t"], 4),
                        "mfma_frac": round(pg["t_mfma"] / pg["t"], 4), "hbm_frac": round(pg["t_hbm"] / pg["t"], 4),
                        "frac_vs_fp32_instruction": round(pg["flops"] / pg["t"] / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                        "avg_launch_us": round(pg["t"] * 1e6, 2)},
                    "step_mfma_floor_ms": round(step_mfma_floor * 1e3, 3),
                    "step_frac_of_matrix_roof": round(step_mfma_floor / (ms_per_step * 1e-3), 4),
                    "step_frac_of_hbm_roof": None if hbm_step is None else round(hbm_step / (PEAK_HBM_GBS * 1e9) / (ms_per_step * 1e-3), 4),
                    "launches_per_step": dom["launches"] // nprof,
                    "avg_launch_us": round(1e3 * dom_iso["total_ms"] / dom_iso["launches"], 2),
                    "avg_launch_us_in_situ": round(1e3 * dom["total_ms"] / dom["launches"], 2),
                    "flops_per_launch": dom_iso["flops"] / dom_iso["launches"],
                    "share_of_gpu_time": round(dom_iso["total_ms"] / (iso_total * nprof), 3),
                    "gpu_ms_per_step_all_kernels_in_situ": round(tot / nprof, 3),
                    "gpu_ms_per_step_all_kernels": round(iso_total, 3),
                    "step_frac_dense_equivalent": round(flops_step / (ms_per_step * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                    "streams": "serialised (--serial)" if a.serial else "agent and hypernet chains overlap on two streams",
                    "measured": f"HIP events on the launch streams, {nprof} steps right after the timed region. `achieved` / `avg_launch_us`: "
                                "the streams serialised, each kernel alone on the GPU = the duration rocprofv3 --kernel-trace reports for it "
                                "(profiles/README.md); FLOPs = what the launch executes (row-list launches: the listed rows, read back "
                                "from the device). `*_in_situ`: both chains overlapping as in the timed region -- a launch shares the GPU "
                                "with the other stream's kernels, a lower bound on kernel quality. `step_frac_executed` = FLOPs the step's launches "
                                "EXECUTE (profiler, listed rows) / step time / peak -- the honest step-level fraction; `step_frac_dense_equivalent` = the "
                                "DENSE algorithmic FLOPs of SURVEY.md section 8d / step time / peak (work the row lists and the algebra of DESIGN.md "
                                "section 5 remove still counts there: a speed-up figure against the dense schedule, not a utilisation). "
                                "`step_frac_of_matrix_roof` = the executed FLOPs priced on the pipes they are issued on (bf16 x 6 products at 416.7, the rest at "
                                "157.3 TFLOP/s) / step time; `step_frac_of_hbm_roof` = PMC bytes per step / 8 TB/s / step time"}
    if world > 1:
        dist.barrier()

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(dims, W["imagine"], data)

    if rank == 0:
        nE, nA = rows["entity_rows"], rows["all_agent_rows"]
        out = {
            "metric": "learner transitions/sec (B x T per full QLearner.train step)", "value": round(value, 1),
            "unit": "transitions/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 3), "median_ms_per_step": round(statistics.median(per_step), 3),
            "min_ms_per_step": round(min(per_step), 3), "loss_step0": loss_step0, "gpu_state": gpu_state(local_rank),
            "host_enqueue_ms_per_step": round(host_enqueue / a.steps * 1e3, 3), "higher_is_better": True, "scaling": a.scaling,
            "filled_transitions_per_s": round(mask_sum * a.steps / elapsed, 1),
            "dense_data": dense, "comm": comm,
            "rank_ms_per_step": {"min": round(min(rank_ms), 3), "max": round(max(rank_ms), 3)},
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "dtype_note": ("fp32 operands, fp32 accumulation, fp32 results everywhere (the reference's arithmetic). Products of the projections: "
                           + ("six bf16 matrix-pipe products of an EXACT 3-way split of both fp32 operands (the three dropped piece products are below "
                              "2^-26 of the product), error against fp64 equal to the fp32 matrix instruction's" if os.environ.get("REFIL_WRES_SPLIT", "6") != "0"
                              else "v_mfma_f32_32x32x2_f32")
                           + "; large weight gradients: " + ("the same bf16 x 6 form" if os.environ.get("REFIL_DW_SPLIT", "6") != "0" else "v_mfma_f32_32x32x2_f32")
                           + " (REFIL_WRES_SPLIT=0 REFIL_DW_SPLIT=0: the fp32 instruction everywhere; both forms are compared with the oracle at every "
                             "production shape, tests/test_gpu_learner.py::PRODUCTION)"),
            "config": {"workload": f"{a.config}: synthetic replay (B={B}/GPU, T={T}, n_entities={dims['ne']}, n_agents={dims['na']}, "
                                   f"d={dims['d']}, hypernet={dims['h']}), {'refil' if W['imagine'] else 'qmix_atten'} learner "
                                   f"({'imagine agent' if W['imagine'] else 'entity_attend_rnn agent'} + flex_qmix), {W['what']}",
                       "global_batch": global_B, "seq_len": T, "parallelism": f"dp{world}",
                       "world_size": dist.get_world_size() if world > 1 else 1, "backend": backend,
                       "algorithmic_gflop_per_step_per_gpu": round(flops_step / 1e9, 2),
                       "batches": f"{a.fresh_batches} x B episodes in a {'pinned-HOST (PCIe-inclusive: the gather reads host memory)' if a.host_buffer else 'device'} ReplayBuffer, a fresh sample(B) per step inside the timed region" if a.fresh_batches
                                  else "one resident minibatch (every step trains on the same episodes)",
                       "padding": "none (--dense-data: every entity alive, full-length episodes)" if a.dense_data else "SC2-law padding / deaths / ragged episode ends",
                       "schedule_autotune": {"chosen": tuned or "built-in defaults", "parity_tested_values": {k: list(v) for k, v in tuning.PARITY_TESTED.items()},
                                             "measured_ms (knob, value, default, candidate)": getattr(learner, "_autotune_log", None),
                                             "note": "bench.py sets REFIL_AUTOTUNE=1: QLearner's first train() call on a shape bucket measures launch-size knobs in situ "
                                                     "(before the warm-up steps), restricted to values with parity coverage (refil_amd/tuning.py; a setting outside "
                                                     "it is refused); the library default is the deterministic built-in schedule"}},
            "rows": {"lists_active": bool(rows["lists"]), "live_step_frac": round(rows["live_steps"] / max(rows["steps"], 1), 4),
                     "entity_rows_frac_agent_nets": round(rows["entity_rows_agent"] / max(nE, 1), 4),
                     "entity_rows_frac_hypernets": round(rows["entity_rows_hyper"] / max(nE, 1), 4),
                     "agent_query_rows_frac": round(rows["agent_rows"] / max(nA, 1), 4),
                     "note": "fractions of the dense (b,t,entity) rows the step processes: rows that cannot influence the loss are "
                             "skipped on the device, results equal the dense schedule (tests/test_gpu_learner.py::test_row_skipping_equals_dense_schedule)"},
            "roofline": roofline, "cpu_baseline": cpu, "kernels": kernels,
        }
        if cpu:
            out["gpu_over_cpu"] = round(value / cpu["value"], 1)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
