"""What one learner step EXECUTES, kernel by kernel, counted on the CPU wavefront emulator (tests/emu, EMU_COUNT=1): wave-level matrix
instructions by type and the bytes moved by the raw buffer instructions -- and the matrix-pipe issue floor those instructions imply.

    python tools/emu_counts.py [PRODUCTION case of tests/test_gpu_learner.py, default cfgT] [--wide]

Why. bench.py prices a kernel with the FLOPs its launcher CREDITS it with (closed forms over the rows a launch walks). The fused
in_trans + attention launch computes whole 16-entity tiles, so its credit includes entity slots that are dead; the
row-list GEMMs pad their last tile; the recurrences run 4-row tiles. The emulator executes the kernel source instruction by
instruction, so the counts below are what the matrix pipe is actually asked to do on the SC2-law batch of the bench -- no closed form.

Issue cost per wave-level instruction on one SIMD (MI355X_MICROARCH.md, "MFMA issue"; 4x4x1_16b: profiles/r05_gru_timing.txt, ~20 cycles back
to back from one wave): the floor is sum(count x cycles) / (1024 SIMDs x clock) -- what the step's matrix work costs if every SIMD issued
matrix instructions back to back with nothing else in the way. It is a static figure (no GPU was available in round 6): a launch bound
by latency, HBM or its VALU work sits above it.  NOT a measurement.
"""
import ctypes
import os
import sys
import time

os.environ["EMU_COUNT"] = "1"
# the library sizes grids and picks kernel variants (4-row / 16-row recurrence tiles, workgroups per weight-gradient launch) from the device's
# CU count: emulate the MI355X's 256. The one-launch row lists would then need 256 co-resident spinning workgroups -- more than the
# emulator's thread pool: the four-launch form builds the same lists.
os.environ.setdefault("EMU_CUS", "256")
os.environ.setdefault("REFIL_LISTS_FUSED", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

NAMES = ["mfma_32x32x2_f32", "mfma_16x16x4_f32", "mfma_4x4x1_f32", "mfma_16x16x32_bf16", "mfma_32x32x16_bf16", "buf_load_bytes", "buf_store_bytes"]
CYCLES = {"mfma_32x32x2_f32": 64, "mfma_16x16x4_f32": 32, "mfma_4x4x1_f32": 20, "mfma_16x16x32_bf16": 17, "mfma_32x32x16_bf16": 32}
# fp32-equivalent FLOPs of one wave-level instruction (a bf16 instruction of the 3-way split carries 1/6 of an fp32 product)
FLOPS = {"mfma_32x32x2_f32": 2 * 32 * 32 * 2, "mfma_16x16x4_f32": 2 * 16 * 16 * 4, "mfma_4x4x1_f32": 2 * 16 * 4 * 4,
         "mfma_16x16x32_bf16": 2 * 16 * 16 * 32 / 6.0, "mfma_32x32x16_bf16": 2 * 32 * 32 * 16 / 6.0}
SIMDS, CLOCK_GHZ = 1024, 2.4


def main():
    import emu_util
    which = next((a for a in sys.argv[1:] if not a.startswith("--")), "cfgT")
    L = emu_util.load_copy("test_gpu_learner", DEV="cpu")
    kw = dict(L.PRODUCTION[which])
    tuned = dict(kw.get("tuned") or {})
    if "--wide" in sys.argv:
        tuned["attn_qkv_wide"] = 1
    with emu_util.active() as lib:
        cfg, batch, bits, agent, mixer, tagent, tmixer = L._oracle_case(
            kw["B"], kw["T"], kw["ne"], seed=40 + kw["B"], imagine=kw["imagine"], d=kw["d"], h=kw["d"], H=kw.get("H", 64), na=kw.get("na"),
            A=kw.get("A"), gm=kw.get("gm", False))
        from refil_amd import _lib
        t0 = time.time()
        # profile=True: the library's own profiler -- the FLOPs each launch is CREDITED with (bench.py's source)
        r = L.run_hip_step(cfg, batch, bits, agent, mixer, tagent, tmixer, tuned=tuned or None, profile=True)
        credited = {e["name"]: e for e in r["profile"]}
        dt = time.time() - t0
        from golden_util import live_steps
        live = live_steps(batch)
        live_frac = float(live.float().mean())
        n = lib.emu_counters_dump(None, 0, 0)
        buf = ctypes.create_string_buffer(n)
        lib.emu_counters_dump(buf, n, 1)
    rows = []
    for line in buf.value.decode().splitlines():
        f = line.split("\t")
        c = dict(zip(NAMES, (int(x) for x in f[3:])))
        cyc = sum(c[k] * CYCLES[k] for k in CYCLES)
        fl = sum(c[k] * FLOPS[k] for k in FLOPS)
        rows.append((f[0], int(f[1]), int(f[2]), c, cyc, fl))
    rows.sort(key=lambda r: -r[4])
    tot_cyc = sum(r[4] for r in rows)
    tot_fl = sum(r[5] for r in rows)
    print(f"# {which}{' +attn_qkv_wide' if '--wide' in sys.argv else ''}: B={kw['B']} T={kw['T']} ne={kw['ne']} d={kw['d']}  (one forward_backward + clip_rmsprop on the emulator, {dt:.0f} s; "
          f"emulated device: {os.environ.get('EMU_CUS')} CUs)")
    print(f"# matrix-pipe issue floor = sum(count x issue cycles) / ({SIMDS} SIMDs x {CLOCK_GHZ} GHz); issue cycles per SIMD: {CYCLES}")
    print(f"{'kernel':78s} {'launches':>8s} {'32x32x2f32':>11s} {'16x16x4f32':>11s} {'4x4x1f32':>10s} {'16x16x32bf16':>13s} {'32x32x16bf16':>13s} "
          f"{'buf ld MB':>10s} {'buf st MB':>10s} {'GFLOP(fp32 eq)':>14s} {'floor us':>9s} {'share':>6s}")
    for name, launches, wgs, c, cyc, fl in rows:
        if cyc == 0 and c["buf_load_bytes"] == 0 and c["buf_store_bytes"] == 0:
            continue
        us = cyc / (SIMDS * CLOCK_GHZ * 1e3)
        print(f"{name[:78]:78s} {launches:8d} {c['mfma_32x32x2_f32']:11d} {c['mfma_16x16x4_f32']:11d} {c['mfma_4x4x1_f32']:10d} {c['mfma_16x16x32_bf16']:13d} "
              f"{c['mfma_32x32x16_bf16']:13d} {c['buf_load_bytes'] / 1e6:10.1f} {c['buf_store_bytes'] / 1e6:10.1f} {fl / 1e9:14.2f} {us:9.1f} {cyc / max(tot_cyc, 1):6.1%}")
    print(f"{'TOTAL':78s} {'':8s} {'':11s} {'':11s} {'':10s} {'':13s} {'':13s} {'':10s} {'':10s} {tot_fl / 1e9:14.2f} {tot_cyc / (SIMDS * CLOCK_GHZ * 1e3):9.1f}")
    # credited (the launchers' closed forms, what bench.py prices a kernel with) against executed (counted above), per kernel family.
    # The attention scopes credit every (b,t) row and bench.py scales them by the live-step fraction: done here as well.
    fam = {}
    for name, launches, wgs, c, cyc, fl in rows:
        fam.setdefault(name.split("<")[0], [0.0, 0])[0] += fl
    print(f"\n# credited vs executed GFLOP per step (fp32-equivalent), live-step fraction {live_frac:.3f}")
    print(f"{'profiler scope':40s} {'launches':>8s} {'credited':>10s} {'executed':>10s} {'executed/credited':>18s}")
    alias = {"gru_fwd_kernel": ("gru_fwd4_kernel", "gru_fwd16_kernel"), "gru_bwd_kernel": ("gru_bwd4_kernel", "gru_bwd16_kernel"),
             "attn_fwd_mfma": ("attn_fwd_pipe",), "attn_bwd_mfma": ("attn_bwd_pipe",)}
    seen = set()
    for pname, e in sorted(credited.items(), key=lambda kv: -kv[1]["flops"]):
        if e["flops"] <= 0:
            continue
        base = pname.split("<")[0]
        if base in seen:
            continue
        cred = sum(x["flops"] for n2, x in credited.items() if n2.split("<")[0] == base)
        nl = sum(x["launches"] for n2, x in credited.items() if n2.split("<")[0] == base)
        if base.startswith("attn_"):
            cred *= live_frac
        ex = sum(fam.get(k, [0.0])[0] for k in (base,) + alias.get(base, ()))
        seen.add(base)
        print(f"{base:40s} {nl:8d} {cred / 1e9:10.2f} {ex / 1e9:10.2f} {ex / cred if cred else float('nan'):18.3f}")


if __name__ == "__main__":
    main()
