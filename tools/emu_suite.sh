#!/bin/bash
# Everything round 6 keeps under profiles/ from the CPU wavefront emulator (tests/emu; no GPU needed). ~3-4 h on 8 cores, most of it the
# PRODUCTION matrix.   bash tools/emu_suite.sh [TAG]      (single parts: PART=ops|learner|production|counts|fuzz bash tools/emu_suite.sh)
TAG=${1:-r06}
OUT=profiles
HEAD=$(git rev-parse HEAD 2>/dev/null)
hdr() { echo "HEAD $HEAD  $(date -u +%FT%TZ)  $1"; }
run() { python -m pytest "$@" -q -W ignore -p no:cacheprovider 2>&1 | grep -v "^$"; }
PART=${PART:-all}
cat > /tmp/test_emu_prod.py <<'PY'
import sys, pytest
sys.path.insert(0, "tests")
import emu_util
_G = emu_util.load_copy("test_gpu_learner", DEV="cpu")
@pytest.fixture(autouse=True)
def _emu():
    with emu_util.active():
        yield
test_production_size_step_matches_oracle = _G.test_production_size_step_matches_oracle
PY
if [ $PART = all ] || [ $PART = ops ]; then
  { hdr "CPU tier incl. the emulator tests: python -m pytest tests -q -m 'not gpu'"; run tests -m "not gpu" --durations=8; } > $OUT/${TAG}_emu_cpu_tier.txt
fi
if [ $PART = all ] || [ $PART = learner ]; then
  { hdr "REFIL_EMU_FULL=1 python -m pytest tests/test_emu_learner.py tests/test_emu_plugin.py tests/test_emu_early.py tests/test_emu_run_loop.py  (every learner-step test of the gpu tier below production size, the whole plugin suite, 40 iterations of the run loop)"; REFIL_EMU_FULL=1 run tests/test_emu_learner.py tests/test_emu_plugin.py tests/test_emu_early.py tests/test_emu_run_loop.py --durations=8; } > $OUT/${TAG}_emu_learner_full.txt
fi
if [ $PART = all ] || [ $PART = production ]; then
  { hdr "tests/test_gpu_learner.py::test_production_size_step_matches_oracle on the emulator: every PRODUCTION case at full size vs the oracle"; run /tmp/test_emu_prod.py --rootdir=. --timeout 6000 --durations=0 -rA | grep -v "^PASSED"; } > $OUT/${TAG}_emu_production.txt
fi
if [ $PART = all ] || [ $PART = counts ]; then
  for c in cfgT cfg2 cfg5_ne48_mmm_law; do python tools/emu_counts.py $c > $OUT/${TAG}_emu_counts_$c.txt 2>/dev/null; done
  python tools/emu_counts.py cfg5_ne48_mmm_law --wide > $OUT/${TAG}_emu_counts_cfg5_ne48_mmm_law_qkvwide.txt 2>/dev/null
fi
if [ $PART = all ] || [ $PART = fuzz ]; then
  { hdr "fuzzers with fresh seeds on the emulator (REFIL_FUZZ_SEED=60606, REFIL_FUZZ_QKV_SEED=70707)"
    REFIL_FUZZ_QKV_N=300 REFIL_FUZZ_QKV_SEED=70707 REFIL_FUZZ_ATTN_N=200 run tests/test_emu_ops.py -k "random_shapes"
    REFIL_EMU_FULL=1 REFIL_FUZZ_SEED=60606 REFIL_FUZZ_N=100 REFIL_FUZZ_VAR_N=60 REFIL_FUZZ_LIST_N=24 REFIL_FUZZ_ACT_N=30 run tests/test_emu_learner.py -k "random_" --timeout 1200; } > $OUT/${TAG}_emu_fuzz.txt
fi
