#!/bin/bash
# The emulator tests under UndefinedBehaviorSanitizer (tests/emu/build_emu.py, EMU_UBSAN=1). Reports go to stderr; the run continues.
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.ubsan_standalone-x86_64.so)
export EMU_UBSAN=1
python tests/emu/build_emu.py > /dev/null || exit 1
export LD_PRELOAD=$RT
export UBSAN_OPTIONS=print_stacktrace=0:halt_on_error=0
if [ $# -eq 0 ]; then set -- tests/test_emu_ops.py tests/test_emu_learner.py tests/test_emu_replay.py; fi
exec python -m pytest "$@" -q -W ignore -p no:cacheprovider
