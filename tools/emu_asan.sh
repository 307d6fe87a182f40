#!/bin/bash
# The emulator tests under AddressSanitizer (tests/emu/build_emu.py, EMU_ASAN=1): out-of-bounds global / LDS accesses of the kernel sources.
#   bash tools/emu_asan.sh [pytest args, default: the op tests and the default learner selection]
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
export EMU_ASAN=1
python tests/emu/build_emu.py > /dev/null || exit 1
export LD_PRELOAD=$RT
export ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1:abort_on_error=1:allocator_may_return_null=1
if [ $# -eq 0 ]; then set -- tests/test_emu_ops.py tests/test_emu_learner.py tests/test_emu_replay.py; fi
exec python -m pytest "$@" -q -W ignore -p no:cacheprovider -x
