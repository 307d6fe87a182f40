#!/bin/bash
# A second build of the library that differs in ONE source file (A/B runs on one GPU box through REFIL_LIB_PATH):
#   tools/build_variant.sh NAME csrc-file.hip "-DFLAG=1 ..." [replaced-object-basename]
# compiles the file with the flags and links it with the objects of the regular build (python -m refil_amd.build first)
# into refil_amd/variants/librefil_NAME.so
set -e
cd "$(dirname "$0")/.."
NAME=$1; SRC=$2; FLAGS=$3; REPL=${4:-$(basename "$SRC" .hip)}
mkdir -p refil_amd/variants refil_amd/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -c "$SRC" -o refil_amd/build/variant_$NAME.o
OBJS=$(ls refil_amd/build/*.o | grep -v "/variant_" | grep -v "/$REPL.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o refil_amd/variants/librefil_$NAME.so $OBJS refil_amd/build/variant_$NAME.o -ldl
echo "built refil_amd/variants/librefil_$NAME.so"
