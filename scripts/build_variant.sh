#!/bin/bash
# A variant of the library for A/B runs in one GPU session (scripts/ab_libs.py, MAG_LIB_PATH): persist.hip recompiled with
# extra -D flags, linked against the objects of the product build.  (The variant is ONE translation unit: it does not define
# MAG_PERSIST_SPLIT_K4, so the four-slot structured instantiation, which the product compiles separately under max-ilp --
# persist_k4.o --, is compiled here with the variant's flags and scheduler like every other instantiation.)
#   bash scripts/build_variant.sh <name> "<-D flags>" [scheduler] [file]     ->  magnetite_amd/ab/libmagnetite_hip_<name>.so
# file: persist.hip (default) or exact.hip (compiled as the Makefile does: -ffp-contract=off, default scheduler)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1
FLAGS=$2
SCHED=${3:-iterative-ilp}
FILE=${4:-persist.hip}
cd "$ROOT/magnetite_amd/csrc"
make -s all
mkdir -p ../ab build
OBJS="build/primitives.o build/symbolic.o build/exact.o build/cg.o build/persist.o build/persist_k4.o build/api.o build/comm.o"
if [ "$FILE" = "exact.hip" ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I../../include -I. -Wall -Wno-unused-result $FLAGS \
        -ffp-contract=off -c exact.hip -o build/exact_$NAME.o
    OBJS=${OBJS/build\/exact.o/build\/exact_$NAME.o}
else
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I../../include -I. -Wall -Wno-unused-result $FLAGS \
        -ffp-contract=off -mllvm -amdgpu-sched-strategy=$SCHED -c persist.hip -o build/persist_$NAME.o
    OBJS=${OBJS/build\/persist.o/build\/persist_$NAME.o}
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../ab/libmagnetite_hip_$NAME.so $OBJS -ldl -Wl,-rpath,/opt/rocm/lib
echo "../ab/libmagnetite_hip_$NAME.so"
