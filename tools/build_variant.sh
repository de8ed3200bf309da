#!/bin/bash
# A/B builds of ONE kernel source: tools/build_variant.sh <name> <source.hip> [extra hipcc flags...] -> tools/_build/libadv_<name>.so (+ _hooks),
# linked with the other objects of the regular build (eval_driving_safety_amd/csrc/_build/*.o).  Select with ADVENGINE_LIB=... (tests / tools only).
set -eu
NAME=$1; SRC=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/eval_driving_safety_amd/csrc
O=$R/tools/_build
mkdir -p $O
BASE=$(basename $SRC .hip)
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -fvisibility=hidden -I$R/include -I$C -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS "$@" -c -o $O/$BASE.$NAME.o $C/$SRC &
/opt/rocm/bin/hipcc $FLAGS -DADV_TEST_HOOKS "$@" -c -o $O/$BASE.$NAME.hooks.o $C/$SRC &
wait
OBJS=$(ls $C/_build/*.o | grep -v hooks | grep -v "/$BASE.o")
HOBJS=$(ls $C/_build/*.hooks.o | grep -v "/$BASE.hooks.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libadv_$NAME.so $OBJS $O/$BASE.$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libadv_${NAME}_hooks.so $HOBJS $O/$BASE.$NAME.hooks.o
ls -la $O/libadv_$NAME.so $O/libadv_${NAME}_hooks.so
