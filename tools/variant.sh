#!/bin/bash
# Build an experimental variant of the library: tools/variant.sh <name> <source.hip> [-DFLAG ...]
# Recompiles one source with extra flags, links it with the in-tree objects of the others into tools/exp/lib<name>.so
# (select it at run time with MMST_LIB=tools/exp/lib<name>.so).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; SRC=$2; shift 2
CS=$ROOT/multimodalspectraltransformer_amd/csrc
mkdir -p $ROOT/tools/exp
BASE=$(basename $SRC .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $CS/$BASE.hip -o $ROOT/tools/exp/$BASE.$NAME.o
OBJS=""
for o in $CS/_build/*.o; do
  if [ "$(basename $o .o)" = "$BASE" ]; then OBJS="$OBJS $ROOT/tools/exp/$BASE.$NAME.o"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -shared -o $ROOT/tools/exp/lib$NAME.so $OBJS
rm -f $ROOT/tools/exp/$BASE.$NAME.o
echo built tools/exp/lib$NAME.so
