#!/bin/bash
# Per-workgroup start / end times of the feed-forward kernels inside a training step.
# Here (CPU):   tools/wglog.sh build      -> tools/exp/libwglog.so (ffn.hip, ffn_wgrad.hip, attn.hip and gemm.hip with -DMMST_WG_LOG)
# On the GPU:   MMST_LIB=$GRAFT_REPO_ROOT/tools/exp/libwglog.so python tools/wglog.py
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/multimodalspectraltransformer_amd/csrc
mkdir -p $ROOT/tools/exp
if [ "$1" = build ]; then
  OBJS=""
  for f in ffn ffn_wgrad attn gemm; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DMMST_WG_LOG -c $CS/$f.hip -o $ROOT/tools/exp/$f.wglog.o
  done
  for o in $CS/_build/*.o; do
    b=$(basename $o .o)
    if [ $b = ffn ] || [ $b = ffn_wgrad ] || [ $b = attn ] || [ $b = gemm ]; then OBJS="$OBJS $ROOT/tools/exp/$b.wglog.o"; else OBJS="$OBJS $o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -shared -o $ROOT/tools/exp/libwglog.so $OBJS
  rm -f $ROOT/tools/exp/*.wglog.o
  echo built tools/exp/libwglog.so
fi
