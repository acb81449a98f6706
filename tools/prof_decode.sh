#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/dec
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
MMST_DECODE_GRAPH=${MMST_DECODE_GRAPH:-0} timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/bench_decode.py 64 128 > $OUT/trace.log 2>&1
tail -2 $OUT/trace.log
