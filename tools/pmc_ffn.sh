#!/bin/bash
# PMC pass over the feed-forward micro-benchmark: tools/pmc_ffn.sh <tag> <counters...>
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT -- python3 $ROOT/tools/bench_ffn.py 32768 > $OUT/log.txt 2>&1
echo rc=$?
