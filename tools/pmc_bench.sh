#!/bin/bash
# one PMC pass over a short bench run: tools/pmc_bench.sh <tag> <counters...>
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmcb_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-tap > $OUT/log.txt 2>&1
echo rc=$?
