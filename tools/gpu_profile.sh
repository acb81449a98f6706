#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace (+ optional HBM counter passes) of the bench command.
#   tools/gpu_profile.sh <tag> [pmc]
set -e
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --repeats 1 > $OUT/trace.log 2>&1
if [ "$2" = "pmc" ]; then
  timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-tap --no-secondary --repeats 1 > $OUT/pmc_fetch.log 2>&1
  timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-tap --no-secondary --repeats 1 > $OUT/pmc_write.log 2>&1
fi
tail -1 $OUT/trace.log | cut -c1-300
