#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace + HBM counter passes of the beam-search decode benchmark.
#   tools/gpu_profile_decode.sh <tag>
set -e
TAG=${1:-r2_decode}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --decode --steps 3 --warmup 1 > $OUT/trace.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --decode --steps 1 --warmup 0 > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --decode --steps 1 --warmup 0 > $OUT/pmc_write.log 2>&1
tail -1 $OUT/trace.log | cut -c1-300
