#!/bin/bash
# Interleaved A/B of two environment settings on ONE box (step times differ by ~1 % between boxes): tools/ab_env.sh "VAR=a" "VAR=b" [reps]
cd $GRAFT_REPO_ROOT
A="$1"; B="$2"; R=${3:-3}
for i in $(seq $R); do
  for cfg in "$A" "$B"; do
    v=$(env $cfg timeout -k 10 150 python bench.py --no-cpu-baseline --no-tap 2>/dev/null | tail -1 | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$cfg $v"
  done
done
