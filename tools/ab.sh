#!/bin/bash
# A/B of environment switches on the training-step bench: tools/ab.sh "VAR=a VAR2=b" "VAR=c" ...   (2 rounds)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/ab.log
: > $OUT
for round in 1 2; do
  for cfg in "$@"; do
    line=$(env $cfg timeout -k 10 200 python -u $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) || exit 1
    echo "$cfg: $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.3f ms/step, dominant %.0f us" % (d["ms_per_step"], d.get("roofline",{}).get("avg_us",0)))')" >> $OUT
  done
done
cat $OUT
