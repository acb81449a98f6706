#!/bin/bash
# Per-kernel times of the decoder micro-bench for the in-tree library and for variant builds (tools/variant.sh):
#   tools/prof_dec.sh [variant ...]     (run on the GPU box)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in base "$@"; do
  if [ "$v" = base ]; then unset MMST_LIB; else export MMST_LIB=$ROOT/tools/exp/lib$v.so; fi
  rm -rf /tmp/pd_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pd_$v -- python3 $ROOT/tools/bench_dec.py > /tmp/pd_$v.log 2>&1 || { tail -5 /tmp/pd_$v.log; exit 1; }
  echo "== $v: $(grep 'decoder' /tmp/pd_$v.log | tr '\n' ' ')"
  python3 - "$v" <<'PY'
import csv, glob, os, sys
f = max(glob.glob('/tmp/pd_%s/*/*kernel_stats.csv' % sys.argv[1]), key=os.path.getmtime)
for r in csv.DictReader(open(f)):
    if 'dec_' in r['Name'] or 'attn_q_kernel<8' in r['Name'] or 'attn_bwd' in r['Name'] or 'attn_kv' in r['Name']:
        print('   %-58s %4s x %6.1f us' % (r['Name'].replace('(anonymous namespace)::', '')[:58], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
