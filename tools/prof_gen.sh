#!/bin/bash
# Per-kernel times of the greedy-decode micro-bench (tools/bench_gen.py): tools/prof_gen.sh   (run on the GPU box)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pg
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg -- python3 $ROOT/tools/bench_gen.py 64 1 > /tmp/pg.log 2>&1 || { tail -5 /tmp/pg.log; exit 1; }
grep positions /tmp/pg.log
python3 - <<'PY'
import csv, glob, os
f = max(glob.glob('/tmp/pg/*/*kernel_stats.csv'), key=os.path.getmtime)
for r in list(csv.DictReader(open(f)))[:14]:
    print('   %-70s %5s x %6.1f us  %5.1f %%' % (r['Name'].replace('(anonymous namespace)::', '')[:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
