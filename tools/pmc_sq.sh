#!/bin/bash
# SQ-counter passes (matrix-pipe utilisation, instruction mix, waits, LDS conflicts, occupancy) over a short run of the
# training-step bench, one rocprofv3 --pmc run per pass of at most 8 SQ counters:   tools/pmc_sq.sh <tag>
# (run on the GPU box; summary: tools/summarize_sq.py <tag>)
TAG=${1:-r3}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${TAG}_sq
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_LEVEL_WAVES SQ_VALU_MFMA_COEXEC_CYCLES"
i=1
for P in "$P1" "$P2"; do
  timeout -s KILL 200 rocprofv3 --pmc $P --output-format csv -d $OUT/pass$i -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-tap --no-secondary --repeats 1 > $OUT/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/pass$i.log; exit 1; }
  # keep what the summary needs, per kernel: (counter, launches, sum)
  python3 - $OUT/pass$i <<'PY'
import collections, csv, glob, json, os, sys
d = sys.argv[1]
f = max(glob.glob(d + "/*/*counter_collection.csv"), key=os.path.getmtime)
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    k = (r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", ""), r["Counter_Name"])
    agg[k][0] += 1
    agg[k][1] += float(r["Counter_Value"])
out = collections.defaultdict(dict)
for (kern, c), (n, s) in agg.items():
    out[kern][c] = {"launches": n, "sum": s}
json.dump(out, open(d + ".json", "w"), indent=0, sort_keys=True)
PY
  rm -rf $OUT/pass$i
  i=$((i+1))
done
ls -la $OUT
