#!/bin/bash
# Kernel trace of training steps with in-training generation (greedy): per-queue timeline of one step.
#   tools/trace_gen.sh [ENV=VALUE ...]      (run on the GPU box)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tg
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tg -- python3 $ROOT/bench.py --gen --steps 4 --warmup 2 --no-cpu-baseline --no-tap --no-secondary --repeats 1 > /tmp/tg.log 2>&1 || { tail -5 /tmp/tg.log; exit 1; }
tail -1 /tmp/tg.log | cut -c1-120
python3 - <<'PY'
import csv, glob, os, collections
f = max(glob.glob('/tmp/tg/*/*kernel_trace.csv'), key=os.path.getmtime)
rows = [r for r in csv.DictReader(open(f))]
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
    r['n'] = r['Kernel_Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')[:40]
rows.sort(key=lambda r: r['s'])
starts = [i for i, r in enumerate(rows) if 'embed_fwd' in r['n'] and 'trg' not in r['n']]
a, b = starts[-2], starts[-1]
step = rows[a:b]
t0 = step[0]['s']
print("step wall %.3f ms, %d kernels" % ((rows[b]['s'] - t0) / 1e6, len(step)))
byq = collections.defaultdict(list)
for r in step:
    byq[r['Queue_Id']].append(r)
for q, rs in byq.items():
    busy = sum(r['e'] - r['s'] for r in rs)
    names = collections.Counter(r['n'][:28] for r in rs).most_common(3)
    print("queue %s: %d kernels, busy %.3f ms, first %.3f last end %.3f  %s" % (q, len(rs), busy / 1e6, (rs[0]['s'] - t0) / 1e6, (max(r['e'] for r in rs) - t0) / 1e6, names))
dec = [r for r in step if 'decode_' in r['n']]
if dec:
    # decode progress: time of every 16th position's first launch
    firsts = [r for r in dec if 'decode_stack_kernel' in r['n']]
    per = len(firsts) // 127 if len(firsts) >= 127 else 1
    marks = [(i // per, (firsts[i]['s'] - t0) / 1e6) for i in range(0, len(firsts), per * 16)]
    print("decode positions at (ms):", ["%d:%.2f" % m for m in marks])
    import statistics
    for nm in ('decode_stack_kernel<1, true>', 'decode_ffn_kernel'):
        for lo, hi, tag in ((0, 5e6, 'first 5 ms'), (8e6, 1e12, 'after 8 ms')):
            d = [r['e'] - r['s'] for r in dec if nm in r['n'] and lo <= r['s'] - t0 < hi]
            if d:
                print("  %-30s %-11s n=%4d avg %.1f us" % (nm, tag, len(d), statistics.mean(d) / 1e3))
PY
