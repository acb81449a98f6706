"""Per-queue timeline of the last full training step in a rocprofv3 kernel trace: tools/timeline.py <tag> [n]"""
import csv, glob, os, sys, collections
tag = sys.argv[1]
f = max(glob.glob(f'/root/repo/gpurun_out/{tag}/trace/*/*kernel_trace.csv'), key=os.path.getmtime)
rows = [r for r in csv.DictReader(open(f))]
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
    r['n'] = r['Kernel_Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')[:34]
rows.sort(key=lambda r: r['s'])
# a step starts at the embed_fwd kernel
starts = [i for i, r in enumerate(rows) if 'embed_fwd' in r['n'] and 'trg' not in r['n']]
a, b = starts[-3], starts[-2]
step = rows[a:b]
t0 = step[0]['s']
print("step wall %.3f ms, %d kernels" % ((rows[b]['s'] - t0) / 1e6, len(step)))
byq = collections.defaultdict(list)
for r in step:
    byq[r['Queue_Id']].append(r)
for q, rs in byq.items():
    busy = 0; last = None
    for r in sorted(rs, key=lambda r: r['s']):
        s, e = r['s'], r['e']
        if last is not None and s < last: s = last
        if e > s: busy += e - s
        last = max(last or 0, e)
    print("queue %s: %d kernels, busy %.3f ms, first %.3f last end %.3f" % (q, len(rs), busy / 1e6, (rs[0]['s'] - t0) / 1e6, (max(r['e'] for r in rs) - t0) / 1e6))
if len(sys.argv) > 2:
    for r in step:
        print("%8.3f %8.3f q%s %7.1f us  %s" % ((r['s'] - t0) / 1e6, (r['e'] - t0) / 1e6, r['Queue_Id'], (r['e'] - r['s']) / 1e3, r['n']))
