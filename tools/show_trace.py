import csv, glob, collections, sys
tag = sys.argv[1]
import os
f = max(glob.glob(f'/root/repo/gpurun_out/{tag}/trace/*/*kernel_trace.csv'), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(list)
for r in rows:
    name = r['Kernel_Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')[:40]
    agg[(name, r['Grid_Size_X'], r['Workgroup_Size_X'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
items = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
tot = sum(sum(v) for v in agg.values())
nsteps = 13
for k, v in items[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print("%-42s grid %8s wg %5s n/step %5.1f avg %8.1f us  ms/step %6.3f %5.1f%%" % (k[0], k[1], k[2], len(v) / nsteps, sum(v) / len(v), sum(v) / 1e3 / nsteps, 100 * sum(v) / tot))
print("total ms/step", tot / 1e3 / nsteps)
