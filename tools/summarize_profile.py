"""Turn the rocprofv3 outputs of one round (gpurun_out/<tag>/{trace,pmc_fetch,pmc_write}) into the
committed summaries under profiles/:  <tag>_kernel_stats.csv, <tag>_hbm_traffic.json, <tag>_summary.md.

HBM bytes follow the gfx950 recipe: FETCH_SIZE and WRITE_SIZE are collected in separate --pmc passes, both
count KiB, and FETCH_SIZE under-reports wide coalesced reads by exactly 2x, so
    traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch.
"""
import collections
import csv
import glob
import json
import os
import sys


def load(pat):
    fs = glob.glob(pat)
    return list(csv.DictReader(open(max(fs, key=os.path.getmtime)))) if fs else []   # newest run of that tag


def short(n):
    return n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")


def main(tag, what, pmc_steps=0, unit="training step"):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "gpurun_out", tag)
    out = os.path.join(root, "profiles")
    os.makedirs(out, exist_ok=True)
    stats = load(os.path.join(src, "trace", "*", "*kernel_stats.csv"))
    with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in stats:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                        r["MinNs"], r["MaxNs"]])
    counters = {}
    for nm, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in load(os.path.join(src, f"pmc_{nm}", "*", "*counter_collection.csv")):
            if r["Counter_Name"] != cname:
                continue
            k = short(r["Kernel_Name"])
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
        counters[cname] = {k: v[1] / v[0] for k, v in agg.items()}
        counters[cname + "_launches"] = {k: v[0] for k, v in agg.items()}
    traffic = {}
    for k in set(counters["FETCH_SIZE"]) | set(counters["WRITE_SIZE"]):
        fe, wr = counters["FETCH_SIZE"].get(k, 0.0), counters["WRITE_SIZE"].get(k, 0.0)
        traffic[k] = {"fetch_size_kib_avg": fe, "write_size_kib_avg": wr,
                      "hbm_bytes_per_launch": (2.0 * fe + wr) * 1024.0,
                      "launches_in_pmc_run": counters["FETCH_SIZE_launches"].get(k, counters["WRITE_SIZE_launches"].get(k, 0))}
    json.dump(traffic, open(os.path.join(out, f"{tag}_hbm_traffic.json"), "w"), indent=1, sort_keys=True)
    tot = sum(float(r["TotalDurationNs"]) for r in stats) or 1.0
    with open(os.path.join(out, f"{tag}_summary.md"), "w") as f:
        f.write(f"# rocprofv3 summary, {tag}\n\n{what}\n\n")
        f.write("| kernel | calls | avg us | % of GPU time | HBM MB / launch (PMC) |\n|---|---|---|---|---|\n")
        for r in stats[:22]:
            k = short(r["Name"])
            t = traffic.get(k, {}).get("hbm_bytes_per_launch")
            f.write(f"| `{k[:90]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | "
                    f"{100 * float(r['TotalDurationNs']) / tot:.1f} | {'' if t is None else f'{t / 1e6:.1f}'} |\n")
        f.write(f"\nTotal GPU kernel time {tot / 1e6:.1f} ms over the run.\n")
        if pmc_steps:
            plural = unit + ("es" if unit.endswith("ch") else "s")
            total = sum(v["hbm_bytes_per_launch"] * v["launches_in_pmc_run"] for v in traffic.values())
            f.write(f"\nHBM traffic (PMC, FETCH_SIZE doubled + WRITE_SIZE) summed over every kernel of the counter run: "
                    f"{total / 1e9:.2f} GB in {pmc_steps} {plural} = {total / 1e9 / pmc_steps:.2f} GB per {unit} (includes the "
                    f"one-off launches of model set-up).\n")
    print("wrote profiles for", tag)


if __name__ == "__main__":
    tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
    if "decode" in tag:
        main(tag, "`rocprofv3 --kernel-trace --stats -- python3 bench.py --decode --steps 3 --warmup 1` (beam search, 256 "
                  "molecules x 3 beams x 64 positions, 6 searches incl. the timing and capture passes, 1 MI355X); counters: "
                  "separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of `bench.py --decode --steps 1 --warmup 0`",
             pmc_steps=3, unit="search")
    else:
        main(tag, "`rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline` (13 timed + 5 "
                  "isolated training steps, 64 molecules each, 1 MI355X); counters: separate `--pmc FETCH_SIZE` / "
                  "`--pmc WRITE_SIZE` runs of `bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-tap` (3 + 5 steps)",
             pmc_steps=8)
