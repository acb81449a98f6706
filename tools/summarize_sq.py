"""profiles/<tag>_sq_counters.md from the two SQ-counter passes of tools/pmc_sq.sh (gpurun_out/<tag>_sq/pass{1,2}.json)
and the kernel times of the round's trace (profiles/<tag>_kernel_stats.csv): tools/summarize_sq.py <tag> [kernels]"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r3"
top = int(sys.argv[2]) if len(sys.argv) > 2 else 8
p1 = json.load(open(os.path.join(ROOT, "gpurun_out", tag + "_sq", "pass1.json")))
p2 = json.load(open(os.path.join(ROOT, "gpurun_out", tag + "_sq", "pass2.json")))
stats = list(csv.DictReader(open(os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"))))
stats = [r for r in stats if not r["Name"].startswith("__amd") and "at::native" not in r["Name"]][:top]


def val(p, kern, c):
    e = p.get(kern, {}).get(c)
    return (e["sum"] / e["launches"]) if e and e["launches"] else float("nan")


with open(os.path.join(ROOT, "profiles", tag + "_sq_counters.md"), "w") as f:
    f.write(f"# SQ counters of the {top} kernels with the most GPU time, {tag}\n\n"
            "`rocprofv3 --pmc <8 SQ counters> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-tap "
            "--no-secondary --repeats 1`, two passes (tools/pmc_sq.sh); values are per launch, averaged over the launches "
            "of the counter run.  Derived columns: *MFMA busy* = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_WAVE_CYCLES) (share of "
            "the waves' lifetime during which the matrix pipe works for them; SQ_WAVE_CYCLES counts quad-cycles), *VALU per "
            "MFMA* = (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA, *wait* = SQ_WAIT_ANY / SQ_WAVE_CYCLES (parked in "
            "s_waitcnt or a barrier), *issue stall* = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES, *LDS conflicts* = "
            "SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, *waves* = SQ_WAVES, the waves of a launch (256 CUs x 4 "
            "SIMDs; SQ_LEVEL_WAVES, the resident-wave counter, reads 0 under this rocprofv3, so residency is what the "
            "launch bounds say: one wave per SIMD for the feed-forward backward kernels, two for the forward ones), *co-exec* = SQ_VALU_MFMA_COEXEC_CYCLES / SQ_VALU_MFMA_BUSY_CYCLES.\n\n")
    f.write("| kernel | avg us | % GPU time | MFMA busy | VALU per MFMA | wait | issue stall | LDS conflicts | waves | co-exec |\n"
            "|---|---|---|---|---|---|---|---|---|---|\n")
    for r in stats:
        k = r["Name"]
        wc = val(p1, k, "SQ_WAVE_CYCLES")
        mf, va = val(p1, k, "SQ_INSTS_MFMA"), val(p1, k, "SQ_INSTS_VALU")
        busy = val(p1, k, "SQ_VALU_MFMA_BUSY_CYCLES")
        row = [busy / (4 * wc), (va - mf) / mf if mf else float("nan"), val(p1, k, "SQ_WAIT_ANY") / wc,
               val(p2, k, "SQ_WAIT_INST_ANY") / wc, val(p2, k, "SQ_LDS_BANK_CONFLICT") / max(val(p2, k, "SQ_LDS_IDX_ACTIVE"), 1.0),
               val(p1, k, "SQ_WAVES"),
               val(p2, k, "SQ_VALU_MFMA_COEXEC_CYCLES") / max(busy, 1.0)]
        f.write(f"| `{k[:70]}` | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} | "
                + " | ".join((f"{x:.0f}" if i == 5 else f"{x:.2f}") for i, x in enumerate(row)) + " |\n")
    f.write("\nRaw per-launch sums (pass 1 / pass 2):\n\n```\n")
    for r in stats:
        k = r["Name"]
        f.write(k[:90] + "\n")
        for p in (p1, p2):
            f.write("   " + "  ".join(f"{c}={e['sum'] / e['launches']:.4g}" for c, e in sorted(p.get(k, {}).items())) + "\n")
    f.write("```\n")
print("wrote profiles/%s_sq_counters.md" % tag)
