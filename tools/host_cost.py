"""Host-side cost of enqueueing one training step: bench at a tiny batch (GPU faster than the host), then a cProfile
of the step loop at the bench batch.  tools/host_cost.py"""
import cProfile, io, os, pstats, subprocess, sys, json
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for b in (2, 64):
    out = subprocess.run([sys.executable, "-u", f"{ROOT}/bench.py", "--steps", "30", "--warmup", "5", "--no-cpu-baseline",
                          "--no-tap", "--batch", str(b)], capture_output=True, text=True, timeout=300).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    print("batch %d: %.3f ms/step, host enqueue %.3f ms/step" % (b, d["ms_per_step"], d["host_enqueue_ms_per_step"]), flush=True)
