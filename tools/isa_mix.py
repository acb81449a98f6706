"""Instruction mix of every backward-branch loop in a gfx950 assembly listing (hipcc -S --cuda-device-only)."""
import collections
import re
import sys

text = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)s_endpgm", text, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pat not in name:
        continue
    lines = [l.strip() for l in body.split("\n")]
    labels = {}
    for i, l in enumerate(lines):
        mm = re.match(r"^(\.LBB\d+_\d+):", l)
        if mm:
            labels[mm.group(1)] = i
    print(name, "instructions:", sum(1 for l in lines if l and l[0] not in ".;"))
    for i, l in enumerate(lines):
        mm = re.match(r"^s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if mm and labels.get(mm.group(1), 1 << 30) < i:
            cnt = collections.Counter()
            for x in lines[labels[mm.group(1)]:i]:
                if not x or x[0] in ".;":
                    continue
                op = x.split()[0]
                if op.startswith("v_mfma"): k = "mfma"
                elif op.startswith(("ds_", "global_", "buffer_", "scratch_")): k = op
                elif op.startswith("v_"): k = "valu"
                elif op.startswith("s_waitcnt"): k = "waitcnt"
                elif op.startswith("s_barrier"): k = "barrier"
                else: k = "salu"
                cnt[k] += 1
            print("  loop", labels[mm.group(1)], i, dict(cnt))
