"""Attention alone at the training shapes (64 molecules): tools/bench_attn.py [cross|modal] [p] [fwd]
Prints the time per launch of mmst_attn_bwd (whatever kernels it selects), or of mmst_attn_fwd with a fourth
argument, measured with events."""
import sys
import torch
sys.path.insert(0, ".")
from multimodalspectraltransformer_amd import ops  # noqa: E402

DEV = "cuda"
which = sys.argv[1] if len(sys.argv) > 1 else "cross"
p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
fwd = len(sys.argv) > 3
launch = ops.attn_fwd_launch if fwd else ops.attn_bwd_launch
B = 64
shapes = [(4, 582, 32)] if which == "cross" else [(16, 129, 8)] * 3 + [(16, 66, 8)]
hd = shapes[0][2]
groups = []
keep = []
for i, (H, S, _) in enumerate(shapes):
    d = 128
    g = torch.Generator().manual_seed(i)
    qkv = (torch.randn(B * S, 3 * d, generator=g) * 0.5).to(DEV).bfloat16()
    O = torch.zeros(B * S, d, dtype=torch.bfloat16, device=DEV)
    lse = torch.zeros(B, H, S, device=DEV)
    dO = (torch.randn(B * S, d, generator=g) * 0.1).to(DEV).bfloat16()
    dqkv = torch.zeros(B * S, 3 * d, dtype=torch.bfloat16, device=DEV)
    delta = torch.zeros(B, H, S, device=DEV)
    bias = torch.zeros(B, S, device=DEV)
    grp = dict(Q=ops.eptr(qkv, 0), K=ops.eptr(qkv, d), V=ops.eptr(qkv, 2 * d), kbias=bias, O=O, lse=lse, ldq=3 * d, ldk=3 * d,
               ldv=3 * d, ldo=d, B=B, H=H, Tq=S, Sk=S, seed=77 + i, dO=dO, delta=delta, dQ=ops.eptr(dqkv, 0),
               dK=ops.eptr(dqkv, d), dV=ops.eptr(dqkv, 2 * d), lddo=d, lddq=3 * d, lddk=3 * d, lddv=3 * d)
    groups.append(grp)
    keep.append((qkv, O, lse, dO, dqkv, delta, bias))
batch = ops.attn_batch(groups, hd, False, p)
ops.attn_fwd_launch(batch)
for _ in range(3):
    launch(batch)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
N = 20
e0.record()
for _ in range(N):
    launch(batch)
e1.record()
torch.cuda.synchronize()
print("%s p=%.2f: %.1f us per %s" % (which, p, e0.elapsed_time(e1) * 1000 / N, "forward" if fwd else "backward"))
