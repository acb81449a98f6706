"""Micro-benchmark of the critical-path GEMM shapes of an encoder layer.  usage: bench_gemm.py [M]"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import multimodalspectraltransformer_amd._lib as L
from multimodalspectraltransformer_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 37248
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g)
bf = lambda t: t.bfloat16()
x = bf(r(M, 128)); wqkv = bf(r(384, 128) * 0.1); bqkv = r(384); qkv = torch.empty(M, 384, device=dev, dtype=torch.bfloat16)
wo = bf(r(128, 128) * 0.1); bo = r(128); y = torch.empty(M, 128, device=dev, dtype=torch.bfloat16); z = torch.empty_like(y)
st = torch.empty(M, 2, device=dev); gam = torch.ones(128, device=dev); bet = torch.zeros(128, device=dev)
dz = torch.empty_like(y); dzd = torch.empty_like(y); dg = torch.zeros(128, device=dev); db = torch.zeros(128, device=dev)
dqkv = bf(r(M, 384)); dy = bf(r(M, 128))
cases = {
  "qkv fwd   (bias, K=128, N=384)": lambda: ops.gemm([dict(P=x, Q=wqkv, C=qkv, bias=bqkv, ldp=128, ldq=128, ldc=384, Pn=M, Qn=384, K=128)], L.EPI_BIAS),
  "out+LN fwd (LN, K=128, N=128)": lambda: ops.gemm([dict(P=x, Q=wo, C=y, bias=bo, R=x, ldr=128, gamma=gam, beta=bet, Z=z, stats=st, ldp=128, ldq=128, ldc=128, Pn=M, Qn=128, K=128, drop_p=0.1, seed=3)], L.EPI_LN),
  "dX out    (bias qt, K=128, N=128)": lambda: ops.gemm([dict(P=dy, Q=wo, C=y, ldp=128, ldq=128, ldc=128, Pn=M, Qn=128, K=128)], L.EPI_BIAS, qt=True),
  "dX qkv+LNbwd (K=384, N=128)": lambda: ops.gemm([dict(P=dqkv, Q=wqkv, C=dz, C2=dzd, R=dy, ldr=128, gamma=gam, Z=z, stats=st, dgamma=dg, dbeta=db, ldp=384, ldq=128, ldc=128, Pn=M, Qn=128, K=384, drop_p=0.1, seed=5)], L.EPI_LNBWD, qt=True),
  "dX qkv+LNbwd no dgamma": lambda: ops.gemm([dict(P=dqkv, Q=wqkv, C=dz, C2=dzd, R=dy, ldr=128, gamma=gam, Z=z, stats=st, ldp=384, ldq=128, ldc=128, Pn=M, Qn=128, K=384, drop_p=0.1, seed=5)], L.EPI_LNBWD, qt=True),
  "dX qkv plain (bias qt, K=384)": lambda: ops.gemm([dict(P=dqkv, Q=wqkv, C=dz, R=dy, ldr=128, ldp=384, ldq=128, ldc=128, Pn=M, Qn=128, K=384)], L.EPI_BIAS, qt=True),
}
cases["out+LN fwd (LN, K=128, N=128)"]()   # fills z / st for the backward cases
for name, fn in cases.items():
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print("M=%d %-36s %.1f us" % (M, name, e0.elapsed_time(e1) * 1000 / 20), flush=True)
