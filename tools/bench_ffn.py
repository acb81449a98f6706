"""Micro-benchmark of the fused feed-forward kernels.  usage: bench_ffn.py [lib.so] [M ...]"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import multimodalspectraltransformer_amd._lib as L
args = sys.argv[1:]
if args and args[0].endswith(".so"):
    L.LIB_PATH = os.path.abspath(args.pop(0))
from multimodalspectraltransformer_amd import ops
if args and args[0].startswith("fuse="):
    L.check(L.lib().mmst_ffn_config(int(args.pop(0)[5:])), "cfg")
Ms = [int(a) for a in args] or [37248, 4032]
dev = "cuda"
DP = float(os.environ.get('FFN_DROP', '0.1'))
for M in Ms:
    g = torch.Generator(device=dev).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)
    x = r(M, 128).bfloat16(); w1 = (r(2048, 128) * 0.05).bfloat16(); w2 = (r(128, 2048) * 0.05).bfloat16()
    b1 = r(2048) * 0.1; b2 = r(128) * 0.1; gam = torch.ones(128, device=dev); bet = torch.zeros(128, device=dev)
    h = torch.empty(M, 2048, device=dev, dtype=torch.bfloat16); z = torch.empty(M, 128, device=dev, dtype=torch.bfloat16)
    y = torch.empty_like(z); st = torch.empty(M, 2, device=dev)
    ws = torch.empty(8, M, 128, device=dev) if os.environ.get('FFN_WS') else None
    hs = torch.zeros(M, 64, device=dev, dtype=torch.int32) if os.environ.get('FFN_SIGNS') else None   # needs FFN_NODH
    fb = ops.ffn_batch([dict(X=x, W1=w1, W2=w2, b1=b1, H=h, Hs=hs, ws=ws, drop_p=DP, seed=7,
                             epi=dict(Pn=M, C=y, ldc=128, bias=b2, R=x, ldr=128, gamma=gam, beta=bet, Z=z, stats=st, drop_p=DP, seed=9))])
    dzd = r(M, 128).bfloat16(); dz = r(M, 128).bfloat16(); dz1 = torch.empty_like(z); dzd1 = torch.empty_like(z)
    dg = torch.zeros(128, device=dev); db = torch.zeros(128, device=dev)
    w2t = w2.t().contiguous(); w1t = w1.t().contiguous()
    dh = torch.empty_like(h)
    bb = ops.ffn_batch([dict(X=dzd, W1=w2t, W2=w1t, H=h, Hs=hs, dH=None if os.environ.get('FFN_NODH') else dh, ws=ws, drop_p=0.1,
                             epi=dict(Pn=M, C=dz1, ldc=128, R=dz, ldr=128, gamma=gam, Z=z, stats=st, C2=dzd1, dgamma=dg, dbeta=db, drop_p=0.1, seed=5))])
    for name, b, bwd in (("fwd", fb, False), ("bwd", bb, True)):
        for _ in range(3): ops.ffn_launch(b, bwd)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.ffn_launch(b, bwd)
        e1.record(); torch.cuda.synchronize()
        print("M=%d %s %.1f us" % (M, name, e0.elapsed_time(e1) * 1000 / 20), flush=True)
    # weight gradients: recomputing kernel vs two split-K GEMMs on a stored hidden gradient
    dW1 = torch.zeros(2048, 128, device=dev); db1 = torch.zeros(2048, device=dev)
    dW2 = torch.zeros(128, 2048, device=dev); db2 = torch.zeros(128, device=dev)
    wg = [dict(dZ=dzd, H=h, X=x, W2T=w2t, dW1=dW1, db1=db1, dW2=dW2, db2=db2, M=M, drop_p=0.1)]
    ks = max(1, min(64, 128 // 16))
    gg = [dict(P=dzd, Q=h, C=dW2, dbias=db2, ldp=128, ldq=2048, ldc=2048, Pn=128, Qn=2048, K=M, ksplit=ks),
          dict(P=dh, Q=x, C=dW1, dbias=db1, ldp=2048, ldq=128, ldc=128, Pn=2048, Qn=128, K=M, ksplit=ks)]
    for name, fn in (("wgrad recompute", lambda: ops.ffn_wgrad(wg)),
                     ("wgrad stored dh", lambda: ops.gemm(gg, L.EPI_ATOMIC, pt=True, qt=True))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        print("M=%d %s %.1f us" % (M, name, e0.elapsed_time(e1) * 1000 / 20), flush=True)
