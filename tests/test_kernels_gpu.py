"""Kernel-level parity on the GPU: every GEMM operand mode / epilogue and the attention
forward/backward kernels against fp32 torch math on the same bf16-rounded inputs.

Tolerances: inputs are rounded to bf16 first, so the only differences are the bf16
rounding of stored outputs (rel 2^-8) and fp32 summation order."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from multimodalspectraltransformer_amd import _lib as L
    from multimodalspectraltransformer_amd import ops

DEV = "cuda"


def bf(x):
    return x.to(torch.bfloat16)


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def close(got, want, rtol=2e-2, atol=2e-2):
    got = got.float()
    want = want.float()
    err = (got - want).abs().max().item()
    ref = want.abs().max().item()
    assert err <= atol * max(ref, 1e-6) + 1e-6, f"max err {err} vs ref scale {ref}"
    # relative Frobenius error is a much sharper check than max-abs
    fro = ((got - want).norm() / (want.norm() + 1e-3)).item()
    assert fro <= rtol, f"relative Frobenius error {fro}"


# --------------------------------------------------------------------------- dropout hash (host copy)
def mix32(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7feb352d)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846ca68b)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return x


def keep_mask(seed, idx, p):
    """numpy restatement of drop_keep() in csrc/common.h"""
    thr = int(p * 65536.0 + 0.5)
    idx = idx.astype(np.uint64)
    h = mix32(((idx >> np.uint64(1)) * np.uint64(0x9E3779B9) + np.uint64(seed)) & np.uint64(0xFFFFFFFF))
    bits = np.where((idx & np.uint64(1)) == 1, h >> np.uint64(16), h & np.uint64(0xFFFF))
    return bits >= thr


def attn_keep_mask(seed, rowid, key, p):
    """numpy restatement of attn_row_seed()/attn_pair_bits() in csrc/common.h"""
    M = np.uint64(0xFFFFFFFF)
    thr = int(p * 65536.0 + 0.5)
    rs = mix32((np.uint64(seed) + rowid.astype(np.uint64) * np.uint64(0x9E3779B9)) & M)
    t = (rs + (key.astype(np.uint64) >> np.uint64(1)) * np.uint64(0x85EBCA6B)) & M
    t ^= t >> np.uint64(16)
    t = ((t & np.uint64(0xFFFFFF)) * np.uint64(0x9E3779)) & M
    t ^= t >> np.uint64(16)
    bits = np.where((key.astype(np.uint64) & np.uint64(1)) == 1, t >> np.uint64(16), t & np.uint64(0xFFFF))
    return bits >= thr


# --------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(300, 384, 128), (128, 128, 2048), (77, 64, 128), (1000, 2048, 128), (64, 128, 1024)])
def test_gemm_bias_relu(M, N, K):
    x, w, b = bf(rnd(M, K, seed=1)), bf(rnd(N, K, scale=0.1, seed=2)), rnd(N, seed=3)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm([dict(P=x, Q=w, C=out, bias=b, ldp=K, ldq=K, ldc=N, Pn=M, Qn=N, K=K, flags=L.GF_RELU)], L.EPI_BIAS)
    close(out, torch.relu(x.float() @ w.float().t() + b))
    out32 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    r = bf(rnd(M, N, seed=4))
    ops.gemm([dict(P=x, Q=w, C=out32, R=r, ldr=N, ldp=K, ldq=K, ldc=N, Pn=M, Qn=N, K=K, flags=L.GF_OUT_F32)],
             L.EPI_BIAS)
    close(out32, x.float() @ w.float().t() + r.float(), rtol=1e-3, atol=1e-3)


def test_gemm_grouped():
    groups, wants, outs = [], [], []
    for i, (M, N, K) in enumerate([(129 * 3, 384, 128), (66 * 3, 384, 128), (200, 128, 128)]):
        x, w, b = bf(rnd(M, K, seed=10 + i)), bf(rnd(N, K, scale=0.1, seed=20 + i)), rnd(N, seed=30 + i)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        groups.append(dict(P=x, Q=w, C=out, bias=b, ldp=K, ldq=K, ldc=N, Pn=M, Qn=N, K=K))
        wants.append(x.float() @ w.float().t() + b)
        outs.append(out)
    ops.gemm(groups, L.EPI_BIAS)
    for o, w_ in zip(outs, wants):
        close(o, w_)


@pytest.mark.parametrize("M,N,K", [(300, 384, 128), (513, 2048, 128), (100, 128, 2048), (90, 64, 128)])
def test_gemm_dx(M, N, K):
    """dX[M,K] = dY[M,N] @ W[N,K]: W is the contraction-major Q operand."""
    dy, w = bf(rnd(M, N, seed=1)), bf(rnd(N, K, scale=0.1, seed=2))
    r = bf(rnd(M, K, seed=3))
    out = torch.empty(M, K, dtype=torch.bfloat16, device=DEV)
    ops.gemm([dict(P=dy, Q=w, C=out, R=r, ldr=K, ldp=N, ldq=K, ldc=K, Pn=M, Qn=K, K=N)], L.EPI_BIAS, qt=True)
    close(out, dy.float() @ w.float() + r.float())


@pytest.mark.parametrize("M,N,K,split", [(300, 384, 128, 1), (1000, 2048, 128, 4), (777, 128, 2048, 3), (70, 64, 128, 2)])
def test_gemm_dw(M, N, K, split):
    """dW[N,K] += dY[M,N]^T @ X[M,K], db[N] += colsum(dY): both operands contraction-major."""
    dy, x = bf(rnd(M, N, seed=1)), bf(rnd(M, K, seed=2))
    dw = torch.zeros(N, K, dtype=torch.float32, device=DEV)
    db = torch.zeros(N, dtype=torch.float32, device=DEV)
    ops.gemm([dict(P=dy, Q=x, C=dw, dbias=db, ldp=N, ldq=K, ldc=K, Pn=N, Qn=K, K=M, ksplit=split)],
             L.EPI_ATOMIC, pt=True, qt=True)
    close(dw, dy.float().t() @ x.float(), rtol=1e-3, atol=1e-3)
    close(db, dy.float().sum(0), rtol=1e-3, atol=1e-3)


def test_gemm_drelu():
    M, N, K = 300, 2048, 128
    dy, w = bf(rnd(M, K, seed=1)), bf(rnd(K, N, scale=0.1, seed=2))   # dH = dY[M,128] @ W2[128,2048]
    h = bf(torch.relu(rnd(M, N, seed=3)))
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm([dict(P=dy, Q=w, C=out, R=h, ldr=N, ldp=K, ldq=N, ldc=N, Pn=M, Qn=N, K=K, drop_p=0.1)],
             L.EPI_DRELU, qt=True)
    want = (dy.float() @ w.float()) * (h.float() > 0) / 0.9
    close(out, want)


@pytest.mark.parametrize("M,K", [(300, 128), (129, 2048)])
def test_gemm_layernorm_fwd_bwd(M, K):
    x, w, b = bf(rnd(M, K, seed=1)), bf(rnd(128, K, scale=0.1, seed=2)), rnd(128, seed=3)
    r = bf(rnd(M, 128, seed=4))
    gam, bet = 1 + 0.1 * rnd(128, seed=5), 0.1 * rnd(128, seed=6)
    y = torch.empty(M, 128, dtype=torch.bfloat16, device=DEV)
    z = torch.empty_like(y)
    st = torch.empty(M, 2, dtype=torch.float32, device=DEV)
    ops.gemm([dict(P=x, Q=w, C=y, bias=b, R=r, ldr=128, gamma=gam, beta=bet, Z=z, stats=st,
                   ldp=K, ldq=K, ldc=128, Pn=M, Qn=128, K=K)], L.EPI_LN)
    zref = (x.float() @ w.float().t() + b + r.float()).requires_grad_(True)
    gref = gam.clone().requires_grad_(True)
    bref = bet.clone().requires_grad_(True)
    yref = torch.nn.functional.layer_norm(zref, (128,), gref, bref, 1e-5)
    close(z, zref.detach())
    close(y, yref.detach())
    close(st[:, 0], zref.detach().mean(1), rtol=1e-3, atol=1e-3)
    # backward: dy = dY2 @ W[Kb,128] + Rg ; here take the K=0 path (dy = R) and a GEMM path
    dy = bf(rnd(M, 128, seed=7))
    yref.backward(dy.float())
    # recompute reference with the kernel's bf16 z to isolate the backward arithmetic
    dz = torch.empty(M, 128, dtype=torch.bfloat16, device=DEV)
    dzd = torch.empty_like(dz)
    dg = torch.zeros(128, device=DEV)
    dbt = torch.zeros(128, device=DEV)
    ops.gemm([dict(C=dz, C2=dzd, R=dy, ldr=128, gamma=gam, Z=z, stats=st, dgamma=dg, dbeta=dbt,
                   ldp=8, ldq=8, ldc=128, Pn=M, Qn=128, K=0, drop_p=0.0)], L.EPI_LNBWD)
    close(dz, zref.grad, rtol=3e-2, atol=3e-2)
    close(dzd, zref.grad, rtol=3e-2, atol=3e-2)
    close(dg, gref.grad, rtol=2e-2, atol=2e-2)
    close(dbt, bref.grad, rtol=1e-2, atol=1e-2)
    # with a GEMM in front: dy = dh[M,256] @ W1[256,128] + Rg
    dh, w1 = bf(rnd(M, 256, seed=8)), bf(rnd(256, 128, scale=0.1, seed=9))
    dy2 = (dh.float() @ w1.float() + dy.float())
    zref2 = z.float().detach().requires_grad_(True)
    torch.nn.functional.layer_norm(zref2, (128,), gam, bet, 1e-5).backward(dy2)
    dg.zero_()
    dbt.zero_()
    ops.gemm([dict(P=dh, Q=w1, C=dz, C2=None, R=dy, ldr=128, gamma=gam, Z=z, stats=st, dgamma=dg, dbeta=dbt,
                   ldp=256, ldq=128, ldc=128, Pn=M, Qn=128, K=256)], L.EPI_LNBWD, qt=True)
    close(dz, zref2.grad, rtol=3e-2, atol=3e-2)


def test_gemm_dropout_consistency():
    """Forward dropout (LN epilogue) and the regenerated mask in the LN-backward epilogue agree with
    the host restatement of the hash, and the keep rate is 1-p."""
    M, K, p, seed = 512, 128, 0.1, 1234
    x, w = bf(rnd(M, K, seed=1)), bf(rnd(128, K, scale=0.1, seed=2))
    y = torch.empty(M, 128, dtype=torch.bfloat16, device=DEV)
    z = torch.empty_like(y)
    st = torch.empty(M, 2, device=DEV)
    gam, bet = torch.ones(128, device=DEV), torch.zeros(128, device=DEV)
    ops.gemm([dict(P=x, Q=w, C=y, gamma=gam, beta=bet, Z=z, stats=st, ldp=K, ldq=K, ldc=128,
                   Pn=M, Qn=128, K=K, drop_p=p, seed=seed)], L.EPI_LN)
    keep = torch.from_numpy(keep_mask(seed, np.arange(M * 128), p).reshape(M, 128)).to(DEV)
    assert abs(keep.float().mean().item() - 0.9) < 0.01
    want = (x.float() @ w.float().t()) * keep / (1 - p)
    close(z, want)
    dz = torch.empty_like(y)
    dzd = torch.empty_like(y)
    dy = bf(rnd(M, 128, seed=3))
    ops.gemm([dict(C=dz, C2=dzd, R=dy, ldr=128, gamma=gam, Z=z, stats=st, ldp=8, ldq=8, ldc=128,
                   Pn=M, Qn=128, K=0, drop_p=p, seed=seed)], L.EPI_LNBWD)
    close(dzd, dz.float() * keep / (1 - p), rtol=1e-2, atol=1e-2)


# --------------------------------------------------------------------------- fused feed-forward block
def ffn_keep_mask(seed, M, N, p):
    """numpy restatement of the row-seeded hidden dropout of the fused forward kernel (csrc/common.h)."""
    m_, n_ = np.meshgrid(np.arange(M), np.arange(N), indexing="ij")
    return attn_keep_mask(seed, m_.reshape(-1), n_.reshape(-1), p).reshape(M, N)


@pytest.mark.parametrize("fused", [True, False, "split"])
@pytest.mark.parametrize("M,groups,p", [(300, 1, 0.0), (1000, 2, 0.1), (129, 3, 0.0), (33000, 1, 0.1), (16500, 2, 0.0)])
def test_ffn_fwd_bwd(fused, M, groups, p):
    """y = LN(x + drop(W2 drop(relu(W1 x + b1)) + b2)) and its activation-gradient pass, fused kernels and the
    two-launch path, against fp32 torch math on the same bf16 operands (hidden dropout mask taken from h).
    The two large cases need more than 256 128-row tiles, i.e. the 160-row workgroup shape."""
    if M > 10000 and fused is not True:
        pytest.skip("large case exercises the fused kernels only")
    L.check(L.lib().mmst_ffn_config(0 if fused else 1 << 20), "cfg")
    try:
        fw, bw, keepers, bw2, outs2 = [], [], [], [], []
        for gi in range(groups):
            Mg = M + 17 * gi
            sd = 10 * gi
            x = bf(rnd(Mg, 128, seed=sd + 1))
            w1, w2 = bf(rnd(2048, 128, scale=0.08, seed=sd + 2)), bf(rnd(128, 2048, scale=0.03, seed=sd + 3))
            b1, b2 = 0.1 * rnd(2048, seed=sd + 4), 0.1 * rnd(128, seed=sd + 5)
            gam, bet = 1 + 0.1 * rnd(128, seed=sd + 6), 0.1 * rnd(128, seed=sd + 7)
            h = torch.full((Mg, 2048), float("nan"), dtype=torch.bfloat16, device=DEV)
            y = torch.empty(Mg, 128, dtype=torch.bfloat16, device=DEV)
            z = torch.empty_like(y)
            st = torch.empty(Mg, 2, device=DEV)
            ws = torch.full((8, Mg, 128), float("nan"), device=DEV) if fused == "split" else None
            hs = torch.full((Mg, 64), 0x5A5A5A5A, dtype=torch.int32, device=DEV)   # sign words of h, every one to be written
            fw.append(dict(X=x, W1=w1, W2=w2, b1=b1, H=h, Hs=hs, ws=ws, drop_p=p, seed=100 + gi,
                           epi=dict(Pn=Mg, C=y, ldc=128, bias=b2, R=x, ldr=128, gamma=gam, beta=bet, Z=z, stats=st,
                                    drop_p=p, seed=200 + gi)))
            dzd, dz = bf(rnd(Mg, 128, seed=sd + 8)), bf(rnd(Mg, 128, seed=sd + 9))
            dz1 = torch.empty_like(y)
            dzd1 = torch.empty_like(y) if p > 0 else None
            dh = torch.empty_like(h)
            dg, db = torch.zeros(128, device=DEV), torch.zeros(128, device=DEV)
            # the GEMM chained behind the epilogue: Cn = (epilogue output after its dropout) Wn
            wn = bf(rnd(128, 128, scale=0.09, seed=sd + 11))
            cn = torch.full((Mg, 128), float("nan"), dtype=torch.bfloat16, device=DEV)
            # backward epilogue = LayerNorm backward of an EARLIER norm: reuse (z, st, gam) of this call as its saved state
            bw.append(dict(X=dzd, W1=w2.t().contiguous(), W2=w1.t().contiguous(), H=h, dH=dh, ws=ws, drop_p=p, Wn=wn, Cn=cn,
                           epi=dict(Pn=Mg, C=dz1, ldc=128, R=dz, ldr=128, gamma=gam, Z=z, stats=st, dgamma=dg, dbeta=db,
                                    C2=dzd1, drop_p=p, seed=300 + gi)))
            keepers.append((x, w1, w2, b1, b2, gam, bet, h, y, z, dzd, dz, dz1, dh, dg, db, dzd1, wn, cn))
            # the same backward from the sign words alone (no stored hidden gradient): h is poisoned before it runs
            o2 = dict(dz1=torch.empty_like(y), dzd1=torch.empty_like(y) if p > 0 else None, cn=torch.empty_like(cn),
                      dg=torch.zeros(128, device=DEV), db=torch.zeros(128, device=DEV))
            bw2.append(dict(bw[-1], H=h, Hs=hs, dH=None if fused is True else torch.empty_like(h), Cn=o2["cn"],
                            epi=dict(bw[-1]["epi"], C=o2["dz1"], C2=o2["dzd1"], dgamma=o2["dg"], dbeta=o2["db"])))
            outs2.append((hs, o2))
        ops.ffn_launch(ops.ffn_batch(fw), backward=False)
        chained = ops.ffn_launch(ops.ffn_batch(bw), backward=True)
        assert chained == (1 if fused is True else 0)
        torch.cuda.synchronize()
        hidden = [k[7].clone() for k in keepers]
        if fused is True:
            for k in keepers:
                k[7].fill_(float("nan"))
        assert ops.ffn_launch(ops.ffn_batch(bw2), backward=True) == chained
        torch.cuda.synchronize()
        for k, hc in zip(keepers, hidden):
            k[7].copy_(hc)
        for (hs, o2), k in zip(outs2, keepers):
            h, dz1, dg, db, dzd1, cn = k[7], k[12], k[14], k[15], k[16], k[18]
            # word 16 q + j  <->  columns 64 c + 32 t + 8 q + e (c = 2 j + odd); flag bit (e odd ? 31 : 15) - 4 t - (e >> 1) - 8 odd
            pos = (h.float() > 0).view(-1, 16, 2, 2, 4, 8).permute(0, 4, 1, 2, 3, 5).reshape(-1, 64, 2, 2, 8).long()
            bit = torch.tensor([[[(31 if e & 1 else 15) - 4 * t - (e >> 1) - 8 * odd for e in range(8)] for t in range(2)]
                                for odd in range(2)], device=DEV)
            words = (pos << bit).sum((-1, -2, -3))
            assert torch.equal(hs.long() & 0xFFFFFFFF, words)
            assert torch.equal(o2["dz1"], dz1)
            close(o2["dg"], dg, rtol=1e-5, atol=1e-5)   # column sums by atomics: the order differs from run to run
            close(o2["db"], db, rtol=1e-5, atol=1e-5)
            assert dzd1 is None or torch.equal(o2["dzd1"], dzd1)
            assert not chained or torch.equal(o2["cn"], cn)
        for gi, (x, w1, w2, b1, b2, gam, bet, h, y, z, dzd, dz, dz1, dh, dg, db, dzd1, wn, cn) in enumerate(keepers):
            Mg = x.shape[0]
            pre = torch.relu(x.float() @ w1.float().t() + b1)
            if p > 0:
                if Mg > 10000:   # too many elements for the numpy hash: take the mask from h, check its rate
                    keep = (h != 0) | (pre < 0.05)
                    assert abs(keep[pre >= 0.05].float().mean().item() - (1 - p)) < 0.002
                elif fused:
                    keep = torch.from_numpy(ffn_keep_mask(100 + gi, Mg, 2048, p)).to(DEV)
                else:
                    keep = torch.from_numpy(keep_mask(100 + gi, np.arange(Mg * 2048), p).reshape(Mg, 2048)).to(DEV)
                assert abs(keep.float().mean().item() - (1 - p)) < 0.01 or Mg > 10000
                pre = pre * keep / (1 - p)
            close(h, pre)
            out = h.float() @ w2.float().t() + b2
            if p > 0:
                keep2 = torch.from_numpy(keep_mask(200 + gi, np.arange(Mg * 128), p).reshape(Mg, 128)).to(DEV)
                out = out * keep2 / (1 - p)
            zref = x.float() + out
            close(z, zref)
            close(y, torch.nn.functional.layer_norm(z.float(), (128,), gam, bet, 1e-5))
            # backward
            dh_ref = (dzd.float() @ w2.float()) * (h.float() > 0) / (1 - p)
            close(dh, dh_ref)
            dy = dh.float() @ w1.float() + dz.float()
            zr = z.float().detach().requires_grad_(True)
            gr = gam.clone().requires_grad_(True)
            torch.nn.functional.layer_norm(zr, (128,), gr, bet, 1e-5).backward(dy)
            close(dz1, zr.grad, rtol=3e-2, atol=3e-2)
            close(dg, gr.grad, rtol=2e-2, atol=2e-2)
            close(db, dy.sum(0), rtol=2e-2, atol=2e-2)
            e_out = dz1
            if p > 0:   # the epilogue's second output: dz1 through the dropout mask in front of the norm
                keep3 = torch.from_numpy(keep_mask(300 + gi, np.arange(Mg * 128), p).reshape(Mg, 128)).to(DEV)
                close(dzd1, dz1.float() * keep3 / (1 - p), rtol=1e-2, atol=1e-2)
                e_out = dzd1
            if chained:
                close(cn, e_out.float() @ wn.float(), rtol=2e-2, atol=2e-2)
            else:
                assert bool(torch.isnan(cn.float()).all())   # untouched: the caller runs that GEMM itself
    finally:
        L.check(L.lib().mmst_ffn_config(-1), "cfg")


@pytest.fixture(params=[1, 2, 3])
def wgrad_shape(request):
    """Both workgroup shapes of the recomputing kernel: setting 1 (4 waves; 3 is the same kernel) and 2 (8 waves)."""
    L.check(L.lib().mmst_ffn_wgrad_config(request.param), "mmst_ffn_wgrad_config")
    assert L.lib().mmst_ffn_wgrad_mode() == request.param
    yield request.param
    L.check(L.lib().mmst_ffn_wgrad_config(-1), "mmst_ffn_wgrad_config")


@pytest.mark.parametrize("M,groups", [(37248, 1), (7200, 4), (4032, 1), (300, 2)])
def test_ffn_fwd_with_prologue_and_chained_projection(M, groups):
    """The fused forward kernel with the linear layer + LayerNorm in front of the block (out_proj + residual + norm1)
    as its prologue and the next layer's QKV projection chained behind its row epilogue, against fp32 torch math on
    the same bf16 operands, dropout off:  x1 = LN1(x + attn Wo^T + bo),  y = LN2(x1 + FFN(x1)),  qkv = y Wq^T + bq.
    37248 rows (the cross encoder at 64 molecules) and 4 x 7200 rows take the fused kernel, which reports the chained
    projection as done; 4032 rows (the decoder's row count) and 300 rows fall back to separate launches."""
    probs, keep = [], []
    for gi in range(groups):
        Mg, sd = M + 9 * gi, 20 * gi
        attn, x = bf(rnd(Mg, 128, seed=sd + 1)), bf(rnd(Mg, 128, seed=sd + 2))
        wo, bo = bf(rnd(128, 128, scale=0.09, seed=sd + 3)), 0.1 * rnd(128, seed=sd + 4)
        g1, be1 = 1 + 0.1 * rnd(128, seed=sd + 5), 0.1 * rnd(128, seed=sd + 6)
        w1, w2 = bf(rnd(2048, 128, scale=0.08, seed=sd + 7)), bf(rnd(128, 2048, scale=0.03, seed=sd + 8))
        b1, b2 = 0.1 * rnd(2048, seed=sd + 9), 0.1 * rnd(128, seed=sd + 10)
        g2, be2 = 1 + 0.1 * rnd(128, seed=sd + 11), 0.1 * rnd(128, seed=sd + 12)
        wq, bq = bf(rnd(384, 128, scale=0.09, seed=sd + 13)), 0.1 * rnd(384, seed=sd + 14)
        nan16 = lambda *shape: torch.full(shape, float("nan"), dtype=torch.bfloat16, device=DEV)   # noqa: E731
        x1, z1, y, z2, h, qkv = nan16(Mg, 128), nan16(Mg, 128), nan16(Mg, 128), nan16(Mg, 128), nan16(Mg, 2048), nan16(Mg, 384)
        st1, st2 = torch.empty(Mg, 2, device=DEV), torch.empty(Mg, 2, device=DEV)
        probs.append(dict(X=x1, W1=w1, W2=w2, b1=b1, H=h, drop_p=0.0, seed=1, A=attn, Wp=wo, Wq=wq, bq=bq, Cq=qkv, nq=384,
                          pre=dict(Pn=Mg, C=x1, ldc=128, bias=bo, R=x, ldr=128, gamma=g1, beta=be1, Z=z1, stats=st1),
                          epi=dict(Pn=Mg, C=y, ldc=128, bias=b2, R=x1, ldr=128, gamma=g2, beta=be2, Z=z2, stats=st2)))
        keep.append((attn, x, wo, bo, g1, be1, w1, w2, b1, b2, g2, be2, wq, bq, x1, z1, y, h, qkv, st1))
    tiles = sum((q["epi"]["Pn"] + 127) // 128 for q in probs)
    rc = ops.ffn_launch(ops.ffn_batch(probs), backward=False)
    fused = tiles >= 96   # the default threshold of the fused kernels (mmst_ffn_config)
    assert rc == (2 if fused else 0)
    torch.cuda.synchronize()
    ln = torch.nn.functional.layer_norm
    for attn, x, wo, bo, g1, be1, w1, w2, b1, b2, g2, be2, wq, bq, x1, z1, y, h, qkv, st1 in keep:
        zr = x.float() + attn.float() @ wo.float().t() + bo
        close(z1, zr, rtol=2e-2, atol=2e-2)
        x1r = ln(zr, (128,), g1, be1, 1e-5)
        close(x1, x1r, rtol=2e-2, atol=2e-2)
        close(st1[:, 0], zr.mean(dim=1), rtol=1e-2, atol=1e-2)
        # downstream stages from the kernel's own (bf16) x1, as the kernel computes them
        hr = torch.relu(x1.float() @ w1.float().t() + b1)
        close(h, hr, rtol=2e-2, atol=2e-2)
        yr = ln(x1.float() + h.float() @ w2.float().t() + b2, (128,), g2, be2, 1e-5)
        close(y, yr, rtol=2e-2, atol=3e-2)
        if rc == 2:
            close(qkv, y.float() @ wq.float().t() + bq, rtol=2e-2, atol=2e-2)
        else:
            assert torch.isnan(qkv.float()).all()   # not produced: the caller launches that projection


@pytest.mark.parametrize("M,groups,p", [(1024, 1, 0.0), (1000, 2, 0.1), (77, 1, 0.1), (8256, 1, 0.1), (33000, 1, 0.1), (37248, 1, 0.1)])
def test_ffn_wgrad_recompute(M, groups, p, wgrad_shape):
    """dW1/db1/dW2/db2 of the feed-forward block from (dz, h, x, W2^T) with dh recomputed in the kernel,
    accumulated on top of existing gradient content, against fp32 torch math."""
    gs, keep = [], []
    for gi in range(groups):
        Mg, sd = M + 40 * gi, 20 * gi
        dz, x = bf(rnd(Mg, 128, seed=sd + 1)), bf(rnd(Mg, 128, seed=sd + 2))
        h = bf(torch.relu(rnd(Mg, 2048, seed=sd + 3)))
        w2 = bf(rnd(128, 2048, scale=0.05, seed=sd + 4))
        dW1, db1 = rnd(2048, 128, seed=sd + 5), rnd(2048, seed=sd + 6)
        dW2, db2 = rnd(128, 2048, seed=sd + 7), rnd(128, seed=sd + 8)
        init = [t.clone() for t in (dW1, db1, dW2, db2)]
        gs.append(dict(dZ=dz, H=h, X=x, W2T=w2.t().contiguous(), dW1=dW1, db1=db1, dW2=dW2, db2=db2, M=Mg, drop_p=p))
        keep.append((dz, x, h, w2, dW1, db1, dW2, db2, init))
    ops.ffn_wgrad(gs)
    torch.cuda.synchronize()
    for dz, x, h, w2, dW1, db1, dW2, db2, init in keep:
        dh = bf((dz.float() @ w2.float()) * (h.float() > 0) / (1 - p)).float()
        close(dW2 - init[2], dz.float().t() @ h.float(), rtol=5e-3, atol=5e-3)
        close(db2 - init[3], dz.float().sum(0), rtol=5e-3, atol=5e-3)
        close(dW1 - init[0], dh.t() @ x.float(), rtol=5e-3, atol=5e-3)
        close(db1 - init[1], dh.sum(0), rtol=5e-3, atol=5e-3)


def test_transpose_bf16():
    import ctypes as C
    src = bf(rnd(3 * 2048 * 128 + 64, seed=5))
    dst = torch.zeros_like(src)
    offs = [0, 2048 * 128 + 64, 2 * 2048 * 128 + 64]
    arr = (C.c_uint32 * 3)(*offs)
    L.check(L.lib().mmst_transpose_bf16(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), 3, arr, 2048, 128,
                                        L.stream_ptr()), "transpose")
    for o in offs:
        assert torch.equal(dst[o:o + 2048 * 128].view(128, 2048), src[o:o + 2048 * 128].view(2048, 128).t())


# --------------------------------------------------------------------------- attention
def attn_ref(q, k, v, bias, causal, hd, keep=None, p=0.0):
    """q (B,T,H,hd) fp32 etc.  Returns O (B,T,H,hd)."""
    s = torch.einsum("bthd,bshd->bhts", q, k) / math.sqrt(hd)
    if bias is not None:
        s = s + bias[:, None, None, :]
    if causal:
        T, S = s.shape[-2:]
        s = s + torch.triu(torch.full((T, S), float("-inf"), device=s.device), diagonal=1)
    a = torch.softmax(s, dim=-1)
    if keep is not None:
        a = a * keep / (1 - p)
    return torch.einsum("bhts,bshd->bthd", a, v)


def run_attn(B, H, T, S, hd, causal, use_bias, p=0.0, seed=77, self_attn=True):
    d = H * hd
    if self_attn:
        qkv = bf(rnd(B * T, 3 * d, seed=1))
        Q, K_, V = qkv, qkv, qkv
        qo, ko, vo, ldq, ldk = 0, d, 2 * d, 3 * d, 3 * d
    else:
        Q = bf(rnd(B * T, d, seed=1))
        K_ = bf(rnd(B * S, 2 * d, seed=2))
        V = K_
        qo, ko, vo, ldq, ldk = 0, 0, d, d, 2 * d
    bias = None
    if use_bias:
        bias = (torch.rand(B, S, generator=torch.Generator().manual_seed(5)) < 0.3).float().to(DEV)
        bias[:, S // 2:S // 2 + 3] = float("-inf")
    O = torch.zeros(B * T, d, dtype=torch.bfloat16, device=DEV)
    lse = torch.zeros(B, H, T, device=DEV)
    grp = dict(Q=ops.eptr(Q, qo), K=ops.eptr(K_, ko), V=ops.eptr(V, vo), kbias=bias, O=O, lse=lse,
               ldq=ldq, ldk=ldk, ldv=ldk, ldo=d, B=B, H=H, Tq=T, Sk=S, seed=seed)
    ops.attn_fwd_launch(ops.attn_batch([grp], hd, causal, p))
    qf = Q.float()[:, qo:qo + d].reshape(B, T, H, hd).clone().requires_grad_(True)
    kf = K_.float()[:, ko:ko + d].reshape(B, S, H, hd).clone().requires_grad_(True)
    vf = V.float()[:, vo:vo + d].reshape(B, S, H, hd).clone().requires_grad_(True)
    keep = None
    if p > 0:
        b_, h_, t_, s_ = np.meshgrid(np.arange(B), np.arange(H), np.arange(T), np.arange(S), indexing="ij")
        rowid = (b_ * H + h_) * T + t_
        keep = torch.from_numpy(attn_keep_mask(seed, rowid.reshape(-1), s_.reshape(-1), p).reshape(B, H, T, S)).to(DEV)
        assert abs(keep.float().mean().item() - (1 - p)) < 0.01
    ref = attn_ref(qf, kf, vf, bias, causal, hd, keep, p)
    close(O.reshape(B, T, H, hd), ref.detach())
    # backward
    dO = bf(rnd(B * T, d, seed=9))
    ref.backward(dO.float().reshape(B, T, H, hd))
    delta = torch.zeros(B, H, T, device=DEV)
    if self_attn:
        dqkv = torch.zeros(B * T, 3 * d, dtype=torch.bfloat16, device=DEV)
        dQ, dK, dV, lq, lk = ops.eptr(dqkv, 0), ops.eptr(dqkv, d), ops.eptr(dqkv, 2 * d), 3 * d, 3 * d
    else:
        dq_t = torch.zeros(B * T, d, dtype=torch.bfloat16, device=DEV)
        dkv_t = torch.zeros(B * S, 2 * d, dtype=torch.bfloat16, device=DEV)
        dQ, dK, dV, lq, lk = ops.eptr(dq_t, 0), ops.eptr(dkv_t, 0), ops.eptr(dkv_t, d), d, 2 * d
    grp.update(dO=dO, delta=delta, dQ=dQ, dK=dK, dV=dV, lddo=d, lddq=lq, lddk=lk, lddv=lk)
    ops.attn_bwd_launch(ops.attn_batch([grp], hd, causal, p))
    if self_attn:
        gq, gk, gv = dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:]
    else:
        gq, gk, gv = dq_t, dkv_t[:, :d], dkv_t[:, d:]
    close(gq.reshape(B, T, H, hd), qf.grad, rtol=3e-2, atol=3e-2)
    close(gk.reshape(B, S, H, hd), kf.grad, rtol=3e-2, atol=3e-2)
    close(gv.reshape(B, S, H, hd), vf.grad, rtol=3e-2, atol=3e-2)


@pytest.fixture(params=[0, 7, 8], ids=["bwd-two-pass", "bwd-one-pass", "bwd-short-seq"])
def attn_bwd_mode(request):
    """The forms of the self-attention backward: two passes (dQ, then dK/dV), the single-pass kernel and the kernel
    that runs both passes of a short sequence in one workgroup (each serves the shapes that fit it; the others fall back
    to the two passes under any setting)."""
    L.check(L.lib().mmst_attn_bwd_config(request.param), "mmst_attn_bwd_config")
    assert L.lib().mmst_attn_bwd_mode() == request.param
    yield request.param
    L.check(L.lib().mmst_attn_bwd_config(-1), "mmst_attn_bwd_config")


@pytest.mark.parametrize("B,H,S,hd", [(2, 16, 129, 8), (3, 16, 66, 8), (2, 4, 582, 32), (1, 4, 453, 32),
                                       (2, 16, 193, 8), (1, 4, 902, 32), (2, 16, 64, 8), (2, 4, 608, 32),
                                       (3, 4, 33, 32), (2, 16, 97, 8)])
def test_attn_self(B, H, S, hd, attn_bwd_mode):
    run_attn(B, H, S, S, hd, causal=False, use_bias=True)


@pytest.mark.parametrize("T", [1, 5, 63, 128, 33, 64])
def test_attn_causal(T, attn_bwd_mode):
    run_attn(2, 16, T, T, 8, causal=True, use_bias=False)


def test_attn_causal_dropout(attn_bwd_mode):
    run_attn(2, 16, 63, 63, 8, causal=True, use_bias=False, p=0.1)


@pytest.mark.parametrize("T,S", [(63, 582), (1, 453), (128, 902), (65, 130), (129, 64), (64, 98), (1, 66), (2, 130), (1, 128), (1, 7), (2, 9)])
def test_attn_cross(T, S):
    run_attn(2, 16, T, S, 8, causal=False, use_bias=True, self_attn=False)


@pytest.mark.parametrize("hd,H,S", [(8, 16, 129), (32, 4, 300), (32, 4, 582), (8, 16, 66)])
def test_attn_dropout(hd, H, S, attn_bwd_mode):
    run_attn(2, H, S, S, hd, causal=False, use_bias=True, p=0.1)
