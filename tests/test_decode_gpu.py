"""mmst_decoder_stack_step (one new target position through every decoder layer in one launch) through the C ABI
against a plain PyTorch fp32 evaluation of the same step on the same bf16 weights and caches, and against
mmst_decoder_layer_step.  The kernel rounds the row to bf16 between stages (as the training kernels do), so the
stated tolerance is max-abs 6e-2 on LayerNorm outputs (6e-2 + 2 % on logits) and cosine >= 0.999."""
import ctypes as C
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

D, FF, NH, HD = 128, 2048, 16, 8

if torch.cuda.is_available():
    from multimodalspectraltransformer_amd import _lib as L


def _layer_params(gen, dev):
    r = lambda *s, sc=1.0: torch.randn(*s, generator=gen, device=dev) * sc  # noqa: E731
    w = dict(w_qkv=r(3 * D, D, sc=0.09), w_o=r(D, D, sc=0.09), w_cqkv=r(3 * D, D, sc=0.09), w_co=r(D, D, sc=0.09),
             w_1=r(FF, D, sc=0.09), w_2=r(D, FF, sc=0.02))
    w = {k: v.bfloat16() for k, v in w.items()}
    f = dict(b_qkv=r(3 * D, sc=0.1), b_o=r(D, sc=0.1), b_cqkv=r(3 * D, sc=0.1), b_co=r(D, sc=0.1), b_1=r(FF, sc=0.1),
             b_2=r(D, sc=0.1))
    for i in (1, 2, 3):
        f[f"g{i}"] = 1.0 + r(D, sc=0.1)
        f[f"be{i}"] = r(D, sc=0.1)
    return {**w, **f}


def _ln(z, g, b):
    mu = z.mean(-1, keepdim=True)
    var = ((z - mu) ** 2).mean(-1, keepdim=True)
    return (z - mu) * torch.rsqrt(var + 1e-5) * g + b


def _attend(q, K, V, bias):
    """q [B][128], K/V [B][S][128] fp32, bias [B][S] or None -> [B][128]"""
    B, S = K.shape[0], K.shape[1]
    qh = q.view(B, NH, 1, HD)
    Kh = K.view(B, S, NH, HD).permute(0, 2, 1, 3)
    Vh = V.view(B, S, NH, HD).permute(0, 2, 1, 3)
    sc = (qh @ Kh.transpose(2, 3)) / math.sqrt(HD)
    if bias is not None:
        sc = sc + bias.view(B, 1, 1, S)
    return (torch.softmax(sc, dim=-1) @ Vh).reshape(B, D)


def _reference_step(layers, x, caches, pos, kvc, bias, group):
    """fp32 evaluation; caches[l] [B][cap][384] are read for positions < pos, the new row is returned per layer."""
    B = x.shape[0]
    rows = []
    for l, p in enumerate(layers):
        f = {k: v.float() for k, v in p.items()}
        qkv = x @ f["w_qkv"].t() + f["b_qkv"]
        rows.append(qkv)
        c = caches[l].float()
        K = torch.cat([c[:, :pos, D:2 * D], qkv[:, None, D:2 * D]], 1)
        V = torch.cat([c[:, :pos, 2 * D:], qkv[:, None, 2 * D:]], 1)
        a = _attend(qkv[:, :D], K, V, None)
        x1 = _ln(x + a @ f["w_o"].t() + f["b_o"], f["g1"], f["be1"])
        qc = x1 @ f["w_cqkv"][:D].t() + f["b_cqkv"][:D]
        mem = kvc[l].float().repeat_interleave(group, dim=0)
        mb = bias.repeat_interleave(group, dim=0) if bias is not None else None
        a2 = _attend(qc, mem[:, :, :D], mem[:, :, D:], mb)
        x2 = _ln(x1 + a2 @ f["w_co"].t() + f["b_co"], f["g2"], f["be2"])
        h = torch.relu(x2 @ f["w_1"].t() + f["b_1"])
        x = _ln(x2 + h @ f["w_2"].t() + f["b_2"], f["g3"], f["be3"])
    return x, rows


def _params_array(layers):
    arr = (L.DecParams * len(layers))()
    for l, p in enumerate(layers):
        for k, v in p.items():
            setattr(arr[l], k, v.data_ptr())
    return arr


@pytest.mark.parametrize("B,group,Lm,cap,pos,nl,masked", [
    (3, 1, 70, 16, 0, 2, False),      # first position: a single key
    (3, 1, 70, 16, 5, 2, True),
    (4, 2, 582, 140, 63, 3, True),    # two sequences per memory block (beams), benchmark memory length
    (9, 3, 582, 64, 20, 2, True),     # three beams per molecule: one workgroup per molecule
    (8, 4, 200, 64, 33, 2, False),    # four beams per molecule
    (1, 1, 9, 200, 131, 1, False),    # more cached positions than one batch of key loads covers
    (64, 1, 130, 64, 17, 6, True),
    (70, 1, 90, 32, 9, 2, True),      # a ragged last 64-row tile of the token-batched feed-forward launch
])
@pytest.mark.parametrize("split", [False, True], ids=["one-launch", "split-ffn"])
def test_stack_step_vs_torch(B, group, Lm, cap, pos, nl, masked, split):
    """split-ffn: the form taken from 256 sequences on when the caller gives the workspace (every layer's feed-forward
    block as a launch of its own over all sequences), forced here for the small batches."""
    L.check(L.lib().mmst_decoder_stack_config(1 if split else 0), "mmst_decoder_stack_config")
    try:
        _stack_step_vs_torch(B, group, Lm, cap, pos, nl, masked, split)
    finally:
        L.check(L.lib().mmst_decoder_stack_config(-1), "mmst_decoder_stack_config")


def _stack_step_vs_torch(B, group, Lm, cap, pos, nl, masked, split):
    dev = "cuda"
    gen = torch.Generator(device=dev).manual_seed(1000 + B + pos)
    layers = [_layer_params(gen, dev) for _ in range(nl)]
    r = lambda *s: torch.randn(*s, generator=gen, device=dev)  # noqa: E731
    V, VP = 43, 64
    embed, pos_row = r(V, D) * 0.5, r(D) * 0.5
    tok = torch.randint(0, V, (B,), generator=gen, device=dev)
    w_out = torch.zeros(VP, D, device=dev)
    w_out[:V] = r(V, D) * 0.2
    w_out = w_out.bfloat16()
    b_out = torch.zeros(VP, device=dev)
    b_out[:V] = r(V) * 0.1
    caches = [(r(B, cap, 3 * D) * 0.7).bfloat16() for _ in range(nl)]
    kvc = [(r(B // group, Lm, 2 * D) * 0.7).bfloat16() for _ in range(nl)]
    bias = None
    if masked:
        bias = torch.zeros(B // group, Lm, device=dev)
        bias[:, Lm // 2:Lm // 2 + 3] = float("-inf")
        bias[0, -1] = float("-inf")
    x0 = (embed[tok] + pos_row).bfloat16().float()
    want, rows = _reference_step(layers, x0, caches, pos, kvc, bias, group)
    want_logits = want @ w_out.float().t() + b_out

    y = torch.zeros(B, D, dtype=torch.bfloat16, device=dev)
    logits = torch.full((B, VP), 7.0, device=dev)
    nxt = torch.full((B,), -1, dtype=torch.long, device=dev)
    prob = torch.zeros(B, device=dev)
    mine = [c.clone() for c in caches]
    s = L.DecStackStep()
    s.tok, s.embed, s.pos_row, s.y = tok.data_ptr(), embed.data_ptr(), pos_row.data_ptr(), y.data_ptr()
    for l in range(nl):
        s.qkv_cache[l], s.kvc[l] = mine[l].data_ptr(), kvc[l].data_ptr()
    s.mem_bias = bias.data_ptr() if bias is not None else None
    s.w_out, s.b_out, s.logits, s.ld_logits, s.n_out = w_out.data_ptr(), b_out.data_ptr(), logits.data_ptr(), VP, VP
    s.next_tok, s.next_prob, s.temperature, s.V = nxt.data_ptr(), prob.data_ptr(), 0.7, V
    s.B, s.pos, s.cap, s.L, s.kv_group = B, pos, cap, Lm, group
    ws = torch.full((((B * D * 2 + 255) // 256 * 256 + 16 * B * D * 4) // 4,), float("nan"), device=dev)   # MMST_DEC_STACK_WS_BYTES
    s.ffn_ws = ws.data_ptr() if split else None
    L.check(L.lib().mmst_decoder_stack_step(nl, _params_array(layers), C.byref(s), NH, L.stream_ptr()), "stack_step")
    torch.cuda.synchronize()
    assert bool(torch.isnan(ws).all()) != split     # the workspace is used by the split form only

    got = y.float()
    assert float((got - want).abs().max()) < 6e-2
    assert float(torch.nn.functional.cosine_similarity(got, want, dim=1).min()) > 0.999
    lerr = (logits[:, :V] - want_logits[:, :V]).abs()
    assert bool((lerr < 6e-2 + 2e-2 * want_logits[:, :V].abs()).all()), float(lerr.max())   # logits reach |6|
    assert float(logits[:, V:].abs().max()) == 0.0          # zero-padded rows of the projection
    for l in range(nl):   # the new cache row, and nothing else, was written
        assert float((mine[l][:, pos].float() - rows[l]).abs().max()) < (4e-2 if l == 0 else 8e-2)
        keep = torch.ones(cap, dtype=torch.bool, device=dev)
        keep[pos] = False
        assert torch.equal(mine[l][:, keep], caches[l][:, keep])
    # greedy choice on the kernel's own logits: first maximum, softmax(logits / T) probability
    z = logits[:, :V] / 0.7
    assert torch.equal(nxt, z.argmax(dim=1))
    assert float((prob - torch.softmax(z, dim=1).max(dim=1).values).abs().max()) < 1e-5


def test_stack_step_equals_layer_steps():
    """Same caches, same inputs: one launch for the whole stack against one mmst_decoder_layer_step per layer
    (GEMM / attention tile kernels); both round to bf16 at the same points, so they agree to bf16 noise."""
    dev, B, Lm, cap, pos, nl = "cuda", 8, 193, 32, 11, 2
    gen = torch.Generator(device=dev).manual_seed(5)
    layers = [_layer_params(gen, dev) for _ in range(nl)]
    r = lambda *s: torch.randn(*s, generator=gen, device=dev)  # noqa: E731
    x = (r(B, D) * 0.8).bfloat16()
    caches = [(r(B, cap, 3 * D) * 0.7).bfloat16() for _ in range(nl)]
    kvc = [(r(B, Lm, 2 * D) * 0.7).bfloat16() for _ in range(nl)]
    bias = torch.zeros(B, Lm, device=dev)
    bias[:, 100:] = float("-inf")
    params = _params_array(layers)
    bf = dict(dtype=torch.bfloat16, device=dev)
    # per-layer path
    c1 = [c.clone() for c in caches]
    rows = [x.clone(), torch.zeros(B, D, **bf)]
    attn, x1, qc, x2 = (torch.zeros(B, D, **bf) for _ in range(4))
    h, lse, ws = torch.zeros(B, FF, **bf), torch.zeros(B, 16, device=dev), torch.zeros(8, B, D, device=dev)
    for l in range(nl):
        st = L.DecStep()
        st.x, st.y = rows[l % 2].data_ptr(), rows[(l + 1) % 2].data_ptr()
        st.qkv_cache, st.kvc, st.mem_bias = c1[l].data_ptr(), kvc[l].data_ptr(), bias.data_ptr()
        st.attn, st.x1, st.qc, st.x2, st.h, st.lse = (t.data_ptr() for t in (attn, x1, qc, x2, h, lse))
        st.ffn_ws = ws.data_ptr()
        st.B, st.pos, st.cap, st.L = B, pos, cap, Lm
        L.check(L.lib().mmst_decoder_layer_step(C.byref(params[l]), C.byref(st), NH, L.stream_ptr()), "layer_step")
    want = rows[nl % 2].float()
    # one launch
    c2 = [c.clone() for c in caches]
    y = torch.zeros(B, D, **bf)
    s = L.DecStackStep()
    s.x, s.y, s.mem_bias = x.data_ptr(), y.data_ptr(), bias.data_ptr()
    for l in range(nl):
        s.qkv_cache[l], s.kvc[l] = c2[l].data_ptr(), kvc[l].data_ptr()
    s.B, s.pos, s.cap, s.L = B, pos, cap, Lm
    L.check(L.lib().mmst_decoder_stack_step(nl, params, C.byref(s), NH, L.stream_ptr()), "stack_step")
    torch.cuda.synchronize()
    assert float((y.float() - want).abs().max()) < 5e-2
    for l in range(nl):
        assert float((c1[l].float() - c2[l].float()).abs().max()) < (2e-2 if l == 0 else 5e-2)   # l = 0: one bf16 ulp


def test_stack_step_rejects_bad_arguments():
    s = L.DecStackStep()
    p = (L.DecParams * 1)()
    lib = L.lib()
    assert lib.mmst_decoder_stack_step(1, p, C.byref(s), NH, None) != 0          # empty step
    s.B, s.pos, s.cap, s.L = 2, 4, 4, 8                                             # pos beyond the cache
    assert lib.mmst_decoder_stack_step(1, p, C.byref(s), NH, None) != 0
    assert lib.mmst_decoder_stack_step(9, p, C.byref(s), NH, None) != 0          # too many layers
    assert lib.mmst_decoder_stack_step(1, p, C.byref(s), 4, None) != 0           # head count


@pytest.mark.parametrize("W", [2, 3, 4])
def test_grouped_stack_step_equals_one_workgroup_per_sequence(W):
    """The beams of a molecule in ONE workgroup (weights and memory keys/values read once for all of them) against
    one workgroup per sequence, on the same inputs; then the same step with the cached prefixes scattered over the
    molecule's sequences and a history table that says where each position's row lives."""
    dev, mols, Lm, cap, pos, nl = "cuda", 5, 301, 48, 21, 3
    B = mols * W
    gen = torch.Generator(device=dev).manual_seed(40 + W)
    layers = [_layer_params(gen, dev) for _ in range(nl)]
    params = _params_array(layers)
    r = lambda *s: torch.randn(*s, generator=gen, device=dev)  # noqa: E731
    V, VP = 43, 64
    embed, pos_row = r(V, D) * 0.5, r(D) * 0.5
    tok = torch.randint(0, V, (B,), generator=gen, device=dev)
    w_out = torch.zeros(VP, D, device=dev)
    w_out[:V] = r(V, D) * 0.2
    w_out, b_out = w_out.bfloat16(), torch.zeros(VP, device=dev)
    caches = [(r(B, cap, 3 * D) * 0.7).bfloat16() for _ in range(nl)]
    kvc = [(r(mols, Lm, 2 * D) * 0.7).bfloat16() for _ in range(nl)]
    bias = torch.zeros(mols, Lm, device=dev)
    bias[:, 17:23] = 1.0
    bias[1, 100:] = float("-inf")

    def run(cs, rows_per_group, hist=None):
        y = torch.zeros(B, D, dtype=torch.bfloat16, device=dev)
        logits = torch.zeros(B, VP, device=dev)
        s = L.DecStackStep()
        s.tok, s.embed, s.pos_row, s.y = tok.data_ptr(), embed.data_ptr(), pos_row.data_ptr(), y.data_ptr()
        for l in range(nl):
            s.qkv_cache[l], s.kvc[l] = cs[l].data_ptr(), kvc[l].data_ptr()
        s.mem_bias = bias.data_ptr()
        s.w_out, s.b_out, s.logits, s.ld_logits, s.n_out, s.V = w_out.data_ptr(), b_out.data_ptr(), logits.data_ptr(), VP, VP, V
        s.B, s.pos, s.cap, s.L, s.kv_group, s.rows_per_group = B, pos, cap, Lm, W, rows_per_group
        s.hist = hist.data_ptr() if hist is not None else None
        L.check(L.lib().mmst_decoder_stack_step(nl, params, C.byref(s), NH, L.stream_ptr()), "stack_step")
        torch.cuda.synchronize()
        return y.float(), logits

    c_ref = [c.clone() for c in caches]
    y_ref, lg_ref = run(c_ref, 1)
    c_grp = [c.clone() for c in caches]
    y_grp, lg_grp = run(c_grp, 0)
    # same operands and rounding points per row; the projections sum in another order (matrix cores against packed
    # dot products) and the online-softmax rescales are batched differently: agreement to a bf16 ulp per stage
    assert float((y_grp - y_ref).abs().max()) < 3e-2
    assert float((lg_grp - lg_ref).abs().max()) < 3e-2
    for l in range(nl):
        assert float((c_grp[l].float() - c_ref[l].float()).abs().max()) < (2e-2 if l == 0 else 4e-2)
    # history: position t of sequence (m, w) is stored in the cache of sequence (m, perm[m, w, t])
    perm = torch.stack([torch.stack([torch.randperm(W, generator=gen, device=dev) for _ in range(cap)], dim=1)
                        for _ in range(mols)])                                  # [mols][W][cap]: slot <- (w, t)
    hist = perm.reshape(B, cap).to(torch.int32).contiguous()
    scattered = []
    for l in range(nl):
        c = caches[l].view(mols, W, cap, 3 * D)
        sc = torch.zeros_like(c)
        idx = perm.unsqueeze(-1).expand(-1, -1, -1, 3 * D)
        sc.scatter_(1, idx, c)           # sc[m, perm[m, w, t], t] = c[m, w, t]
        scattered.append(sc.view(B, cap, 3 * D).contiguous())
    y_h, lg_h = run(scattered, 0, hist)
    assert float((y_h - y_grp).abs().max()) == 0.0 and float((lg_h - lg_grp).abs().max()) == 0.0
    # the new row goes to the sequence's OWN slot at `pos`
    for l in range(nl):
        assert torch.equal(scattered[l][:, pos], c_grp[l][:, pos])


def test_beam_step_kernel_follows_the_reference_bookkeeping():
    """mmst_beam_step against a host transcription of the reference's beam_search_step (validate_generate_MMT_v15_4.py
    :995-1053) on random logits over several steps: sequences, lengths, double-precision scores, probabilities, the
    tokens fed to the next decoder step and the history tables."""
    dev, B, W, V, cap, eos, sos = "cuda", 7, 3, 43, 16, 2, 3
    gen = torch.Generator(device=dev).manual_seed(9)
    lib = L.lib()

    def state():
        t = dict(seqs=torch.zeros(B, W, cap + 1, dtype=torch.int32, device=dev),
                 probs=torch.zeros(B, W, cap, device=dev), scores=torch.zeros(B, W, dtype=torch.float64, device=dev),
                 lens=torch.zeros(B, W, dtype=torch.int32, device=dev), nbeams=torch.zeros(B, dtype=torch.int32, device=dev),
                 hist=torch.zeros(B * W, cap, dtype=torch.int32, device=dev))
        st = L.BeamState()
        for k, v in t.items():
            setattr(st, k, v.data_ptr())
        return t, st
    (ta, sa), (tb, sb) = state(), state()
    tok = torch.zeros(B * W, dtype=torch.long, device=dev)
    L.check(lib.mmst_beam_init(C.byref(sa), C.c_void_p(tok.data_ptr()), B, W, cap, sos, L.stream_ptr()), "beam_init")
    beams = [[(1, [sos], [])] for _ in range(B)]
    slots = [[0] for _ in range(B)]                 # slot of every reference beam in the device arrays
    cur, nxt = (ta, sa), (tb, sb)
    for pos in range(10):
        logits = torch.randn(B * W, 64, generator=gen, device=dev) * 2.0
        logits[:, eos] += 1.5                       # make finished beams common
        L.check(lib.mmst_beam_step(C.c_void_p(logits.data_ptr()), 64, V, C.byref(cur[1]), C.byref(nxt[1]),
                                   C.c_void_p(tok.data_ptr()), B, W, cap, pos, eos, L.stream_ptr()), "beam_step")
        torch.cuda.synchronize()
        probs = torch.softmax(logits[:, :V], dim=1).cpu()
        for i in range(B):
            new_beam = []
            for j, (score, seq, pr) in enumerate(beams[i]):
                if seq[-1] == eos:
                    new_beam.append((score, seq, pr, j))
                    continue
                tp, ti = torch.topk(probs[i * W + j], W)
                for k in range(W):
                    p = float(tp[k])
                    new_beam.append((score * p, seq + [int(ti[k])], pr + [p], j))
            new_beam.sort(key=lambda x: x[0], reverse=True)
            new_beam = new_beam[:W]
            got_len = nxt[0]["lens"][i].tolist()
            for w, (score, seq, pr, parent) in enumerate(new_beam):
                assert got_len[w] == len(seq)
                assert nxt[0]["seqs"][i, w, :len(seq)].tolist() == seq, (pos, i, w)
                assert abs(float(nxt[0]["scores"][i, w]) - score) <= 1e-5 * abs(score)
                np_ = nxt[0]["probs"][i, w, :len(pr)].tolist()
                assert all(abs(a - b) < 1e-5 for a, b in zip(np_, pr))
                assert int(tok[i * W + w]) == seq[-1]
                h = nxt[0]["hist"][i * W + w].tolist()
                assert h[pos] == parent and h[:pos] == cur[0]["hist"][i * W + parent, :pos].tolist()
            beams[i] = [(b[0], b[1], b[2]) for b in new_beam]
            assert int(nxt[0]["nbeams"][i]) == W
        cur, nxt = nxt, cur
    assert any(b[1][-1] == eos for bm in beams for b in bm)


def _sample(logits, V, seed, counter, temperature=1.0):
    B = logits.shape[0]
    tok = torch.full((B,), -1, dtype=torch.int64, device="cuda")
    prob = torch.zeros(B, dtype=torch.float32, device="cuda")
    L.check(L.lib().mmst_sample_select(C.c_void_p(logits.data_ptr()), logits.shape[1], V, B, 1, C.c_float(temperature),
                                       C.c_uint32(seed), C.c_uint32(counter), C.c_void_p(tok.data_ptr()),
                                       C.c_void_p(prob.data_ptr()), L.stream_ptr()), "mmst_sample_select")
    return tok, prob


def test_sample_select_distribution_and_reproducibility():
    """mmst_sample_select against torch.softmax: 2048 draws per row (counter 0..2047) have the frequencies of the
    row's distribution (4.5 standard deviations of a binomial proportion), the returned probability is that of the
    chosen token, an index of probability zero is never drawn, and a draw is a pure function of (seed, counter, row)."""
    gen = torch.Generator(device="cuda").manual_seed(3)
    V, B, N = 43, 24, 2048
    logits = torch.randn(B, 64, generator=gen, device="cuda") * 2.0
    logits[:, 5] = -float("inf")              # masked out: never drawn
    logits[3, :V] = -float("inf")
    logits[3, 17] = 0.5                       # a one-point distribution
    logits[:, V:] = 50.0                      # padding columns beyond the vocabulary are not part of it
    T = 0.7
    p = torch.softmax(logits[:, :V] / T, dim=1)
    counts = torch.zeros(B, V, device="cuda")
    for c in range(N):
        tok, prob = _sample(logits, V, 1234, c, T)
        assert int(tok.min()) >= 0 and int(tok.max()) < V
        counts.scatter_add_(1, tok.unsqueeze(1), torch.ones(B, 1, device="cuda"))
        if c < 4:
            assert torch.allclose(prob, p.gather(1, tok.unsqueeze(1)).squeeze(1), atol=1e-5)
    freq = (counts / N).cpu()
    pc = p.cpu()
    sigma = torch.sqrt(pc * (1 - pc) / N)
    assert bool(((freq - pc).abs() <= 4.5 * sigma + 1e-3).all()), float(((freq - pc).abs() - 4.5 * sigma).max())
    assert float(freq[:, 5].max()) == 0.0 and float(freq[3, 17]) == 1.0
    a, _ = _sample(logits, V, 99, 7, T)
    b, _ = _sample(logits, V, 99, 7, T)
    c_, _ = _sample(logits, V, 99, 8, T)
    d, _ = _sample(logits, V, 100, 7, T)
    assert torch.equal(a, b) and not torch.equal(a, c_) and not torch.equal(a, d)
    # rows are independent: the same logits in every row do not give the same token in every row
    same = logits[:1].expand(B, 64).contiguous()
    t, _ = _sample(same, V, 5, 0, T)
    assert len(set(t.tolist())) > 1
    with pytest.raises(L.MMSTError):
        L.check(L.lib().mmst_sample_select(C.c_void_p(logits.data_ptr()), 64, 65, B, 1, C.c_float(1.0), C.c_uint32(0),
                                           C.c_uint32(0), C.c_void_p(a.data_ptr()), C.c_void_p(prob.data_ptr()),
                                           L.stream_ptr()), "mmst_sample_select")
