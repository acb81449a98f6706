"""Inference drivers: encoder half, greedy / multinomial decode, beam search.

Counterparts of utils_MMT/validate_generate_MMT_v15_4.py (run_model :95, greedy_sequence :723,
multinomial_sequence :841, beam_search :1056) and of _generate_sequences (models_MMT_v15_4.py:1222-1297).  The decode
loop keeps the reference's outputs (stop when every next token is <PAD>) but not its cost: the reference
re-runs the decoder on the whole prefix for every token; here each step computes only the new position
(self-attention keys/values cached per layer, cross-attention keys/values of the memory projected once) and
selects the token on the device, one launch per token (mmst_decoder_stack_step; MMST_DECODE_FUSED=0 runs the
same step as one mmst_decoder_layer_step per layer).  MMST_DECODE_CACHED=0 (or cached=False) selects the prefix
re-run, kept as the cross-check of the cached path.
"""
import ctypes as C
import os

import torch

from . import _lib as L
from . import helper_functions as hf
from .engine import VPAD, _P


def run_model(model, data_dict, config):
    """Encoder half for a collated batch: returns (memory, src_padding_mask, trg_enc_SMI, fingerprint,
    src_HSQC, src_COSY) like the reference, plus keeps the encoded plan on model._last_plan."""
    x = data_dict
    model.eval()
    with torch.no_grad():
        pl = model.run_encoder_half(x, T=0)
    model._last_plan = pl
    memory, mask, fingerprint = model.outputs(pl)
    dev = model._dev()
    src_HSQC = x.get('src_HSQC') if "HSQC" in config.training_mode else x.get('src_HSQC_')
    src_COSY = x.get('src_COSY') if "COSY" in config.training_mode else x.get('src_COSY_')
    return memory, mask, x['trg_enc_SMI'].to(dev), fingerprint, src_HSQC, src_COSY


D = 128
# measurement hook (bench.py): a list that receives a (start, end) pair of torch.cuda.Event(enable_timing=True) around
# every decoder-stack launch of the device beam search; None = off
STEP_EVENTS = None


class _StepState:
    """Per-plan buffers and prebuilt structs of the cached decode: one Q|K|V cache per decoder layer, two
    ping-pong row buffers, shared scratch.  rep > 1 (beam search): every molecule owns `rep` consecutive
    sequences, which read copies of its memory keys/values and key bias (refresh_memory)."""

    def __init__(self, model, pl, rep=1):
        from . import ops
        dev, B, cap = pl.cv.buf[torch.bfloat16].device, pl.B * rep, pl.cap_T
        ndl = model.num_decoder_layers
        self.rep, self.B = rep, B
        self.copies = rep > 1 and os.environ.get("MMST_DECODE_FUSED", "1") == "0"   # only the per-layer step needs them
        if self.copies:
            self.kvc = [torch.empty(B, pl.Lmem, 2 * D, dtype=torch.bfloat16, device=dev) for _ in range(ndl)]
            self.mem_bias = torch.empty(B, pl.Lmem, dtype=torch.float32, device=dev)
        bf = dict(dtype=torch.bfloat16, device=dev)
        self.cache = [torch.zeros(B, cap, 3 * D, **bf) for _ in range(ndl)]
        self.rows = [torch.zeros(B, D, **bf) for _ in range(2)]
        self.attn, self.x1, self.qc, self.x2 = (torch.zeros(B, D, **bf) for _ in range(4))
        self.h = torch.zeros(B, 2048, **bf)
        self.lse = torch.zeros(B, 16, dtype=torch.float32, device=dev)
        self.ws = torch.zeros(8, B, D, dtype=torch.float32, device=dev)
        self.logits = torch.zeros(B, VPAD, dtype=torch.float32, device=dev)
        self.steps = []
        for l in range(ndl):
            s = L.DecStep()
            s.x, s.y = self.rows[l % 2].data_ptr(), self.rows[(l + 1) % 2].data_ptr()
            s.qkv_cache = self.cache[l].data_ptr()
            s.kvc = self.kvc[l].data_ptr() if self.copies else _P(pl.dec_acts[l]["kvc"])
            s.mem_bias = self.mem_bias.data_ptr() if self.copies else _P(pl.kbias_glob)
            for f in ("attn", "x1", "qc", "x2", "h", "lse"):
                setattr(s, f, getattr(self, f).data_ptr())
            s.ffn_ws = self.ws.data_ptr()
            s.B, s.pos, s.cap, s.L = B, 0, cap, pl.Lmem
            self.steps.append(s)
        self.out = self.rows[ndl % 2]
        # the same position through all layers in one launch (mmst_decoder_stack_step)
        self.params = (type(pl.dec_p[0]) * ndl)(*pl.dec_p)
        ar = model._engine.arena
        ss = self.stack = L.DecStackStep()
        for l in range(ndl):
            ss.qkv_cache[l] = self.cache[l].data_ptr()
            ss.kvc[l] = _P(pl.dec_acts[l]["kvc"])     # the sequences of a molecule share its memory rows (kv_group)
        ss.mem_bias = _P(pl.kbias_glob)
        ss.kv_group = rep
        ss.rows_per_group = 1      # host-ranked beam search and greedy decode: one workgroup per sequence
        ss.embed = ar.p32("embed_trg.weight")
        ss.y = self.out.data_ptr()
        ss.w_out, ss.b_out = ar.pb("fc_out.weight"), ar.p32("fc_out.bias")
        ss.logits, ss.ld_logits, ss.n_out = self.logits.data_ptr(), VPAD, VPAD
        ss.V = model.out_size
        ss.B, ss.pos, ss.cap, ss.L = B, 0, cap, pl.Lmem
        # workspace of the split token step (MMST_DEC_STACK_WS_BYTES): bf16 rows, then 16 fp32 partial slabs
        self.stack_ws = torch.zeros(((B * D * 2 + 255) // 256 * 256 + 16 * B * D * 4 + 3) // 4, dtype=torch.float32, device=dev)
        ss.ffn_ws = self.stack_ws.data_ptr()
        self.ndl = ndl
        self.ops = ops
        self.graphs = {}   # (position, temperature) -> captured token step

    def stack_step(self, tok_ptr, pe_ptr, pos, next_tok=None, next_prob=None, temperature=1.0):
        """One position through the whole stack: embedding of the tokens at tok_ptr, all layers, logits, and (if
        next_tok is given) the greedy choice."""
        ss = self.stack
        ss.tok, ss.pos_row, ss.pos = tok_ptr, pe_ptr + 4 * D * pos, pos
        ss.next_tok, ss.next_prob, ss.temperature = next_tok, next_prob, float(temperature)
        L.check(L.lib().mmst_decoder_stack_step(self.ndl, self.params, C.byref(ss), 16, L.stream_ptr()),
                "mmst_decoder_stack_step")

    def refresh_memory(self, model, pl):
        """Copies of the projected memory keys/values and of the key bias for each of a molecule's sequences (for
        the per-layer token step; the one-launch step reads the molecule's rows directly)."""
        if not self.copies:
            return
        for l in range(model.num_decoder_layers):
            src = pl.cv.view(pl.dec_acts[l]["kvc"], pl.B, pl.Lmem, 2 * D)
            self.kvc[l].view(pl.B, self.rep, pl.Lmem, 2 * D).copy_(src.unsqueeze(1).expand(-1, self.rep, -1, -1))
        kb = pl.cv.view(pl.kbias_glob, pl.B, pl.Lmem)
        self.mem_bias.view(pl.B, self.rep, pl.Lmem).copy_(kb.unsqueeze(1).expand(-1, self.rep, -1))


def _cross_kv(model, pl):
    """K|V projection of the encoder memory for every decoder layer, one grouped launch."""
    from . import ops
    ar = model._engine.arena
    groups = []
    for l in range(model.num_decoder_layers):
        k = f"decoder.layers.{l}.multihead_attn."
        groups.append(dict(P=_P(pl.memory), Q=ar.pb(k + "in_proj_weight") + 2 * D * D, C=_P(pl.dec_acts[l]["kvc"]),
                           bias=ar.p32(k + "in_proj_bias") + 4 * D, ldp=D, ldq=D, ldc=2 * D,
                           Pn=pl.B * pl.Lmem, Qn=2 * D, K=D))
    ops.gemm(groups, L.EPI_BIAS)


def _decode_plan_cached(model, pl, max_len, temperature, sos, early_stop, sample=False, generator=None,
                        keep_logits=None, kv_ready=False):
    """Decode loop with cached keys/values: per token one new row through the decoder.

    With MMST_DECODE_GRAPH=1 greedy decoding replays one hipGraph per target position (captured on first use, kept
    with the plan): a token is ~50 small launches whose arguments depend only on the position.  Measured on an
    MI355X at 64 molecules x 128 tokens the replay is no faster than issuing the launches directly (52 ms either
    way: the token step is bound by the latency of its chain of small kernels, not by the host), so it is off by
    default.  The all-<PAD> stop test needs the tokens on the host, so it runs every 8th position; positions
    decoded past the stop are cut off, which gives exactly the reference's output."""
    eng = model._engine
    ar = eng.arena
    B = pl.B
    if max_len > pl.cap_T:
        raise L.MMSTError(f"decode length {max_len} exceeds the plan's capacity of {pl.cap_T} target positions")
    lib, st = L.lib(), L.stream_ptr
    stt = getattr(pl, "_step_state", None)
    if stt is None:
        stt = pl._step_state = _StepState(model, pl)
    tok = pl.cv.view(pl.tok, model.max_len + 1, B)
    prob = pl.cv.view(pl.next_prob, model.max_len + 1, B)
    tok[0].fill_(sos)
    was_training = model.training
    model.eval()
    if not kv_ready:   # kv_ready: the plan's last forward pass has projected this memory with the current weights
        _cross_kv(model, pl)
    emb, pe = ar.p32("embed_trg.weight"), ar.p32("pe_trg.weight")
    use_graph = (not sample) and os.environ.get("MMST_DECODE_GRAPH", "0") == "1"
    # multinomial decode: the draw is part of the token step's launch (mmst_sample_select's rule: a pure function of
    # seed, position and sequence); the seed of a decode comes from the caller's generator (torch's default one
    # without), so seeded callers get the same tokens again.  MMST_SAMPLE_TORCH=1: softmax / multinomial / gather in
    # torch per position instead.
    sample_dev = sample and os.environ.get("MMST_SAMPLE_TORCH", "0") != "1"
    seed = 0
    if sample_dev:
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,), generator=generator,
                                 device=generator.device if generator is not None else "cpu"))

    fused = os.environ.get("MMST_DECODE_FUSED", "1") != "0"

    def enqueue(pos, select):
        if fused:
            # this position's logits go straight to row `pos` of the caller's [max_len][B][64] buffer -- but never under
            # graph capture or replay: a captured launch keeps the pointer it was captured with, so there the logits
            # always land in the plan's own buffer and are copied out behind the replay
            direct = keep_logits is not None and not use_graph
            if direct:
                stt.stack.logits = keep_logits[pos].data_ptr()
            stt.stack.sample, stt.stack.sample_seed = int(sample_dev), seed
            stt.stack_step(tok[pos].data_ptr(), pe, pos, tok[pos + 1].data_ptr() if select else None,
                           prob[pos].data_ptr() if select else None, temperature)
            stt.stack.logits = stt.logits.data_ptr()
            return
        L.check(lib.mmst_trg_embed_fwd(C.c_void_p(tok[pos].data_ptr()), B, C.c_void_p(emb), C.c_void_p(pe + 4 * D * pos),
                                       C.c_void_p(stt.rows[0].data_ptr()), B, 1, C.c_float(0.0), C.c_uint32(0), st()),
                "mmst_trg_embed_fwd")
        for l, s in enumerate(stt.steps):
            s.pos = pos
            L.check(lib.mmst_decoder_layer_step(C.byref(pl.dec_p[l]), C.byref(s), 16, st()), "mmst_decoder_layer_step")
        stt.ops.gemm([dict(P=stt.out, Q=ar.pb("fc_out.weight"), C=stt.logits, bias=ar.p32("fc_out.bias"),
                           ldp=D, ldq=D, ldc=VPAD, Pn=B, Qn=VPAD, K=D, flags=L.GF_OUT_F32)], L.EPI_BIAS)
        if select and sample_dev:
            L.check(lib.mmst_sample_select(C.c_void_p(stt.logits.data_ptr()), VPAD, model.out_size, B, 1,
                                           C.c_float(float(temperature)), C.c_uint32(seed), C.c_uint32(pos),
                                           C.c_void_p(tok[pos + 1].data_ptr()), C.c_void_p(prob[pos].data_ptr()), st()),
                    "mmst_sample_select")
        elif select:
            L.check(lib.mmst_greedy_select(C.c_void_p(stt.logits.data_ptr()), VPAD, model.out_size, B, 1,
                                           C.c_float(float(temperature)), C.c_void_p(tok[pos + 1].data_ptr()),
                                           C.c_void_p(prob[pos].data_ptr()), st()), "mmst_greedy_select")
        if keep_logits is not None and not use_graph:
            keep_logits[pos].copy_(stt.logits)

    steps = 0
    for t in range(1, max_len + 1):
        pos = t - 1
        if use_graph and pos > 0:       # position 0 runs directly: it also loads every kernel before a capture
            key = (pos, float(temperature))
            g = stt.graphs.get(key)
            if g is None:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    enqueue(pos, True)
                stt.graphs[key] = g
            g.replay()
            if keep_logits is not None:
                keep_logits[pos].copy_(stt.logits)
        else:
            enqueue(pos, (not sample) or sample_dev)
            if use_graph and keep_logits is not None:
                keep_logits[pos].copy_(stt.logits)
            if sample and not sample_dev:
                p = torch.softmax(stt.logits[:, :model.out_size] / temperature, dim=1)
                nxt = torch.multinomial(p, 1, generator=generator).squeeze(1)
                tok[t].copy_(nxt)
                prob[t - 1].copy_(p.gather(1, nxt.unsqueeze(1)).squeeze(1))
        steps = t
        if early_stop and (t % 8 == 0 or t == max_len):
            done = (tok[1:t + 1] == 0).all(dim=1).nonzero()
            if done.numel():
                steps = int(done[0]) + 1
                break
    model.train(was_training)
    return tok[:steps + 1].clone(), prob[:steps].clone()


def _decode_plan(model, pl, max_len, temperature, sos, early_stop, sample=False, generator=None, cached=None,
                 keep_logits=None, kv_ready=False):
    """Shared loop on an encoded plan.  Returns tokens (steps+1, B) incl. <SOS> and probs (steps, B).
    keep_logits: optional fp32 [max_len][B][64] buffer that receives every position's logits (cached path only)."""
    if cached is None:
        cached = os.environ.get("MMST_DECODE_CACHED", "1") != "0"
    if keep_logits is not None and not cached:
        raise L.MMSTError("keep_logits needs the cached decode")
    if cached:
        return _decode_plan_cached(model, pl, max_len, temperature, sos, early_stop, sample, generator, keep_logits,
                                   kv_ready)
    eng = model._engine
    B = pl.B
    lib, st = L.lib(), L.stream_ptr
    tok = pl.cv.view(pl.tok, model.max_len + 1, B)
    prob = pl.cv.view(pl.next_prob, model.max_len + 1, B)
    tok[0].fill_(sos)
    steps = 0
    was_training = model.training
    model.eval()
    for t in range(1, max_len + 1):
        eng.decode(pl, False, T=t, kv_ready=1 if t > 1 else 0)
        if not sample:
            L.check(lib.mmst_greedy_select(C.c_void_p(_P(pl.logits)), VPAD, model.out_size, B, t,
                                           C.c_float(float(temperature)), C.c_void_p(tok[t].data_ptr()),
                                           C.c_void_p(prob[t - 1].data_ptr()), st()), "mmst_greedy_select")
        else:
            lg = pl.cv.view(pl.logits, B, model.max_len if False else t, VPAD)[:, t - 1, :model.out_size]
            p = torch.softmax(lg / temperature, dim=1)
            nxt = torch.multinomial(p, 1, generator=generator).squeeze(1)
            tok[t].copy_(nxt)
            prob[t - 1].copy_(p.gather(1, nxt.unsqueeze(1)).squeeze(1))
        steps = t
        if early_stop and bool((tok[t] == 0).all()):
            break
    model.train(was_training)
    return tok[:steps + 1].clone(), prob[:steps].clone()


def greedy_sequence(model, stoi, itos, memory, src_padding_mask, config):
    """Reference signature.  `memory` must come from run_model() / MultimodalTransformer.forward() (it is held in
    the plan buffers, see _plan_for); tokens (steps, B) without <SOS> and probabilities with the first
    row dropped, as validate_generate_MMT_v15_4.py:768-775 returns them."""
    pl = _plan_for(model, memory)
    toks, probs = _decode_plan(model, pl, config.max_len, config.temperature, stoi["<SOS>"], early_stop=True)
    return toks[1:, :], probs[1:, :]


def greedy_sequence_2(model, stoi, itos, memory, src_padding_mask, config):
    """mmt_result_test_functions_15_4.py:984-1032 variant: keeps every probability row."""
    pl = _plan_for(model, memory)
    toks, probs = _decode_plan(model, pl, config.max_len, config.temperature, stoi["<SOS>"], early_stop=True)
    return toks[1:, :], probs


def multinomial_sequence(model, stoi, memory, src_padding_mask, config, generator=None):
    pl = _plan_for(model, memory)
    toks, probs = _decode_plan(model, pl, config.max_len, config.temperature, stoi["<SOS>"], early_stop=False,
                               sample=True, generator=generator)
    return toks[1:], probs.transpose(0, 1)


def multinomial_sequence_multi_2(model, memory, src_padding_mask, stoi, config, generator=None):
    """mmt_result_test_functions_15_4.py:791-829: multinomial decode of max_len tokens for every column of `memory`
    (callers duplicate the batch to draw several samples per molecule); returns tokens without <SOS> (max_len, N)
    and the chosen tokens' probabilities with the first row dropped (max_len - 1, N), as the reference does."""
    pl = _plan_for(model, memory)
    toks, probs = _decode_plan(model, pl, config.max_len, config.temperature, stoi["<SOS>"], early_stop=False,
                               sample=True, generator=generator)
    return toks[1:, :], probs[1:, :]


def predict_prop_correct_max_sequence(model, stoi, memory, src_padding_mask, trg_enc_SMI, gen_num, config,
                                      generator=None):
    """validate_generate_MMT_v15_4.py:309-431: per-token probabilities under teacher forcing.  The reference re-runs
    the decoder on a growing prefix of the TRUE tokens for every position; under the causal mask the distribution it
    reads at step idx is row idx of ONE teacher-forced pass over [<SOS>, true tokens...], which is what runs here.
    Returns (trg_tensor (N, T), corr_token_prob (T, N), trg_tensor_max (N, T), max_token_prob (T, N),
    multinom_token_prob (T * N,)) with T = trg_enc_SMI.shape[1] - 1; gen_num is ignored as in the reference (:386)."""
    pl = _plan_for(model, memory)
    eng = model._engine
    B = pl.B
    trg = trg_enc_SMI.to(eng.device, torch.int64)
    T = trg.shape[1] - 1
    if T > pl.cap_T:
        raise L.MMSTError(f"target length {T} exceeds the plan's capacity of {pl.cap_T} positions")
    was_training = model.training
    model.eval()
    real = trg.t()[1:, :]                                     # (T, N) true next tokens
    tok = pl.cv.view(pl.tok, model.max_len + 1, B)
    tok[0].fill_(stoi["<SOS>"])
    tok[1:T].copy_(real[:T - 1])
    eng.decode(pl, False, T=T)
    logits = pl.cv.view(pl.logits, B, T, VPAD)[:, :, :model.out_size].transpose(0, 1)   # (T, N, V)
    p = torch.softmax(logits / config.temperature, dim=2)
    nxt = torch.argmax(p, dim=2)
    max_prob = p.gather(2, nxt.unsqueeze(-1)).squeeze(-1)
    corr = p.gather(2, real.unsqueeze(-1)).squeeze(-1)
    sampled = torch.multinomial(p.reshape(-1, model.out_size), 1, generator=generator)
    multinom = p.reshape(-1, model.out_size).gather(1, sampled).reshape(-1)
    model.train(was_training)
    return real.t().contiguous(), corr, nxt.t().contiguous(), max_prob, multinom


def generate_sequences(model, pl, config, stoi, itos, temperature=1, max_len=None, generator=None, kv_ready=False):
    """In-training generation, TransformerMultiGPU._generate_sequences (models_MMT_v15_4.py:1222-1297), on an encoded
    plan: max_len decode steps without early stop; returns the reference's (token tensor incl. <SOS>, SMILES list,
    list of per-molecule confidence tensors).  config.sampling_method:
      "greedy"      argmax decode (:1248-1256).
      "multinomial" the reference only ever feeds the GREEDY tensor to the decoder (:1241) and never grows it in
                    this mode, so every step samples from the distribution of position 0 given <SOS> alone (:1258-1264);
                    one decoder position is computed and sampled max_len times.
      "mix"         greedy decode; the multinomial tokens are sampled from the greedy-prefix distributions, and the
                    MULTINOMIAL confidences are the ones cut to the greedy strings' lengths (:1294); combine_gen_sims
                    then keeps the greedy result for every molecule (its except-branch needs Chem.MolFromSmiles to
                    raise, which it does not for strings: it returns None).
    kv_ready: the plan's last forward pass ran with the current weights, so the decoder layers' projections of the
    memory (keys/values) are already in the plan and are not recomputed.
    """
    sos = stoi["<SOS>"] if stoi else 3
    max_len = int(max_len or config.max_len)
    method = getattr(config, "sampling_method", "greedy")
    if method not in ("greedy", "multinomial", "mix"):
        raise ValueError(f"sampling_method {method!r}")
    V = model.out_size
    if method == "multinomial":
        keep = torch.empty(1, pl.B, VPAD, dtype=torch.float32, device=model._engine.device)
        _decode_plan(model, pl, 1, temperature, sos, early_stop=False, cached=True, keep_logits=keep, kv_ready=kv_ready)
        p0 = torch.softmax(keep[0, :, :V] / temperature, dim=1)
        nxt = torch.multinomial(p0, max_len, replacement=True, generator=generator)        # (B, max_len)
        toks = torch.cat((torch.full((1, pl.B), sos, dtype=torch.long, device=nxt.device), nxt.t()), dim=0)
        probs = p0.gather(1, nxt).t().contiguous()
    else:
        keep = None
        if method == "mix":
            keep = torch.empty(max_len, pl.B, VPAD, dtype=torch.float32, device=model._engine.device)
        toks, probs = _decode_plan(model, pl, max_len, temperature, sos, early_stop=False,
                                   cached=True if method == "mix" or kv_ready else None, keep_logits=keep,
                                   kv_ready=kv_ready)
        if method == "mix":
            p = torch.softmax(keep[:, :, :V] / temperature, dim=2)
            nxt = torch.multinomial(p.reshape(-1, V), 1, generator=generator)
            probs = p.reshape(-1, V).gather(1, nxt).reshape(max_len, pl.B)
    smiles, conf = (None, None)
    if itos:
        smiles, conf = hf.tensor_to_smiles_and_prob_2(toks[1:], probs, itos)
    return toks, smiles, conf


class _BeamBuffers:
    """Two alternating beam states (include/mmst.h: mmst_beam_state) for B molecules x W beams."""

    def __init__(self, dev, B, W, cap):
        i32 = dict(dtype=torch.int32, device=dev)
        self.seqs = [torch.zeros(B, W, cap + 1, **i32) for _ in range(2)]
        self.probs = [torch.zeros(B, W, cap, dtype=torch.float32, device=dev) for _ in range(2)]
        self.scores = [torch.zeros(B, W, dtype=torch.float64, device=dev) for _ in range(2)]
        self.lens = [torch.zeros(B, W, **i32) for _ in range(2)]
        self.nbeams = [torch.zeros(B, **i32) for _ in range(2)]
        self.hist = [torch.zeros(B * W, cap, **i32) for _ in range(2)]
        self.tok = torch.zeros(B * W, dtype=torch.long, device=dev)
        self.state = []
        for i in range(2):
            st = L.BeamState()
            st.seqs, st.probs, st.scores = self.seqs[i].data_ptr(), self.probs[i].data_ptr(), self.scores[i].data_ptr()
            st.lens, st.nbeams, st.hist = self.lens[i].data_ptr(), self.nbeams[i].data_ptr(), self.hist[i].data_ptr()
            self.state.append(st)
        self.graphs = {}


def _beam_search_device(model, pl, W, gen_len, sos, eos):
    """Beam search without a host round trip per step: per position ONE decoder launch for all molecules x beams (the
    beams of a molecule share a workgroup, the weights it streams and the molecule's memory keys/values) and ONE
    bookkeeping launch (mmst_beam_step).  A beam's prefix is never copied: its history table names the cache rows.
    The whole loop is captured in a hipGraph on first use and replayed afterwards (MMST_DECODE_GRAPH=0: plain
    launches)."""
    lib = L.lib()
    B, cap = pl.B, pl.cap_T
    key = "_beam_device_%d" % W
    stt = getattr(pl, key, None)
    if stt is None:
        stt = _StepState(model, pl, rep=W)
        stt.beam = _BeamBuffers(stt.logits.device, B, W, cap)
        setattr(pl, key, stt)
    bb = stt.beam
    ar = model._engine.arena
    pe = ar.p32("pe_trg.weight")
    V = model.out_size

    def run():
        _cross_kv(model, pl)
        L.check(lib.mmst_beam_init(C.byref(bb.state[0]), C.c_void_p(bb.tok.data_ptr()), B, W, cap, sos, L.stream_ptr()),
                "mmst_beam_init")
        ss = stt.stack
        ss.kv_group, ss.rows_per_group = W, 0
        for idx in range(gen_len):
            ss.hist = bb.hist[idx & 1].data_ptr()
            ev = None
            if STEP_EVENTS is not None and not torch.cuda.is_current_stream_capturing():
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            stt.stack_step(bb.tok.data_ptr(), pe, idx)
            if ev is not None:
                ev[1].record()
                STEP_EVENTS.append(ev)
            L.check(lib.mmst_beam_step(C.c_void_p(stt.logits.data_ptr()), VPAD, V, C.byref(bb.state[idx & 1]),
                                       C.byref(bb.state[(idx + 1) & 1]), C.c_void_p(bb.tok.data_ptr()), B, W, cap, idx,
                                       eos, L.stream_ptr()), "mmst_beam_step")
        ss.hist = None

    gkey = (gen_len, sos, eos, ar.weights_serial)
    g = bb.graphs.get(gkey)
    if g is not None:
        g.replay()
    else:
        run()   # eager: also loads every kernel before a capture
        if os.environ.get("MMST_DECODE_GRAPH", "1") != "0":
            torch.cuda.synchronize()
            out = [t[gen_len & 1].clone() for t in (bb.seqs, bb.lens, bb.scores, bb.probs)]
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    run()
                bb.graphs = {gkey: g}
            except Exception:      # capture is an optimisation: plain launches stay correct
                bb.graphs = {}
            for t, o in zip((bb.seqs, bb.lens, bb.scores, bb.probs), out):
                t[gen_len & 1].copy_(o)   # a capture executes nothing, but keep the eager results in any case
    f = gen_len & 1
    seqs, lens = bb.seqs[f].cpu().tolist(), bb.lens[f].cpu().tolist()
    scores, probs = bb.scores[f].cpu().tolist(), bb.probs[f].cpu().tolist()
    beams = []
    for i in range(B):
        beams.append([(scores[i][w], seqs[i][w][:lens[i][w]], probs[i][w][:lens[i][w] - 1]) for w in range(W)])
    return beams


def beam_search(model, stoi, memory, src_padding_mask, config, beam_size):
    """validate_generate_MMT_v15_4.py:1056-1094 with the reference's bookkeeping (:995-1053: de-duplication,
    finished beams carried over, score = product of the ranking probabilities, stable sort) but not its cost.
    The reference runs the decoder once per molecule, beam and step on the whole prefix; here all
    molecules x beams advance together by one cached position per step (mmst_decoder_layer_step on B * beam_size
    sequences), the top-k of the softmax is taken on the device, and only those k probabilities and indices per
    sequence go to the host, where the beams are re-ranked exactly like the reference does.  A beam inherits the
    key/value cache of the sequence it extends (one gather per layer and step).
    Returns beams[i] = list of (score, token list incl. <SOS>, probability list), best first."""
    pl = _plan_for(model, memory)
    eng, ar = model._engine, model._engine.arena
    B, W = pl.B, int(beam_size)
    sos, eos = stoi["<SOS>"], stoi["<EOS>"]
    gen_len = int(config.gen_len)
    if gen_len > pl.cap_T:
        raise L.MMSTError(f"gen_len {gen_len} exceeds the plan's capacity of {pl.cap_T} target positions")
    lib, st = L.lib(), L.stream_ptr
    fused = os.environ.get("MMST_DECODE_FUSED", "1") != "0"
    if fused and W <= L.MAX_BEAM and os.environ.get("MMST_BEAM_HOST", "0") != "1":
        was_training = model.training
        model.eval()
        out = _beam_search_device(model, pl, W, gen_len, sos, eos)
        model.train(was_training)
        return out
    key = "_beam_state_%d_%d" % (W, fused)
    stt = getattr(pl, key, None)
    if stt is None:
        stt = _StepState(model, pl, rep=W)
        setattr(pl, key, stt)
    was_training = model.training
    model.eval()
    _cross_kv(model, pl)
    stt.refresh_memory(model, pl)
    dev = stt.logits.device
    emb, pe = ar.p32("embed_trg.weight"), ar.p32("pe_trg.weight")
    tok = torch.zeros(B * W, dtype=torch.long, device=dev)
    beams = [[(1, [sos], [])] for _ in range(B)]
    for idx in range(gen_len):
        # slots of this step: the distinct, unfinished beams of every molecule (the reference's `seen` filter)
        slots = []            # per molecule: list of (slot or None for carried-over beams, beam)
        toks = [0] * (B * W)
        for i in range(B):
            seen, row = set(), []
            for beam in beams[i]:
                t = tuple(beam[1])
                if t in seen:
                    continue
                seen.add(t)
                if beam[1][-1] == eos:
                    row.append((None, beam))
                else:
                    slot = i * W + sum(1 for s_, _ in row if s_ is not None)
                    toks[slot] = beam[1][-1]
                    row.append((slot, beam))
            slots.append(row)
        tok.copy_(torch.tensor(toks, dtype=torch.long), non_blocking=False)
        if fused:
            stt.stack_step(tok.data_ptr(), pe, idx)
        else:
            L.check(lib.mmst_trg_embed_fwd(C.c_void_p(tok.data_ptr()), B * W, C.c_void_p(emb),
                                           C.c_void_p(pe + 4 * D * idx), C.c_void_p(stt.rows[0].data_ptr()), B * W, 1,
                                           C.c_float(0.0), C.c_uint32(0), st()), "mmst_trg_embed_fwd")
            for l, s in enumerate(stt.steps):
                s.pos = idx
                L.check(lib.mmst_decoder_layer_step(C.byref(pl.dec_p[l]), C.byref(s), 16, st()),
                        "mmst_decoder_layer_step")
            stt.ops.gemm([dict(P=stt.out, Q=ar.pb("fc_out.weight"), C=stt.logits, bias=ar.p32("fc_out.bias"),
                               ldp=D, ldq=D, ldc=VPAD, Pn=B * W, Qn=VPAD, K=D, flags=L.GF_OUT_F32)], L.EPI_BIAS)
        probs = torch.softmax(stt.logits[:, :model.out_size], dim=1)      # :1038, no temperature
        top_p, top_i = torch.topk(probs, W, dim=1)
        top_p, top_i = top_p.cpu().tolist(), top_i.cpu().tolist()
        parent = list(range(B * W))
        for i in range(B):
            new_beam, seen = [], set()
            for slot, (score, sequence, prob_sequence) in slots[i]:
                seen.add(tuple(sequence))
                if slot is None:
                    new_beam.append((score, sequence, prob_sequence, None))
                    continue
                for k in range(W):
                    new_sequence = sequence + [top_i[slot][k]]
                    t = tuple(new_sequence)
                    if t in seen:
                        continue
                    seen.add(t)
                    new_beam.append((score * top_p[slot][k], new_sequence, prob_sequence + [top_p[slot][k]], slot))
            new_beam.sort(key=lambda x: x[0], reverse=True)
            new_beam = new_beam[:W]
            beams[i] = [(b[0], b[1], b[2]) for b in new_beam]
            # next step's slot order = order of the unfinished beams in the new list (duplicates cannot occur here)
            nslot = i * W
            for b in new_beam:
                if b[1][-1] != eos:
                    parent[nslot] = b[3]
                    nslot += 1
        if idx + 1 < gen_len and parent != list(range(B * W)):
            pidx = torch.tensor(parent, dtype=torch.long, device=dev)
            for c in stt.cache:
                c[:, :idx + 1] = c.index_select(0, pidx)[:, :idx + 1]
    model.train(was_training)
    return beams


def _plan_for(model, memory):
    """The plan whose buffers hold `memory`.  Tensors handed out by MultimodalTransformer.forward()/run_model() carry
    their plan and the serial number of the encoder pass that produced them; a plan that has been re-encoded since
    (another batch of the same shape) no longer holds this memory, and decoding it would silently use the newer
    batch, so that is refused.  A tensor without the stamp (a clone, a copy on another device, a slice put together
    again) is accepted only if it IS the memory of the model's last plan: 64 sampled elements must equal the plan's
    (the reference's decode functions read the tensor they are given; here the tensor names the buffers to read, so one
    with other contents is refused instead of being ignored)."""
    stamp = getattr(memory, "_mmst_plan", None)
    if stamp is not None:
        pl, serial = stamp
        if pl.eng is not model._engine or pl.encode_serial != serial:
            raise L.MMSTError("this memory belongs to an earlier encoder pass: its plan buffers have been reused by a "
                              "later run_model()/forward() call of the same shape")
    else:
        pl = getattr(model, "_last_plan", None)
    if pl is None or memory.dim() != 3 or memory.shape[1] != pl.B or memory.shape[0] != pl.Lmem or memory.shape[2] != D:
        raise L.MMSTError("decoding needs the plan of the preceding run_model()/forward() call")
    if stamp is None:
        Lm, B = pl.Lmem, pl.B
        n = Lm * B * D
        idx = torch.linspace(0, n - 1, 64, device=pl.eng.device).long()
        l, b, d = idx // (B * D), (idx // D) % B, idx % D
        held = pl.cv.view(pl.memory, B, Lm, D)[b, l, d].float()
        given = memory.to(pl.eng.device)[l, b, d].float()
        if not bool(((held - given).abs() <= 1e-6 + 2e-3 * held.abs()).all()):
            raise L.MMSTError("this memory tensor is not the encoder output held by the model's last plan: decode "
                              "functions read the plan's buffers, so a tensor with other contents cannot be decoded "
                              "(run run_model()/forward() on the batch it belongs to first)")
    return pl
