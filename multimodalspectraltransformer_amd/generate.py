"""Inference drivers: encoder half, greedy / multinomial decode.

Counterparts of utils_MMT/validate_generate_MMT_v15_4.py (run_model :95, greedy_sequence :723,
multinomial_sequence :841) and of _generate_sequences (models_MMT_v15_4.py:1222-1297).  The decode
loop keeps the reference's outputs (stop when every next token is <PAD>) but not its cost: the reference
re-runs the decoder on the whole prefix for every token; here each step computes only the new position
(mmst_decoder_layer_step: self-attention keys/values cached per layer, cross-attention keys/values of the
memory projected once) and selects the token on the device.  MMST_DECODE_CACHED=0 (or cached=False) selects
the prefix re-run, kept as the cross-check of the cached path.
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


class _StepState:
    """Per-plan buffers and prebuilt structs of the cached decode: one Q|K|V cache per decoder layer, two
    ping-pong row buffers, shared scratch."""

    def __init__(self, model, pl):
        from . import ops
        dev, B, cap = pl.cv.buf[torch.bfloat16].device, pl.B, pl.cap_T
        ndl = model.num_decoder_layers
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
            s.kvc = _P(pl.dec_acts[l]["kvc"])
            s.mem_bias = _P(pl.kbias_glob)
            for f in ("attn", "x1", "qc", "x2", "h", "lse"):
                setattr(s, f, getattr(self, f).data_ptr())
            s.ffn_ws = self.ws.data_ptr()
            s.B, s.pos, s.cap, s.L = B, 0, cap, pl.Lmem
            self.steps.append(s)
        self.out = self.rows[ndl % 2]
        self.ops = ops
        self.graphs = {}   # (position, temperature) -> captured token step


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


def _decode_plan_cached(model, pl, max_len, temperature, sos, early_stop, sample=False, generator=None):
    """Decode loop with cached keys/values: per token one new row through the decoder.

    With MMST_DECODE_GRAPH=1 greedy decoding replays one hipGraph per target position (captured on first use, kept
    with the plan): a token is ~50 small launches whose arguments depend only on the position.  Measured on an
    MI355X at 64 molecules x 128 tokens the replay is no faster than issuing the launches directly (67 ms either
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
    _cross_kv(model, pl)
    emb, pe = ar.p32("embed_trg.weight"), ar.p32("pe_trg.weight")
    use_graph = (not sample) and os.environ.get("MMST_DECODE_GRAPH", "0") == "1"

    def enqueue(pos, select):
        L.check(lib.mmst_trg_embed_fwd(C.c_void_p(tok[pos].data_ptr()), B, C.c_void_p(emb), C.c_void_p(pe + 4 * D * pos),
                                       C.c_void_p(stt.rows[0].data_ptr()), B, 1, C.c_float(0.0), C.c_uint32(0), st()),
                "mmst_trg_embed_fwd")
        for l, s in enumerate(stt.steps):
            s.pos = pos
            L.check(lib.mmst_decoder_layer_step(C.byref(pl.dec_p[l]), C.byref(s), 16, st()), "mmst_decoder_layer_step")
        stt.ops.gemm([dict(P=stt.out, Q=ar.pb("fc_out.weight"), C=stt.logits, bias=ar.p32("fc_out.bias"),
                           ldp=D, ldq=D, ldc=VPAD, Pn=B, Qn=VPAD, K=D, flags=L.GF_OUT_F32)], L.EPI_BIAS)
        if select:
            L.check(lib.mmst_greedy_select(C.c_void_p(stt.logits.data_ptr()), VPAD, model.out_size, B, 1,
                                           C.c_float(float(temperature)), C.c_void_p(tok[pos + 1].data_ptr()),
                                           C.c_void_p(prob[pos].data_ptr()), st()), "mmst_greedy_select")

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
        else:
            enqueue(pos, not sample)
            if sample:
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


def _decode_plan(model, pl, max_len, temperature, sos, early_stop, sample=False, generator=None, cached=None):
    """Shared loop on an encoded plan.  Returns tokens (steps+1, B) incl. <SOS> and probs (steps, B)."""
    if cached is None:
        cached = os.environ.get("MMST_DECODE_CACHED", "1") != "0"
    if cached:
        return _decode_plan_cached(model, pl, max_len, temperature, sos, early_stop, sample, generator)
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
    """Reference signature.  `memory` must be the memory of the model's last run_model()/forward() call
    (it is held in the plan buffers); tokens (steps, B) without <SOS> and probabilities with the first
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


def generate_sequences(model, pl, config, stoi, itos, temperature=1):
    """In-training generation (models_MMT_v15_4.py:1222-1297), greedy branch: no early stop, max_len steps."""
    sos = stoi["<SOS>"] if stoi else 3
    toks, probs = _decode_plan(model, pl, config.max_len, temperature, sos, early_stop=False)
    smiles, conf = (None, None)
    if itos:
        smiles, conf = hf.tensor_to_smiles_and_prob_2(toks[1:], probs, itos)
    return toks, smiles, conf


def _plan_for(model, memory):
    pl = getattr(model, "_last_plan", None)
    if pl is None or memory.shape[1] != pl.B or memory.shape[0] != pl.Lmem:
        raise L.MMSTError("greedy decode needs the plan of the preceding run_model()/forward() call")
    return pl
