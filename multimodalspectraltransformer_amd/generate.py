"""Inference drivers: encoder half, greedy / multinomial decode.

Counterparts of utils_MMT/validate_generate_MMT_v15_4.py (run_model :95, greedy_sequence :723,
multinomial_sequence :841) and of _generate_sequences (models_MMT_v15_4.py:1222-1297).  The decode
loop keeps the reference's algorithm (the whole prefix is re-run through the decoder every step;
stop when every next token is <PAD>), but runs it in the model's preallocated plan buffers with
the per-layer K/V projection of the memory computed once, and selects tokens on the device.
"""
import ctypes as C

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


def _decode_plan(model, pl, max_len, temperature, sos, early_stop, sample=False, generator=None):
    """Shared loop on an encoded plan.  Returns tokens (steps+1, B) incl. <SOS> and probs (steps, B)."""
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
