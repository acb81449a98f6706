"""Drop-in proof on the SUBMODULE surface: the decode loop of the reference's validation code
(utils_MMT/validate_generate_MMT_v15_4.py:738-775) reaches into the model -- embed_trg, pe_trg, dropout2,
generate_square_subsequent_mask, decoder(tgt, memory, tgt_mask=, memory_key_padding_mask=), fc_out -- instead of
calling a decode function.  `reference_style_greedy` below is this repository's own restatement of that call
sequence (prefix re-run, no key/value cache, softmax / argmax in torch); run against the package's model it must
produce the reference's tokens and SMILES of the fitted fixtures, and the package's own decode function must agree.
Also here: a memory tensor that has lost its plan stamp is decoded only if it still is the plan's memory."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.util import GOLDEN, load_golden
from tests.test_model_gpu import build

pytestmark = pytest.mark.gpu

from multimodalspectraltransformer_amd import generate as gen  # noqa: E402
from multimodalspectraltransformer_amd import helper_functions as hf  # noqa: E402
from multimodalspectraltransformer_amd import _lib as L  # noqa: E402


def reference_style_greedy(model, stoi, memory, src_padding_mask, config):
    """The loop of validate_generate_MMT_v15_4.py:738-775 through the model's submodules."""
    model.eval()
    N = memory.size(1)
    dev = memory.device
    seq = torch.full((1, N), stoi["<SOS>"], dtype=torch.long, device=dev)
    probs = []
    with torch.no_grad():
        for _ in range(config.max_len):
            n = seq.size(0)
            positions = torch.arange(n, device=dev).unsqueeze(1).expand(n, N)
            emb = model.dropout2(model.embed_trg(seq) + model.pe_trg(positions))
            tgt_mask = model.generate_square_subsequent_mask(n).to(dev)
            out = model.decoder(emb, memory, tgt_mask=tgt_mask, memory_key_padding_mask=src_padding_mask)
            out = model.fc_out(out)
            p = F.softmax(out / config.temperature, dim=2)
            nxt = torch.argmax(p[-1, :, :], dim=1)
            probs.append(p[-1, :, :].gather(1, nxt.unsqueeze(-1)).squeeze())
            seq = torch.cat((seq, nxt.unsqueeze(0)), dim=0)
            if (nxt == 0).all():
                break
    return seq[1:, :], torch.stack(probs)[1:, :]


@pytest.mark.parametrize("name,mode", [("greedy_c2_L2", "1H_HSQC_COSY_IR_MF_MW"), ("greedy_c1_L2", "1H")])
def test_reference_decode_loop_through_submodules(name, mode):
    g = load_golden(name)
    assert float(g["margins"].min()) > 2.0
    model, config, cfg, sd, batch, db = build(g, mode, gain=float(g["gain"]))
    stoi = json.load(open(os.path.join(GOLDEN, "stoi.json")))
    itos = json.load(open(os.path.join(GOLDEN, "itos.json")))
    db["src_HSQC_"], db["src_COSY_"] = db["src_HSQC"], db["src_COSY"]
    memory, mask, *_ = gen.run_model(model, db, config)
    toks, probs = reference_style_greedy(model, stoi, memory, mask, config)
    want = g["tokens"]
    assert tuple(toks.shape) == want.shape, (tuple(toks.shape), want.shape)
    assert np.array_equal(toks.cpu().numpy(), want)
    assert hf.tensor_to_smiles(toks, itos) == [str(s_) for s_ in g["smiles"]]
    assert np.abs(probs.float().cpu().numpy() - g["probs"]).max() < 2e-2
    # ... and the package's decode function (KV-cached) says the same
    toks2, _ = gen.greedy_sequence(model, stoi, itos, memory, mask, config)
    assert torch.equal(toks2.cpu(), toks.cpu())


def test_unstamped_memory_is_checked_against_the_plan():
    name, mode = "greedy_c2_L2", "1H_HSQC_COSY_IR_MF_MW"
    g = load_golden(name)
    model, config, cfg, sd, batch, db = build(g, mode, gain=float(g["gain"]))
    stoi = json.load(open(os.path.join(GOLDEN, "stoi.json")))
    itos = json.load(open(os.path.join(GOLDEN, "itos.json")))
    db["src_HSQC_"], db["src_COSY_"] = db["src_HSQC"], db["src_COSY"]
    memory, mask, *_ = gen.run_model(model, db, config)
    want, _ = gen.greedy_sequence(model, stoi, itos, memory, mask, config)
    clone = memory.clone()                      # same contents, no stamp: accepted
    assert not hasattr(clone, "_mmst_plan")
    got, _ = gen.greedy_sequence(model, stoi, itos, clone, mask, config)
    assert torch.equal(got, want)
    with pytest.raises(L.MMSTError):            # same shape, other contents: refused, not silently ignored
        gen.greedy_sequence(model, stoi, itos, memory.clone().mul_(1.01), mask, config)
    with pytest.raises(L.MMSTError):
        gen.greedy_sequence(model, stoi, itos, torch.cat([memory[:, 1:], memory[:, :1]], dim=1).contiguous(), mask, config)


def test_mix_then_greedy_then_mix_under_decode_graphs(monkeypatch):
    """Per-position decode graphs keep the pointers they were captured with: the logits a 'mix' generation keeps must
    not depend on whether an earlier call captured the graphs with or without a keep buffer."""
    monkeypatch.setenv("MMST_DECODE_GRAPH", "1")
    monkeypatch.setenv("MMST_DECODE_FUSED", "1")
    name, mode = "greedy_c2_L2", "1H_HSQC_COSY_IR_MF_MW"
    g = load_golden(name)
    model, config, cfg, sd, batch, db = build(g, mode, gain=float(g["gain"]))
    db["src_HSQC_"], db["src_COSY_"] = db["src_HSQC"], db["src_COSY"]
    memory, mask, *_ = gen.run_model(model, db, config)
    pl = gen._plan_for(model, memory)
    T = 12
    outs = []
    for keep in (True, False, True):
        buf = torch.full((T, pl.B, 64), float("nan"), device="cuda") if keep else None
        toks, _ = gen._decode_plan(model, pl, T, 1.0, 3, early_stop=False, keep_logits=buf)
        torch.cuda.synchronize()
        outs.append((toks.clone(), None if buf is None else buf.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][0], outs[2][0])
    assert torch.isfinite(outs[0][1][:, :, :43]).all() and torch.isfinite(outs[2][1][:, :, :43]).all()
    assert torch.equal(outs[0][1][:, :, :43], outs[2][1][:, :, :43])
    monkeypatch.setenv("MMST_DECODE_GRAPH", "0")
    buf = torch.zeros(T, pl.B, 64, device="cuda")
    toks, _ = gen._decode_plan(model, pl, T, 1.0, 3, early_stop=False, keep_logits=buf)
    assert torch.equal(toks, outs[0][0])
    assert float((buf[:, :, :43] - outs[0][1][:, :, :43]).abs().max()) < 1e-5
