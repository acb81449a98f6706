"""The CPU oracle against fixtures produced by the reference implementation itself
(oracle/make_golden.py).  fp32 tolerance: logits atol 2e-5 (measured thread-count noise of
the reference is 1.2e-6)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import mmt_oracle as O
from tests.util import apply_trained, GOLDEN, digest, digest_close, load_golden

CASES = [
    ("fwd_c2_L2", "1H_HSQC_COSY_IR_MF_MW"),
    ("fwd_c1_L2", "1H"),
    ("fwd_all_L1", "1H_13C_HSQC_COSY_IR_MF_MW"),
    ("fwd_ms_L1", "1H_13C_HSQC_COSY_IR_MF_MS_MW"),
    ("fwd_nomf_L1", "1H_HSQC_COSY_IR"),
    ("fwd_c2_L6", "1H_HSQC_COSY_IR_MF_MW"),
]


def _setup(g, mode):
    nl = int(g["nl"])
    cfg = O.default_cfg(training_mode=mode, num_encoder_layers=nl, num_decoder_layers=nl)
    sd = O.make_state_dict(cfg, int(g["wseed"]))
    batch = O.synthetic_batch(int(g["B"]), int(g["seed"]), mode)
    return cfg, sd, batch


@pytest.mark.parametrize("name,mode", CASES)
def test_forward_matches_reference(name, mode):
    g = load_golden(name)
    cfg, sd, batch = _setup(g, mode)
    with torch.no_grad():
        loss, logits, fp, memory, mask = O.training_forward(sd, cfg, batch)
        mem2, emb_src, mask2, fp2 = O.forward(sd, cfg, batch, None)
    np.testing.assert_allclose(logits.numpy(), g["logits"], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(fp.numpy(), g["fingerprint"], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(mask.float().numpy(), g["mask"])
    digest_close(digest(memory, 512), g["memory"], 1e-4, 2e-5)
    digest_close(digest(emb_src, 512), g["embedding_src"], 1e-5, 1e-6)
    assert abs(float(loss) - float(g["loss"])) < 1e-5


@pytest.mark.parametrize("name,mode", CASES[:5])
def test_gradients_match_reference(name, mode):
    g = load_golden(name)
    cfg, sd, batch = _setup(g, mode)
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    loss = O.training_forward(sd, cfg, batch)[0]
    loss.backward()
    nograd = set(str(k) for k in g["nograd"])
    checked = 0
    for k, p in sd.items():
        if k in nograd:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        digest_close(digest(p.grad), g["grad/" + k], 2e-3, 1e-7)
        checked += 1
    assert checked > 50


def test_use_real_data_branch_matches_reference():
    g = load_golden("fwd_rd_L1")
    nl = int(g["nl"])
    cfg = O.default_cfg(training_mode="1H_HSQC_COSY_IR_MF_MW", num_encoder_layers=nl, num_decoder_layers=nl,
                        use_real_data=True)
    sd = {k: v.clone().requires_grad_(True) for k, v in O.make_state_dict(cfg, int(g["wseed"])).items()}
    batch = O.synthetic_batch(int(g["B"]), int(g["seed"]), cfg["training_mode"])
    loss, logits, fp, memory, mask = O.training_forward(sd, cfg, batch)
    np.testing.assert_allclose(logits.detach().numpy(), g["logits"], atol=2e-5, rtol=1e-4)
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    loss.backward()
    for k in ("real_data_linear.weight", "real_data_linear.bias", "fc_out.weight", "encoder_cross.layers.0.linear1.weight"):
        digest_close(digest(sd[k].grad), g["grad/" + k], 2e-3, 1e-7)


def test_adamw_two_steps():
    g = load_golden("fwd_c2_L2")
    cfg, sd, batch = _setup(g, "1H_HSQC_COSY_IR_MF_MW")
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    m = {k: torch.zeros_like(v) for k, v in sd.items()}
    v = {k: torch.zeros_like(v_) for k, v_ in sd.items()}
    losses = []
    for step in (1, 2):
        for p in params.values():
            p.grad = None
        loss = O.training_forward(params, cfg, batch)[0]
        loss.backward()
        losses.append(float(loss))
        with torch.no_grad():
            O.adamw_step(params, {k: p.grad for k, p in params.items()}, m, v, step, cfg["lr_pretraining"])
    np.testing.assert_allclose(losses, g["step_losses"], atol=2e-5)
    for k in g.files:
        if k.startswith("post/"):
            digest_close(digest(params[k[5:]]), g[k], 1e-5, 1e-6)


def test_full_size_step_matches_reference():
    """The benchmark configuration at its real size (64 molecules, 6 + 6 layers) against the reference's own step:
    loss, per-molecule losses, logits digest, every parameter gradient, one AdamW step."""
    g = load_golden("step_c2_full")
    cfg, sd, batch = _setup(g, "1H_HSQC_COSY_IR_MF_MW")
    assert int(g["B"]) == 64 and int(g["nl"]) == 6
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    loss, logits, fp, memory, mask = O.training_forward(params, cfg, batch)
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    assert tuple(logits.shape) == tuple(g["logits_shape"])
    digest_close(digest(logits, 4096), g["logits_digest"], 1e-4, 2e-5)
    V = logits.shape[-1]
    trg = batch["trg_enc_SMI"].t()
    per_pos = torch.nn.functional.cross_entropy(logits.detach().reshape(-1, V), trg[1:, :].reshape(-1), reduction="none")
    np.testing.assert_allclose(per_pos.reshape(logits.shape[0], -1).mean(0).numpy(), g["loss_per_molecule"], atol=2e-5)
    np.testing.assert_allclose(fp.detach().numpy(), g["fingerprint"], atol=2e-5, rtol=1e-4)
    nograd = set(str(k) for k in g["nograd"])
    checked = 0
    for k, p in params.items():
        if k in nograd:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        digest_close(digest(p.grad), g["grad/" + k], 2e-3, 1e-7)
        checked += 1
    assert checked > 400
    m = {k: torch.zeros_like(v) for k, v in sd.items()}
    v = {k: torch.zeros_like(v_) for k, v_ in sd.items()}
    with torch.no_grad():
        O.adamw_step(params, {k: p.grad for k, p in params.items()}, m, v, 1, cfg["lr_pretraining"])
    for k in g.files:
        if k.startswith("post/"):
            digest_close(digest(params[k[5:]]), g[k], 1e-5, 1e-6)


GREEDY = [("greedy_c2_L2", "1H_HSQC_COSY_IR_MF_MW"), ("greedy_c1_L2", "1H"), ("greedy_c2_L6", "1H_HSQC_COSY_IR_MF_MW")]


@pytest.mark.parametrize("name,mode", GREEDY)
def test_greedy_tokens_identical(name, mode):
    """Reference greedy_sequence on a model with confident decisions (fitted readout, see make_golden.train_readout):
    tokens, probabilities and SMILES strings identical end to end, stop rule included."""
    g = load_golden(name)
    nl = int(g["nl"])
    cfg = O.default_cfg(training_mode=mode, num_encoder_layers=nl, num_decoder_layers=nl)
    sd = apply_trained(O.make_state_dict(cfg, int(g["wseed"]), logit_gain=float(g["gain"])), g)
    batch = O.synthetic_batch(int(g["B"]), int(g["seed"]), mode)
    with torch.no_grad():
        memory, _, mask, fp = O.forward(sd, cfg, batch, None)
        toks, probs, margins = O.greedy_sequence(sd, cfg, memory, mask)
    assert np.array_equal(toks.numpy(), g["tokens"])
    assert float(g["margins"].min()) > 2.0 and len(toks) < cfg["max_len"]   # separated decisions; stopped on all-<PAD>
    np.testing.assert_allclose(probs.numpy(), g["probs"], atol=1e-5)
    itos = json.load(open(os.path.join(GOLDEN, "itos.json")))
    assert O.tensor_to_smiles(toks, itos) == [str(s) for s in g["smiles"]]


def test_clip_head_matches_reference():
    """Oracle CLIP head + loss against the reference's CLIPModel.forward (models_CLIP_v15_4.py:243-302) with the
    ChemBERTa features given: loss, per-sample losses, gradients."""
    g = load_golden("clip_c2_L1")
    mode, nl, B = "1H_HSQC_COSY_IR_MF_MW", int(g["nl"]), int(g["B"])
    cfg = O.default_cfg(training_mode=mode, num_encoder_layers=nl, num_decoder_layers=nl)
    sd = {k: v.requires_grad_(True) for k, v in O.make_state_dict(cfg, int(g["wseed"])).items()}
    head = {k: v.requires_grad_(True) for k, v in O.make_clip_head_state_dict(int(g["hseed"])).items()}
    batch = O.synthetic_batch(B, int(g["seed"]), mode)
    rng = np.random.Generator(np.random.PCG64(int(g["seed"]) + 1000))
    cb = torch.from_numpy(rng.standard_normal((B, O.CLIP_DIMS["CB_embedding"])).astype(np.float32))
    loss, losses = O.clip_forward(sd, head, cfg, batch, cb, 1.0)
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) < 1e-4 * float(g["loss"])
    np.testing.assert_allclose(losses.detach().numpy(), g["losses"], rtol=1e-3, atol=1e-4)
    checked = 0
    for k in g.files:
        if not k.startswith("grad/"):
            continue
        name = k[5:]
        p = head[name] if name in head else sd[name[len("MT_model."):]]
        digest_close(digest(p.grad), g[k], 5e-3, 1e-6)
        checked += 1
    assert checked >= 15


def test_beam_search_identical():
    """Oracle beam search against the reference's beams (validate_generate_MMT_v15_4.py:1056)."""
    g = load_golden("beam_c2_L2")
    mode, nl = "1H_HSQC_COSY_IR_MF_MW", int(g["nl"])
    cfg = O.default_cfg(training_mode=mode, num_encoder_layers=nl, num_decoder_layers=nl)
    sd = O.make_state_dict(cfg, int(g["wseed"]), logit_gain=float(g["gain"]))
    batch = O.synthetic_batch(int(g["B"]), int(g["seed"]), mode)
    with torch.no_grad():
        memory, _, mask, fp = O.forward(sd, cfg, batch, None)
        beams, margins = O.beam_search(sd, cfg, memory, mask, int(g["beam_size"]), int(g["gen_len"]))
    for i, beam in enumerate(beams):
        for j, (score, seq, probs) in enumerate(beam):
            assert seq == g["seqs"][i, j, :g["lens"][i, j]].tolist(), (i, j)
            assert abs(score - g["scores"][i, j]) <= 1e-4 * g["scores"][i, j]
            assert len(probs) == len(seq) - 1
    np.testing.assert_allclose(margins, g["margins"], rtol=1e-2, atol=1e-5)


def test_tokenizer_and_padding():
    t = json.load(open(os.path.join(GOLDEN, "text.json")))
    stoi = json.load(open(os.path.join(GOLDEN, "stoi.json")))
    itos = json.load(open(os.path.join(GOLDEN, "itos.json")))
    stoi_MF = json.load(open(os.path.join(GOLDEN, "stoi_MF.json")))
    for smi, toks, ids in zip(t["smiles"], t["tokens"], t["ids"]):
        assert O.tokenize_smiles(smi) == toks
        assert O.encode_smiles(stoi, smi) == ids
    for f, ids in zip(t["formulas"], t["mf_ids"]):
        assert O.encode_MF(stoi_MF, f) == ids
    assert O.tensor_to_smiles(torch.tensor(t["decode_input"]), itos) == t["decoded"]
    for key, (data, padded, mask) in t["pads"].items():
        dims = int(key[3])
        if dims == 1 and len(data) == 0:
            continue
        p, m = O.zero_pad(data, 64, dims)
        np.testing.assert_allclose(p.numpy(), np.array(padded, np.float32), rtol=1e-6)
        assert m.tolist() == mask
