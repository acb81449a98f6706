"""The LightningModule-shaped wrapper and the torch-facing module surface on the GPU: in-training generation
(TransformerMultiGPU._generate_sequences), validation_step, a plain torch optimizer around the autograd path,
and the guards of the decode drivers.  Fixtures come from the reference (oracle/make_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import mmt_oracle as O
from tests.test_model_gpu import GREEDY, build, call, make_config
from tests.util import GOLDEN, digest, digest_close, load_golden

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from multimodalspectraltransformer_amd import _lib as L
    from multimodalspectraltransformer_amd import generate as gen
    from multimodalspectraltransformer_amd.train import TransformerMultiGPU, run_training_MMT

MODE = "1H_HSQC_COSY_IR_MF_MW"


def _wrapper_on(g, mode):
    model, config, cfg, sd, batch, db = build(g, mode, gain=float(g["gain"]))
    tm = TransformerMultiGPU.__new__(TransformerMultiGPU)   # the fitted model of the fixture, not fresh weights
    tm.config, tm.model = config, model
    stoi = json.load(open(os.path.join(GOLDEN, "stoi.json")))
    itos = json.load(open(os.path.join(GOLDEN, "itos.json")))
    db["src_HSQC_"], db["src_COSY_"] = db["src_HSQC"], db["src_COSY"]
    memory, mask, trg, fp, _, _ = gen.run_model(model, db, config)
    return tm, model, config, memory, mask, stoi, itos


@pytest.mark.parametrize("name,mode", GREEDY[:2])
def test_generate_sequences_greedy_vs_reference(name, mode):
    """_generate_sequences(model, config, memory, mask, max_len, temperature, stoi, itos), sampling_method greedy:
    the reference's token tensor (incl. <SOS>, no early stop), SMILES strings and confidence lengths, identical."""
    g = load_golden(name)
    tm, model, config, memory, mask, stoi, itos = _wrapper_on(g, mode)
    config.sampling_method = "greedy"
    n = int(g["gen/len"])
    toks, smiles, conf = tm._generate_sequences(model, config, memory, mask, n, 1, stoi, itos)
    assert model.training   # the reference leaves the model in train mode (:1268)
    assert np.array_equal(toks.cpu().numpy(), g["gen/tokens"])
    assert smiles == [str(s_) for s_ in g["gen/smiles"]]
    assert [len(c) for c in conf] == g["gen/conf_len"].tolist()
    np.testing.assert_allclose([float(c[0]) for c in conf], g["gen/conf0"], atol=2e-2)


def test_generate_sequences_multinomial_and_mix():
    """The two sampling quirks of the reference (models_MMT_v15_4.py:1241, :1258-1264, :1294): "multinomial" samples
    every step from the distribution of position 0 given <SOS> (the decoder input never grows); "mix" returns the
    greedy tokens with the multinomial confidences cut to the greedy strings' lengths."""
    name, mode = GREEDY[0]
    g = load_golden(name)
    tm, model, config, memory, mask, stoi, itos = _wrapper_on(g, mode)
    n, B = int(g["gen/len"]), int(g["B"])
    p0 = g["gen/p0"]
    config.sampling_method = "multinomial"
    torch.manual_seed(5)
    toks, smiles, conf = tm._generate_sequences(model, config, memory, mask, 128, 1, stoi, itos)
    toks = toks.cpu().numpy()
    assert toks.shape == (129, B) and (toks[0] == stoi["<SOS>"]).all() and len(smiles) == B
    counts = np.stack([np.bincount(toks[1:, b], minlength=p0.shape[1]) for b in range(B)]) / 128.0
    # 128 draws per molecule from p0: frequencies within 4.5 standard deviations of a binomial proportion
    sigma = np.sqrt(p0 * (1 - p0) / 128.0)
    assert (np.abs(counts - p0) <= 4.5 * sigma + 2e-2).all()
    assert not np.array_equal(toks[1:n + 1], g["gen/tokens"][1:])   # not the greedy continuation
    # the confidences are p0[token]: the position-0 distribution, whatever was sampled before
    for b in range(B):
        c = conf[b].cpu().numpy()
        np.testing.assert_allclose(c, p0[b, toks[1:1 + len(c), b]], atol=2e-2)
    config.sampling_method = "mix"
    toks2, smiles2, conf2 = tm._generate_sequences(model, config, memory, mask, n, 1, stoi, itos)
    assert np.array_equal(toks2.cpu().numpy(), g["gen/tokens"])
    assert smiles2 == [str(s_) for s_ in g["gen/smiles"]]
    assert [len(c) for c in conf2] == g["gen/conf_len"].tolist()
    assert all(float(c.min()) > 0 and float(c.max()) <= 1 for c in conf2 if len(c))
    # the fitted model is confident: a token sampled from the greedy-prefix distribution is nearly always the greedy
    # one, so the multinomial confidences are close to the greedy ones
    assert np.median([float(c.mean()) for c in conf2 if len(c)]) > 0.8


def test_training_step_generates_with_the_weights_of_its_forward_pass(monkeypatch, tmp_path):
    """The reference generates between the forward pass and the loss (models_MMT_v15_4.py:1064-1077): with the weights
    the step STARTED with, on the train-mode memory of that step.  The step's generation (optimizer behind it) must
    return what _generate_sequences returns for a twin of the model that never takes the optimizer step, on both step
    paths (one call into the library / launch by launch); and the weights after the step are those of the step without
    generation."""
    name, mode = GREEDY[0]
    g = load_golden(name)
    stoi = json.load(open(os.path.join(GOLDEN, "stoi.json")))
    itos = json.load(open(os.path.join(GOLDEN, "itos.json")))

    def wrapper(gen_on):
        model, config, cfg, sd, batch, db = build(g, mode, gain=float(g["gain"]))
        config.model_save_dir = str(tmp_path)
        tm = TransformerMultiGPU.__new__(TransformerMultiGPU)
        tm.__dict__.update(config=config, model=model, lr=1e-3, stoi=stoi, itos=itos, optimizer=None, scheduler=None,
                           reducer=None, world=1, global_step=0, current_epoch=0, last_train_loss=None)
        config.gen_SMI_sequence, config.sampling_method = gen_on, "greedy"
        model._engine.step_seed = 4242
        return tm, model, config, db

    outs = []
    for c_step in ("1", "0"):
        monkeypatch.setenv("MMST_C_STEP", c_step)
        tm, model, config, db = wrapper(True)
        out = tm.training_step(dict(db), 0)
        toks, smiles, conf = out["gen"]
        outs.append((toks.cpu().clone(), smiles, float(out["loss"]), {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}))
        assert model.training
    assert torch.equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1]
    # the twin: same weights, same dropout stream, forward in train mode only, then the reference-signature generation
    tm, model, config, db = wrapper(False)
    eng = model._engine
    model.train()
    eng.step_seed = (4242 * 1664525 + 1013904223) & 0xFFFFFFFF
    db["src_HSQC_"], db["src_COSY_"] = db["src_HSQC"], db["src_COSY"]
    trg = db["trg_enc_SMI"].t()
    with torch.no_grad():
        logits, fp, memory, mask = call(model, mode, db, trg[:-1, :])
    config.sampling_method = "greedy"
    toks_t, smiles_t, conf_t = tm._generate_sequences(model, config, memory, mask, config.max_len, 1, stoi, itos)
    same = float((toks_t.cpu() == outs[0][0]).float().mean())
    assert same == 1.0, same
    assert smiles_t == outs[0][1]
    # ... and the optimizer step itself is the one of a step without generation
    monkeypatch.setenv("MMST_C_STEP", "1")
    tm, model, config, db = wrapper(False)
    out = tm.training_step(dict(db), 0)
    assert abs(float(out["loss"]) - outs[0][2]) < 1e-6
    post = model.state_dict()
    for k in ("decoder.layers.1.linear1.weight", "encoder_cross.layers.0.self_attn.in_proj_weight", "fc_out.weight",
              "embed_trg.weight"):
        assert float((post[k].cpu() - outs[0][3][k]).abs().max()) < 1e-6, k
        assert float((post[k].cpu() - outs[1][3][k]).abs().max()) < 2e-4, k


def test_validation_step_loss_vs_reference():
    g = load_golden("fwd_c2_L2")
    model, config, cfg, sd, batch, db = build(g, MODE)
    tm = TransformerMultiGPU.__new__(TransformerMultiGPU)
    tm.config, tm.model = config, model
    out = tm.validation_step(dict(db), 0)
    assert not model.training
    assert abs(float(out["loss"]) - float(g["loss"])) < 5e-3
    # the returned loss is a value of its own, not a view of the slot the next step overwrites
    keep = float(out["loss"])
    other = {k: v.flip(0) if k == "trg_enc_SMI" else v for k, v in db.items()}
    out2 = tm.validation_step(other, 1)
    assert float(out["loss"]) == keep and float(out2["loss"]) != keep
    assert out["loss"].data_ptr() != out2["loss"].data_ptr()


def test_torch_optimizer_around_autograd_path_vs_reference():
    """`zero_grad(); model(...); loss.backward(); optimizer.step()` with torch.optim.AdamW and its default
    zero_grad(set_to_none=True), two steps: step losses and post-step weights of the reference's own two AdamW steps
    (gradients must not pile up in the flat buffer), then eval-mode forward sees the updated weights."""
    g = load_golden("fwd_c2_L2")
    model, config, cfg, sd, batch, db = build(g, MODE)
    model.eval()   # the fixture was taken in eval mode (dropout off), gradients enabled
    opt = torch.optim.AdamW(model.parameters(), lr=cfg["lr_pretraining"])
    trg = db["trg_enc_SMI"].t()
    losses = []
    for step in range(2):
        opt.zero_grad()
        logits, fp, memory, mask = call(model, MODE, db, trg[:-1, :])
        loss = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.shape[-1]), trg[1:, :].reshape(-1))
        loss.backward()
        if step == 1:   # second-step gradients against the oracle at the current weights
            cur = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
            O.training_forward(cur, cfg, batch)[0].backward()
            for k in ("decoder.layers.1.linear1.weight", "encoder_cross.layers.0.self_attn.in_proj_weight",
                      "encoder_HSQC.layers.1.linear2.weight", "fc_out.weight", "embed_trg.weight"):
                a = dict(model.named_parameters())[k].grad.detach().cpu().reshape(-1).double()
                b = cur[k].grad.reshape(-1).double()
                assert float(a @ b / (a.norm() * b.norm())) > 0.99 and abs(float(a.norm() / b.norm()) - 1) < 0.1, k
        opt.step()
        losses.append(float(loss.detach()))
    np.testing.assert_allclose(losses, g["step_losses"], atol=6e-3)
    post = model.state_dict()
    for k in g.files:
        if k.startswith("post/"):
            digest_close(digest(post[k[5:]]), g[k], 2e-4, 2e-4)
    # eval-mode inference after an external optimizer step runs on the updated weights
    with torch.no_grad():
        l2 = call(model, MODE, db, trg[:-1, :])[0]
        want = O.training_forward({k: v.cpu() for k, v in post.items()}, cfg, batch)[1]
    assert float((l2.cpu() - want).abs().max()) < 4e-2
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(1.01)
        l3 = call(model, MODE, db, trg[:-1, :])[0]
    assert float((l3 - l2).abs().max()) > 1e-3


def test_epoch_mean_is_the_mean_of_the_step_losses(tmp_path):
    """run_training_MMT reports (and schedules the learning rate on) the mean of the epoch's step losses; the losses
    training_step returns are values of their own, not views of a slot that later steps overwrite."""
    config, cfg = make_config(MODE, 1, drop_out=0.0, model_save_dir=str(tmp_path), num_epochs=1)
    batches = [{k: v.cuda() for k, v in O.synthetic_batch(4, 60 + i, MODE).items()} for i in range(3)]
    torch.manual_seed(11)
    tm = TransformerMultiGPU(config).to("cuda")
    tm.configure_optimizers()
    kept = [tm.training_step(b, i)["loss"] for i, b in enumerate(batches)]
    vals = [float(x) for x in kept]
    assert len(set(round(v, 4) for v in vals)) == 3      # three different batches, three different losses
    logged = []
    torch.manual_seed(11)                                 # same initial weights, same batches, same dropout seeds
    run_training_MMT(config, lambda epoch: batches, num_epochs=1, log=logged.append)
    mean = float(logged[-1].split("loss")[1].split()[0])
    assert abs(mean - float(np.mean(vals))) < 5e-3, (mean, vals)
    assert abs(mean - vals[-1]) > 1e-3                    # in particular not the last step's loss


def test_decode_refuses_a_stale_memory():
    g = load_golden("fwd_c2_L2")
    model, config, cfg, sd, batch, db = build(g, MODE)
    stoi = json.load(open(os.path.join(GOLDEN, "stoi.json")))
    db["src_HSQC_"], db["src_COSY_"] = db["src_HSQC"], db["src_COSY"]
    mem_a, mask_a, *_ = gen.run_model(model, db, config)
    other = {k: v.flip(0) for k, v in db.items()}
    mem_b, mask_b, *_ = gen.run_model(model, other, config)
    short = type(config)(**dict(vars(config), max_len=6))
    with pytest.raises(L.MMSTError):
        gen.greedy_sequence(model, stoi, None, mem_a, mask_a, short)
    toks, _ = gen.greedy_sequence(model, stoi, None, mem_b, mask_b, short)
    assert toks.shape[1] == int(g["B"])
    # memory returned by forward() is decodable as well (forward registers its plan)
    model.eval()
    with torch.no_grad():
        mem_c, _, mask_c, _ = call(model, MODE, db, None)
    toks_c, _ = gen.greedy_sequence(model, stoi, None, mem_c, mask_c, short)
    assert toks_c.shape == toks.shape


def test_decoder_submodule_mask_handling_and_cache():
    """model.decoder(...): tgt_mask=None means no causal mask (last row of a one-layer decoder is unchanged, earlier
    rows are not), an arbitrary mask is refused, and the cached memory projections never serve another memory."""
    g = load_golden("fwd_all_L1")
    model, config, cfg, sd, batch, db = build(g, "1H_13C_HSQC_COSY_IR_MF_MW")
    model.eval()
    gen_ = torch.Generator().manual_seed(2)
    mem = torch.randn(40, 3, 128, generator=gen_).cuda()
    tgt = torch.randn(9, 3, 128, generator=gen_).cuda()
    with torch.no_grad():
        causal = model.decoder(tgt, mem, tgt_mask=model.generate_square_subsequent_mask(9))
        full = model.decoder(tgt, mem, tgt_mask=None)
        assert float((causal[-1] - full[-1]).abs().max()) < 1e-2
        assert float((causal[0] - full[0]).abs().max()) > 5e-2
        want = O.decoder(tgt.cpu(), mem.cpu(), sd, 1, 16, None)
        assert float((causal.cpu() - want).abs().max()) < 6e-2
        with pytest.raises(L.MMSTError):
            model.decoder(tgt, mem, tgt_mask=torch.zeros(9, 9).cuda())
        # same shape, new values written IN PLACE into the same tensor: the cached K/V projections must not be reused
        mem.copy_(torch.randn(40, 3, 128, generator=gen_))
        again = model.decoder(tgt, mem, tgt_mask=model.generate_square_subsequent_mask(9))
        want2 = O.decoder(tgt.cpu(), mem.cpu(), sd, 1, 16, None)
        assert float((again.cpu() - want2).abs().max()) < 6e-2
        # and a different tensor object of the same shape
        mem3 = torch.randn(40, 3, 128, generator=gen_).cuda()
        third = model.decoder(tgt, mem3, tgt_mask=model.generate_square_subsequent_mask(9))
        assert float((third.cpu() - O.decoder(tgt.cpu(), mem3.cpu(), sd, 1, 16, None)).abs().max()) < 6e-2


def test_timed_checkpoints_and_resume(tmp_path):
    """Wall-clock saves (models_MMT_v15_4.py:1196-1210: a .pth and a Lightning-layout .ckpt once model_save_interval
    seconds have passed) and resuming from the .ckpt: weights, AdamW moments, step count and learning rate come back,
    so the next step of the resumed run equals the next step of the run that wrote the file."""
    config, cfg = make_config(MODE, 1, drop_out=0.0, model_save_dir=str(tmp_path), model_save_interval=1e9)
    batch = {k: v.cuda() for k, v in O.synthetic_batch(4, 91, MODE).items()}
    torch.manual_seed(4)
    tm = TransformerMultiGPU(config).to("cuda")
    tm.configure_optimizers()
    tm.model._engine.step_seed = 99
    for i in range(2):
        tm.training_step(batch, i)
    assert tm.on_batch_end() is None                       # interval not reached: nothing written
    config.model_save_interval = 0.0
    pth, ckpt = tm.on_batch_end()
    assert pth.endswith(".pth") and ckpt.endswith(".ckpt") and "MultimodalTransformer_time_" in pth and "_Loss_" in pth
    raw = torch.load(ckpt, map_location="cpu", weights_only=True)
    assert raw["global_step"] == 2 and all(k.startswith("model.") for k in raw["state_dict"])
    plain = torch.load(pth, map_location="cpu", weights_only=True)
    assert list(plain.keys()) == [k[len("model."):] for k in raw["state_dict"]]
    config.model_save_interval = 1e9
    tm2 = TransformerMultiGPU.load_from_checkpoint(ckpt, config=config, resume_optimizer=True).to("cuda")
    assert tm2.restore_optimizer() and tm2.optimizer.step_count == 2 and tm2.global_step == 2
    seed = tm.model._engine.step_seed
    tm2.model._engine.step_seed = seed
    a = float(tm.training_step(batch, 2)["loss"])
    b = float(tm2.training_step(batch, 2)["loss"])
    assert abs(a - b) < 1e-4
    sa, sb = tm.model.state_dict(), tm2.model.state_dict()
    for k in ("decoder.layers.0.linear1.weight", "encoder_cross.layers.0.self_attn.in_proj_weight", "fc_out.weight",
              "embed_trg.weight"):
        assert float((sa[k] - sb[k]).abs().max()) < 1e-6, k


def test_teacher_forced_probabilities_and_multi_sample_multinomial():
    """predict_prop_correct_max_sequence (validate_generate_MMT_v15_4.py:309-431) as ONE teacher-forced pass against
    the oracle's prefix-by-prefix evaluation, and multinomial_sequence_multi_2 (mmt_result_test_functions_15_4.py:791)
    on the fitted fixture model: shapes of the reference's return values, probabilities within 2e-2, argmax tokens
    identical (every decision of that model is separated by more than 2 logit units)."""
    name, mode = GREEDY[0]
    g = load_golden(name)
    tm, model, config, memory, mask, stoi, itos = _wrapper_on(g, mode)
    model.eval()
    batch = O.synthetic_batch(int(g["B"]), int(g["seed"]), mode)
    trg = batch["trg_enc_SMI"]
    N, T = trg.shape[0], trg.shape[1] - 1
    out = gen.predict_prop_correct_max_sequence(model, stoi, memory, mask, trg.cuda(), 1, config)
    trg_tensor, corr, trg_max, max_prob, multinom = out
    assert trg_tensor.shape == (N, T) and corr.shape == (T, N) and trg_max.shape == (N, T) and max_prob.shape == (T, N)
    assert multinom.shape == (T * N,)
    assert torch.equal(trg_tensor.cpu(), trg[:, 1:])
    cfg = O.default_cfg(training_mode=mode, num_encoder_layers=int(g["nl"]), num_decoder_layers=int(g["nl"]))
    from tests.util import apply_trained
    sd = apply_trained(O.make_state_dict(cfg, int(g["wseed"])), g)
    with torch.no_grad():
        omem, _, omask, _ = O.forward(sd, cfg, batch, None)
        want = torch.softmax(O.decode_logits(sd, cfg, trg.t()[:-1, :], omem, omask), dim=2)    # (T, N, V)
    real = trg.t()[1:, :]
    np.testing.assert_allclose(corr.cpu().numpy(), want.gather(2, real.unsqueeze(-1)).squeeze(-1).numpy(), atol=2e-2)
    np.testing.assert_allclose(max_prob.cpu().numpy(), want.max(dim=2).values.numpy(), atol=2e-2)
    fitted = 57        # positions the fixture's readout was fitted on
    assert torch.equal(trg_max.cpu()[:, :fitted], want.argmax(dim=2).t()[:, :fitted])
    assert torch.equal(trg_max.cpu()[:, :fitted], trg[:, 1:1 + fitted])          # the fitted model predicts its targets
    toks, probs = gen.multinomial_sequence_multi_2(model, memory, mask, stoi, config)
    assert toks.shape == (config.max_len, N) and probs.shape == (config.max_len - 1, N)
    assert float(probs.min()) > 0 and float(probs.max()) <= 1
    # the draw is part of the token step's launch and a pure function of (seed, position, sequence): a seeded generator
    # gives the same tokens again, another seed other tokens; the per-layer token step with the stand-alone selection
    # launch draws the same tokens from its logits
    g1 = torch.Generator(device="cuda").manual_seed(77)
    ta, pa = gen.multinomial_sequence_multi_2(model, memory, mask, stoi, config, generator=g1)
    tb, pb = gen.multinomial_sequence_multi_2(model, memory, mask, stoi, config,
                                              generator=torch.Generator(device="cuda").manual_seed(77))
    tc, _ = gen.multinomial_sequence_multi_2(model, memory, mask, stoi, config,
                                             generator=torch.Generator(device="cuda").manual_seed(78))
    assert torch.equal(ta, tb) and torch.equal(pa, pb) and not torch.equal(ta, tc)
    os.environ["MMST_DECODE_FUSED"] = "0"
    try:
        td, _ = gen.multinomial_sequence_multi_2(model, memory, mask, stoi, config,
                                                 generator=torch.Generator(device="cuda").manual_seed(77))
    finally:
        del os.environ["MMST_DECODE_FUSED"]
    assert float((td[:fitted] == ta[:fitted]).float().mean()) > 0.97
    # a confident model: sampling mostly reproduces the greedy tokens (one unlucky draw sends a molecule elsewhere)
    assert float((toks.cpu()[:fitted] == torch.from_numpy(g["tokens"])[:fitted]).float().mean()) > 0.6


def _clip_setup(g, tmp_path):
    from multimodalspectraltransformer_amd.clip_functions import CLIPMultiGPU
    model, config, cfg, sd, batch, db = build(g, MODE)
    for k, v in dict(CB_embedding=1024, MT_embedding=512, projection_dim=512, dropout=0.1, CLIP_temperature=1,
                     CLIP_model_save_dir=str(tmp_path), model_save_interval=10 ** 9, CLIP_continue_training=False).items():
        setattr(config, k, v)
    config.CB_lr, config.MT_lr, config.MT_projection_lr, config.weight_decay = (float(v) for v in g["lrs"])

    E, B = 1024, int(g["B"])
    rng = np.random.Generator(np.random.PCG64(int(g["seed"]) + 1000))     # the generator's draws, in its order
    raw = rng.standard_normal((B, E)).astype(np.float32)
    ll1_w = (np.eye(E) + 0.02 * rng.standard_normal((E, E))).astype(np.float32)
    ll1_b = (0.02 * rng.standard_normal(E)).astype(np.float32)
    assert abs(float(ll1_w.sum()) - float(g["ll1_weight_sum"])) < 1e-3 * E

    class StandIn(torch.nn.Module):            # the fixture's stand-in for the SMILES model: one trainable layer on given features
        def __init__(self):
            super().__init__()
            self.LL1 = torch.nn.Linear(E, E)
            self.register_buffer("raw", torch.from_numpy(raw))

        def forward(self, ids, mask):
            return None, self.LL1(self.raw)

    cb = StandIn()
    cb.LL1.load_state_dict({"weight": torch.from_numpy(ll1_w), "bias": torch.from_numpy(ll1_b)})
    tm = CLIPMultiGPU(config, CB_model=cb, tokenizer=None, MT_model=model)
    head = O.make_clip_head_state_dict(int(g["hseed"]))
    tm.CLIP_model.load_state_dict(head, strict=False)
    return tm, db, cb


def test_clip_training_steps_vs_reference_fixture(tmp_path):
    """Three steps of the contrastive training loop (loss, backward through heads / fp1 / encoders / the caller's
    SMILES layer, three-group AdamW) against the same steps of the reference's CLIPModel under the optimizer CLIP_make
    builds (eval mode, scaled learning rates: oracle/make_golden.py clip_train_case).  Losses within 6 % (the loss is a
    log-softmax over dot products of 512-wide LayerNorm rows, see test_clip_head_vs_reference_fixture); the weight
    movement of every sampled tensor points the way the reference's does (Adam's steps are sign-like, so the cosine
    of the sampled deltas is the meaningful figure); parameters without a gradient stay untouched."""
    g = load_golden("clip_train_c2_L1")
    tm, db, cb = _clip_setup(g, tmp_path)
    clip = tm.CLIP_model
    before = {k: v.detach().clone() for k, v in clip.state_dict().items()}
    ref = g["step_losses"]
    losses = []
    for i in range(int(g["steps"])):
        out = tm.training_step(db, i, deterministic=True)
        losses.append(float(out["loss"]))
    final = float(tm.validation_step(db)["loss"])
    losses.append(final)
    print("clip step losses", losses, "reference", ref.tolist())
    np.testing.assert_allclose(losses, ref, rtol=0.06)
    assert losses[0] > losses[1] > losses[2] > losses[3]
    after = clip.state_dict()
    for k in ("MT_projection.fc.weight", "CB_projection.projection.weight", "MT_projection.layer_norm.bias",
              "MT_model.fp1.weight", "MT_model.encoder_cross.layers.0.linear1.weight",
              "MT_model.encoder_1H.layers.0.self_attn.in_proj_weight", "CB_model.LL1.weight"):
        d_ref = (g["post/" + k] - g["pre/" + k])[3:]
        d_got = (digest(after[k]) - digest(before[k]))[3:]
        assert np.abs(d_ref).max() > 0
        c = float(d_ref @ d_got / (np.linalg.norm(d_ref) * np.linalg.norm(d_got) + 1e-30))
        print(k, "delta cosine", round(c, 4), "size", np.abs(d_got).max(), np.abs(d_ref).max())
        assert c > 0.9, (k, c)
        assert 0.7 < np.abs(d_got).max() / np.abs(d_ref).max() < 1.4, k
    for k in ("MT_model.decoder.layers.0.linear1.weight", "MT_model.fc_out.weight", "MT_model.embed_trg.weight"):
        assert torch.equal(after[k].cpu(), before[k].cpu()), k


def test_clip_epoch_loop_saves_and_resumes(tmp_path):
    """run_training_CLIP: epoch loop over weight-sorted batches, per-epoch .pth / .ckpt in the reference's naming,
    the written configuration, the plateau scheduler on the epoch mean, and a resume from the .ckpt."""
    from multimodalspectraltransformer_amd.clip_functions import CLIPMultiGPU, run_training_CLIP
    g = load_golden("clip_train_c2_L1")
    tm0, db, cb = _clip_setup(g, tmp_path)
    config = tm0.config
    seen = []
    tm = run_training_CLIP(lambda e: [db, db], lambda e: [db], config, CB_model=cb, MT_model=tm0.CLIP_model.MT_model,
                           num_epochs=2, log=seen.append)
    assert len(seen) == 2 and "train_loss" in seen[0] and "test_loss" in seen[0], seen
    files = sorted(os.listdir(tmp_path))
    assert sum(f.startswith("MultimodalCLIP_Epoch_") and f.endswith(".pth") for f in files) == 2, files
    assert sum(f.startswith("MultimodalCLIP_Epoch_") and f.endswith(".ckpt") for f in files) == 2, files
    assert any(f.startswith("config_") and f.endswith(".json") for f in files)
    assert tm.global_step == 4
    ck = [f for f in files if f.startswith("MultimodalCLIP_Epoch_1_") and f.endswith(".ckpt")][0]
    sd = torch.load(os.path.join(tmp_path, ck), map_location="cpu", weights_only=True)
    assert all(k.startswith("CLIP_model.") for k in sd["state_dict"])
    assert {"CLIP_model.MT_model.fp1.weight", "CLIP_model.MT_projection.fc.weight", "CLIP_model.CB_model.LL1.weight"} <= set(sd["state_dict"])
    config.CLIP_continue_training, config.CLIP_model_path = True, os.path.join(tmp_path, ck)
    tm2 = CLIPMultiGPU.load_from_checkpoint(config.CLIP_model_path, config=config, CB_model=cb)
    a, b = tm.CLIP_model.state_dict(), tm2.CLIP_model.state_dict()
    for k in ("MT_model.fp1.weight", "MT_projection.fc.weight", "MT_model.encoder_cross.layers.0.linear1.weight"):
        assert torch.equal(a[k].cpu(), b[k].cpu()), k
    assert tm2.current_epoch == 1
