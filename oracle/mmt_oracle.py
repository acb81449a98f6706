"""CPU oracle for the MultimodalTransformer forward/backward path.

TEST INFRASTRUCTURE ONLY.  This file is a plain fp32 restatement (explicit
matmul / softmax / layer-norm arithmetic on CPU tensors) of the reference's
hot path.  It is imported only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py, as the checker; the product package never
imports it.

Parity status: PINNED.  oracle/make_golden.py imports the reference module
itself in the authoring container and stores its outputs under tests/golden/;
tests/test_oracle_golden.py checks every function here against those files.

Reference anchors (paths relative to the reference checkout):
  utils_MMT/models_MMT_v15_4.py:358-546   embeddings and module construction
  utils_MMT/models_MMT_v15_4.py:549-976   sequence assembly and forward
  utils_MMT/models_MMT_v15_4.py:1222-1297 in-training sequence generation
  utils_MMT/models_MMT_v15_4.py:200-355   CE loss (weight_SMI only)
  utils_MMT/validate_generate_MMT_v15_4.py:723-775  greedy_sequence
  utils_MMT/helper_functions_pl_v15_4.py:177-209,239-269  tokenizer / decode
  utils_MMT/dataloaders_pl_v15_4.py:267-299,395-421,665-712 padding / encode / collate
The transformer layers themselves are torch.nn.TransformerEncoderLayer /
TransformerDecoderLayer defaults (post-norm, ReLU, dim_feedforward 2048,
eps 1e-5); their arithmetic is restated in encoder_layer() / decoder_layer().
"""
import math
import re

import torch

NEG_INF = float("-inf")

# ----------------------------------------------------------------------------
# parameter inventory (names and shapes of the reference state_dict)
# ----------------------------------------------------------------------------

ENCODERS = ("1H", "13C", "HSQC", "COSY", "IR", "cross")
FFN_DIM = 2048  # torch default dim_feedforward; config.forward_expansion is never read


def param_shapes(cfg):
    """Ordered {key: shape} of MultimodalTransformer.state_dict()
    (models_MMT_v15_4.py:495-546)."""
    d = cfg["hidden_size"]
    out = {}
    out["linear_spec_embedding_1H.point_embedding_layer_1H.fc_H.weight"] = (d, cfg["input_dim_1H"])
    out["linear_spec_embedding_1H.point_embedding_layer_1H.fc_H.bias"] = (d,)
    out["linear_spec_embedding_13C.point_embedding_layer_13C.fc_C.weight"] = (d, cfg["input_dim_13C"])
    out["linear_spec_embedding_13C.point_embedding_layer_13C.fc_C.bias"] = (d,)
    out["linear_spec_embedding_HSQC.point_embedding_layer_HSQC.fc_HSQC.weight"] = (d, cfg["input_dim_HSQC"])
    out["linear_spec_embedding_HSQC.point_embedding_layer_HSQC.fc_HSQC.bias"] = (d,)
    out["linear_spec_embedding_COSY.point_embedding_layer_COSY.fc_COSY.weight"] = (d, cfg["input_dim_COSY"])
    out["linear_spec_embedding_COSY.point_embedding_layer_COSY.fc_COSY.bias"] = (d,)
    out["linear_spec_embedding_IR.linear_spec_embedding_IR.weight"] = (d, cfg["input_dim_IR"])
    out["linear_spec_embedding_IR.linear_spec_embedding_IR.bias"] = (d,)
    out["linear_embedding_MF.embedding.weight"] = (cfg["MF_vocab_size"], d)
    out["linear_embedding_MS.embedding.weight"] = (cfg["MS_vocab_size"], d)
    out["linear_embedding_MW.linear_spec_embedding_MW.weight"] = (d, 1)
    out["linear_embedding_MW.linear_spec_embedding_MW.bias"] = (d,)
    out["embed_trg.weight"] = (cfg["in_size"], d)
    out["pe_trg.weight"] = (cfg["max_len"], d)

    def enc_layer(prefix):
        out[prefix + "self_attn.in_proj_weight"] = (3 * d, d)
        out[prefix + "self_attn.in_proj_bias"] = (3 * d,)
        out[prefix + "self_attn.out_proj.weight"] = (d, d)
        out[prefix + "self_attn.out_proj.bias"] = (d,)
        out[prefix + "linear1.weight"] = (FFN_DIM, d)
        out[prefix + "linear1.bias"] = (FFN_DIM,)
        out[prefix + "linear2.weight"] = (d, FFN_DIM)
        out[prefix + "linear2.bias"] = (d,)
        out[prefix + "norm1.weight"] = (d,)
        out[prefix + "norm1.bias"] = (d,)
        out[prefix + "norm2.weight"] = (d,)
        out[prefix + "norm2.bias"] = (d,)

    for name in ENCODERS:
        for i in range(cfg["num_encoder_layers"]):
            enc_layer(f"encoder_{name}.layers.{i}.")
    for i in range(cfg["num_decoder_layers"]):
        p = f"decoder.layers.{i}."
        out[p + "self_attn.in_proj_weight"] = (3 * d, d)
        out[p + "self_attn.in_proj_bias"] = (3 * d,)
        out[p + "self_attn.out_proj.weight"] = (d, d)
        out[p + "self_attn.out_proj.bias"] = (d,)
        out[p + "multihead_attn.in_proj_weight"] = (3 * d, d)
        out[p + "multihead_attn.in_proj_bias"] = (3 * d,)
        out[p + "multihead_attn.out_proj.weight"] = (d, d)
        out[p + "multihead_attn.out_proj.bias"] = (d,)
        out[p + "linear1.weight"] = (FFN_DIM, d)
        out[p + "linear1.bias"] = (FFN_DIM,)
        out[p + "linear2.weight"] = (d, FFN_DIM)
        out[p + "linear2.bias"] = (d,)
        for n in ("norm1", "norm2", "norm3"):
            out[p + n + ".weight"] = (d,)
            out[p + n + ".bias"] = (d,)
    out["fp1.weight"] = (cfg["fingerprint_size"], d)
    out["fp1.bias"] = (cfg["fingerprint_size"],)
    out["fc_out.weight"] = (cfg["out_size"], d)
    out["fc_out.bias"] = (cfg["out_size"],)
    out["real_data_linear.weight"] = (cfg["out_size"], d)
    out["real_data_linear.bias"] = (cfg["out_size"],)
    return out


def make_state_dict(cfg, seed, logit_gain=1.0):
    """Deterministic, torch-independent weights for every state_dict key.

    Values come from numpy's PCG64 stream in key order, scaled like the default
    torch initialisers (uniform +-1/sqrt(fan_in) for matrices and biases,
    N(0,1) for embedding tables, ones/zeros(+noise) for LayerNorm), so the same
    seed reproduces the same 100 MB of weights on any machine.  `logit_gain`
    multiplies fc_out so that greedy decode has well separated logits.
    """
    import numpy as np

    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    for key, shape in param_shapes(cfg).items():
        if ".norm" in key:
            if key.endswith("weight"):
                v = 1.0 + 0.1 * rng.standard_normal(shape)
            else:
                v = 0.1 * rng.standard_normal(shape)
        elif key in ("linear_embedding_MF.embedding.weight", "linear_embedding_MS.embedding.weight",
                     "embed_trg.weight", "pe_trg.weight"):
            v = rng.standard_normal(shape)
            if "linear_embedding_M" in key:
                v[0] = 0.0  # padding_idx=0 row (models_MMT_v15_4.py:449,459)
        else:
            fan_in = shape[1] if len(shape) == 2 else None
            if fan_in is None:
                # bias of the matrix generated just before it: use its fan_in
                fan_in = last_fan_in
            bound = 1.0 / math.sqrt(fan_in)
            v = rng.uniform(-bound, bound, size=shape)
            if len(shape) == 2:
                last_fan_in = fan_in
        if key.startswith("fc_out."):
            v = v * logit_gain
        sd[key] = torch.from_numpy(np.asarray(v, dtype=np.float32)).contiguous()
    return sd


# ----------------------------------------------------------------------------
# elementary arithmetic
# ----------------------------------------------------------------------------

def linear(x, w, b=None):
    y = x @ w.t()
    return y if b is None else y + b


def layer_norm(x, g, b, eps=1e-5):
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    return xc / torch.sqrt(var + eps) * g + b


def additive_mask(mask):
    """Key-padding mask -> additive float bias, the way torch's
    multi_head_attention_forward treats it: a bool mask means -inf on True, a
    floating (or integer, as produced by the reference collate_fn) mask is
    added to the attention logits as it stands."""
    if mask is None:
        return None
    if mask.dtype == torch.bool:
        return torch.zeros(mask.shape, dtype=torch.float32).masked_fill(mask, NEG_INF)
    return mask.to(torch.float32)


# Train mode.  torch's layers apply dropout at fixed sites (nn.MultiheadAttention: the attention probabilities;
# nn.Transformer{En,De}coderLayer: dropout1/2/3 on the sub-block outputs and `dropout` on the hidden activation;
# models_MMT_v15_4.py:961: dropout2 on the target embedding).  The functions below are those layers with dropout OFF
# unless DROPOUT is set: DROPOUT(x, (module prefix, kind)) -> x with the caller's mask and scale applied, kind in
# "attn" (x = probabilities (B,H,T,S)), "out1" / "out2" / "out3", "hidden", "embed" (x sequence-first (T,B,n)).  A test
# that knows which mask a run used at every site (the HIP path's masks are pure functions of seed and index) can so
# replay that run's train-mode step here and compare by value.
DROPOUT = None


def _drop(x, prefix, kind):
    return x if DROPOUT is None else DROPOUT(x, (prefix, kind))


def attention(q_in, kv_in, w_in, b_in, w_out, b_out, nhead, key_bias=None, causal=False, site=None):
    """nn.MultiheadAttention (batch_first=False) on sequence-first tensors.

    q_in (T,B,d), kv_in (S,B,d); key_bias (B,S) additive float or None.
    """
    T, B, d = q_in.shape
    S = kv_in.shape[0]
    hd = d // nhead
    q = linear(q_in, w_in[:d], b_in[:d])
    k = linear(kv_in, w_in[d:2 * d], b_in[d:2 * d])
    v = linear(kv_in, w_in[2 * d:], b_in[2 * d:])
    q = q.reshape(T, B, nhead, hd).permute(1, 2, 0, 3)  # B,H,T,hd
    k = k.reshape(S, B, nhead, hd).permute(1, 2, 0, 3)
    v = v.reshape(S, B, nhead, hd).permute(1, 2, 0, 3)
    s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(hd))
    if key_bias is not None:
        s = s + key_bias[:, None, None, :]
    if causal:
        cm = torch.triu(torch.full((T, S), NEG_INF), diagonal=1)
        s = s + cm
    p = torch.softmax(s, dim=-1)
    if site is not None:
        p = _drop(p, site, "attn")
    o = p @ v  # B,H,T,hd
    o = o.permute(2, 0, 1, 3).reshape(T, B, d)
    return linear(o, w_out, b_out)


def encoder_layer(x, sd, pre, nhead, key_bias):
    """torch.nn.TransformerEncoderLayer, norm_first=False, activation relu,
    dropout off (eval / p=0)."""
    sa = attention(x, x, sd[pre + "self_attn.in_proj_weight"], sd[pre + "self_attn.in_proj_bias"],
                   sd[pre + "self_attn.out_proj.weight"], sd[pre + "self_attn.out_proj.bias"],
                   nhead, key_bias, site=pre + "self_attn")
    x = layer_norm(x + _drop(sa, pre, "out1"), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"])
    h = _drop(torch.relu(linear(x, sd[pre + "linear1.weight"], sd[pre + "linear1.bias"])), pre, "hidden")
    ff = linear(h, sd[pre + "linear2.weight"], sd[pre + "linear2.bias"])
    return layer_norm(x + _drop(ff, pre, "out2"), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"])


def encoder(x, sd, name, nlayers, nhead, mask):
    kb = additive_mask(mask)
    for i in range(nlayers):
        x = encoder_layer(x, sd, f"encoder_{name}.layers.{i}.", nhead, kb)
    return x


def decoder_layer(x, mem, sd, pre, nhead, mem_bias):
    sa = attention(x, x, sd[pre + "self_attn.in_proj_weight"], sd[pre + "self_attn.in_proj_bias"],
                   sd[pre + "self_attn.out_proj.weight"], sd[pre + "self_attn.out_proj.bias"],
                   nhead, None, causal=True, site=pre + "self_attn")
    x = layer_norm(x + _drop(sa, pre, "out1"), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"])
    ca = attention(x, mem, sd[pre + "multihead_attn.in_proj_weight"], sd[pre + "multihead_attn.in_proj_bias"],
                   sd[pre + "multihead_attn.out_proj.weight"], sd[pre + "multihead_attn.out_proj.bias"],
                   nhead, mem_bias, site=pre + "multihead_attn")
    x = layer_norm(x + _drop(ca, pre, "out2"), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"])
    h = _drop(torch.relu(linear(x, sd[pre + "linear1.weight"], sd[pre + "linear1.bias"])), pre, "hidden")
    ff = linear(h, sd[pre + "linear2.weight"], sd[pre + "linear2.bias"])
    return layer_norm(x + _drop(ff, pre, "out3"), sd[pre + "norm3.weight"], sd[pre + "norm3.bias"])


def decoder(x, mem, sd, nlayers, nhead, mem_mask):
    mb = additive_mask(mem_mask)
    for i in range(nlayers):
        x = decoder_layer(x, mem, sd, f"decoder.layers.{i}.", nhead, mb)
    return x


# ----------------------------------------------------------------------------
# embeddings and sequence assembly (models_MMT_v15_4.py:733-792, 549-731)
# ----------------------------------------------------------------------------

_PE = {"1H": "linear_spec_embedding_1H.point_embedding_layer_1H.fc_H",
       "13C": "linear_spec_embedding_13C.point_embedding_layer_13C.fc_C",
       "HSQC": "linear_spec_embedding_HSQC.point_embedding_layer_HSQC.fc_HSQC",
       "COSY": "linear_spec_embedding_COSY.point_embedding_layer_COSY.fc_COSY"}


def embed_spectrum(sd, name, spectrum, mask):
    x = spectrum.unsqueeze(-1) if name == "13C" else spectrum
    e = torch.relu(linear(x.float(), sd[_PE[name] + ".weight"], sd[_PE[name] + ".bias"]))
    return e.permute(1, 0, 2), mask.to(torch.bool)


def embed_IR(sd, spectrum):
    e = torch.relu(linear(spectrum.float(), sd["linear_spec_embedding_IR.linear_spec_embedding_IR.weight"],
                          sd["linear_spec_embedding_IR.linear_spec_embedding_IR.bias"]))
    return e, torch.zeros(len(e), dtype=torch.bool)


def embed_tokens(sd, which, tokens, mask):
    e = torch.relu(sd[f"linear_embedding_{which}.embedding.weight"][tokens])
    return e.permute(1, 0, 2), mask


def embed_MW(sd, mw):
    e = torch.relu(linear(mw.float(), sd["linear_embedding_MW.linear_spec_embedding_MW.weight"],
                          sd["linear_embedding_MW.linear_spec_embedding_MW.bias"]))
    return e, torch.zeros((mw.shape[0], 1), dtype=torch.bool)


def forward(sd, cfg, batch, trg_SMI_input=None):
    """MultimodalTransformer.forward (models_MMT_v15_4.py:803-976), eval mode.

    `batch` holds the reference's keyword names (src_1H, mask_1H, ... trg_MW).
    Returns what the reference returns: (logits, fingerprint, memory, mask) or,
    without targets, (memory, embedding_src, mask, fingerprint).
    """
    mode = cfg["training_mode"]
    nl, nh = cfg["num_encoder_layers"], cfg["num_heads"]
    mw = batch["trg_MW"].unsqueeze(1)
    emb = {}
    bsz = False
    for name in ("1H", "13C", "HSQC", "COSY"):
        if name in mode:
            emb[name] = embed_spectrum(sd, name, batch["src_" + name], batch["mask_" + name])
            bsz = emb[name][0].shape[1]
    e_IR = m_IR = None
    if "IR" in mode:
        e_IR, m_IR = embed_IR(sd, batch["src_IR"])
        e_IR, m_IR = e_IR.unsqueeze(0), m_IR.unsqueeze(-1)
    e_MF, m_MF = embed_tokens(sd, "MF", batch["src_MF"], batch["mask_MF"]) if "MF" in mode else (None, None)
    e_MS, m_MS = embed_tokens(sd, "MS", batch["src_MS"], batch["mask_MS"]) if "MS" in mode else (None, None)
    e_MW, m_MW = embed_MW(sd, mw) if "MW" in mode else (None, None)
    if e_MW is not None:
        e_MW = e_MW.unsqueeze(0)

    def assemble(e, m):
        es, ms = [e], [m]
        for ee, mm in ((e_MF, m_MF), (e_MS, m_MS), (e_MW, m_MW)):
            if ee is not None:
                es.append(ee)
                ms.append(mm)
        return torch.cat(es, dim=0), torch.cat(ms, dim=1)

    feature_dim = 193 if "MS" in mode else 129
    feature_dim_IR = 130 if "MS" in mode else 66
    memory, embedding_src, masks = [], [], []
    for name in ("1H", "13C", "HSQC", "COSY"):
        if name in emb:
            e, m = assemble(*emb[name])
            memory.append(encoder(e, sd, name, nl, nh, m))
            embedding_src.append(e)
            masks.append(m)
        else:
            rows = 65 if name == "COSY" else feature_dim
            memory.append(torch.zeros(rows, bsz, 128))
            embedding_src.append(torch.zeros(rows, bsz, 128))
            masks.append(torch.ones(bsz, rows))
    if e_IR is not None:
        e, m = assemble(e_IR, m_IR)
        memory.append(encoder(e, sd, "IR", nl, nh, m))
        embedding_src.append(e)
        masks.append(m)
    else:
        memory.append(torch.zeros(feature_dim_IR, bsz, 128))
        embedding_src.append(torch.zeros(feature_dim_IR, bsz, 128))
        masks.append(torch.full((bsz, feature_dim_IR), False, dtype=torch.bool))
    memory = torch.cat(memory, dim=0)
    embedding_src = torch.cat(embedding_src, dim=0)
    mask = torch.cat(masks, dim=1)
    memory = encoder(memory, sd, "cross", nl, int(nh / 4), mask)
    fingerprint = linear(memory.mean(dim=0), sd["fp1.weight"], sd["fp1.bias"])
    if trg_SMI_input is None:
        return memory, embedding_src, mask, fingerprint
    logits = decode_logits(sd, cfg, trg_SMI_input, memory, mask)
    if cfg.get("use_real_data", False):
        extra = linear(memory.mean(dim=0), sd["real_data_linear.weight"], sd["real_data_linear.bias"])
        logits = (logits + extra.unsqueeze(0)) / 2
    return logits, fingerprint, memory, mask


def decode_logits(sd, cfg, trg, memory, mask):
    """Target embedding + decoder + vocabulary projection
    (models_MMT_v15_4.py:955-974)."""
    T, _ = trg.shape
    e = _drop(sd["embed_trg.weight"][trg] + sd["pe_trg.weight"][torch.arange(T)].unsqueeze(1), "embed_trg", "embed")
    out = decoder(e, memory, sd, cfg["num_decoder_layers"], cfg["num_heads"], mask)
    return linear(out, sd["fc_out.weight"], sd["fc_out.bias"])


def ce_loss(logits, trg_enc_SMI):
    """SMI loss of calculate_total_loss (models_MMT_v15_4.py:268-271): plain
    CrossEntropyLoss (mean over every position, PAD included) between logits
    (T,B,V) and trg_enc_SMI[:, 1:] given batch-first (B,64)."""
    T, B, V = logits.shape
    target = trg_enc_SMI.t()[1:, :].reshape(-1)
    lg = logits.reshape(-1, V)
    lse = torch.logsumexp(lg, dim=-1)
    picked = lg.gather(1, target.unsqueeze(1)).squeeze(1)
    return (lse - picked).mean()


def training_forward(sd, cfg, batch):
    """training_step up to the loss (models_MMT_v15_4.py:1051-1064, 1086)."""
    trg = batch["trg_enc_SMI"].t()
    logits, fp, memory, mask = forward(sd, cfg, batch, trg[:-1, :])
    return ce_loss(logits, batch["trg_enc_SMI"]), logits, fp, memory, mask


def adamw_step(params, grads, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8, wd=0.01):
    """torch.optim.AdamW defaults (models_MMT_v15_4.py:1213-1214), in place."""
    for k in params:
        g = grads.get(k)
        if g is None:
            continue
        params[k].mul_(1 - lr * wd)
        m[k].mul_(b1).add_(g, alpha=1 - b1)
        v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** step
        bc2 = 1 - b2 ** step
        denom = (v[k].sqrt() / math.sqrt(bc2)).add_(eps)
        params[k].addcdiv_(m[k], denom, value=-lr / bc1)


# ----------------------------------------------------------------------------
# greedy decode (validate_generate_MMT_v15_4.py:723-775; the same loop is
# models_MMT_v15_4.py:1222-1275 with temperature 1 and no early stop)
# ----------------------------------------------------------------------------

def greedy_sequence(sd, cfg, memory, mask, sos=3, temperature=1.0, early_stop=True, max_len=None):
    N = memory.shape[1]
    toks = torch.full((1, N), sos, dtype=torch.long)
    probs = []
    margins = []
    for _ in range(max_len or cfg["max_len"]):
        logits = decode_logits(sd, cfg, toks, memory, mask)
        p = torch.softmax(logits / temperature, dim=2)[-1]
        nxt = torch.argmax(p, dim=1)
        probs.append(p.gather(1, nxt.unsqueeze(-1)).squeeze(-1))
        top2 = torch.topk(logits[-1], 2, dim=1).values
        margins.append(top2[:, 0] - top2[:, 1])
        toks = torch.cat((toks, nxt.unsqueeze(0)), dim=0)
        if early_stop and bool((nxt == 0).all()):
            break
    probs = torch.stack(probs)
    return toks[1:], probs[1:], torch.stack(margins)


# ----------------------------------------------------------------------------
# CLIP head on the encoder half (models_CLIP_v15_4.py:224-302, 390-426).  The ChemBERTa branch needs weights
# that are not in the repository: its output features are an INPUT here.
# ----------------------------------------------------------------------------

CLIP_DIMS = dict(CB_embedding=1024, MT_embedding=512, projection_dim=512)   # config_V8.json


def make_clip_head_state_dict(seed, dims=CLIP_DIMS):
    """Seeded weights of CLIPModel.CB_projection / MT_projection (keys as in the reference's state_dict)."""
    import numpy as np
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    for name, e in (("CB_projection.", dims["CB_embedding"]), ("MT_projection.", dims["MT_embedding"])):
        p = dims["projection_dim"]
        for key, shape, fan in (("projection.weight", (p, e), e), ("projection.bias", (p,), e),
                                ("fc.weight", (p, p), p), ("fc.bias", (p,), p)):
            b = 1.0 / math.sqrt(fan)
            sd[name + key] = torch.from_numpy(rng.uniform(-b, b, size=shape).astype(np.float32))
        sd[name + "layer_norm.weight"] = torch.from_numpy((1.0 + 0.1 * rng.standard_normal(p)).astype(np.float32))
        sd[name + "layer_norm.bias"] = torch.from_numpy((0.1 * rng.standard_normal(p)).astype(np.float32))
    return sd


def projection_head(x, sd, pre):
    """ProjectionHead.forward (:419-426) with dropout off: LN(fc(gelu(proj(x))) + proj(x)), exact (erf) GELU."""
    projected = linear(x, sd[pre + "projection.weight"], sd[pre + "projection.bias"])
    h = 0.5 * projected * (1.0 + torch.erf(projected / math.sqrt(2.0)))
    h = linear(h, sd[pre + "fc.weight"], sd[pre + "fc.bias"])
    return layer_norm(h + projected, sd[pre + "layer_norm.weight"], sd[pre + "layer_norm.bias"])


def clip_loss(CB_emb, MT_emb, temperature=1.0):
    """Symmetric soft-target cross entropy (:293-302).  The targets are NOT detached in the reference: gradients
    flow through them as well."""
    logits = CB_emb @ MT_emb.T / temperature
    targets = torch.softmax((MT_emb @ MT_emb.T + CB_emb @ CB_emb.T) / 2 * temperature, dim=-1)
    texts = (-targets * torch.log_softmax(logits, dim=-1)).sum(1)
    images = (-targets.T * torch.log_softmax(logits.T, dim=-1)).sum(1)
    loss = (images + texts) / 2.0
    return loss.mean(), loss


def clip_forward(sd, head_sd, cfg, batch, CB_features, temperature=1.0):
    """CLIPModel.forward (:243-302) given the ChemBERTa features: MT features = the fingerprint output of the
    encoder half (fourth return value of MultimodalTransformer.forward without a target, :278-285)."""
    memory, _, mask, fingerprint = forward(sd, cfg, batch, None)
    CB_emb = projection_head(CB_features, head_sd, "CB_projection.")
    MT_emb = projection_head(fingerprint, head_sd, "MT_projection.")
    return clip_loss(CB_emb, MT_emb, temperature)


# ----------------------------------------------------------------------------
# beam search (validate_generate_MMT_v15_4.py:995-1094)
# ----------------------------------------------------------------------------

def beam_search_step(sd, cfg, beam, memory, mask, beam_size, eos=2):
    """One step for one molecule (:995-1053).  beam = list of (score, token list, probability list).
    Also returns the decision margins of the step: the smallest relative score gap between neighbours in the
    sorted candidate list down to the first dropped candidate (a reduced-precision run is only comparable with
    this one while those gaps are large)."""
    new_beam, seen = [], set()
    for score, sequence, prob_sequence in beam:
        key = tuple(sequence)
        if key in seen:                       # :1015
            continue
        seen.add(key)
        if sequence[-1] == eos:               # :1019 finished beams are carried over unchanged
            new_beam.append((score, sequence, prob_sequence))
            continue
        trg = torch.tensor(sequence, dtype=torch.long).unsqueeze(1)
        logits = decode_logits(sd, cfg, trg, memory, mask)
        probs = torch.softmax(logits[-1], dim=-1)          # :1038 no temperature on the ranking distribution
        top_probs, top_idx = torch.topk(probs, beam_size, dim=-1)
        for i in range(beam_size):
            new_sequence = sequence + [int(top_idx[0, i])]
            if tuple(new_sequence) in seen:   # :1046
                continue
            seen.add(tuple(new_sequence))
            p = float(top_probs[0, i])
            new_beam.append((score * p, new_sequence, prob_sequence + [p]))
    new_beam.sort(key=lambda x: x[0], reverse=True)        # :1052 stable sort, ties keep generation order
    margin = float("inf")
    for a, b in zip(new_beam[:beam_size], new_beam[1:beam_size + 1]):
        margin = min(margin, (a[0] - b[0]) / max(a[0], 1e-300))
    return new_beam[:beam_size], margin


def beam_search(sd, cfg, memory, mask, beam_size, gen_len, sos=3, eos=2):
    """:1056-1094.  Returns (beams, margins): beams[i] = list of (score, tokens incl. <SOS>, probabilities) for
    molecule i, best first; margins[i] = smallest decision margin over all steps of molecule i."""
    N = memory.shape[1]
    beams = [[(1, [sos], [])] for _ in range(N)]
    margins = [float("inf")] * N
    for _ in range(gen_len):
        for i in range(N):
            mask_i = mask[i:i + 1] if mask is not None else None
            beams[i], m = beam_search_step(sd, cfg, beams[i], memory[:, i:i + 1, :], mask_i, beam_size, eos)
            margins[i] = min(margins[i], m)
    return beams, margins


# ----------------------------------------------------------------------------
# tokenizer / detokenizer / collate helpers
# ----------------------------------------------------------------------------

TWO_CHAR_SYMBOLS = [
    'He', 'Li', 'Be', 'Ne', 'Na', 'Mg', 'Al', 'Si', 'Cl', 'Ar', 'Ca', 'Ti', 'Cr', 'Mn', 'Fe', 'Ni', 'Cu', 'Zn',
    'Ga', 'Ge', 'As', 'Se', 'Br', 'Kr', 'Rb', 'Sr', 'Zr', 'Nb', 'Mo', 'Tc', 'Ru', 'Rh', 'Pd', 'Ag', 'Cd', 'In',
    'Sb', 'Te', 'I', 'Xe', 'Ba', 'La', 'Ce', 'Pr', 'Nd', 'Pm', 'Sm', 'Eu', 'Gd', 'Tb', 'Dy', 'Ho', 'Er', 'Tm',
    'Yb', 'Lu', 'Hf', 'Ta', 'W', 'Re', 'Os', 'Ir', 'Pt', 'Au', 'Hg', 'Tl', 'Pb', 'Bi', 'Th', 'Pa', 'U', 'Np',
    'Pu', 'Am', 'Cm', 'Bk', 'Cf', 'Es', 'Fm', 'Md', 'No', 'Lr', 'Rf', 'Db', 'Sg', 'Bh', 'Hs', 'Mt', 'Ds', 'Rg',
    'Nh', 'Fl', 'Mc', 'Lv', 'Ts', 'Og']


def tokenize_smiles(smiles, two_letter_atoms=TWO_CHAR_SYMBOLS):
    """helper_functions_pl_v15_4.py:188-209: greedy two-characters-first scan."""
    tokens, i = [], 0
    while i < len(smiles):
        if i < len(smiles) - 1 and smiles[i:i + 2] in two_letter_atoms:
            tokens.append(smiles[i:i + 2])
            i += 2
        else:
            tokens.append(smiles[i])
            i += 1
    return tokens


def encode_smiles(stoi, smi):
    """dataloaders_pl_v15_4.py:395-408."""
    return [stoi.get(t, stoi.get('<UNK>')) for t in tokenize_smiles(smi)]


def encode_MF(stoi_MF, mf):
    """dataloaders_pl_v15_4.py:411-421."""
    tokens = []
    for element, count in re.findall(r'([A-Z][a-z]*)(\d{1,2})?', mf):
        tokens.append(element)
        if count:
            tokens.append(count)
    return [stoi_MF.get(t, stoi_MF.get('<UNK>')) for t in tokens]


def tensor_to_smiles(tensor, itos):
    """helper_functions_pl_v15_4.py:239-269: cut every column at the first <EOS>."""
    if tensor.dim() > 1:
        out = []
        for i in range(tensor.shape[1]):
            seq = []
            for j in range(tensor.shape[0]):
                tok = itos[str(int(tensor[j, i]))]
                if tok == "<EOS>":
                    break
                seq.append(tok)
            out.append("".join(seq))
        return out
    seq = []
    for j in range(len(tensor)):
        tok = itos[str(int(tensor[j]))]
        if tok == "<EOS>":
            break
        seq.append(tok)
    return "".join(seq)


def zero_pad(data, pad_length, dimensions):
    """dataloaders_pl_v15_4.py:267-299 (including the 1-D >= pad_length case
    that leaves the mask all ones)."""
    mask = torch.ones(pad_length, dtype=torch.long)
    if dimensions == 1:
        t = torch.tensor(data, dtype=torch.float32)
        if len(data) >= pad_length:
            return t[:pad_length], mask
        mask[:len(data)] = 0
        return torch.cat((t, torch.zeros(pad_length - len(data)))), mask
    rows = [list(r) for r in data]
    mask[:len(rows)] = 0
    while len(rows) < pad_length:
        rows.append([0, 0])
    return torch.tensor(rows[:pad_length], dtype=torch.float32), mask


# ----------------------------------------------------------------------------
# seeded synthetic batches (shape and ranges of the 1000-molecule CSV fixtures)
# ----------------------------------------------------------------------------

def synthetic_batch(B, seed, mode="1H_HSQC_COSY_IR_MF_MW", float_masks=True):
    """A collate_fn-shaped batch dict of synthetic spectra.  Masks of MF/MS are
    float32 (same 0/1 values as the reference's int64) because torch >= 2
    refuses integer key-padding masks."""
    import numpy as np
    rng = np.random.default_rng(seed)
    b = {}

    def spectrum2d(lo, hi):
        x = np.zeros((B, 64, 2), np.float32)
        m = np.ones((B, 64), np.float32)
        for i in range(B):
            n = int(rng.integers(lo, hi + 1))
            x[i, :n] = rng.uniform(0, 1, (n, 2))
            m[i, :n] = 0
        return torch.from_numpy(x), torch.from_numpy(m)

    b["src_1H"], b["mask_1H"] = spectrum2d(2, 64)
    b["src_1H"][:, :, 1] = torch.round(b["src_1H"][:, :, 1] * 3) * (b["mask_1H"] == 0)
    b["src_HSQC"], b["mask_HSQC"] = spectrum2d(2, 25)
    b["src_COSY"], b["mask_COSY"] = spectrum2d(2, 54)
    x = np.zeros((B, 64), np.float32)
    m = np.ones((B, 64), np.float32)
    for i in range(B):
        n = int(rng.integers(8, 22))
        x[i, :n] = rng.uniform(0, 1, n)
        m[i, :n] = 0
    b["src_13C"], b["mask_13C"] = torch.from_numpy(x), torch.from_numpy(m)
    b["src_IR"] = torch.from_numpy(rng.uniform(0, 1, (B, 1000)).astype(np.float32))
    b["mask_IR"] = torch.zeros(B, 1000)
    mf = np.zeros((B, 64), np.int64)
    mfm = np.ones((B, 64), np.int64)
    ms = np.zeros((B, 64), np.int64)
    msm = np.ones((B, 64), np.int64)
    for i in range(B):
        k = int(rng.integers(4, 13))
        mf[i, 0] = 3
        mf[i, 1:1 + k] = rng.integers(5, 212, k)
        mf[i, 1 + k] = 2
        mfm[i, :k + 2] = 0
        ell = int(rng.integers(25, 61))
        ms[i, 0] = 3
        ms[i, 1:1 + ell] = rng.integers(5, 43, ell)
        ms[i, 1 + ell] = 2
        msm[i, :ell + 2] = 0
    b["src_MF"] = torch.from_numpy(mf)
    b["src_MS"] = torch.from_numpy(ms)
    if float_masks:
        b["mask_MF"] = torch.from_numpy(mfm).float()
        b["mask_MS"] = torch.from_numpy(msm).float()
    else:
        b["mask_MF"] = torch.from_numpy(mfm)
        b["mask_MS"] = torch.from_numpy(msm)
    b["trg_MW"] = torch.from_numpy(rng.uniform(150, 500, B).astype(np.float32))
    b["trg_enc_SMI"] = b["src_MS"].clone()
    b["trg_FP"] = torch.from_numpy((rng.uniform(0, 1, (B, 512)) < 0.1).astype(np.float32))
    return b


def default_cfg(**over):
    cfg = dict(input_dim_1H=2, input_dim_13C=1, input_dim_HSQC=2, input_dim_COSY=2, input_dim_IR=1000,
               MF_vocab_size=212, MS_vocab_size=43, in_size=43, hidden_size=128, out_size=43,
               num_encoder_layers=6, num_decoder_layers=6, num_heads=16, max_len=128, drop_out=0.1,
               fingerprint_size=512, use_real_data=False, training_mode="1H_HSQC_COSY_IR_MF_MW",
               temperature=1, lr_pretraining=1e-4)
    cfg.update(over)
    return cfg
