"""CLIP contrastive head on the accelerated encoder half.

Counterpart of utils_MMT/models_CLIP_v15_4.py: ProjectionHead (:403-426), CLIPModel.forward (:243-302) and
cross_entropy (:390-396), with the same constructor arguments, attribute and state_dict names
(CB_projection.*, MT_projection.*, MT_model.*).  The spectra branch is the MultimodalTransformer of this
package (its fingerprint output are the "MT features", :278-285); the SMILES branch (ChemBERTa) is whatever
module the caller passes as CB_model, exactly as in the reference: it is called as CB_model(input_ids, mask) and
must return (_, features).  Its weights are not part of the reference repository, so tests feed given features.

forward() returns (loss.mean(), loss) like the reference.  loss_and_backward() also runs the backward pass in
HIP kernels: gradients of both projection heads (.grad of their parameters), of fp1 and of every encoder
parameter (flat gradient buffer of the MultimodalTransformer), and of the CB features (returned, for the caller's
own backward through its SMILES model).  The head runs in fp32 (csrc/clip.hip); there is no CPU fallback.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib as L
from . import helper_functions as hf
from .engine import _P

D = 128


def _p(t):
    return C.c_void_p(t.data_ptr())


def _sgemm(A, sai, sak, Bm, sbk, sbj, Cm, ldc, bias, M, N, K, alpha=1.0, beta=0.0):
    L.check(L.lib().mmst_sgemm(_p(A), C.c_long(sai), C.c_long(sak), _p(Bm), C.c_long(sbk), C.c_long(sbj), _p(Cm), ldc,
                               _p(bias) if bias is not None else None, M, N, K, C.c_float(alpha), C.c_float(beta),
                               L.stream_ptr()), "mmst_sgemm")


def _nt(X, W, bias, out):
    """out[M][N] = X[M][K] W[N][K]^T + bias"""
    M, K = X.shape
    N = W.shape[0]
    _sgemm(X, K, 1, W, 1, K, out, N, bias, M, N, K)


def _nn(X, W, out, alpha=1.0, beta=0.0):
    """out[M][N] (+)= alpha * X[M][K] W[K][N]"""
    M, K = X.shape
    N = W.shape[1]
    _sgemm(X, K, 1, W, N, 1, out, N, None, M, N, K, alpha, beta)


def _tn(X, Y, out, alpha=1.0, beta=0.0):
    """out[M][N] (+)= alpha * X[K][M]^T Y[K][N]"""
    K, M = X.shape
    N = Y.shape[1]
    _sgemm(X, 1, M, Y, N, 1, out, N, None, M, N, K, alpha, beta)


class ProjectionHead(nn.Module):
    """LayerNorm(dropout(fc(gelu(projection(x)))) + projection(x)), models_CLIP_v15_4.py:403-426."""

    def __init__(self, config, embedding_type="MT"):
        super().__init__()
        self.embedding_dim = config.CB_embedding if embedding_type == "CB" else config.MT_embedding
        self.projection_dim = config.projection_dim
        self.drop_p = float(config.dropout)
        self.projection = nn.Linear(self.embedding_dim, self.projection_dim)
        self.fc = nn.Linear(self.projection_dim, self.projection_dim)
        self.layer_norm = nn.LayerNorm(self.projection_dim)
        if self.projection_dim > 1024:
            raise L.MMSTError("projection_dim above 1024 is not supported by the row kernels")
        self._saved = None

    def _check(self, x):
        if x.device.type != "cuda":
            raise L.MMSTError("the CLIP head runs on the GPU only (no CPU fallback)")
        if self.projection.weight.device != x.device:
            raise L.MMSTError("ProjectionHead parameters and input are on different devices")

    def forward(self, x, seed=0):
        """x [B][embedding_dim] fp32 -> [B][projection_dim] fp32; keeps what the backward needs."""
        x = x.contiguous().float()
        self._check(x)
        B, P = x.shape[0], self.projection_dim
        lib, st = L.lib(), L.stream_ptr()
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=x.device)  # noqa: E731
        projected, h, f, out, xhat, rstd = new(B, P), new(B, P), new(B, P), new(B, P), new(B, P), new(B)
        p = self.drop_p if self.training else 0.0
        _nt(x, self.projection.weight.data, self.projection.bias.data, projected)
        L.check(lib.mmst_gelu_fwd(_p(projected), _p(h), C.c_long(B * P), st), "mmst_gelu_fwd")
        _nt(h, self.fc.weight.data, self.fc.bias.data, f)
        L.check(lib.mmst_head_out_fwd(_p(f), _p(projected), _p(self.layer_norm.weight.data), _p(self.layer_norm.bias.data),
                                      _p(out), _p(xhat), _p(rstd), B, P, C.c_float(p), C.c_uint32(seed), st),
                "mmst_head_out_fwd")
        self._saved = (x, projected, h, xhat, rstd, p, seed)
        return out

    def backward(self, dout):
        """Accumulates the parameter gradients into .grad and returns d(input)."""
        x, projected, h, xhat, rstd, p, seed = self._saved
        B, P = dout.shape
        lib, st = L.lib(), L.stream_ptr()
        for q in self.parameters():
            if q.grad is None:
                q.grad = torch.zeros_like(q.data)
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=x.device)  # noqa: E731
        df, dproj, dh, dx = new(B, P), new(B, P), new(B, P), new(B, self.embedding_dim)
        ones = torch.ones(B, 1, dtype=torch.float32, device=x.device)
        L.check(lib.mmst_head_out_bwd(_p(dout.contiguous()), _p(xhat), _p(rstd), _p(self.layer_norm.weight.data), _p(df),
                                      _p(dproj), _p(self.layer_norm.weight.grad), _p(self.layer_norm.bias.grad), B, P,
                                      C.c_float(p), C.c_uint32(seed), st), "mmst_head_out_bwd")
        _tn(df, h, self.fc.weight.grad, beta=1.0)                       # dWfc += df^T h
        _tn(ones, df, self.fc.bias.grad.view(1, P), beta=1.0)
        _nn(df, self.fc.weight.data, dh)                                # dh = df Wfc
        L.check(lib.mmst_gelu_bwd(_p(projected), _p(dh), _p(dproj), C.c_long(B * P), st), "mmst_gelu_bwd")
        _tn(dproj, x, self.projection.weight.grad, beta=1.0)            # dWp += dproj^T x
        _tn(ones, dproj, self.projection.bias.grad.view(1, P), beta=1.0)
        _nn(dproj, self.projection.weight.data, dx)                     # dx = dproj Wp
        return dx


class CLIPModel(nn.Module):
    def __init__(self, CB_model, MT_model, tokenizer, itos, config):
        super().__init__()
        self.CB_model, self.MT_model, self.tokenizer, self.itos, self.config = CB_model, MT_model, tokenizer, itos, config
        self.CB_projection = ProjectionHead(config, "CB").to(config.device)
        self.MT_projection = ProjectionHead(config, "MT").to(config.device)
        self.temperature = float(config.CLIP_temperature)

    # ------------------------------------------------------------------
    def _cb_features(self, x):
        """ChemBERTa branch as the reference drives it (:265-277).  Without a tokenizer the SMILES model is called
        with the target token ids and their padding mask as they are (a caller's own SMILES encoder)."""
        dev = self.config.device
        trg = x['trg_enc_SMI'].to(dev).transpose(0, 1)
        if self.tokenizer is None:
            ids = trg.transpose(0, 1)
            _, feats = self.CB_model(ids, (ids != 0).long())
            return feats
        smiles = hf.tensor_to_smiles(trg[1:, :], self.itos)
        tok = self.tokenizer(smiles, return_tensors='pt', padding=True)
        _, feats = self.CB_model(tok['input_ids'].to(dev), tok['attention_mask'].to(dev))
        return feats

    def _forward(self, x, CB_features, want_grad):
        mt = self.MT_model
        dev = mt._dev()
        pl = mt.run_encoder_half(x, T=0)
        B = pl.B
        fingerprint = pl.cv.view(pl.fingerprint, B, mt.fingerprint_size)
        if CB_features is None:
            CB_features = self._cb_features(x)
        seed = (mt._engine.step_seed * 2654435761 + 77) & 0xFFFFFFFF
        CB = self.CB_projection(CB_features.to(dev), seed=seed)
        MT = self.MT_projection(fingerprint, seed=(seed + 1) & 0xFFFFFFFF)
        P = CB.shape[1]
        lib, st = L.lib(), L.stream_ptr()
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
        logits, Z = new(B, B), new(B, B)
        tau = self.temperature
        _sgemm(CB, P, 1, MT, 1, P, logits, B, None, B, B, P, 1.0 / tau)            # CB MT^T / tau (:293)
        _sgemm(MT, P, 1, MT, 1, P, Z, B, None, B, B, P, 0.5 * tau)                 # (MT MT^T + CB CB^T) / 2 * tau (:294-298)
        _sgemm(CB, P, 1, CB, 1, P, Z, B, None, B, B, P, 0.5 * tau, 1.0)
        ws = new(5 * B * B + 3 * B)
        losses, mean = new(B), torch.zeros(1, dtype=torch.float32, device=dev)
        dlogits, dZ = new(B, B), new(B, B)
        L.check(lib.mmst_clip_loss(_p(logits), _p(Z), B, _p(ws), _p(losses), _p(mean), _p(dlogits), _p(dZ), st),
                "mmst_clip_loss")
        self._state = (pl, CB, MT, dlogits, dZ) if want_grad else None
        return mean[0], losses

    def forward(self, x, CB_features=None):
        """(loss.mean(), loss) as models_CLIP_v15_4.py:302.  CB_features: optional given SMILES-branch features."""
        return self._forward(x, CB_features, False)

    def loss_and_backward(self, x, CB_features=None, on_segment=None):
        """forward + backward of loss.mean().  Returns (loss mean, per-sample losses, d(CB_features))."""
        mean, losses = self._forward(x, CB_features, True)
        pl, CB, MT, dlogits, dZ = self._state
        mt, eng = self.MT_model, self.MT_model._engine
        ar = eng.arena
        B, P = CB.shape
        dev = CB.device
        lib, st = L.lib(), L.stream_ptr()
        tau = self.temperature
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
        dS, dCB, dMT = new(B, B), new(B, P), new(B, P)
        L.check(lib.mmst_symmetrize(_p(dZ), _p(dS), B, st), "mmst_symmetrize")
        _nn(dlogits, MT, dCB, alpha=1.0 / tau)                 # d logits -> CB
        _nn(dS, CB, dCB, alpha=0.5 * tau, beta=1.0)            # d(CB CB^T): (dZ + dZ^T) CB * tau / 2
        _tn(dlogits, CB, dMT, alpha=1.0 / tau)                 # d logits^T -> MT
        _nn(dS, MT, dMT, alpha=0.5 * tau, beta=1.0)
        dCB_features = self.CB_projection.backward(dCB)
        dfp = self.MT_projection.backward(dMT)                 # [B][512] gradient of the fingerprint
        # fp1 (models_MMT_v15_4.py:543, :948) and the mean over the memory rows (:946)
        mean_rows = pl.cv.view(pl.mean, B, D).float()
        Wg = torch.as_strided(ar.G, (mt.fingerprint_size, D), (D, 1), ar.offset["fp1.weight"])
        bg = torch.as_strided(ar.G, (1, mt.fingerprint_size), (mt.fingerprint_size, 1), ar.offset["fp1.bias"])
        Wp = torch.as_strided(ar.P, (mt.fingerprint_size, D), (D, 1), ar.offset["fp1.weight"])
        ones = torch.ones(B, 1, dtype=torch.float32, device=dev)
        _tn(dfp, mean_rows, Wg, beta=1.0)
        _tn(ones, dfp, bg, beta=1.0)
        dmean = new(B, D)
        _nn(dfp, Wp, dmean)
        L.check(lib.mmst_seq_mean_bwd(_p(dmean), C.c_void_p(_P(pl.dmem)), B, pl.Lmem, st), "mmst_seq_mean_bwd")
        eng.backward(pl, mt.training, on_segment, skip_decoder=True)
        mt._last_plan = pl
        return mean, losses, dCB_features
