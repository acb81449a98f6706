"""Host-side execution engine: parameter arena, activation plans and the launch sequences that
drive libmmst.so for the MultimodalTransformer forward / backward path.

torch is used for device memory, streams and (in train.py) torch.distributed only; every
arithmetic step of the hot path is a HIP kernel behind the C ABI of include/mmst.h.
"""
import ctypes as C
import math
import os

import torch

from . import _lib as L
from . import ops

D = 128
FF = 2048
SPEC = ("1H", "13C", "HSQC", "COSY")
ALL_MODS = ("1H", "13C", "HSQC", "COSY", "IR")
# Encoder layers: enqueue all of a layer's weight gradients after its data phase (0, default), or start its
# feed-forward weight gradients beside its own attention backward (1).  Measured on the 64-molecule training step:
# (1) makes the weight-gradient kernel faster (94 us instead of 133 us per launch) and the attention kernels
# slower by more than that, 8.80 ms instead of 8.49 ms per step.
_INTERLEAVE_WGRADS = os.environ.get("MMST_WGRAD_INTERLEAVE", "0") == "1"
LAYER_DROPOUT = 0.1  # torch default of nn.Transformer{En,De}coderLayer; the reference never overrides it
VPAD = 64            # padded vocabulary width of the logits / fc_out rows


def _align(n, a=64):
    return (n + a - 1) // a * a


class Arena:
    """All parameters in ONE flat fp32 buffer (plus flat gradient, AdamW moments and bf16 copy).

    Order = execution order (embeddings, modality-encoder layers interleaved by depth, cross
    encoder, decoder, heads) so that backward completes gradient memory from the tail towards the
    head and the data-parallel all-reduce can run on contiguous tail segments while earlier layers
    are still in backward."""

    def __init__(self, model, device):
        self.device = device
        named = dict(model.named_parameters())
        nl, ndl = model.num_encoder_layers, model.num_decoder_layers
        order = [k for k in named if not (k.startswith("encoder_") or k.startswith("decoder.") or
                                          k.startswith("fp1.") or k.startswith("fc_out.") or
                                          k.startswith("real_data_linear."))]
        self.seg_embed_end_keys = len(order)
        layer_first = {}            # index in `order` of the first parameter of each modality-encoder depth
        for l in range(nl):
            layer_first[len(order)] = l
            for m in ALL_MODS:
                order += [k for k in named if k.startswith(f"encoder_{m}.layers.{l}.")]
        mark_modal = len(order)
        for l in range(nl):
            order += [k for k in named if k.startswith(f"encoder_cross.layers.{l}.")]
        mark_cross = len(order)
        for l in range(ndl):
            order += [k for k in named if k.startswith(f"decoder.layers.{l}.")]
        order += [k for k in named if k.startswith("fp1.") or k.startswith("fc_out.") or
                  k.startswith("real_data_linear.")]
        assert len(order) == len(named) and len(set(order)) == len(named)
        self.offset, self.numel = {}, {}
        off = 0
        marks = {}
        for i, k in enumerate(order):
            if i == mark_modal:
                marks["cross"] = off
            if i == mark_cross:
                marks["decoder"] = off
            if i == self.seg_embed_end_keys:
                marks["modal"] = off
            if i in layer_first:
                marks[f"modal_l{layer_first[i]}"] = off
            p = named[k]
            n = p.numel()
            slots = n
            if k in ("fc_out.weight", "real_data_linear.weight"):
                slots = VPAD * D          # rows V..63 stay zero: the GEMMs run on a 64-row matrix
            elif k in ("fc_out.bias", "real_data_linear.bias"):
                slots = VPAD
            self.offset[k] = off
            self.numel[k] = n
            off += _align(slots)
        self.total = off
        self.marks = marks
        self.order = order
        self.P = torch.zeros(off, dtype=torch.float32, device=device)
        self.G = torch.zeros(off, dtype=torch.float32, device=device)
        self.Pb = torch.zeros(off, dtype=torch.bfloat16, device=device)
        # transposed bf16 copies of the feed-forward weights (same offsets as in Pb), read by the backward pass
        self.PbT = torch.zeros(off, dtype=torch.bfloat16, device=device)
        # fragment-major images of the decoder's weight matrices and of their transposes (same offsets as in Pb): the
        # decoder kernels that keep a molecule's rows in LDS read their weight operands from these
        self.PbF = torch.zeros(off, dtype=torch.bfloat16, device=device)
        self.PbTF = torch.zeros(off, dtype=torch.bfloat16, device=device)
        self.frag_descs = []
        for l in range(ndl):
            pre = f"decoder.layers.{l}."
            for name, rows, cols, trows in (("self_attn.in_proj_weight", 3 * D, D, 3 * D), ("self_attn.out_proj.weight", D, D, D),
                                            ("multihead_attn.in_proj_weight", 3 * D, D, D),
                                            ("multihead_attn.out_proj.weight", D, D, D),
                                            ("linear1.weight", FF, D, FF), ("linear2.weight", D, FF, D)):
                o = 2 * self.offset[pre + name]
                self.frag_descs.append((self.Pb.data_ptr() + o, self.PbF.data_ptr() + o, rows, cols, cols, 0))
                # the image of the transpose of the first `trows` rows: [cols][trows]
                self.frag_descs.append((self.Pb.data_ptr() + o, self.PbTF.data_ptr() + o, cols, trows, cols, 1))
        self.ffn1_offsets = [self.offset[k] for k in order if k.endswith("linear1.weight")]
        self.ffn2_offsets = [self.offset[k] for k in order if k.endswith("linear2.weight")]
        self.M = None
        self.V = None
        for k in order:
            p = named[k]
            view = self.P[self.offset[k]: self.offset[k] + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
        self.named = named
        self.use_real_data = bool(getattr(model.config, "use_real_data", False))
        self.shadow_stale = True
        self.grads_zeroed = False      # True right after a step whose fused optimizer left the active gradients zeroed
        self.weights_serial = 0        # counts rewrites of the bf16 weight copy (caches of projected tensors key on it)
        self.transposed_ready = None   # event after the last refresh of PbT on another stream than the caller's

    def owns(self, model):
        for k in (self.order[0], self.order[-1]):
            p = self.named[k]
            if p.data_ptr() != self.P.data_ptr() + 4 * self.offset[k] or p.device != self.P.device:
                return False
        return True

    def p32(self, k):
        return self.P.data_ptr() + 4 * self.offset[k]

    def g32(self, k):
        return self.G.data_ptr() + 4 * self.offset[k]

    def pb(self, k):
        return self.Pb.data_ptr() + 2 * self.offset[k]

    def pbt(self, k):
        return self.PbT.data_ptr() + 2 * self.offset[k]

    def pbf(self, k):
        return self.PbF.data_ptr() + 2 * self.offset[k]

    def pbtf(self, k):
        return self.PbTF.data_ptr() + 2 * self.offset[k]

    def frag_array(self):
        arr = (L.FragDesc * max(len(self.frag_descs), 1))()
        for i, (src, dst, rows, cols, ld, tr) in enumerate(self.frag_descs):
            arr[i].src, arr[i].dst, arr[i].rows, arr[i].cols, arr[i].ld, arr[i].transpose = src, dst, rows, cols, ld, tr
        return arr

    def refresh_fragments(self):
        """PbF / PbTF <- fragment-major images of the decoder weights in Pb."""
        if self.frag_descs:
            L.check(L.lib().mmst_fragment_bf16(len(self.frag_descs), self.frag_array(), L.stream_ptr()), "mmst_fragment_bf16")

    def refresh_transposed(self):
        """PbT <- transposes of the linear1 / linear2 weights in Pb (one launch per shape, 128 matrices each)."""
        for offs, rows, cols in ((self.ffn1_offsets, FF, D), (self.ffn2_offsets, D, FF)):
            for i in range(0, len(offs), 128):
                part = offs[i:i + 128]
                arr = (C.c_uint32 * len(part))(*part)
                L.check(L.lib().mmst_transpose_bf16(C.c_void_p(self.Pb.data_ptr()), C.c_void_p(self.PbT.data_ptr()),
                                                    len(part), arr, rows, cols, L.stream_ptr()), "mmst_transpose_bf16")

    def grad_view(self, k):
        p = self.named[k]
        return self.G[self.offset[k]: self.offset[k] + p.numel()].view(p.shape)

    def refresh_shadow(self):
        L.check(L.lib().mmst_cast_bf16(C.c_void_p(self.P.data_ptr()), C.c_void_p(self.Pb.data_ptr()), 1,
                                       self.total, self.total, L.stream_ptr()), "mmst_cast_bf16")
        self.refresh_transposed()
        self.refresh_fragments()
        self.shadow_stale = False
        self.weights_serial += 1

    def active_keys(self, mode):
        """Parameters that receive a gradient in this training_mode (the reference leaves the others at
        grad None, so AdamW never touches them, weight decay included): encoders / embeddings of absent
        modalities, fp1 (the fingerprint is not part of the loss) and real_data_linear are excluded."""
        tags = set(_mode_tags(mode))
        for t in ("MF", "MS", "MW"):
            if t in mode:
                tags.add(t)
        out = []
        for k in self.order:
            if k.startswith("fp1.") or (k.startswith("real_data_linear.") and not self.use_real_data):
                continue
            if not self.named[k].requires_grad:   # frozen (MT_functions_v15_4.py:124-133 freezes all but real_data_linear)
                continue
            skip = False
            for t in ALL_MODS:
                if (k.startswith(f"encoder_{t}.") or k.startswith(f"linear_spec_embedding_{t}.")) and t not in tags:
                    skip = True
            for t in ("MF", "MS", "MW"):
                if k.startswith(f"linear_embedding_{t}.") and t not in tags:
                    skip = True
            if not skip:
                out.append(k)
        return out

    def reduce_segments(self):
        """Gradient-reduction segments {name: (lo, hi)} in the order backward completes them: 'decoder' (decoder
        layers + heads), 'cross', one per modality-encoder depth from the top ('modal_l5' ... 'modal_l0'), 'embed'."""
        m = self.marks
        nl = sum(1 for k in m if k.startswith("modal_l"))
        segs = {"decoder": (m["decoder"], self.total), "cross": (m["cross"], m["decoder"])}
        for l in reversed(range(nl)):
            hi = m[f"modal_l{l + 1}"] if l + 1 < nl else m["cross"]
            segs[f"modal_l{l}"] = (m[f"modal_l{l}"], hi)
        segs["embed"] = (0, m["modal"])
        return segs

    def active_ranges(self, mode):
        """Contiguous [lo, hi) element ranges of the flat buffers covering the active parameters."""
        act = set(self.active_keys(mode))
        ranges = []
        for i, k in enumerate(self.order):
            if k not in act:
                continue
            lo = self.offset[k]
            hi = self.offset[self.order[i + 1]] if i + 1 < len(self.order) else self.total
            if ranges and ranges[-1][1] == lo:
                ranges[-1][1] = hi
            else:
                ranges.append([lo, hi])
        return [tuple(r) for r in ranges]

    def prepare_grads(self, mode):
        """torch.autograd semantics for the flat gradient buffer before a backward pass ACCUMULATES into it: a
        parameter whose .grad is None (optimizer.zero_grad() sets it to None by default) starts from zero, one whose
        .grad already is its slice of the buffer keeps accumulating, a foreign .grad tensor is taken over."""
        act = self.active_keys(mode)
        if all(self.named[k].grad is None for k in act):
            for lo, hi in self.active_ranges(mode):
                self.G[lo:hi].zero_()
        else:
            for k in act:
                p, view = self.named[k], self.grad_view(k)
                if p.grad is None:
                    view.zero_()
                elif p.grad.data_ptr() != view.data_ptr():
                    view.copy_(p.grad)
        self.attach_grads(mode)

    def param_signature(self):
        """Changes whenever a torch optimizer (or any in-place operation through the nn.Parameters) has written
        the fp32 weights since the last call."""
        return sum(p._version for p in self.named.values())

    def attach_grads(self, mode=None):
        act = set(self.active_keys(mode)) if mode is not None else None
        for k, p in self.named.items():
            p.grad = self.grad_view(k) if (act is None or k in act) else None


class _Carver:
    def __init__(self):
        self.items = []
        self.size = {torch.bfloat16: 0, torch.float32: 0, torch.int64: 0}

    def take(self, n, dtype=torch.bfloat16):
        off = self.size[dtype]
        self.size[dtype] += _align(n, 128)
        h = [dtype, off, n, None]
        self.items.append(h)
        return h

    def finish(self, device):
        self.buf = {dt: torch.zeros(max(n, 1), dtype=dt, device=device) for dt, n in self.size.items()}
        for h in self.items:
            h[3] = self.buf[h[0]].data_ptr() + h[1] * self.buf[h[0]].element_size()

    def view(self, h, *shape):
        """Tensor view of the leading prod(shape) elements of a carved buffer (buffers may hold spare capacity)."""
        n = 1
        for d in shape:
            n *= d
        if n > h[2]:
            raise L.MMSTError(f"view of {n} elements exceeds the buffer's {h[2]}")
        return self.buf[h[0]][h[1]: h[1] + n].view(*shape)


def _P(h):
    return h[3]


class Plan:
    """Buffers and prebuilt C structs for one (batch size, target length, training_mode)."""

    def __init__(self, eng, B, T, mode):
        self.eng, self.B, self.T, self.mode = eng, B, T, mode
        # decoder buffers are carved for this many target positions: a full-length decode fits every plan
        self.cap_T = max(T, 1, int(eng.model.max_len))
        m = eng.model
        ar = eng.arena
        nl, ndl = m.num_encoder_layers, m.num_decoder_layers
        self.has_mf, self.has_ms, self.has_mw = "MF" in mode, "MS" in mode, "MW" in mode
        tail = (64 if self.has_mf else 0) + (64 if self.has_ms else 0) + (1 if self.has_mw else 0)
        feature_dim = 193 if self.has_ms else 129          # models_MMT_v15_4.py:834-835
        feature_dim_IR = 130 if self.has_ms else 66
        self.mods = []          # (name, S, offset in memory)
        self.blank = []         # (name, rows, offset)
        off = 0
        for name in ALL_MODS:
            if name in _mode_tags(mode):
                S = (1 if name == "IR" else 64) + tail
                self.mods.append((name, S, off))
                off += S
            else:
                rows = 65 if name == "COSY" else (feature_dim_IR if name == "IR" else feature_dim)
                self.blank.append((name, rows, off))
                off += rows
        self.Lmem = off
        Lm = self.Lmem
        cv = _Carver()
        f32, bf = torch.float32, torch.bfloat16
        # inputs / embeddings
        self.ir_bf16 = cv.take(B * 1000)
        self.xseq = {n: cv.take(B * S * D) for n, S, _ in self.mods}
        self.kbias = {n: cv.take(B * S, f32) for n, S, _ in self.mods}
        self.kbias_glob = cv.take(B * Lm, f32)
        self.mem_in = cv.take(B * Lm * D)
        self.memory = None

        def enc_acts(M, Hh, S_):
            return dict(qkv=cv.take(M * 3 * D), attn=cv.take(M * D), lse=cv.take(B * Hh * S_, f32), z1=cv.take(M * D),
                        st1=cv.take(M * 2, f32), x1=cv.take(M * D), h=cv.take(M * FF), z2=cv.take(M * D),
                        st2=cv.take(M * 2, f32), y=cv.take(M * D), hsign=cv.take(M * 64, f32))

        self.modal_acts = [[enc_acts(B * S, 16, S) for _, S, _ in self.mods] for _ in range(nl)]
        self.cross_acts = [enc_acts(B * Lm, 4, Lm) for _ in range(nl)]
        self.mean = cv.take(B * D)
        self.fingerprint = cv.take(B * m.fingerprint_size, f32)
        Tm = self.cap_T
        M = B * Tm
        self.tok = cv.take((m.max_len + 1) * B, torch.int64)
        self.trg_emb = cv.take(M * D)
        self.dec_acts = []
        for _ in range(ndl):
            self.dec_acts.append(dict(
                qkv=cv.take(M * 3 * D), attn1=cv.take(M * D), lse1=cv.take(B * 16 * Tm, f32), z1=cv.take(M * D),
                st1=cv.take(M * 2, f32), x1=cv.take(M * D), qc=cv.take(M * D), kvc=cv.take(B * Lm * 2 * D),
                attn2=cv.take(M * D), lse2=cv.take(B * 16 * Tm, f32), z2=cv.take(M * D), st2=cv.take(M * 2, f32),
                x2=cv.take(M * D), h=cv.take(M * FF), z3=cv.take(M * D), st3=cv.take(M * 2, f32), y=cv.take(M * D)))
        self.ffn_ws = cv.take(8 * M * D, f32)   # hidden-split partial sums of the decoder feed-forward blocks
        self.logits = cv.take(M * VPAD, f32)
        self.dlogits = cv.take(M * VPAD)
        self.loss = cv.take(1, f32)
        # use_real_data branch: real_data_linear(mean) [B][64], its gradient, fp32 mean, gradient at the mean
        self.rd_extra, self.rd_dextra = cv.take(B * VPAD, f32), cv.take(B * VPAD, f32)
        self.rd_mean32, self.rd_dmean = cv.take(B * D, f32), cv.take(B * D, f32)
        self.next_prob = cv.take((m.max_len + 1) * B, f32)

        # backward scratch: one set PER LAYER, never reused inside a step, so the weight-gradient launches
        # can run on a second stream with no write-after-read hazards (HBM is not the scarce resource here)
        def bwd_scratch(Mr, Hh, S_):
            return dict(dh=cv.take(Mr * FF), dz1=cv.take(Mr * D), dzd1=cv.take(Mr * D), da=cv.take(Mr * D),
                        dqkv=cv.take(Mr * 3 * D), delta=cv.take(B * Hh * S_, f32),
                        dz=cv.take(Mr * D), dzd=cv.take(Mr * D))

        self.modal_bwd = [[bwd_scratch(B * S, 16, S) for _, S, _ in self.mods] for _ in range(nl)]
        self.dxseq = [cv.take(B * S * D) for _, S, _ in self.mods]
        self.cross_bwd = [bwd_scratch(B * Lm, 4, Lm) for _ in range(nl)]
        self.dec_bwd = []
        for _ in range(ndl):
            sc = bwd_scratch(M, 16, max(Tm, 1))
            sc.update(dz2=cv.take(M * D), dzd2=cv.take(M * D), dqc=cv.take(M * D), da2=cv.take(M * D),
                      dkvc=cv.take(B * Lm * 2 * D), delta2=cv.take(B * 16 * max(Tm, 1), f32))
            self.dec_bwd.append(sc)
        self.dmem = cv.take(B * Lm * D)
        self.dmem_in = cv.take(B * Lm * D)
        self.dtrg = cv.take(M * D)
        self.ir_dpre = cv.take(B * D)
        cv.finish(eng.device)
        self.cv = cv
        # constant parts of the global key bias: blank blocks (float ones -> +1; blank IR is bool False -> 0)
        kg = cv.view(self.kbias_glob, B, Lm)
        for name, rows, o in self.blank:
            kg[:, o:o + rows] = 0.0 if name == "IR" else 1.0
        self._build_structs()

    # ------------------------------------------------------------------
    def _enc_params(self, prefix):
        ar = self.eng.arena
        p = L.EncParams()
        g = L.EncGrads()
        names = dict(w_qkv="self_attn.in_proj_weight", b_qkv="self_attn.in_proj_bias", w_o="self_attn.out_proj.weight",
                     b_o="self_attn.out_proj.bias", w_1="linear1.weight", b_1="linear1.bias", w_2="linear2.weight",
                     b_2="linear2.bias", g1="norm1.weight", be1="norm1.bias", g2="norm2.weight", be2="norm2.bias")
        for f, k in names.items():
            key = prefix + k
            setattr(p, f, ar.pb(key) if f.startswith("w_") else ar.p32(key))
            setattr(g, f, ar.g32(key))
        p.w_1t = ar.pbt(prefix + "linear1.weight")
        p.w_2t = ar.pbt(prefix + "linear2.weight")
        return p, g

    def _dec_params(self, prefix):
        ar = self.eng.arena
        p = L.DecParams()
        g = L.DecGrads()
        names = dict(w_qkv="self_attn.in_proj_weight", b_qkv="self_attn.in_proj_bias", w_o="self_attn.out_proj.weight",
                     b_o="self_attn.out_proj.bias", w_cqkv="multihead_attn.in_proj_weight",
                     b_cqkv="multihead_attn.in_proj_bias", w_co="multihead_attn.out_proj.weight",
                     b_co="multihead_attn.out_proj.bias", w_1="linear1.weight", b_1="linear1.bias",
                     w_2="linear2.weight", b_2="linear2.bias", g1="norm1.weight", be1="norm1.bias",
                     g2="norm2.weight", be2="norm2.bias", g3="norm3.weight", be3="norm3.bias")
        for f, k in names.items():
            key = prefix + k
            setattr(p, f, ar.pb(key) if f.startswith("w_") else ar.p32(key))
            setattr(g, f, ar.g32(key))
        p.w_1t = ar.pbt(prefix + "linear1.weight")
        p.w_2t = ar.pbt(prefix + "linear2.weight")
        for f, k in (("qkv", "self_attn.in_proj_weight"), ("o", "self_attn.out_proj.weight"),
                     ("cqkv", "multihead_attn.in_proj_weight"), ("co", "multihead_attn.out_proj.weight"),
                     ("1", "linear1.weight"), ("2", "linear2.weight")):
            setattr(p, "f_" + f, ar.pbf(prefix + k))
            setattr(p, "ft_" + f, ar.pbtf(prefix + k))
        return p, g

    def _build_structs(self):
        m, ar, B, T, Lm = self.eng.model, self.eng.arena, self.B, max(self.T, 1), self.Lmem
        nl, ndl = m.num_encoder_layers, m.num_decoder_layers
        n = len(self.mods)
        self.n_groups = n
        # ---- modality encoder layers (grouped) ----
        self.modal_p, self.modal_g, self.modal_a, self.modal_w = [], [], [], []
        self.modal_dzd, self.cross_dzd, self.dec_dzd = [], [], []   # (dz, dzd) input pointers per layer
        for l in range(nl):
            self.modal_dzd.append([])
            P_, G_, A_, W_ = (L.EncParams * n)(), (L.EncGrads * n)(), (L.EncActs * n)(), (L.EncBwd * n)()
            for i, (name, S, off) in enumerate(self.mods):
                P_[i], G_[i] = self._enc_params(f"encoder_{name}.layers.{l}.")
                a, acts = A_[i], self.modal_acts[l][i]
                a.x = _P(self.xseq[name]) if l == 0 else _P(self.modal_acts[l - 1][i]["y"])
                last = l == nl - 1
                a.y = _P(self.mem_in) if last else _P(acts["y"])
                a.y_L, a.y_off = (Lm, off) if last else (0, 0)
                a.kbias = _P(self.kbias[name])
                for f in ("qkv", "attn", "lse", "z1", "st1", "x1", "h", "z2", "st2", "hsign"):
                    setattr(a, f, _P(acts[f]))
                a.B, a.S = B, S
                w, sc = W_[i], self.modal_bwd[l][i]
                w.dz2, w.dzd2 = _P(sc["dz"]), _P(sc["dzd"])
                self.modal_dzd[l].append((w.dz2, w.dzd2))
                for f in ("dh", "dz1", "dzd1", "da", "dqkv", "delta"):
                    setattr(w, f, _P(sc[f]))
                if l > 0:
                    w.dx, w.dxd = _P(self.modal_bwd[l - 1][i]["dz"]), _P(self.modal_bwd[l - 1][i]["dzd"])
                else:
                    w.dx = _P(self.dxseq[i])
                if l > 0:
                    pa = self.modal_acts[l - 1][i]
                    pk = f"encoder_{name}.layers.{l - 1}."
                    w.prev_z2, w.prev_st2 = _P(pa["z2"]), _P(pa["st2"])
                    w.prev_g2 = ar.p32(pk + "norm2.weight")
                    w.prev_dg2, w.prev_dbe2 = ar.g32(pk + "norm2.weight"), ar.g32(pk + "norm2.bias")
            self.modal_p.append(P_); self.modal_g.append(G_); self.modal_a.append(A_); self.modal_w.append(W_)
        # ---- cross encoder ----
        self.cross_p, self.cross_g, self.cross_a, self.cross_w = [], [], [], []
        for l in range(nl):
            P_, G_, A_, W_ = (L.EncParams * 1)(), (L.EncGrads * 1)(), (L.EncActs * 1)(), (L.EncBwd * 1)()
            P_[0], G_[0] = self._enc_params(f"encoder_cross.layers.{l}.")
            a, acts = A_[0], self.cross_acts[l]
            a.x = _P(self.mem_in) if l == 0 else _P(self.cross_acts[l - 1]["y"])
            a.y = _P(acts["y"])
            a.kbias = _P(self.kbias_glob)
            for f in ("qkv", "attn", "lse", "z1", "st1", "x1", "h", "z2", "st2", "hsign"):
                setattr(a, f, _P(acts[f]))
            a.B, a.S = B, Lm
            w, sc = W_[0], self.cross_bwd[l]
            w.dz2, w.dzd2 = _P(sc["dz"]), _P(sc["dzd"])
            self.cross_dzd.append((w.dz2, w.dzd2))
            for f in ("dh", "dz1", "dzd1", "da", "dqkv", "delta"):
                setattr(w, f, _P(sc[f]))
            if l > 0:
                w.dx, w.dxd = _P(self.cross_bwd[l - 1]["dz"]), _P(self.cross_bwd[l - 1]["dzd"])
                pa, pk = self.cross_acts[l - 1], f"encoder_cross.layers.{l - 1}."
                w.prev_z2, w.prev_st2 = _P(pa["z2"]), _P(pa["st2"])
                w.prev_g2 = ar.p32(pk + "norm2.weight")
                w.prev_dg2, w.prev_dbe2 = ar.g32(pk + "norm2.weight"), ar.g32(pk + "norm2.bias")
            else:
                w.dx = _P(self.dmem_in)
            self.cross_p.append(P_); self.cross_g.append(G_); self.cross_a.append(A_); self.cross_w.append(W_)
        self.memory = self.cross_acts[nl - 1]["y"]
        # ---- decoder ----
        self.dec_p, self.dec_g, self.dec_a, self.dec_w = [], [], [], []
        for l in range(ndl):
            P_, G_ = self._dec_params(f"decoder.layers.{l}.")
            a, acts = L.DecActs(), self.dec_acts[l]
            a.x = _P(self.trg_emb) if l == 0 else _P(self.dec_acts[l - 1]["y"])
            a.y = _P(acts["y"])
            a.mem = _P(self.memory)
            a.mem_bias = _P(self.kbias_glob)
            for f in ("qkv", "attn1", "lse1", "z1", "st1", "x1", "qc", "kvc", "attn2", "lse2", "z2", "st2", "x2", "h",
                      "z3", "st3"):
                setattr(a, f, _P(acts[f]))
            a.B, a.T, a.L = B, T, Lm
            a.ffn_ws = _P(self.ffn_ws)
            w, sc = L.DecBwd(), self.dec_bwd[l]
            w.dz3, w.dzd3 = _P(sc["dz"]), _P(sc["dzd"])
            self.dec_dzd.append((w.dz3, w.dzd3))
            for f in ("dh", "dz2", "dzd2", "da", "da2", "dqc", "dkvc", "dz1", "dzd1", "dqkv", "delta", "delta2"):
                setattr(w, f, _P(sc[f]))
            w.dmem = _P(self.dmem)
            w.dmem_accumulate = 0 if l == ndl - 1 else 1
            if l > 0:
                w.dx, w.dxd = _P(self.dec_bwd[l - 1]["dz"]), _P(self.dec_bwd[l - 1]["dzd"])
                pa, pk = self.dec_acts[l - 1], f"decoder.layers.{l - 1}."
                w.prev_z3, w.prev_st3 = _P(pa["z3"]), _P(pa["st3"])
                w.prev_g3 = ar.p32(pk + "norm3.weight")
                w.prev_dg3, w.prev_dbe3 = ar.g32(pk + "norm3.weight"), ar.g32(pk + "norm3.bias")
            else:
                w.dx = _P(self.dtrg)
            self.dec_p.append(P_); self.dec_g.append(G_); self.dec_a.append(a); self.dec_w.append(w)

    def set_T(self, T, kv_ready=0):
        """Decode loops re-run the decoder with a growing prefix in the same buffers."""
        if T > self.cap_T:
            raise L.MMSTError(f"decode length {T} exceeds the plan's capacity of {self.cap_T} target positions")
        for a in self.dec_a:
            a.T = T
            a.kv_ready = kv_ready
        self.T = T


def _mode_tags(mode):
    """Substring gating exactly as the reference does it (`"1H" in training_mode`, ...)."""
    return [t for t in ALL_MODS if t in mode]


_EMB_KEYS = {
    "1H": "linear_spec_embedding_1H.point_embedding_layer_1H.fc_H",
    "13C": "linear_spec_embedding_13C.point_embedding_layer_13C.fc_C",
    "HSQC": "linear_spec_embedding_HSQC.point_embedding_layer_HSQC.fc_HSQC",
    "COSY": "linear_spec_embedding_COSY.point_embedding_layer_COSY.fc_COSY",
    "IR": "linear_spec_embedding_IR.linear_spec_embedding_IR",
}


class Engine:
    def __init__(self, model):
        self.model = model
        self.arena = None
        self.plans = {}
        self.device = None
        self.step_seed = 0x1234567
        self.interleave_wgrads = _INTERLEAVE_WGRADS
        # decoder backward: the memory's dK/dV pass and its projection into dmem run on the weight-gradient stream
        self.decoder_kv_on_wstream = os.environ.get("MMST_DEC_KV_WSTREAM", "1") != "0"

    # ------------------------------------------------------------------
    def ensure(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise L.MMSTError("the MultimodalTransformer hot path runs on an MI355X only (device must be cuda); "
                              "there is no CPU implementation in this package")
        L.lib()
        if self.arena is None or self.device != device or not self.arena.owns(self.model):
            self.device = device
            self.arena = Arena(self.model, device)
            self.plans = {}
        ar = self.arena
        if self.model._external_optimizer:
            # a torch optimizer updates the fp32 weights in place, in train or eval mode alike: refresh the bf16
            # copies whenever the parameters' version counters have moved (the fused AdamW keeps them itself)
            sig = ar.param_signature()
            if sig != getattr(ar, "_seen_signature", None):
                ar.shadow_stale = True
                ar._seen_signature = sig
        if ar.shadow_stale:
            ar.refresh_shadow()

    def plan(self, B, T, mode):
        key = (B, T, mode)
        pl = self.plans.get(key)
        if pl is None:
            pl = Plan(self, B, T, mode)
            self.plans[key] = pl
        return pl

    def layer_seed(self, uid):
        return (self.step_seed * 2654435761 + uid * 64) & 0xFFFFFFFF

    # ------------------------------------------------------------------
    def _embed_structs(self, pl, batch, backward=False):
        ar, n, B = self.arena, pl.n_groups, pl.B
        G_ = (L.EmbedGroup * n)()
        keep = []

        def f32(t):
            t = t.to(device=self.device, dtype=torch.float32).contiguous()
            keep.append(t)
            return t.data_ptr()

        mode = pl.mode
        mf_mask, ms_mask = batch.get("mask_MF"), batch.get("mask_MS")
        mf_bool = bool(pl.has_mf and mf_mask is not None and mf_mask.dtype == torch.bool)
        ms_bool = bool(pl.has_ms and ms_mask is not None and ms_mask.dtype == torch.bool)
        comps_bool = (not pl.has_mf or mf_bool) and (not pl.has_ms or ms_bool)
        # per-modality concatenated mask: bool only if every component is bool (spectrum, IR and MW masks are)
        mod_bool = comps_bool
        # global mask: additionally every blank 1H/13C/HSQC/COSY block is a float block of ones
        glob_bool = comps_bool and all(name == "IR" for name, _, _ in pl.blank)
        for i, (name, S, off) in enumerate(pl.mods):
            g = G_[i]
            g.xseq, g.kbias = _P(pl.xseq[name]), _P(pl.kbias[name])
            g.kbias_glob = _P(pl.kbias_glob)
            g.S, g.L_glob, g.off_glob, g.mod_bool = S, pl.Lmem, off, int(mod_bool)
            key = _EMB_KEYS[name]
            if name == "IR":
                g.in_dim, g.n_spec = 0, 1
                g.ir_dpre = _P(pl.ir_dpre)
            else:
                g.in_dim, g.n_spec = (1 if name == "13C" else 2), 64
                g.src, g.mask = f32(batch["src_" + name]), f32(batch["mask_" + name])
                g.w, g.b = ar.p32(key + ".weight"), ar.p32(key + ".bias")
                g.dw, g.db = ar.g32(key + ".weight"), ar.g32(key + ".bias")
        s = L.EmbedShared()
        s.B, s.glob_bool = B, int(glob_bool)
        s.has_mf, s.has_ms, s.has_mw = int(pl.has_mf), int(pl.has_ms), int(pl.has_mw)
        if pl.has_mf:
            t = batch["src_MF"].to(self.device, torch.int64).contiguous()
            keep.append(t)
            s.mf_tok, s.mf_mask, s.mf_bool = t.data_ptr(), f32(mf_mask), int(mf_bool)
            s.mf_table = ar.p32("linear_embedding_MF.embedding.weight")
            s.d_mf_table = ar.g32("linear_embedding_MF.embedding.weight")
        if pl.has_ms:
            t = batch["src_MS"].to(self.device, torch.int64).contiguous()
            keep.append(t)
            s.ms_tok, s.ms_mask, s.ms_bool = t.data_ptr(), f32(ms_mask), int(ms_bool)
            s.ms_table = ar.p32("linear_embedding_MS.embedding.weight")
            s.d_ms_table = ar.g32("linear_embedding_MS.embedding.weight")
        if pl.has_mw:
            s.mw = f32(batch["trg_MW"].reshape(-1))
            k = "linear_embedding_MW.linear_spec_embedding_MW"
            s.mw_w, s.mw_b = ar.p32(k + ".weight"), ar.p32(k + ".bias")
            s.d_mw_w, s.d_mw_b = ar.g32(k + ".weight"), ar.g32(k + ".bias")
        return G_, s, keep, glob_bool

    # ------------------------------------------------------------------
    def encode(self, pl, batch, train):
        """Embeddings + modality encoders + cross encoder + fingerprint head."""
        lib, st, ar, B, Lm = L.lib(), L.stream_ptr(), self.arena, pl.B, pl.Lmem
        m = self.model
        p_drop = LAYER_DROPOUT if train else 0.0
        G_, s, keep, glob_bool = self._embed_structs(pl, batch)
        pl.encode_serial = getattr(pl, "encode_serial", 0) + 1
        pl.embed_structs = (G_, s, keep)
        pl.mask_is_bool = glob_bool
        L.check(lib.mmst_embed_fwd(pl.n_groups, G_, C.byref(s), st), "mmst_embed_fwd")
        names = [n for n, _, _ in pl.mods]
        if "IR" in names:
            S_ir = dict((n, S) for n, S, _ in pl.mods)["IR"]
            src = batch["src_IR"].to(self.device, torch.float32).contiguous()
            keep.append(src)
            L.check(lib.mmst_cast_bf16(C.c_void_p(src.data_ptr()), C.c_void_p(_P(pl.ir_bf16)), B, 1000, 1000, st),
                    "mmst_cast_bf16")
            k = _EMB_KEYS["IR"]
            ops.gemm([dict(P=_P(pl.ir_bf16), Q=ar.pb(k + ".weight"), C=_P(pl.xseq["IR"]), bias=ar.p32(k + ".bias"),
                           ldp=1000, ldq=1000, ldc=S_ir * D, Pn=B, Qn=D, K=1000, flags=L.GF_RELU)], L.EPI_BIAS)
        nl = m.num_encoder_layers
        for l in range(nl):
            L.check(lib.mmst_encoder_layer_fwd(pl.n_groups, pl.modal_p[l], pl.modal_a[l], 16, C.c_float(p_drop),
                                               C.c_uint32(self.layer_seed(l)), st), "mmst_encoder_layer_fwd")
        for l in range(nl):
            L.check(lib.mmst_encoder_layer_fwd(1, pl.cross_p[l], pl.cross_a[l], m.cross_heads, C.c_float(p_drop),
                                               C.c_uint32(self.layer_seed(32 + l)), st), "mmst_encoder_layer_fwd")
        L.check(lib.mmst_seq_mean(C.c_void_p(_P(pl.memory)), C.c_void_p(_P(pl.mean)), B, Lm, st), "mmst_seq_mean")
        ops.gemm([dict(P=_P(pl.mean), Q=ar.pb("fp1.weight"), C=_P(pl.fingerprint), bias=ar.p32("fp1.bias"),
                       ldp=D, ldq=D, ldc=m.fingerprint_size, Pn=B, Qn=m.fingerprint_size, K=D,
                       flags=L.GF_OUT_F32)], L.EPI_BIAS)

    def decode(self, pl, train, T=None, embed=True, kv_ready=0, tok_ptr=None):
        """Target embedding + decoder + vocabulary projection on the tokens held in pl.tok[:T]."""
        lib, st, ar, B = L.lib(), L.stream_ptr(), self.arena, pl.B
        m = self.model
        T = pl.T if T is None else T
        if T != pl.dec_a[0].T or kv_ready != pl.dec_a[0].kv_ready:
            pl.set_T(T, kv_ready)
        p_drop = LAYER_DROPOUT if train else 0.0
        p_drop2 = float(m.config.drop_out) if train else 0.0
        if embed:
            L.check(lib.mmst_trg_embed_fwd(C.c_void_p(tok_ptr or _P(pl.tok)), B, C.c_void_p(ar.p32("embed_trg.weight")),
                                           C.c_void_p(ar.p32("pe_trg.weight")), C.c_void_p(_P(pl.trg_emb)), B, T,
                                           C.c_float(p_drop2), C.c_uint32(self.layer_seed(100)), st),
                    "mmst_trg_embed_fwd")
        if ar.transposed_ready is not None:   # fragment-major weight images refreshed on the second stream
            torch.cuda.current_stream().wait_event(ar.transposed_ready)
        ndl = m.num_decoder_layers
        pp = (C.c_void_p * ndl)(*[C.addressof(pl.dec_p[l]) for l in range(ndl)])
        aa = (C.c_void_p * ndl)(*[C.addressof(pl.dec_a[l]) for l in range(ndl)])
        seeds = (C.c_uint32 * ndl)(*[self.layer_seed(64 + l) for l in range(ndl)])
        L.check(lib.mmst_decoder_stack_fwd(ndl, pp, aa, 16, C.c_float(p_drop), seeds, st, None), "mmst_decoder_stack_fwd")
        out = pl.dec_acts[m.num_decoder_layers - 1]["y"]
        ops.gemm([dict(P=_P(out), Q=ar.pb("fc_out.weight"), C=_P(pl.logits), bias=ar.p32("fc_out.bias"),
                       ldp=D, ldq=D, ldc=VPAD, Pn=B * T, Qn=VPAD, K=D, flags=L.GF_OUT_F32)], L.EPI_BIAS)
        if getattr(m.config, "use_real_data", False):
            # models_MMT_v15_4.py:965-971: logits = (fc_out(decoder) + real_data_linear(mean(memory))) / 2
            ops.gemm([dict(P=_P(pl.mean), Q=ar.pb("real_data_linear.weight"), C=_P(pl.rd_extra),
                           bias=ar.p32("real_data_linear.bias"), ldp=D, ldq=D, ldc=VPAD, Pn=B, Qn=VPAD, K=D,
                           flags=L.GF_OUT_F32)], L.EPI_BIAS)
            L.check(lib.mmst_real_data_mix(C.c_void_p(_P(pl.logits)), C.c_void_p(_P(pl.rd_extra)), B, T, st),
                    "mmst_real_data_mix")

    def real_data_backward(self, pl, into_memory=True):
        """Backward of the use_real_data branch from pl.dlogits (which must already carry the factor 1/2 of the
        average): gradients of real_data_linear and, if into_memory, the gradient at mean(memory) in pl.rd_dmean."""
        lib, st, ar, B, T = L.lib(), L.stream_ptr(), self.arena, pl.B, pl.T
        vp = C.c_void_p
        L.check(lib.mmst_real_data_rows_bwd(vp(_P(pl.dlogits)), vp(_P(pl.mean)), vp(_P(pl.rd_dextra)),
                                            vp(_P(pl.rd_mean32)), vp(ar.g32("real_data_linear.bias")), B, T, st),
                "mmst_real_data_rows_bwd")
        one, zero = C.c_float(1.0), C.c_float(0.0)
        # dW[c][j] += sum_b d_extra[b][c] mean[b][j]
        L.check(lib.mmst_sgemm(vp(_P(pl.rd_dextra)), C.c_long(1), C.c_long(VPAD), vp(_P(pl.rd_mean32)), C.c_long(D),
                               C.c_long(1), vp(ar.g32("real_data_linear.weight")), D, None, VPAD, D, B, one, one, st),
                "mmst_sgemm")
        if into_memory:   # d_mean[b][j] = sum_c d_extra[b][c] W[c][j]
            L.check(lib.mmst_sgemm(vp(_P(pl.rd_dextra)), C.c_long(VPAD), C.c_long(1),
                                   vp(ar.p32("real_data_linear.weight")), C.c_long(D), C.c_long(1), vp(_P(pl.rd_dmean)),
                                   D, None, B, D, VPAD, one, zero, st), "mmst_sgemm")

    def ce_loss(self, pl, trg_enc_SMI, grad_scale=1.0, want_grad=True):
        """CrossEntropyLoss over all positions (PAD included); fills pl.dlogits."""
        lib, st = L.lib(), L.stream_ptr()
        trg = trg_enc_SMI.to(self.device, torch.int64).contiguous()
        pl._trg_keep = trg
        pl.cv.view(pl.loss, 1).zero_()
        L.check(lib.mmst_ce_loss(C.c_void_p(_P(pl.logits)), VPAD, self.model.out_size, C.c_void_p(trg.data_ptr()),
                                 trg.shape[1], pl.B, pl.T, C.c_void_p(_P(pl.loss)),
                                 C.c_void_p(_P(pl.dlogits)) if want_grad else None, C.c_float(grad_scale), st),
                "mmst_ce_loss")
        return pl.cv.view(pl.loss, 1)

    # ------------------------------------------------------------------
    def step_plan(self, pl):
        """The C description of `pl` for mmst_train_step (built once per plan; the ctypes arrays of the plan stay
        referenced from it)."""
        sp = getattr(pl, "_step_plan", None)
        if sp is not None:
            return sp
        m, ar = self.model, self.arena
        nl, ndl, n = m.num_encoder_layers, m.num_decoder_layers, pl.n_groups
        if nl > L.MAX_LAYERS or ndl > L.MAX_LAYERS:
            raise L.MMSTError(f"mmst_train_step supports at most {L.MAX_LAYERS} layers per stack")
        sp = L.StepPlan()
        sp.B, sp.T, sp.L, sp.n_groups = pl.B, max(pl.T, 1), pl.Lmem, n
        sp.n_enc_layers, sp.n_dec_layers, sp.cross_heads = nl, ndl, m.cross_heads
        sp.vocab, sp.in_size, sp.fingerprint_size = m.out_size, m.config.in_size, m.fingerprint_size
        sp.ir_group = -1
        for i, (name, S, off) in enumerate(pl.mods):
            g = sp.embed_g[i]
            g.xseq, g.kbias, g.kbias_glob = _P(pl.xseq[name]), _P(pl.kbias[name]), _P(pl.kbias_glob)
            g.S, g.L_glob, g.off_glob = S, pl.Lmem, off
            key = _EMB_KEYS[name]
            if name == "IR":
                g.in_dim, g.n_spec, g.ir_dpre = 0, 1, _P(pl.ir_dpre)
                sp.ir_group = i
                sp.ir_bf16 = _P(pl.ir_bf16)
                sp.ir_w, sp.ir_b = ar.pb(key + ".weight"), ar.p32(key + ".bias")
                sp.ir_dw, sp.ir_db = ar.g32(key + ".weight"), ar.g32(key + ".bias")
            else:
                g.in_dim, g.n_spec = (1 if name == "13C" else 2), 64
                g.w, g.b = ar.p32(key + ".weight"), ar.p32(key + ".bias")
                g.dw, g.db = ar.g32(key + ".weight"), ar.g32(key + ".bias")
            sp.dxseq[i] = _P(pl.dxseq[i])
        es = sp.embed_s
        es.B = pl.B
        es.has_mf, es.has_ms, es.has_mw = int(pl.has_mf), int(pl.has_ms), int(pl.has_mw)
        if pl.has_mf:
            es.mf_table = ar.p32("linear_embedding_MF.embedding.weight")
            es.d_mf_table = ar.g32("linear_embedding_MF.embedding.weight")
        if pl.has_ms:
            es.ms_table = ar.p32("linear_embedding_MS.embedding.weight")
            es.d_ms_table = ar.g32("linear_embedding_MS.embedding.weight")
        if pl.has_mw:
            k = "linear_embedding_MW.linear_spec_embedding_MW"
            es.mw_w, es.mw_b = ar.p32(k + ".weight"), ar.p32(k + ".bias")
            es.d_mw_w, es.d_mw_b = ar.g32(k + ".weight"), ar.g32(k + ".bias")
        adr = C.addressof
        for l in range(nl):
            sp.modal_p[l], sp.modal_g[l] = adr(pl.modal_p[l]), adr(pl.modal_g[l])
            sp.modal_a[l], sp.modal_w[l] = adr(pl.modal_a[l]), adr(pl.modal_w[l])
            sp.cross_p[l], sp.cross_g[l] = adr(pl.cross_p[l]), adr(pl.cross_g[l])
            sp.cross_a[l], sp.cross_w[l] = adr(pl.cross_a[l]), adr(pl.cross_w[l])
            for i in range(n):
                sp.modal_dzd[l][i] = pl.modal_dzd[l][i][1]
            sp.cross_dzd[l] = pl.cross_dzd[l][1]
        for l in range(ndl):
            sp.dec_p[l], sp.dec_g[l] = adr(pl.dec_p[l]), adr(pl.dec_g[l])
            sp.dec_a[l], sp.dec_w[l] = adr(pl.dec_a[l]), adr(pl.dec_w[l])
            sp.dec_dzd[l] = pl.dec_dzd[l][1]
        sp.memory, sp.mean = _P(pl.memory), _P(pl.mean)
        sp.fp1_w, sp.fp1_b, sp.fingerprint = ar.pb("fp1.weight"), ar.p32("fp1.bias"), _P(pl.fingerprint)
        sp.trg_emb = _P(pl.trg_emb)
        sp.embed_trg, sp.pe_trg = ar.p32("embed_trg.weight"), ar.p32("pe_trg.weight")
        sp.d_embed_trg, sp.d_pe_trg = ar.g32("embed_trg.weight"), ar.g32("pe_trg.weight")
        sp.fc_out_w, sp.fc_out_b = ar.pb("fc_out.weight"), ar.p32("fc_out.bias")
        sp.d_fc_out_w, sp.d_fc_out_b = ar.g32("fc_out.weight"), ar.g32("fc_out.bias")
        sp.logits, sp.dlogits, sp.loss = _P(pl.logits), _P(pl.dlogits), _P(pl.loss)
        sp.dmem, sp.dmem_in, sp.dtrg = _P(pl.dmem), _P(pl.dmem_in), _P(pl.dtrg)
        if ar.M is None:
            ar.M, ar.V = torch.zeros_like(ar.P), torch.zeros_like(ar.P)
        sp.P, sp.G, sp.M, sp.V = ar.P.data_ptr(), ar.G.data_ptr(), ar.M.data_ptr(), ar.V.data_ptr()
        sp.Pb, sp.PbT = ar.Pb.data_ptr(), ar.PbT.data_ptr()
        ranges = ar.active_ranges(pl.mode)
        if len(ranges) > L.MAX_RANGES or len(ar.ffn1_offsets) > L.MAX_FFN:
            raise L.MMSTError("too many parameter ranges / feed-forward blocks for mmst_train_step")
        sp.n_ranges = len(ranges)
        for i, (lo, hi) in enumerate(ranges):
            sp.range_lo[i], sp.range_hi[i] = lo, hi
        sp.n_ffn = len(ar.ffn1_offsets)
        for i, (o1, o2) in enumerate(zip(ar.ffn1_offsets, ar.ffn2_offsets)):
            sp.ffn1_off[i], sp.ffn2_off[i] = o1, o2
        if len(ar.frag_descs) > L.MAX_FRAG:
            raise L.MMSTError("too many fragment-major weight images for mmst_step_plan")
        sp.n_frag = len(ar.frag_descs)
        for i, d in enumerate(ar.frag_array()[:sp.n_frag]):
            sp.frag[i] = d
        mk = ar.marks
        sp.seg_decoder, sp.seg_cross = mk["decoder"], mk["cross"]
        for l in range(nl):
            sp.seg_modal[l] = mk[f"modal_l{l}"]
        sp.seg_modal[nl] = mk["cross"]
        pl._step_plan = sp
        pl._step_args = L.StepArgs()
        return sp

    def train_step(self, pl, batch, train=True, zero_grads=True, adam=None, grad_scale=1.0, loss_grad_scale=1.0,
                   on_segment=None):
        """forward + CE + backward (+ AdamW when `adam` = (step count, lr, betas, eps, weight decay)) of one batch in ONE
        call into the library (mmst_train_step).  Same launches as encode() / decode() / ce_loss() / backward() and
        FusedAdamW.step(); returns the loss as a view of the plan's loss slot."""
        sp = self.step_plan(pl)
        a = pl._step_args
        keep = []

        def dev(t, dtype):
            t = t.to(device=self.device, dtype=dtype)
            t = t if t.is_contiguous() else t.contiguous()
            keep.append(t)
            return t.data_ptr()

        f32 = torch.float32
        mf_mask, ms_mask = batch.get("mask_MF"), batch.get("mask_MS")
        mf_bool = bool(pl.has_mf and mf_mask is not None and mf_mask.dtype == torch.bool)
        ms_bool = bool(pl.has_ms and ms_mask is not None and ms_mask.dtype == torch.bool)
        comps_bool = (not pl.has_mf or mf_bool) and (not pl.has_ms or ms_bool)
        glob_bool = comps_bool and all(name == "IR" for name, _, _ in pl.blank)
        for i, (name, S, off) in enumerate(pl.mods):
            if name == "IR":
                a.src[i], a.mask[i] = None, None
                a.src_ir = dev(batch["src_IR"], f32)
            else:
                a.src[i], a.mask[i] = dev(batch["src_" + name], f32), dev(batch["mask_" + name], f32)
        if pl.has_mf:
            a.mf_tok, a.mf_mask = dev(batch["src_MF"], torch.int64), dev(mf_mask, f32)
        if pl.has_ms:
            a.ms_tok, a.ms_mask = dev(batch["src_MS"], torch.int64), dev(ms_mask, f32)
        if pl.has_mw:
            a.mw = dev(batch["trg_MW"].reshape(-1), f32)
        a.mf_bool, a.ms_bool, a.mod_bool, a.glob_bool = int(mf_bool), int(ms_bool), int(comps_bool), int(glob_bool)
        trg = batch["trg_enc_SMI"]
        a.trg, a.trg_ld = dev(trg, torch.int64), trg.shape[1]
        a.step_seed = self.step_seed & 0xFFFFFFFF
        a.drop_layers = LAYER_DROPOUT if train else 0.0
        a.drop_embed = float(self.model.config.drop_out) if train else 0.0
        a.zero_grads, a.loss_grad_scale = int(zero_grads), float(loss_grad_scale)
        if adam is None:
            a.adam_step = 0
        else:
            a.adam_step, a.lr, (a.beta1, a.beta2), a.eps, a.weight_decay = adam
        a.grad_scale = float(grad_scale)
        failure = []

        def guarded(seg, user):   # an exception must not vanish inside the foreign-function callback
            try:
                on_segment(seg, user)
            except BaseException as e:  # noqa: BLE001
                if not failure:
                    failure.append(e)

        cb = L.SEGMENT_CALLBACK(guarded) if on_segment is not None else L.SEGMENT_CALLBACK()
        a.on_segment = cb
        main = torch.cuda.current_stream()
        ws = self._side_stream(main)
        pl.mask_is_bool = glob_bool
        pl.encode_serial = getattr(pl, "encode_serial", 0) + 1
        pl._step_keep = (keep, cb)     # inputs stay referenced until the next step has been enqueued
        self.arena.grads_zeroed = False
        L.check(L.lib().mmst_train_step(C.byref(sp), C.byref(a), C.c_void_p(main.cuda_stream),
                                        C.c_void_p(ws.cuda_stream)), "mmst_train_step")
        if failure:
            raise failure[0]   # e.g. a collective that failed in a segment callback: the step must not go on unreduced
        self.arena.grads_zeroed = adam is not None   # the fused optimizer clears what it has consumed
        if adam is not None:
            # the step refreshed the transposed weights and the fragment-major images on the second stream: whoever reads
            # them outside mmst_train_step (Engine.decode, Engine.backward) waits for this event first
            self.arena.transposed_ready = torch.cuda.Event()
            self.arena.transposed_ready.record(ws)
        return pl.cv.view(pl.loss, 1)

    def _side_stream(self, main):
        if getattr(self, "wstream", None) is None or self.wstream.device != main.device:
            self.wstream = torch.cuda.Stream(device=main.device)
        return self.wstream

    # ------------------------------------------------------------------
    def backward(self, pl, train, on_segment=None, skip_decoder=False):
        """Backward of decode() + encode() from pl.dlogits; accumulates into the flat gradient buffer.
        skip_decoder=True starts at the encoder memory instead: the caller has filled pl.dmem (the CLIP head).

        Two streams: the critical path (activation gradients) runs on the current stream; each layer's
        parameter gradients (one grouped split-K launch) run on a second stream as soon as that layer's
        data phase is done, filling the CUs the small critical-path launches leave idle.
        on_segment(name) is called when the launches that finish the gradient memory of a segment of
        Arena.reduce_segments() ('decoder', 'cross', 'modal_l5' ... 'modal_l0', 'embed') have been ENQUEUED;
        self.wstream then holds the last of them."""
        lib, ar, B, T, Lm = L.lib(), self.arena, pl.B, pl.T, pl.Lmem
        ar.grads_zeroed = False
        m = self.model
        nl, ndl = m.num_encoder_layers, m.num_decoder_layers
        p_drop = LAYER_DROPOUT if train else 0.0
        p_drop2 = float(m.config.drop_out) if train else 0.0
        M = B * T
        main = torch.cuda.current_stream()
        ws = self._side_stream(main)
        st, wst = C.c_void_p(main.cuda_stream), C.c_void_p(ws.cuda_stream)
        ws.wait_stream(main)   # gradients were zeroed / forward finished on the main stream
        if ar.transposed_ready is not None:   # FusedAdamW refreshes the transposed weights on the second stream
            main.wait_event(ar.transposed_ready)

        def weights_after_data():
            ev = torch.cuda.Event()
            ev.record(main)
            ws.wait_event(ev)

        def encoder_layer_bwd(n, p, g, a, w, heads, seed):
            L.check(lib.mmst_encoder_layer_bwd_streams(n, p, g, a, w, heads, C.c_float(p_drop), C.c_uint32(seed),
                                                       1 if self.interleave_wgrads else 0, st, wst),
                    "mmst_encoder_layer_bwd_streams")

        if not skip_decoder:
            if getattr(m.config, "use_real_data", False):
                self.real_data_backward(pl)
            # vocabulary projection
            last = pl.dec_acts[ndl - 1]
            sc = pl.dec_bwd[ndl - 1]
            ops.gemm([dict(P=_P(pl.dlogits), Q=ar.pb("fc_out.weight"), C=_P(sc["dz"]),
                           C2=_P(sc["dzd"]) if p_drop > 0 else None, Z=_P(last["z3"]), stats=_P(last["st3"]),
                           gamma=ar.p32(f"decoder.layers.{ndl - 1}.norm3.weight"),
                           dgamma=ar.g32(f"decoder.layers.{ndl - 1}.norm3.weight"),
                           dbeta=ar.g32(f"decoder.layers.{ndl - 1}.norm3.bias"),
                           ldp=VPAD, ldq=D, ldc=D, Pn=M, Qn=D, K=VPAD, drop_p=p_drop,
                           seed=(self.layer_seed(64 + ndl - 1) + 5) & 0xFFFFFFFF)], L.EPI_LNBWD, qt=True)
            with torch.cuda.stream(ws):
                ops.gemm([dict(P=_P(pl.dlogits), Q=_P(last["y"]), C=ar.g32("fc_out.weight"), dbias=ar.g32("fc_out.bias"),
                               ldp=VPAD, ldq=D, ldc=D, Pn=VPAD, Qn=D, K=M, ksplit=max(1, min(16, M // 256)))],
                         L.EPI_ATOMIC, pt=True, qt=True)
            for l in range(ndl):
                w = pl.dec_w[l]
                w.dzd3 = pl.dec_dzd[l][1] if p_drop > 0 else pl.dec_dzd[l][0]
                w.prev_seed = self.layer_seed(64 + l - 1) if l > 0 else 0
            if self.decoder_kv_on_wstream:
                arr = lambda xs: (C.c_void_p * ndl)(*[C.addressof(x) for x in xs])  # noqa: E731
                seeds = (C.c_uint32 * ndl)(*[self.layer_seed(64 + l) for l in range(ndl)])
                L.check(lib.mmst_decoder_stack_bwd_streams(ndl, arr(pl.dec_p), arr(pl.dec_g), arr(pl.dec_a), arr(pl.dec_w),
                                                           16, C.c_float(p_drop), seeds, st, wst),
                        "mmst_decoder_stack_bwd_streams")
            for l in reversed(range(ndl)):
                w = pl.dec_w[l]
                if self.decoder_kv_on_wstream:
                    break
                L.check(lib.mmst_decoder_layer_bwd_data(C.byref(pl.dec_p[l]), C.byref(pl.dec_g[l]), C.byref(pl.dec_a[l]),
                                                        C.byref(w), 16, C.c_float(p_drop),
                                                        C.c_uint32(self.layer_seed(64 + l)), st),
                        "mmst_decoder_layer_bwd_data")
                weights_after_data()
                L.check(lib.mmst_decoder_layer_bwd_weights(C.byref(pl.dec_g[l]), C.byref(pl.dec_a[l]), C.byref(w),
                                                           C.c_float(p_drop), wst), "mmst_decoder_layer_bwd_weights")
            # target-embedding gradients only feed parameters: off the critical path
            weights_after_data()
            L.check(lib.mmst_trg_embed_bwd(C.c_void_p(_P(pl.tok)), B, C.c_void_p(_P(pl.dtrg)),
                                           C.c_void_p(ar.g32("embed_trg.weight")), C.c_void_p(ar.g32("pe_trg.weight")),
                                           B, T, m.config.in_size, C.c_float(p_drop2), C.c_uint32(self.layer_seed(100)),
                                           wst), "mmst_trg_embed_bwd")
            if getattr(m.config, "use_real_data", False):
                # the gradient at mean(memory) joins the decoder's memory gradient before the cross encoder reads it
                L.check(lib.mmst_seq_mean_bwd_add(C.c_void_p(_P(pl.rd_dmean)), C.c_void_p(_P(pl.dmem)), B, Lm, st),
                        "mmst_seq_mean_bwd_add")
            if on_segment:
                on_segment("decoder")
        # cross encoder
        sc = pl.cross_bwd[nl - 1]
        lastc = pl.cross_acts[nl - 1]
        pk = f"encoder_cross.layers.{nl - 1}."
        L.check(lib.mmst_layernorm_bwd(B * Lm, C.c_void_p(_P(pl.dmem)), 0, 0, 0, C.c_void_p(_P(lastc["z2"])),
                                       C.c_void_p(_P(lastc["st2"])), C.c_void_p(ar.p32(pk + "norm2.weight")),
                                       C.c_void_p(_P(sc["dz"])), C.c_void_p(_P(sc["dzd"])),
                                       C.c_void_p(ar.g32(pk + "norm2.weight")), C.c_void_p(ar.g32(pk + "norm2.bias")),
                                       C.c_float(p_drop), C.c_uint32((self.layer_seed(32 + nl - 1) + 3) & 0xFFFFFFFF),
                                       st), "mmst_layernorm_bwd")
        for l in reversed(range(nl)):
            pl.cross_w[l][0].prev_seed = self.layer_seed(32 + l - 1) if l > 0 else 0
            pl.cross_w[l][0].dzd2 = pl.cross_dzd[l][1] if p_drop > 0 else pl.cross_dzd[l][0]
            encoder_layer_bwd(1, pl.cross_p[l], pl.cross_g[l], pl.cross_a[l], pl.cross_w[l], m.cross_heads,
                              self.layer_seed(32 + l))
        if on_segment:
            on_segment("cross")
        # modality encoders: top LayerNorm backward reads its slice of d(memory_in)
        groups = []
        for i, (name, S, off) in enumerate(pl.mods):
            acts, scm = pl.modal_acts[nl - 1][i], pl.modal_bwd[nl - 1][i]
            pk = f"encoder_{name}.layers.{nl - 1}."
            groups.append(dict(C=_P(scm["dz"]), C2=_P(scm["dzd"]) if p_drop > 0 else None,
                               R=_P(pl.dmem_in), ldr=D, rS=S, rL=Lm, roff=off, Z=_P(acts["z2"]), stats=_P(acts["st2"]),
                               gamma=ar.p32(pk + "norm2.weight"), dgamma=ar.g32(pk + "norm2.weight"),
                               dbeta=ar.g32(pk + "norm2.bias"), ldp=8, ldq=8, ldc=D, Pn=B * S, Qn=D, K=0,
                               drop_p=p_drop, seed=(self.layer_seed(nl - 1) + 8 * i + 3) & 0xFFFFFFFF))
        ops.gemm(groups, L.EPI_LNBWD)
        for l in reversed(range(nl)):
            for i in range(pl.n_groups):
                pl.modal_w[l][i].prev_seed = self.layer_seed(l - 1) if l > 0 else 0
                pl.modal_w[l][i].dzd2 = pl.modal_dzd[l][i][1] if p_drop > 0 else pl.modal_dzd[l][i][0]
            encoder_layer_bwd(pl.n_groups, pl.modal_p[l], pl.modal_g[l], pl.modal_a[l], pl.modal_w[l], 16,
                              self.layer_seed(l))
            if on_segment:
                # depth l is final: its norm2 gradients were written by the data pass of depth l+1 (or by the top
                # LayerNorm backward), everything else by the two launches above
                on_segment(f"modal_l{l}")
        # embeddings
        G_, s, keep = pl.embed_structs
        dx = (C.c_void_p * pl.n_groups)(*[_P(pl.dxseq[i]) for i in range(pl.n_groups)])
        L.check(lib.mmst_embed_bwd(pl.n_groups, G_, C.byref(s), dx, st), "mmst_embed_bwd")
        names = [n for n, _, _ in pl.mods]
        if "IR" in names:
            k = _EMB_KEYS["IR"]
            ops.gemm([dict(P=_P(pl.ir_dpre), Q=_P(pl.ir_bf16), C=ar.g32(k + ".weight"), dbias=ar.g32(k + ".bias"),
                           ldp=D, ldq=1000, ldc=1000, Pn=D, Qn=1000, K=B, ksplit=1)], L.EPI_ATOMIC, pt=True, qt=True)
        if on_segment:
            on_segment("embed")
        main.wait_stream(ws)   # everything after backward (optimizer, readers of .grad) sees all gradients
