"""ctypes binding of libmmst.so (the HIP kernels behind a C ABI).

There is NO fallback: if the shared library is missing or a launch fails, the
calling op raises.  Build with `python -c "import __graft_entry__ as g; g.build()"`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MMST_LIB") or os.path.join(_HERE, "libmmst.so")   # MMST_LIB: another build of the library

EPI_BIAS, EPI_DRELU, EPI_LN, EPI_LNBWD, EPI_ATOMIC = range(5)
GF_RELU, GF_OUT_F32 = 1, 2
MAX_GROUPS = 6          # sequence groups per layer call / attention launch
GEMM_MAX_GROUPS = 16    # problems per grouped GEMM launch


class GemmProb(C.Structure):
    _fields_ = [
        ("P", C.c_void_p), ("Q", C.c_void_p),
        ("C", C.c_void_p), ("bias", C.c_void_p), ("R", C.c_void_p),
        ("gamma", C.c_void_p), ("beta", C.c_void_p),
        ("Z", C.c_void_p), ("stats", C.c_void_p), ("C2", C.c_void_p),
        ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("dbias", C.c_void_p),
        ("ldp", C.c_int), ("ldq", C.c_int), ("ldc", C.c_int), ("ldr", C.c_int),
        ("Pn", C.c_int), ("Qn", C.c_int), ("K", C.c_int),
        ("flags", C.c_int),
        ("drop_p", C.c_float), ("seed", C.c_uint32),
        ("cS", C.c_int), ("cL", C.c_int), ("coff", C.c_int), ("rS", C.c_int), ("rL", C.c_int), ("roff", C.c_int),
        ("ksplit", C.c_int), ("tile_begin", C.c_int),
        ("ptiles", C.c_int), ("qtiles", C.c_int),
    ]


class GemmBatch(C.Structure):
    _fields_ = [("n", C.c_int), ("lds_pad", C.c_int), ("p", GemmProb * GEMM_MAX_GROUPS)]


class AttnProb(C.Structure):
    _fields_ = [
        ("Q", C.c_void_p), ("K", C.c_void_p), ("V", C.c_void_p),
        ("kbias", C.c_void_p), ("O", C.c_void_p), ("lse", C.c_void_p),
        ("dO", C.c_void_p), ("delta", C.c_void_p),
        ("dQ", C.c_void_p), ("dK", C.c_void_p), ("dV", C.c_void_p),
        ("ldq", C.c_int), ("ldk", C.c_int), ("ldv", C.c_int), ("ldo", C.c_int),
        ("lddo", C.c_int), ("lddq", C.c_int), ("lddk", C.c_int), ("lddv", C.c_int),
        ("B", C.c_int), ("H", C.c_int), ("Tq", C.c_int), ("Sk", C.c_int),
        ("seed", C.c_uint32), ("block_begin", C.c_int), ("blocks_per_bh", C.c_int), ("kv_rows", C.c_int), ("tail_o", C.c_int), ("tail_s", C.c_int),
    ]


class AttnBatch(C.Structure):
    _fields_ = [("n", C.c_int), ("hd", C.c_int), ("causal", C.c_int), ("drop_p", C.c_float),
                ("p", AttnProb * MAX_GROUPS)]


class FfnProb(C.Structure):
    _fields_ = [("epi", GemmProb), ("X", C.c_void_p), ("W1", C.c_void_p), ("W2", C.c_void_p), ("b1", C.c_void_p),
                ("H", C.c_void_p), ("dH", C.c_void_p), ("ws", C.c_void_p), ("Wn", C.c_void_p), ("Cn", C.c_void_p),
                ("Hs", C.c_void_p), ("drop_p", C.c_float), ("seed", C.c_uint32),
                ("tile_begin", C.c_int), ("hsplit", C.c_int), ("A", C.c_void_p), ("Wp", C.c_void_p), ("pre", GemmProb),
                ("Wq", C.c_void_p), ("bq", C.c_void_p), ("Cq", C.c_void_p), ("nq", C.c_int)]


class FfnBatch(C.Structure):
    _fields_ = [("n", C.c_int), ("p", FfnProb * MAX_GROUPS)]


class FfnWgradProb(C.Structure):
    _fields_ = _P8 = [(n, C.c_void_p) for n in ("dZ", "H", "X", "W2T", "dW1", "db1", "dW2", "db2")] + [
        ("M", C.c_int), ("drop_p", C.c_float), ("tile_begin", C.c_int), ("rs", C.c_int), ("skew", C.c_float)]


class FfnWgradBatch(C.Structure):
    _fields_ = [("n", C.c_int), ("p", FfnWgradProb * MAX_GROUPS)]


_lib = None


class MMSTError(RuntimeError):
    pass


def lib():
    """Load libmmst.so once; raise loudly if it is absent (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MMSTError(f"{LIB_PATH} not built: run __graft_entry__.build() (hipcc, gfx950) first")
        _lib = C.CDLL(LIB_PATH)
        if _lib.mmst_sizeof_gemm_batch() != C.sizeof(GemmBatch) or \
                _lib.mmst_sizeof_attn_batch() != C.sizeof(AttnBatch) or \
                _lib.mmst_sizeof_ffn_batch() != C.sizeof(FfnBatch) or \
                _lib.mmst_sizeof_ffn_wgrad_batch() != C.sizeof(FfnWgradBatch) or \
                _lib.mmst_sizeof_dec_stack_step() != C.sizeof(DecStackStep) or \
                _lib.mmst_sizeof_step_plan() != C.sizeof(StepPlan) or \
                _lib.mmst_sizeof_step_args() != C.sizeof(StepArgs):
            raise MMSTError("struct layout mismatch between _lib.py and libmmst.so")
    return _lib


def check(rc, what):
    if rc != 0:
        raise MMSTError(f"{what} failed with code {rc}")


def ptr(t):
    return None if t is None else t.data_ptr()


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# ---------------------------------------------------------------------------------------------
# layer-level structs of include/mmst.h
# ---------------------------------------------------------------------------------------------
def _ptrs(*names):
    return [(n, C.c_void_p) for n in names]


class EncParams(C.Structure):
    _fields_ = _ptrs("w_qkv", "b_qkv", "w_o", "b_o", "w_1", "b_1", "w_2", "b_2", "g1", "be1", "g2", "be2",
                     "w_1t", "w_2t")


class EncGrads(C.Structure):
    _fields_ = _ptrs("w_qkv", "b_qkv", "w_o", "b_o", "w_1", "b_1", "w_2", "b_2", "g1", "be1", "g2", "be2")


class EncActs(C.Structure):
    _fields_ = _ptrs("x", "y", "kbias", "qkv", "attn", "lse", "z1", "st1", "x1", "h", "z2", "st2") + [
        ("B", C.c_int), ("S", C.c_int), ("y_L", C.c_int), ("y_off", C.c_int), ("hsign", C.c_void_p)]


class EncBwd(C.Structure):
    _fields_ = _ptrs("dz2", "dzd2", "dh", "dz1", "dzd1", "da", "dqkv", "delta", "dx", "dxd",
                     "prev_z2", "prev_st2", "prev_g2", "prev_dg2", "prev_dbe2") + [("prev_seed", C.c_uint32)]


_DEC_P = ("w_qkv", "b_qkv", "w_o", "b_o", "w_cqkv", "b_cqkv", "w_co", "b_co", "w_1", "b_1", "w_2", "b_2",
          "g1", "be1", "g2", "be2", "g3", "be3")


_DEC_W = ("qkv", "o", "cqkv", "co", "1", "2")


class DecParams(C.Structure):
    _fields_ = _ptrs(*(_DEC_P + ("w_1t", "w_2t") + tuple("f_" + w for w in _DEC_W) + tuple("ft_" + w for w in _DEC_W)))


class DecGrads(C.Structure):
    _fields_ = _ptrs(*_DEC_P)


class DecActs(C.Structure):
    _fields_ = _ptrs("x", "y", "mem", "mem_bias", "qkv", "attn1", "lse1", "z1", "st1", "x1", "qc", "kvc", "attn2",
                     "lse2", "z2", "st2", "x2", "h", "z3", "st3") + [
        ("B", C.c_int), ("T", C.c_int), ("L", C.c_int), ("kv_ready", C.c_int), ("ffn_ws", C.c_void_p), ("no_causal", C.c_int)]


class DecStep(C.Structure):
    _fields_ = _ptrs("x", "y", "qkv_cache", "kvc", "mem_bias", "attn", "x1", "qc", "x2", "h", "lse", "ffn_ws") + [
        ("B", C.c_int), ("pos", C.c_int), ("cap", C.c_int), ("L", C.c_int)]


MAX_DEC_LAYERS = 8


class DecStackStep(C.Structure):
    _fields_ = _ptrs("x", "tok", "embed", "pos_row", "y") + [
        ("qkv_cache", C.c_void_p * MAX_DEC_LAYERS), ("kvc", C.c_void_p * MAX_DEC_LAYERS)] + _ptrs(
        "mem_bias", "w_out", "b_out", "logits") + [("ld_logits", C.c_int), ("n_out", C.c_int)] + _ptrs(
        "next_tok", "next_prob") + [("temperature", C.c_float), ("V", C.c_int),
                                    ("B", C.c_int), ("pos", C.c_int), ("cap", C.c_int), ("L", C.c_int),
                                    ("kv_group", C.c_int), ("rows_per_group", C.c_int), ("hist", C.c_void_p),
                                    ("ffn_ws", C.c_void_p), ("sample", C.c_int), ("sample_seed", C.c_uint32)]


class DecBwd(C.Structure):
    _fields_ = _ptrs("dz3", "dzd3", "dh", "dz2", "dzd2", "da", "dqc", "dkvc", "dz1", "dzd1", "dqkv", "delta",
                     "da2", "dmem") + [("dmem_accumulate", C.c_int)] + _ptrs(
        "dx", "dxd", "prev_z3", "prev_st3", "prev_g3", "prev_dg3", "prev_dbe3") + [("prev_seed", C.c_uint32)] + _ptrs("delta2")


class EmbedGroup(C.Structure):
    _fields_ = _ptrs("xseq", "kbias", "kbias_glob", "src", "mask", "w", "b", "dw", "db", "ir_dpre") + [
        ("in_dim", C.c_int), ("n_spec", C.c_int), ("S", C.c_int), ("L_glob", C.c_int), ("off_glob", C.c_int),
        ("mod_bool", C.c_int)]


class EmbedShared(C.Structure):
    _fields_ = (_ptrs("mf_tok", "mf_mask", "mf_table", "d_mf_table") + [("has_mf", C.c_int), ("mf_bool", C.c_int)] +
                _ptrs("ms_tok", "ms_mask", "ms_table", "d_ms_table") + [("has_ms", C.c_int), ("ms_bool", C.c_int)] +
                _ptrs("mw", "mw_w", "mw_b", "d_mw_w", "d_mw_b") + [("has_mw", C.c_int), ("B", C.c_int),
                                                                    ("glob_bool", C.c_int)])


# ---------------------------------------------------------------------------------------------
# whole training step (mmst_train_step)
# ---------------------------------------------------------------------------------------------
MAX_LAYERS, MAX_RANGES, MAX_FFN, MAX_FRAG = 8, 16, 128, 96


class FragDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("ld", C.c_int),
                ("transpose", C.c_int)]
SEG_DECODER, SEG_CROSS, SEG_EMBED, SEG_MODAL0 = 0, 1, 2, 16


def _layer_ptrs(*names):
    return [(n, C.c_void_p * MAX_LAYERS) for n in names]


class StepPlan(C.Structure):
    _fields_ = ([(n, C.c_int) for n in ("B", "T", "L", "n_groups", "n_enc_layers", "n_dec_layers", "cross_heads",
                                        "vocab", "in_size", "fingerprint_size")] +
                [("embed_g", EmbedGroup * MAX_GROUPS), ("embed_s", EmbedShared), ("ir_group", C.c_int)] +
                _ptrs("ir_bf16", "ir_w", "ir_b", "ir_dw", "ir_db") + [("dxseq", C.c_void_p * MAX_GROUPS)] +
                _layer_ptrs("modal_p", "modal_g", "modal_a", "modal_w", "cross_p", "cross_g", "cross_a", "cross_w",
                            "dec_p", "dec_g", "dec_a", "dec_w") +
                [("modal_dzd", (C.c_void_p * MAX_GROUPS) * MAX_LAYERS)] + _layer_ptrs("cross_dzd", "dec_dzd") +
                _ptrs("memory", "mean", "fp1_w", "fp1_b", "fingerprint", "trg_emb", "embed_trg", "pe_trg", "d_embed_trg",
                      "d_pe_trg", "fc_out_w", "fc_out_b", "d_fc_out_w", "d_fc_out_b", "logits", "dlogits", "loss",
                      "dmem", "dmem_in", "dtrg", "P", "G", "M", "V", "Pb", "PbT") +
                [("n_ranges", C.c_int), ("range_lo", C.c_long * MAX_RANGES), ("range_hi", C.c_long * MAX_RANGES),
                 ("n_ffn", C.c_int), ("ffn1_off", C.c_uint32 * MAX_FFN), ("ffn2_off", C.c_uint32 * MAX_FFN),
                 ("n_frag", C.c_int), ("frag", FragDesc * MAX_FRAG),
                 ("seg_decoder", C.c_long), ("seg_cross", C.c_long), ("seg_modal", C.c_long * (MAX_LAYERS + 1))])


SEGMENT_CALLBACK = C.CFUNCTYPE(None, C.c_int, C.c_void_p)


class StepArgs(C.Structure):
    _fields_ = ([("src", C.c_void_p * MAX_GROUPS), ("mask", C.c_void_p * MAX_GROUPS)] +
                _ptrs("src_ir", "mf_tok", "mf_mask", "ms_tok", "ms_mask", "mw") +
                [(n, C.c_int) for n in ("mf_bool", "ms_bool", "mod_bool", "glob_bool")] +
                [("trg", C.c_void_p), ("trg_ld", C.c_int), ("step_seed", C.c_uint32), ("drop_layers", C.c_float),
                 ("drop_embed", C.c_float), ("zero_grads", C.c_int), ("loss_grad_scale", C.c_float),
                 ("adam_step", C.c_int)] +
                [(n, C.c_float) for n in ("lr", "beta1", "beta2", "eps", "weight_decay", "grad_scale")] +
                [("on_segment", SEGMENT_CALLBACK), ("user", C.c_void_p)])


class BeamState(C.Structure):
    _fields_ = _ptrs("seqs", "probs", "scores", "lens", "nbeams", "hist")


MAX_BEAM = 4
