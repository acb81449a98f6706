"""ctypes binding of libmmst.so (the HIP kernels behind a C ABI).

There is NO fallback: if the shared library is missing or a launch fails, the
calling op raises.  Build with `python -c "import __graft_entry__ as g; g.build()"`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmmst.so")

EPI_BIAS, EPI_DRELU, EPI_LN, EPI_LNBWD, EPI_ATOMIC = range(5)
GF_RELU, GF_OUT_F32 = 1, 2
MAX_GROUPS = 6


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
        ("ksplit", C.c_int), ("tile_begin", C.c_int),
        ("ptiles", C.c_int), ("qtiles", C.c_int),
    ]


class GemmBatch(C.Structure):
    _fields_ = [("n", C.c_int), ("p", GemmProb * MAX_GROUPS)]


class AttnProb(C.Structure):
    _fields_ = [
        ("Q", C.c_void_p), ("K", C.c_void_p), ("V", C.c_void_p),
        ("kbias", C.c_void_p), ("O", C.c_void_p), ("lse", C.c_void_p),
        ("dO", C.c_void_p), ("delta", C.c_void_p),
        ("dQ", C.c_void_p), ("dK", C.c_void_p), ("dV", C.c_void_p),
        ("ldq", C.c_int), ("ldk", C.c_int), ("ldv", C.c_int), ("ldo", C.c_int),
        ("lddo", C.c_int), ("lddq", C.c_int), ("lddk", C.c_int), ("lddv", C.c_int),
        ("B", C.c_int), ("H", C.c_int), ("Tq", C.c_int), ("Sk", C.c_int),
        ("seed", C.c_uint32), ("block_begin", C.c_int), ("blocks_per_bh", C.c_int),
    ]


class AttnBatch(C.Structure):
    _fields_ = [("n", C.c_int), ("hd", C.c_int), ("causal", C.c_int), ("drop_p", C.c_float),
                ("p", AttnProb * MAX_GROUPS)]


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
                _lib.mmst_sizeof_attn_batch() != C.sizeof(AttnBatch):
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
