"""Thin Python wrappers that fill the C structs of libmmst.so from torch tensors.

Every function enqueues HIP kernels on torch's current stream and returns
immediately; nothing here computes on the host.
"""
import ctypes as C

import torch

from . import _lib as L

_PTR_FIELDS = ("P", "Q", "C", "bias", "R", "gamma", "beta", "Z", "stats", "C2", "dgamma", "dbeta", "dbias")


def gemm_batch(groups):
    """groups: list of dicts with GemmProb field names; tensors for pointer fields."""
    if not 1 <= len(groups) <= L.GEMM_MAX_GROUPS:
        raise L.MMSTError("gemm: 1..%d groups" % L.GEMM_MAX_GROUPS)
    b = L.GemmBatch()
    b.n = len(groups)
    for i, g in enumerate(groups):
        p = b.p[i]
        for k, v in g.items():
            if k in _PTR_FIELDS:
                setattr(p, k, L.ptr(v) if isinstance(v, torch.Tensor) else v)
            else:
                setattr(p, k, v)
    return b


def gemm_launch(batch, epi, pt=False, qt=False):
    L.check(L.lib().mmst_gemm(C.byref(batch), epi, int(pt), int(qt), L.stream_ptr()), "mmst_gemm")


def gemm(groups, epi, pt=False, qt=False):
    gemm_launch(gemm_batch(groups), epi, pt, qt)


_APTR = ("Q", "K", "V", "kbias", "O", "lse", "dO", "delta", "dQ", "dK", "dV")


def attn_batch(groups, hd, causal=False, drop_p=0.0):
    if not 1 <= len(groups) <= L.MAX_GROUPS:
        raise L.MMSTError("attn: 1..%d groups" % L.MAX_GROUPS)
    b = L.AttnBatch()
    b.n = len(groups)
    b.hd = hd
    b.causal = int(causal)
    b.drop_p = float(drop_p)
    for i, g in enumerate(groups):
        p = b.p[i]
        for k, v in g.items():
            if k in _APTR:
                setattr(p, k, v if (v is None or isinstance(v, int)) else L.ptr(v))
            else:
                setattr(p, k, v)
    return b


def attn_fwd_launch(batch):
    L.check(L.lib().mmst_attn_fwd(C.byref(batch), L.stream_ptr()), "mmst_attn_fwd")


def attn_bwd_launch(batch):
    L.check(L.lib().mmst_attn_bwd(C.byref(batch), L.stream_ptr()), "mmst_attn_bwd")


_FPTR = ("X", "W1", "W2", "b1", "H", "dH", "ws", "Wn", "Cn", "Hs", "A", "Wp", "Wq", "bq", "Cq")


def ffn_batch(groups):
    """groups: list of dicts; key "epi" holds a dict of GemmProb fields for the final row epilogue."""
    if not 1 <= len(groups) <= L.MAX_GROUPS:
        raise L.MMSTError("ffn: 1..%d groups" % L.MAX_GROUPS)
    b = L.FfnBatch()
    b.n = len(groups)
    for i, g in enumerate(groups):
        p = b.p[i]
        for k, v in g.items():
            if k in ("epi", "pre"):
                sub = getattr(p, k)
                for ek, ev in v.items():
                    setattr(sub, ek, (L.ptr(ev) if isinstance(ev, torch.Tensor) else ev) if ek in _PTR_FIELDS else ev)
            elif k in _FPTR:
                setattr(p, k, L.ptr(v) if isinstance(v, torch.Tensor) else v)
            else:
                setattr(p, k, v)
    return b


def ffn_launch(batch, backward=False):
    """Returns 1 when a backward launch also wrote the chained Cn = E Wn of every problem, else 0."""
    rc = L.lib().mmst_ffn(C.byref(batch), int(backward), L.stream_ptr())
    if rc < 0:
        L.check(rc, "mmst_ffn")
    return rc


def ffn_wgrad(groups):
    """groups: list of dicts with FfnWgradProb fields (tensors for pointers)."""
    b = L.FfnWgradBatch()
    b.n = len(groups)
    for i, g in enumerate(groups):
        for k, v in g.items():
            setattr(b.p[i], k, L.ptr(v) if isinstance(v, torch.Tensor) else v)
    L.check(L.lib().mmst_ffn_wgrad(C.byref(b), L.stream_ptr()), "mmst_ffn_wgrad")


def eptr(t, offset_elems):
    """Device address of element `offset_elems` of tensor t."""
    return t.data_ptr() + offset_elems * t.element_size()
