"""mmst_decoder_layer_step_static -- the token step whose scratch rows live in a static device array inside the
library (for callers that cannot allocate) -- through the C ABI against mmst_decoder_layer_step with caller-provided
scratch: same kernels on the same inputs, so the output row and the appended Q|K|V cache row must be equal bit for
bit.  Two calls in a row share the array, the second with fewer sequences; sizes beyond the array are refused."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

D, FF, NH = 128, 2048, 16

if torch.cuda.is_available():
    from multimodalspectraltransformer_amd import _lib as L
    from tests.test_decode_gpu import _layer_params, _params_array


def _both(layer, B, pos, cap, Lm, gen, masked):
    dev = "cuda"
    r = lambda *s: torch.randn(*s, generator=gen, device=dev)  # noqa: E731
    x = (r(B, D) * 0.8).bfloat16()
    cache0 = (r(B, cap, 3 * D) * 0.7).bfloat16()
    kvc = (r(B, Lm, 2 * D) * 0.7).bfloat16()
    bias = None
    if masked:
        bias = torch.zeros(B, Lm, device=dev)
        bias[:, Lm // 3:Lm // 3 + 2] = float("-inf")
        bias[:, -1] = 1.0
    P = _params_array([layer])
    lib = L.lib()
    # (a) the existing entry point, scratch from the caller
    bf = dict(dtype=torch.bfloat16, device=dev)
    sc = dict(attn=torch.empty(B, D, **bf), x1=torch.empty(B, D, **bf), qc=torch.empty(B, D, **bf),
              x2=torch.empty(B, D, **bf), h=torch.empty(B, FF, **bf), lse=torch.empty(B, 16, device=dev),
              ffn_ws=torch.empty(8, B, D, device=dev))
    ya, cache_a = torch.zeros(B, D, **bf), cache0.clone()
    s = L.DecStep()
    s.x, s.y, s.qkv_cache, s.kvc = x.data_ptr(), ya.data_ptr(), cache_a.data_ptr(), kvc.data_ptr()
    s.mem_bias = bias.data_ptr() if bias is not None else None
    for k, v in sc.items():
        setattr(s, k, v.data_ptr())
    s.B, s.pos, s.cap, s.L = B, pos, cap, Lm
    L.check(lib.mmst_decoder_layer_step(C.byref(P[0]), C.byref(s), NH, L.stream_ptr()), "mmst_decoder_layer_step")
    # (b) the static-scratch entry point
    yb, cache_b = torch.zeros(B, D, **bf), cache0.clone()
    fn = lib.mmst_decoder_layer_step_static
    fn.argtypes = [C.c_void_p] * 6 + [C.c_int] * 5 + [C.c_void_p]
    rc = fn(C.cast(C.byref(P[0]), C.c_void_p), x.data_ptr(), yb.data_ptr(), cache_b.data_ptr(), kvc.data_ptr(),
            bias.data_ptr() if bias is not None else None, B, pos, cap, Lm, NH, L.stream_ptr())
    L.check(rc, "mmst_decoder_layer_step_static")
    torch.cuda.synchronize()
    return ya, yb, cache_a, cache_b, cache0


def test_static_scratch_token_step_equals_the_allocating_one():
    gen = torch.Generator(device="cuda").manual_seed(77)
    layer = _layer_params(gen, "cuda")
    # two calls in a row on the same array: 200 sequences, then 5 (leftovers of the first call beyond row 5)
    for B, pos, cap, Lm, masked in ((200, 9, 24, 582, True), (5, 3, 16, 70, False), (256, 0, 8, 130, True)):
        ya, yb, ca, cb, c0 = _both(layer, B, pos, cap, Lm, gen, masked)
        assert float(ya.float().abs().max()) > 0.1 and bool(torch.isfinite(ya.float()).all())
        assert torch.equal(ya, yb), (B, float((ya.float() - yb.float()).abs().max()))
        assert torch.equal(ca, cb)
        assert not torch.equal(ca[:, pos], c0[:, pos])           # the new Q|K|V row was written ...
        keep = [i for i in range(cap) if i != pos]
        assert torch.equal(cb[:, keep], c0[:, keep])             # ... and nothing else
    # beyond the array's capacity: refused, nothing launched
    lib = L.lib()
    fn = lib.mmst_decoder_layer_step_static
    P = _params_array([layer])
    x = torch.zeros(257, D, dtype=torch.bfloat16, device="cuda")
    y = torch.full((257, D), 3.0, dtype=torch.bfloat16, device="cuda")
    cache = torch.zeros(257, 4, 3 * D, dtype=torch.bfloat16, device="cuda")
    kvc = torch.zeros(257, 8, 2 * D, dtype=torch.bfloat16, device="cuda")
    rc = fn(C.cast(C.byref(P[0]), C.c_void_p), x.data_ptr(), y.data_ptr(), cache.data_ptr(), kvc.data_ptr(), None,
            257, 0, 4, 8, NH, L.stream_ptr())
    torch.cuda.synchronize()
    assert rc != 0 and float(y.float().min()) == 3.0
    assert fn(C.cast(C.byref(P[0]), C.c_void_p), x.data_ptr(), y.data_ptr(), cache.data_ptr(), kvc.data_ptr(), None,
              0, 0, 4, 8, NH, L.stream_ptr()) != 0


def test_static_scratch_token_step_under_graph_capture():
    """The use the entry point exists for: after ONE eager call on the device (which resolves the address of the static
    array; that query must not happen while a stream is capturing) the call is captured into a HIP graph and
    replayed; the replay writes the same bits as the eager call."""
    gen = torch.Generator(device="cuda").manual_seed(78)
    layer = _layer_params(gen, "cuda")
    B, pos, cap, Lm = 48, 5, 16, 130
    ya, yb, ca, cb, c0 = _both(layer, B, pos, cap, Lm, gen, True)      # eager warm-up (and the reference result)
    assert torch.equal(ya, yb)
    dev = "cuda"
    g2 = torch.Generator(device="cuda").manual_seed(78)
    layer2 = _layer_params(g2, dev)                                       # same parameters, same inputs as _both drew
    r = lambda *s: torch.randn(*s, generator=g2, device=dev)  # noqa: E731
    x = (r(B, D) * 0.8).bfloat16()
    cache = (r(B, cap, 3 * D) * 0.7).bfloat16()
    kvc = (r(B, Lm, 2 * D) * 0.7).bfloat16()
    bias = torch.zeros(B, Lm, device=dev)
    bias[:, Lm // 3:Lm // 3 + 2] = float("-inf")
    bias[:, -1] = 1.0
    y = torch.zeros(B, D, dtype=torch.bfloat16, device=dev)
    P = _params_array([layer2])
    fn = L.lib().mmst_decoder_layer_step_static
    fn.argtypes = [C.c_void_p] * 6 + [C.c_int] * 5 + [C.c_void_p]
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        rc = fn(C.cast(C.byref(P[0]), C.c_void_p), x.data_ptr(), y.data_ptr(), cache.data_ptr(), kvc.data_ptr(),
                bias.data_ptr(), B, pos, cap, Lm, NH, L.stream_ptr())
    assert rc == 0
    for _ in range(2):
        y.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, ya), float((y.float() - ya.float()).abs().max())
        assert torch.equal(cache, ca)
