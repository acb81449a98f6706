"""CPU-only checks: host-side logic of the package, the C-ABI surface of libmmst.so (symbols only, no
compute without a GPU) and the data-parallel gradient reducer over gloo with two processes."""
import argparse
import ctypes
import json
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import mmt_oracle as O
from tests.util import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "multimodalspectraltransformer_amd")


def test_library_exports_every_declared_symbol():
    """Every entry point declared in include/*.h is exported by libmmst.so, and the library exports nothing that
    the headers do not declare."""
    declared = set()
    for h in ("mmst.h", "mmst_lowlevel.h"):
        hdr = open(os.path.join(ROOT, "include", h)).read()
        declared |= set(re.findall(r"^\s*int\s+(mmst_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 48
    so = os.path.join(PKG, "libmmst.so")
    lib = ctypes.CDLL(so)
    for n in sorted(declared):
        assert hasattr(lib, n), f"libmmst.so does not export {n}"
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\sT\s+(mmst_\w+)", out))
    assert exported - declared == set(), f"exported but not declared in include/: {sorted(exported - declared)}"
    assert lib.mmst_version() >= 2


def test_bench_refuses_a_rank_count_that_differs_from_gpus():
    """`--gpus N` must never silently measure another number of ranks."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True,
                         text=True, timeout=120)
    assert out.returncode != 0 and "--gpus 4" in out.stderr and "{" not in out.stdout


def test_ctypes_structs_match_library():
    from multimodalspectraltransformer_amd import _lib as L
    lib = L.lib()   # raises on a layout mismatch
    assert lib.mmst_sizeof_gemm_batch() == ctypes.sizeof(L.GemmBatch)
    assert lib.mmst_sizeof_attn_batch() == ctypes.sizeof(L.AttnBatch)


def test_argument_validation_without_gpu():
    """Entry points reject bad arguments before touching the device."""
    lib = ctypes.CDLL(os.path.join(PKG, "libmmst.so"))
    assert lib.mmst_gemm(None, 0, 0, 0, None) != 0
    assert lib.mmst_attn_fwd(None, None) != 0
    assert lib.mmst_encoder_layer_fwd(0, None, None, 16, ctypes.c_float(0), 0, None) != 0
    assert lib.mmst_encoder_layer_fwd(1, None, None, 7, ctypes.c_float(0), 0, None) != 0
    assert lib.mmst_seq_mean(None, None, 1, 1, None) != 0
    assert lib.mmst_ce_loss(None, 64, 43, None, 64, 1, 63, None, None, ctypes.c_float(1), None) != 0


def test_tokenizer_decoder_and_padding_match_reference_fixture():
    from multimodalspectraltransformer_amd import helper_functions as hf
    t = json.load(open(os.path.join(GOLDEN, "text.json")))
    stoi = json.load(open(os.path.join(GOLDEN, "stoi.json")))
    itos = json.load(open(os.path.join(GOLDEN, "itos.json")))
    stoi_MF = json.load(open(os.path.join(GOLDEN, "stoi_MF.json")))
    for smi, toks, ids in zip(t["smiles"], t["tokens"], t["ids"]):
        assert hf.tokenize_smiles(smi) == toks
        assert hf.tokenize_smiles(smi, list(hf.two_char_symbols)) == toks
        assert hf.encode_smiles(stoi, smi) == ids
    for f, ids in zip(t["formulas"], t["mf_ids"]):
        assert hf.encode_MF(stoi_MF, f) == ids
    dec_in = torch.tensor(t["decode_input"])
    assert hf.tensor_to_smiles(dec_in, itos) == t["decoded"]
    assert hf.tensor_to_smiles(dec_in[:, 0], itos) == t["decoded"][0]
    probs = torch.rand(dec_in.shape)
    smi, cut = hf.tensor_to_smiles_and_prob_2(dec_in, probs, itos)
    eos_pos = [int((dec_in[:, i] == 2).nonzero()[0]) for i in range(dec_in.shape[1])]
    assert smi == t["decoded"] and [len(c) for c in cut] == eos_pos
    s1, c1 = hf.tensor_to_smiles_and_prob_2(dec_in[:, 0], probs[:, 0], itos)
    assert s1 == t["decoded"][0] and len(c1) == eos_pos[0]
    for key, (data, padded, mask) in t["pads"].items():
        dims = int(key[3])
        if dims == 1 and len(data) == 0:
            continue
        p, m = hf.zero_pad(data, 64, dims)
        np.testing.assert_allclose(p.numpy(), np.array(padded, np.float32), rtol=1e-6)
        assert m.tolist() == mask
    assert hf.tokenize_smiles("") == [] and hf.tensor_to_smiles(torch.zeros(0, 2, dtype=torch.long), itos) == ["", ""]


def test_collate_fn_dtypes_and_shapes():
    from multimodalspectraltransformer_amd import helper_functions as hf
    items = []
    for n in (3, 5):
        items.append({"src_1H": torch.rand(64, 2), "mask_1H": torch.ones(64).long(), "src_IR": torch.rand(1000).double(),
                      "mask_IR": torch.zeros(1000).long(), "trg_MW": torch.tensor(123.4).double(),
                      "trg_enc_SMI": torch.arange(n).long(), "trg_FP": torch.zeros(512),
                      "src_MF": torch.zeros(64).long(), "mask_MF": torch.ones(64).long(),
                      "src_MS": torch.zeros(64).long(), "mask_MS": torch.ones(64).long()})
    out = hf.collate_fn(items)
    assert out["src_1H"].dtype == torch.float32 and out["mask_1H"].dtype == torch.float32
    assert out["src_IR"].dtype == torch.float32 and out["trg_MW"].dtype == torch.float32
    assert out["trg_enc_SMI"].shape == (2, 64) and out["trg_enc_SMI"].dtype == torch.int64
    assert out["src_MF"].dtype == torch.int64 and out["mask_MF"].dtype == torch.int64


def _cpu_model(mode="1H_HSQC_COSY_IR_MF_MW", nl=2):
    from multimodalspectraltransformer_amd.models_MMT import MultimodalTransformer
    cfg = O.default_cfg(training_mode=mode, num_encoder_layers=nl, num_decoder_layers=nl)
    cfg.update(device="cpu")
    return MultimodalTransformer(argparse.Namespace(**cfg)), cfg


def test_state_dict_keys_and_shapes_match_reference():
    model, cfg = _cpu_model()
    shapes = O.param_shapes(cfg)
    sd = model.state_dict()
    assert list(sd.keys()) == list(shapes.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k
    full, _ = _cpu_model(nl=6)
    assert sum(p.numel() for p in full.parameters()) == 25566294


def test_checkpoint_layout_matches_reference_wrapper(tmp_path):
    """The wrapper's state_dict / .ckpt layout against the key lists of the reference's LightningModule (fixture
    ckpt_keys.json written by oracle/make_golden.py from the reference itself): 'model.'-prefixed keys in the same
    order, same shapes; a written .ckpt loads with weights_only=True and round-trips through load_from_checkpoint."""
    from multimodalspectraltransformer_amd.train import TransformerMultiGPU
    want = json.load(open(os.path.join(GOLDEN, "ckpt_keys.json")))
    cfg = O.default_cfg(num_encoder_layers=1, num_decoder_layers=1)
    cfg.update(device="cpu", training_setup="pretraining", lr_finetuning=5e-5, model_save_dir=str(tmp_path))
    torch.manual_seed(0)
    tm = TransformerMultiGPU(argparse.Namespace(**cfg))
    sd = tm.state_dict()
    assert list(sd.keys()) == want["wrapper_keys"]
    assert list(tm.model.state_dict().keys()) == want["model_keys"]
    assert {k: list(v.shape) for k, v in sd.items()} == want["shapes"]
    path = tm.save_checkpoint(str(tmp_path / "m.ckpt"))
    raw = torch.load(path, map_location="cpu", weights_only=True)
    assert list(raw["state_dict"].keys()) == want["wrapper_keys"] and raw["epoch"] == 0 and raw["global_step"] == 0
    tm2 = TransformerMultiGPU.load_from_checkpoint(path, config=argparse.Namespace(**cfg))
    for k, v in tm2.model.state_dict().items():
        assert torch.equal(v, sd["model." + k]), k
    pth = tm.on_epoch_end(3, 1.23456)
    assert os.path.basename(pth) == "MultimodalTransformer_Epoch_3_Loss1.235.pth"
    tm3 = TransformerMultiGPU.load_from_checkpoint(pth, config=argparse.Namespace(**cfg))
    assert all(torch.equal(v, sd["model." + k]) for k, v in tm3.model.state_dict().items())


def test_synthetic_workload_equals_oracle_generator():
    """bench.py draws its batches from the package (the oracle is test infrastructure); both generators must stay
    the same function of the seed so that the CPU baseline and the GPU run see the same workload."""
    from multimodalspectraltransformer_amd import synthetic as S
    for mode in ("1H_HSQC_COSY_IR_MF_MW", "1H_13C_HSQC_COSY_IR_MF_MS_MW"):
        a, b = O.synthetic_batch(5, 7, mode), S.synthetic_batch(5, 7, mode)
        assert sorted(a) == sorted(b) and all(torch.equal(a[k], b[k]) and a[k].dtype == b[k].dtype for k in a)
    assert O.default_cfg() == S.default_config()


def test_arena_layout_segments_and_active_ranges():
    from multimodalspectraltransformer_amd.engine import Arena
    model, cfg = _cpu_model()
    sd = O.make_state_dict(cfg, 3)
    model.load_state_dict(sd)
    ar = Arena(model, "cpu")
    # parameters became views of the flat buffer and kept their values
    for k, p in model.named_parameters():
        assert p.data_ptr() == ar.P.data_ptr() + 4 * ar.offset[k]
        assert torch.equal(p.detach(), sd[k])
    assert 0 < ar.marks["modal"] < ar.marks["cross"] < ar.marks["decoder"] < ar.total
    # every offset 16-byte aligned in the bf16 copy
    assert all(o % 8 == 0 for o in ar.offset.values())
    mode = "1H_HSQC_COSY_IR_MF_MW"
    act = set(ar.active_keys(mode))
    assert not any(k.startswith("encoder_13C") or k.startswith("fp1.") or k.startswith("real_data_linear.")
                   or k.startswith("linear_embedding_MS.") or k.startswith("linear_spec_embedding_13C") for k in act)
    assert "encoder_1H.layers.0.linear1.weight" in act and "linear_embedding_MW.linear_spec_embedding_MW.weight" in act
    covered = np.zeros(ar.total, bool)
    for lo, hi in ar.active_ranges(mode):
        assert not covered[lo:hi].any()
        covered[lo:hi] = True
    for k in ar.order:
        seg = covered[ar.offset[k]: ar.offset[k] + ar.numel[k]]
        assert seg.all() if k in act else not seg.any(), k
    # reduction segments tile the whole buffer, in the order backward completes them, one per encoder depth
    segs = ar.reduce_segments()
    names = list(segs)
    nl = model.num_encoder_layers
    assert names == ["decoder", "cross"] + [f"modal_l{l}" for l in reversed(range(nl))] + ["embed"]
    tiles = sorted(segs.values())
    assert tiles[0][0] == 0 and tiles[-1][1] == ar.total and all(a[1] == b[0] for a, b in zip(tiles, tiles[1:]))
    for l in range(nl):
        lo, hi = segs[f"modal_l{l}"]
        inside = [k for k in ar.order if lo <= ar.offset[k] < hi]
        assert inside and all(f".layers.{l}." in k and k.startswith("encoder_") and not k.startswith("encoder_cross")
                              for k in inside)
    # load_state_dict writes through the views
    sd2 = O.make_state_dict(cfg, 4)
    model.load_state_dict(sd2)
    assert torch.equal(ar.P[ar.offset["fc_out.weight"]: ar.offset["fc_out.weight"] + 43 * 128].view(43, 128),
                       sd2["fc_out.weight"])


def test_plan_sequence_layout_matches_reference_lengths():
    """Memory length per training_mode: 582 / 582 / 453 / 322 / 902 rows (blank blocks included)."""
    from multimodalspectraltransformer_amd.engine import _mode_tags
    want = {"1H_HSQC_COSY_IR_MF_MW": 582, "1H_13C_HSQC_COSY_IR_MF_MW": 582, "1H": 453, "1H_HSQC_COSY_IR": 322,
            "1H_13C_HSQC_COSY_IR_MF_MS_MW": 902}
    for mode, L_ in want.items():
        has_mf, has_ms, has_mw = "MF" in mode, "MS" in mode, "MW" in mode
        tail = 64 * has_mf + 64 * has_ms + has_mw
        fd, fdir = (193, 130) if has_ms else (129, 66)
        tot = 0
        for name in ("1H", "13C", "HSQC", "COSY", "IR"):
            if name in _mode_tags(mode):
                tot += (1 if name == "IR" else 64) + tail
            else:
                tot += 65 if name == "COSY" else (fdir if name == "IR" else fd)
        assert tot == L_, mode


def test_cpu_forward_raises_no_fallback():
    model, cfg = _cpu_model("1H", 1)
    b = O.synthetic_batch(2, 1, "1H")
    with pytest.raises(Exception) as e:
        model(b["src_1H"], b["mask_1H"], None, None, None, None, None, None, None, None, None, None, None, None,
              b["trg_MW"], None)
    assert "cuda" in str(e.value).lower() or "MI355X" in str(e.value)


def test_shard_indices_like_distributed_sampler():
    from multimodalspectraltransformer_amd.train import shard_indices
    from torch.utils.data.distributed import DistributedSampler
    for n, world in ((10, 2), (4000, 8), (7, 3), (1, 2)):
        for r in range(world):
            ds = DistributedSampler(list(range(n)), num_replicas=world, rank=r, shuffle=False, drop_last=False)
            assert shard_indices(n, r, world) == list(iter(ds))


def test_reduce_lr_on_plateau():
    from multimodalspectraltransformer_amd.train import ReduceLROnPlateau

    class Opt:
        lr = 1e-4
    o = Opt()
    s = ReduceLROnPlateau(o)
    ref_p = torch.nn.Parameter(torch.zeros(1))
    ref_o = torch.optim.SGD([ref_p], lr=1e-4)
    ref_s = torch.optim.lr_scheduler.ReduceLROnPlateau(ref_o, mode="min", factor=0.5, patience=2)
    for m in [1.0, 0.9, 0.95, 0.96, 0.97, 0.98, 0.5, 0.6, 0.6, 0.6, 0.6]:
        s.step(m)
        ref_s.step(m)
        assert abs(o.lr - ref_o.param_groups[0]["lr"]) < 1e-12


_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from multimodalspectraltransformer_amd.train import GradReducer, shard_indices
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
n = 1000
flat = torch.arange(n, dtype=torch.float32) * (rank + 1)
segs = {"decoder": (700, n), "cross": (400, 700), "modal": (0, 400)}
red = GradReducer(flat, segs, world)
for name in ("decoder", "cross", "modal"):          # order in which backward finishes the segments
    red.on_segment(name)
scale = red.finish()
want = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
assert torch.equal(flat, want), "all-reduce result wrong"
assert abs(scale - 1.0 / world) < 1e-12
# active ranges: only their intersection with a segment is reduced, the rest is left alone
flat2 = torch.arange(n, dtype=torch.float32) * (rank + 1)
red2 = GradReducer(flat2, segs, world, active=[(100, 450), (650, 900)])
for name in ("decoder", "cross", "modal"):
    red2.on_segment(name)
red2.finish()
i = torch.arange(n)
act = ((i >= 100) & (i < 450)) | ((i >= 650) & (i < 900))
assert torch.equal(flat2, torch.where(act, want, torch.arange(n, dtype=torch.float32) * (rank + 1)))
# the two pieces of the cross segment (400..450, 650..700) went out as ONE collective: three per step, 600 floats
assert red2.collectives == 3 and red2.bytes_reduced == 4 * 600, (red2.collectives, red2.bytes_reduced)
assert red2.exposed_wait_ms() >= 0.0
red2.begin_step()
assert red2.collectives == 0 and red2.bytes_reduced == 0
# every molecule is seen exactly once per epoch across ranks (up to wrap-around padding)
idx = shard_indices(37, rank, world)
gathered = [None] * world
dist.all_gather_object(gathered, idx)
allidx = sorted(i for g in gathered for i in g)
assert set(allidx) == set(range(37)) and len(allidx) == (37 + world - 1) // world * world
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_grad_reducer_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=120)
        assert p.returncode == 0, out.decode()
