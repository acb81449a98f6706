"""RDKit-free data path against the reference's own dataset code (fixtures of oracle/make_golden.py `data`:
MultimodalData.__getitem__ + collate_fn of the reference run on the first molecules of its CSV fixtures, with the
RDKit results supplied from a sidecar) and against hand-checked formulae / monoisotopic masses.

Parity status of mol_formula / exact_mol_weight: RDKit is not installed anywhere this build runs, so they are
pinned by the known-answer table below only (formula strings and masses from standard tables), not by RDKit."""
import argparse
import json
import os
import random

import numpy as np
import pytest
import torch

from multimodalspectraltransformer_amd import dataloaders as D
from multimodalspectraltransformer_amd import helper_functions as hf

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CSV = os.path.join(GOLDEN, "csv")


def cfg(mode, blank):
    return argparse.Namespace(
        training_mode=mode, padding_points_number=64, input_dim_1H=2, input_dim_13C=1, input_dim_HSQC=2,
        input_dim_COSY=2, input_dim_IR=1000, blank_percentage=blank, fingerprint_size=512, IR_data_folder="/nonexistent",
        csv_1H_path_SGNN=os.path.join(CSV, "1H.csv"), csv_13C_path_SGNN=os.path.join(CSV, "13C.csv"),
        csv_HSQC_path_SGNN=os.path.join(CSV, "HSQC.csv"), csv_COSY_path_SGNN=os.path.join(CSV, "COSY.csv"),
        ref_data_type="1H", data_size=1000, tr_te_split=0.9)


@pytest.mark.parametrize("tag,mode,blank,seed", [("full", "1H_13C_HSQC_COSY_MF_MW", 0.0, 1),
                                                 ("two", "1H_HSQC_MF_MW", 0.0, 2),
                                                 ("blank", "1H_13C_HSQC_COSY_IR_MF_MS_MW", 0.5, 3)])
def test_items_and_collate_match_reference(tag, mode, blank, seed):
    g = np.load(os.path.join(GOLDEN, "dataloader.npz"))
    stoi = json.load(open(os.path.join(GOLDEN, "stoi.json")))
    stoi_MF = json.load(open(os.path.join(GOLDEN, "stoi_MF.json")))
    c = cfg(mode, blank)
    ds = D.MultimodalData(c, stoi, stoi_MF, "all")
    assert ds.ids == [str(s) for s in g["ids"]]
    assert [D.mol_formula(ds.smiles[s]) for s in ds.ids] == [str(f) for f in g["formulas"]]
    random.seed(seed)
    items = [ds[i] for i in range(len(ds))]
    want_dt = json.loads(str(g[f"{tag}/item0_dtypes"]))
    for k, v in want_dt.items():
        assert str(items[0][k].dtype) == v, (k, items[0][k].dtype, v)
    batch = hf.collate_fn(items)
    keys = sorted(k.split("/")[1] for k in g.files if k.startswith(tag + "/") and k.count("/") == 1 and
                  not k.endswith("item0_dtypes"))
    assert sorted(batch) == keys
    for k in keys:
        want = g[f"{tag}/{k}"]
        assert str(batch[k].dtype) == str(g[f"{tag}/{k}/dtype"]), k
        assert tuple(batch[k].shape) == want.shape, (k, batch[k].shape, want.shape)
        np.testing.assert_allclose(batch[k].numpy(), want, rtol=1e-6, atol=1e-7, err_msg=k)


def test_ir_binning_matches_reference():
    g = np.load(os.path.join(GOLDEN, "dataloader.npz"))
    ir, mask = D.load_ir(os.path.join(CSV, "IR_example.csv"), 1000)
    assert ir.dtype == torch.float64 and mask.dtype == torch.int64
    np.testing.assert_allclose(ir.numpy(), g["ir"], rtol=1e-12, atol=0)
    assert np.array_equal(mask.numpy(), g["ir_mask"])


KNOWN = [
    ("Cn1cnc2c1c(=O)n(C)c(=O)n2C", "C8H10N4O2", 194.080376),     # caffeine
    ("CC(=O)Oc1ccccc1C(=O)O", "C9H8O4", 180.042259),              # aspirin
    ("c1ccncc1", "C5H5N", 79.042199),
    ("c1cc[nH]c1", "C4H5N", 67.042199),
    ("Cn1cccc1", "C5H7N", 81.057849),
    ("c1ccsc1", "C4H4S", 84.003371),
    ("c1ccoc1", "C4H4O", 68.026215),
    ("C[N+](C)(C)CC(=O)[O-]", "C5H11NO2", 117.078979),
    ("ClC(Cl)(Cl)Cl", "CCl4", 151.875411),
    ("O", "H2O", 18.010565),
    ("[NH4+]", "H4N+", 18.033826),
    ("CS(C)(=O)=O", "C2H6O2S", 94.008851),
    ("OP(=O)(O)O", "H3O4P", 97.976895),
    ("C[N+](=O)[O-]", "CH3NO2", 61.016378),
    ("FC(F)(F)c1ccccc1I", "C7H4F3I", 271.930982),
    ("C#CCCCCC(=O)N[C@@H]1CCc2nc(NC(=O)CN)sc2C1", "C16H22N4O2S", 334.146347),
    ("C/C=C/C", "C4H8", 56.0626),
    ("BrCCBr", "C2H4Br2", 185.867975),
    ("C1CC1.Cl", "C3H7Cl", 78.023628),
    ("[2H]C([2H])([2H])O", "CH4O", 35.045044),                    # isotopes fold into the element in the formula
    ("C%12CCCCC%12", "C6H12", 84.0939),
]


@pytest.mark.parametrize("smi,formula,mass", KNOWN)
def test_formula_and_exact_mass_known_answers(smi, formula, mass):
    assert D.mol_formula(smi) == formula
    assert abs(D.exact_mol_weight(smi) - mass) < 2e-4


def test_parse_shifts_equals_literal_eval():
    import ast
    for mod, dims in (("1H", 2), ("13C", 1), ("HSQC", 2), ("COSY", 2)):
        import csv
        for row in csv.reader(open(os.path.join(CSV, mod + ".csv"))):
            want = np.array(ast.literal_eval(row[1]), dtype=np.float64)
            got = D.parse_shifts(row[1], dims)
            assert got.shape == want.shape and np.array_equal(got, want)
    assert D.parse_shifts("[1e-3, -2.5E+1, .5]", 1).tolist() == [0.001, -25.0, 0.5]


def test_iter_batches_shards_like_distributed_sampler():
    stoi = json.load(open(os.path.join(GOLDEN, "stoi.json")))
    stoi_MF = json.load(open(os.path.join(GOLDEN, "stoi_MF.json")))
    ds = D.MultimodalData(cfg("1H_HSQC_COSY_MF_MW", 0.0), stoi, stoi_MF, "all")
    n = len(ds)
    seen = []
    for r in range(2):
        got = list(D.iter_batches(ds, 2, rank=r, world=2))
        assert all(b["mask_MF"].dtype == torch.float32 and b["src_MF"].dtype == torch.int64 for b in got)
        assert all(b["trg_enc_SMI"].shape[1] == 64 for b in got)
        seen.append(sum(b["src_1H"].shape[0] for b in got))
    assert seen == [(n + 1) // 2, (n + 1) // 2]
    tr = D.MultimodalData(cfg("1H", 0.0), stoi, stoi_MF, "train")
    te = D.MultimodalData(cfg("1H", 0.0), stoi, stoi_MF, "test")
    assert len(tr) == int(0.9 * n) and len(tr) + len(te) == n


def test_weight_sorted_batches_match_reference_sampler():
    """Fixture: the reference's WeightSortedBatchSampler (clip_functions_v15_4.py:117-152) run with random.seed(5)."""
    import random
    import numpy as np
    from multimodalspectraltransformer_amd.dataloaders import weight_sorted_batches
    from tests.util import load_golden
    g = load_golden("clip_c2_L1")
    mw = g["sampler_mw"].tolist()
    n = 0
    for key in g.files:
        if not key.startswith("sampler/"):
            continue
        bs, ws, dl = key[len("sampler/"):].split("_")
        random.seed(5)
        got = list(weight_sorted_batches(mw, int(bs), bool(int(dl)), None if ws == "None" else int(ws), rng=random))
        want = [[i for i in row if i >= 0] for row in g[key].tolist()]
        assert got == want, key
        n += 1
    assert n == 3
