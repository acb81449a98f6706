"""RDKit-free data path: CSV spectra -> padded tensors of the model's batch dictionary.

Counterpart of utils_MMT/dataloaders_pl_v15_4.py (`MultimodalData` :144-661, `collate_fn` :665-712).  The
reference parses every spectrum string with `ast.literal_eval` inside `__getitem__` and asks RDKit for the
molecular formula, the exact mass, the canonical SMILES and the Morgan fingerprint of every sample, every epoch.
Here the CSV files are parsed ONCE into numpy arrays, `__getitem__` only normalises / pads, and the three RDKit
quantities the model consumes are computed from the SMILES string by a small parser (`mol_formula`,
`exact_mol_weight`); nothing in this module imports RDKit.

What is NOT reproduced without RDKit (stated, not hidden):
* canonicalisation / randomisation of the SMILES: the string of the CSV is used as it is (the reference's generated
  files hold RDKit-canonical SMILES; `config.smi_randomizer` must be off);
* the Morgan fingerprint `trg_FP`: zeros unless a sidecar `{sample-id: 512 floats}` is supplied.  The reference's
  loss gives it weight 0 (`weight_FP`, config_V8.json), so training is unaffected.

File formats (dataloaders_pl_v15_4.py:168-262, 324-346): per modality a CSV with columns
`SMILES, shifts, sample-id` (with or without a header line, an optional unnamed index column is dropped);
`shifts` is a Python-literal list: 1H [(ppm, intensity), ...], 13C [ppm, ...], HSQC [[1H ppm, 13C ppm], ...],
COSY [(ppm, ppm), ...]; IR is one file `<folder>/<sample-id>.csv` with a `spectra` column.
"""
import csv
import os
import random
import re

import numpy as np
import torch

from . import helper_functions as hf

_NUM = re.compile(r'[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?')
_DIMS = {"1H": 2, "13C": 1, "HSQC": 2, "COSY": 2}
SPECTRA = ["1H", "13C", "HSQC", "COSY", "IR"]


# ------------------------------------------------------------------------------------------------------------
# parsing
# ------------------------------------------------------------------------------------------------------------
def parse_shifts(text, dims):
    """'[(1.2, 0.5), (3.4, 1.0)]' / '[17.3, 23.1]' / '[[1.6, 25.9], ...]' -> float64 array (n, dims) or (n,)."""
    vals = np.array(_NUM.findall(text), dtype=np.float64)
    if dims == 1:
        return vals
    if vals.size % dims:
        raise ValueError("shift list does not hold whole %d-tuples" % dims)
    return vals.reshape(-1, dims)


def read_spectrum_csv(path, modality):
    """-> ({sample-id: SMILES}, {sample-id: array}).  Header line and a leading index column are optional."""
    smiles, shifts = {}, {}
    with open(path, newline="") as f:
        for row in csv.reader(f):
            if not row:
                continue
            if len(row) == 4:            # pandas index column ('Unnamed: 0')
                row = row[1:]
            if row[0] == "SMILES" and row[-1] == "sample-id":
                continue
            smi, txt, sid = row[0], row[1], row[2]
            smiles[sid] = smi
            shifts[sid] = parse_shifts(txt, _DIMS[modality])
    return smiles, shifts


def normalise(modality, arr):
    """1H: ppm/10 (intensity kept); 13C: ppm/200; HSQC: (1H/10, 13C/200); COSY: (/10, /10)
    (dataloaders_pl_v15_4.py:459, 485, 348-366)."""
    a = np.array(arr, dtype=np.float64, copy=True)
    if modality == "1H":
        a[:, 0] /= 10.0
    elif modality == "13C":
        a /= 200.0
    elif modality == "HSQC":
        a[:, 0] /= 10.0
        a[:, 1] /= 200.0
    elif modality == "COSY":
        a /= 10.0
    return a


def load_ir(path, input_dim):
    """Bin a measured IR spectrum to `input_dim` points and scale by its maximum (dataloaders_pl_v15_4.py:324-346):
    bin i is the mean of samples round(i*span) .. round((i+1)*span), span = len/input_dim accumulated in floating
    point exactly as the reference's loop does; returns (float64 spectrum, int64 zero mask)."""
    with open(path, newline="") as f:
        rd = csv.reader(f)
        header = next(rd)
        col = header.index("spectra")
        spec = np.array([float(r[col]) for r in rd if r], dtype=np.float64)
    mx = spec.max()
    span = len(spec) / input_dim
    out = np.zeros(input_dim)
    start = 0
    for i in range(input_dim):
        end = start + span
        out[i] = np.mean(spec[round(start):round(end)]) / mx
        start = end
    return torch.tensor(out), torch.tensor(np.zeros(input_dim, dtype=int))


# ------------------------------------------------------------------------------------------------------------
# formula and exact mass from a SMILES string
# ------------------------------------------------------------------------------------------------------------
_ORGANIC = {"B": (3,), "C": (4,), "N": (3, 5), "O": (2,), "P": (3, 5), "S": (2, 4, 6), "F": (1,), "Cl": (1,),
            "Br": (1,), "I": (1,)}
_AROMATIC = {"b": "B", "c": "C", "n": "N", "o": "O", "p": "P", "s": "S"}
_BOND = {"-": 1, "=": 2, "#": 3, "$": 4, ":": 1, "/": 1, "\\": 1}
_BRACKET = re.compile(r'^(\d+)?([A-Z][a-z]?|[a-z][a-z]?)(@{0,2}(?:[A-Z]{2}\d+)?)?(H\d*)?([+-]+\d*|[+-]\d+)?(?::\d+)?$')

# monoisotopic masses (most abundant isotope), u
_MASS = {"H": 1.00782503223, "He": 4.00260325413, "Li": 7.0160034366, "Be": 9.012183065, "B": 11.00930536,
         "C": 12.0, "N": 14.00307400443, "O": 15.99491461957, "F": 18.99840316273, "Ne": 19.9924401762,
         "Na": 22.989769282, "Mg": 23.985041697, "Al": 26.98153853, "Si": 27.97692653465, "P": 30.97376199842,
         "S": 31.9720711744, "Cl": 34.968852682, "Ar": 39.9623831237, "K": 38.9637064864, "Ca": 39.962590863,
         "Fe": 55.93493633, "Cu": 62.92959772, "Zn": 63.92914201, "As": 74.92159457, "Se": 79.9165218,
         "Br": 78.9183376, "Sn": 119.90220163, "I": 126.9044719}
_ISOTOPE = {("H", 2): 2.01410177812, ("H", 3): 3.0160492779, ("C", 13): 13.00335483507, ("C", 14): 14.0032419884,
            ("N", 15): 15.00010889888, ("O", 17): 16.99913175650, ("O", 18): 17.99915961286,
            ("S", 34): 33.967867004, ("Cl", 37): 36.965902602, ("Br", 81): 80.9162897}


def _atoms_of(smiles):
    """Tokenise a SMILES string into atoms with (element, isotope, explicit H or None, charge, aromatic, bond-order
    sum of explicit neighbours).  Implicit hydrogens follow the organic-subset rule: lowest normal valence that
    fits the bond-order sum (aromatic atoms: lowest valence only, one valence used by the ring system)."""
    atoms = []          # dicts
    stack, prev = [], None
    ring = {}           # ring-closure label -> (atom index, bond order or None)
    pending = None      # explicit bond symbol before the next atom / ring closure
    i, n = 0, len(smiles)

    def add_bond(a, b, order):
        if order is None:
            order = 1
        atoms[a]["bo"] += order
        atoms[b]["bo"] += order

    while i < n:
        ch = smiles[i]
        if ch == "[":
            j = smiles.index("]", i)
            m = _BRACKET.match(smiles[i + 1:j])
            if not m:
                raise ValueError("unsupported bracket atom %r" % smiles[i:j + 1])
            iso, sym, _, hs, chg = m.groups()
            arom = sym[0].islower()
            el = sym.capitalize() if arom else sym
            h = 0 if hs is None else (1 if hs == "H" else int(hs[1:]))
            c = 0
            if chg:
                sign = 1 if chg[0] == "+" else -1
                digits = chg.lstrip("+-")
                c = sign * (int(digits) if digits else len(chg))
            atoms.append(dict(el=el, iso=int(iso) if iso else None, h=h, chg=c, arom=arom, bo=0))
            i = j + 1
        elif ch in "()":
            if ch == "(":
                stack.append(prev)
            else:
                prev = stack.pop()
            i += 1
            continue
        elif ch in _BOND:
            pending = _BOND[ch]
            i += 1
            continue
        elif ch == ".":
            prev, pending = None, None
            i += 1
            continue
        elif ch.isdigit() or ch == "%":
            if ch == "%":
                label, i = smiles[i + 1:i + 3], i + 3
            else:
                label, i = ch, i + 1
            if label in ring:
                other, order = ring.pop(label)
                add_bond(prev, other, pending if pending is not None else order)
            else:
                ring[label] = (prev, pending)
            pending = None
            continue
        else:
            two = smiles[i:i + 2]
            if two in ("Cl", "Br"):
                sym, i = two, i + 2
            else:
                sym, i = ch, i + 1
            if sym in _AROMATIC:
                atoms.append(dict(el=_AROMATIC[sym], iso=None, h=None, chg=0, arom=True, bo=0))
            elif sym in _ORGANIC:
                atoms.append(dict(el=sym, iso=None, h=None, chg=0, arom=False, bo=0))
            else:
                raise ValueError("unsupported SMILES symbol %r in %r" % (sym, smiles))
        cur = len(atoms) - 1
        if prev is not None:
            add_bond(prev, cur, pending)
        pending = None
        prev = cur
    for a in atoms:
        if a["h"] is None:
            val = _ORGANIC[a["el"]]
            if a["arom"]:
                a["h"] = max(0, val[0] - a["bo"] - 1)
            else:
                a["h"] = next((v - a["bo"] for v in val if v >= a["bo"]), 0)
    return atoms


def mol_formula(smiles):
    """Molecular formula in Hill order (C, H, then alphabetical; all alphabetical without carbon), net charge
    appended as '+', '-', '+2' ...: the string rdMolDescriptors.CalcMolFormula returns for ordinary organic
    molecules (dataloaders_pl_v15_4.py:439)."""
    count, charge = {}, 0
    for a in _atoms_of(smiles):
        count[a["el"]] = count.get(a["el"], 0) + 1
        if a["h"]:
            count["H"] = count.get("H", 0) + a["h"]
        charge += a["chg"]
    keys = sorted(count)
    if "C" in count:
        keys = ["C"] + (["H"] if "H" in count else []) + [k for k in keys if k not in ("C", "H")]
    s = "".join(k + (str(count[k]) if count[k] > 1 else "") for k in keys)
    if charge:
        s += ("+" if charge > 0 else "-") + (str(abs(charge)) if abs(charge) > 1 else "")
    return s


def exact_mol_weight(smiles):
    """Monoisotopic mass (rdMolDescriptors.CalcExactMolWt, dataloaders_pl_v15_4.py:443): sum of the most abundant
    isotope's mass of every atom, implicit hydrogens included; a bracket isotope label selects that isotope; the
    electron mass is subtracted per unit of net positive charge (no effect on the neutral molecules and zwitterions
    of the training sets)."""
    total = 0.0
    for a in _atoms_of(smiles):
        total -= a["chg"] * 0.000548579909
        if a["iso"] is not None:
            total += _ISOTOPE.get((a["el"], a["iso"]), float(a["iso"]))
        else:
            total += _MASS[a["el"]]
        total += a["h"] * _MASS["H"]
    return total


# ------------------------------------------------------------------------------------------------------------
# dataset
# ------------------------------------------------------------------------------------------------------------
def _empty(points, dimensions):
    """dataloaders_pl_v15_4.py:369-392: zeros and an all-ones (fully padded) mask."""
    shape = (points,) if dimensions == 1 else (points, dimensions)
    return torch.zeros(shape), torch.ones(points, dtype=torch.long)


class MultimodalData(torch.utils.data.Dataset):
    """`MultimodalData(config, stoi, stoi_MF, mode)` with the reference's item dictionary (keys, shapes, dtypes).

    config fields read: training_mode, csv_{1H,13C,HSQC,COSY}_path_SGNN, IR_data_folder, ref_data_type, data_size,
    tr_te_split, padding_points_number, input_dim_{1H,13C,HSQC,COSY,IR}, blank_percentage, fingerprint_size;
    mode: "train" / "test" (split by tr_te_split) / "all"; "val" reads config.csv_path_val for the sample list.
    The per-sample blanking draws `random.random()` once per present modality in the reference's order."""

    def __init__(self, config, stoi, stoi_MF, mode, fingerprints=None):
        self.config, self.stoi, self.stoi_MF, self.mode = config, stoi, stoi_MF, mode
        self.fingerprints = fingerprints
        self.fingerprint_size = config.fingerprint_size
        paths = {"1H": getattr(config, "csv_1H_path_SGNN", None), "13C": getattr(config, "csv_13C_path_SGNN", None),
                 "HSQC": getattr(config, "csv_HSQC_path_SGNN", None), "COSY": getattr(config, "csv_COSY_path_SGNN", None)}
        self.smiles, self.shifts = {}, {}
        order = None
        for mod, path in paths.items():
            # every file that exists is read (parsing is done once): the reference's pickle cache likewise holds all
            # modalities, and the HSQC_/COSY_ ablation copies are filled from it even outside training_mode
            if path and os.path.exists(path):
                smi, sh = read_spectrum_csv(path, mod)
                self.shifts[mod] = sh
                for k, v in smi.items():
                    self.smiles.setdefault(k, v)
                if mod == getattr(config, "ref_data_type", "1H"):
                    order = list(smi.keys())
        if mode == "val" and getattr(config, "csv_path_val", None):
            order = list(read_spectrum_csv(config.csv_path_val, "13C" if "13C" in config.csv_path_val else "1H")[0])
        if order is None:
            order = list(self.smiles.keys())
        order = order[:int(getattr(config, "data_size", len(order)))]
        if mode in ("train", "test"):
            cut = int(config.tr_te_split * len(order))
            order = order[:cut] if mode == "train" else order[cut:]
        self.ids = order
        # formula / mass are functions of the SMILES only: once per sample, not once per epoch
        self._mf, self._mw = {}, {}

    def __len__(self):
        return len(self.ids)

    def _spectrum(self, mod, sid):
        arr = self.shifts.get(mod, {}).get(sid)
        if arr is None:
            raise KeyError(sid)
        data = normalise(mod, arr)
        dims = getattr(self.config, "input_dim_" + mod, _DIMS[mod])
        return hf.zero_pad(data.tolist(), self.config.padding_points_number, dimensions=dims)

    def __getitem__(self, idx):
        cfg, item = self.config, {}
        sid = self.ids[idx]
        smi = self.smiles[sid]
        if sid not in self._mf:
            self._mf[sid], self._mw[sid] = mol_formula(smi), exact_mol_weight(smi)
        item["trg_MW"] = torch.tensor(self._mw[sid])
        pts = cfg.padding_points_number
        actual = [s for s in SPECTRA if s in cfg.training_mode]
        for mod in ("1H", "13C", "HSQC", "COSY"):
            dims = getattr(cfg, "input_dim_" + mod, _DIMS[mod]) if mod != "COSY" else getattr(cfg, "input_dim_HSQC", 2)
            if mod in actual:
                try:
                    src, mask = self._spectrum(mod, sid)
                    chance = cfg.blank_percentage
                except Exception:
                    chance = 1
                if random.random() < chance:
                    src, mask = _empty(pts, dims)
                    actual.remove(mod)
                item["src_" + mod], item["mask_" + mod] = src, mask
                if mod in ("HSQC", "COSY"):
                    item["src_" + mod + "_"], item["mask_" + mod + "_"] = src, mask
            else:
                src, mask = _empty(pts, dims)
                item["src_" + mod], item["mask_" + mod] = src, mask
                if mod in ("HSQC", "COSY"):       # kept for the error-analysis ablations (:520-535, :558-572)
                    try:
                        s2, m2 = self._spectrum(mod, sid)
                    except Exception:
                        s2, m2 = src, mask
                    item["src_" + mod + "_"], item["mask_" + mod + "_"] = s2, m2
        if "IR" in actual:
            try:
                src, mask = load_ir(os.path.join(cfg.IR_data_folder, f"{sid}.csv"), cfg.input_dim_IR)
                chance = cfg.blank_percentage
            except Exception:
                chance = 1
            if random.random() < chance:
                src, mask = _empty(cfg.input_dim_IR, 1)
                actual.remove("IR")
        else:
            src, mask = _empty(cfg.input_dim_IR, 1)
        item["src_IR"], item["mask_IR"] = src, mask
        if not actual:                               # everything blanked: fall back to 1H and 13C (:593-622)
            for mod in ("1H", "13C"):
                try:
                    item["src_" + mod], item["mask_" + mod] = self._spectrum(mod, sid)
                except Exception:
                    pass
        # target SMILES (also the MS input) and molecular formula, framed by the SMILES vocabulary's <SOS>/<EOS>
        enc = [self.stoi["<SOS>"]] + hf.encode_smiles(self.stoi, smi) + [self.stoi["<EOS>"]]
        pad = pts - len(enc)
        item["src_MS"] = torch.tensor(enc + [0] * pad)
        item["mask_MS"] = torch.tensor([0] * len(enc) + [1] * pad)
        item["trg_enc_SMI"] = torch.tensor(enc, dtype=torch.long)
        mf = [self.stoi["<SOS>"]] + hf.encode_MF(self.stoi_MF, self._mf[sid]) + [self.stoi["<EOS>"]]
        pad = pts - len(mf)
        item["src_MF"] = torch.tensor(mf + [0] * pad)
        item["mask_MF"] = torch.tensor([0] * len(mf) + [1] * pad)
        fp = None if self.fingerprints is None else self.fingerprints.get(sid)
        item["trg_FP"] = (torch.zeros(self.fingerprint_size) if fp is None
                          else torch.tensor(fp, dtype=torch.float32))
        return item


def iter_batches(dataset, batch_size, rank=0, world=1, shuffle=False, seed=0, mask_float=True):
    """Collated batches of this rank's shard (DistributedSampler semantics without dropping the tail).
    mask_float: hand `mask_MF` / `mask_MS` over as float32 (what the model needs under torch >= 2; the reference's
    collate leaves them int64, which torch 1.9.1 treated as additive)."""
    from .train import shard_indices
    idx = list(range(len(dataset)))
    if shuffle:
        random.Random(seed).shuffle(idx)
    mine = [idx[i] for i in shard_indices(len(idx), rank, world)]
    for i in range(0, len(mine), batch_size):
        batch = hf.collate_fn([dataset[j] for j in mine[i:i + batch_size]])
        if mask_float:
            for k in ("mask_MF", "mask_MS"):
                if k in batch:
                    batch[k] = batch[k].float()
        yield batch


def weight_sorted_batches(mol_weights, batch_size, drop_last=False, window_size=None, rng=None):
    """Batches of indices of similar molecular weight for contrastive training: WeightSortedBatchSampler of
    clip_functions_v15_4.py:117-152.  Indices are sorted by weight (stable, like DataFrame.sort_values), cut into
    windows of `window_size` (default max(batch_size, 4)) that are shuffled one by one with `rng.shuffle`
    (the reference uses the global `random` module: pass `random` to reproduce a seeded run of it), and a batch
    is emitted whenever exactly batch_size indices have been collected; leftovers form a last batch unless
    drop_last.  Like the reference, a window size that does not divide batch_size makes the collected count skip
    over batch_size, and then everything ends up in the final batch."""
    import random as _random
    rng = rng or _random
    window_size = window_size if window_size is not None else max(batch_size, 4)
    order = sorted(range(len(mol_weights)), key=lambda i: mol_weights[i])
    batch, window = [], []
    for idx in order:
        window.append(idx)
        if len(window) == window_size:
            rng.shuffle(window)
            batch.extend(window)
            window = []
            if len(batch) == batch_size:
                yield batch
                batch = []
    if window:
        rng.shuffle(window)
        batch.extend(window)
    if not drop_last and batch:
        yield batch
