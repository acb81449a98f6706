"""SMILES / molecular-formula tokenisation and token -> SMILES decoding.

Host-side string work of the path (integer/byte arithmetic only), behaviour-identical to
utils_MMT/helper_functions_pl_v15_4.py:177-209, 239-269, 390-419 and
utils_MMT/dataloaders_pl_v15_4.py:267-299, 395-421, 665-712.
"""
import re

import torch
import torch.nn.functional as F

two_char_symbols = [
    'He', 'Li', 'Be', 'Ne', 'Na', 'Mg', 'Al', 'Si', 'Cl', 'Ar', 'Ca', 'Ti', 'Cr', 'Mn', 'Fe', 'Ni', 'Cu', 'Zn',
    'Ga', 'Ge', 'As', 'Se', 'Br', 'Kr', 'Rb', 'Sr', 'Zr', 'Nb', 'Mo', 'Tc', 'Ru', 'Rh', 'Pd', 'Ag', 'Cd', 'In',
    'Sb', 'Te', 'I', 'Xe', 'Ba', 'La', 'Ce', 'Pr', 'Nd', 'Pm', 'Sm', 'Eu', 'Gd', 'Tb', 'Dy', 'Ho', 'Er', 'Tm',
    'Yb', 'Lu', 'Hf', 'Ta', 'W', 'Re', 'Os', 'Ir', 'Pt', 'Au', 'Hg', 'Tl', 'Pb', 'Bi', 'Th', 'Pa', 'U', 'Np',
    'Pu', 'Am', 'Cm', 'Bk', 'Cf', 'Es', 'Fm', 'Md', 'No', 'Lr', 'Rf', 'Db', 'Sg', 'Bh', 'Hs', 'Mt', 'Ds', 'Rg',
    'Nh', 'Fl', 'Mc', 'Lv', 'Ts', 'Og']
_TWO = frozenset(two_char_symbols)


def tokenize_smiles(smiles, two_letter_atoms=two_char_symbols):
    """Greedy scan, two-character symbols first (the list also holds 'I', 'W', 'U', which can only
    match through the single-character branch, exactly as in the reference)."""
    two = _TWO if two_letter_atoms is two_char_symbols else two_letter_atoms
    tokens, i, n = [], 0, len(smiles)
    while i < n:
        if i < n - 1 and smiles[i:i + 2] in two:
            tokens.append(smiles[i:i + 2])
            i += 2
        else:
            tokens.append(smiles[i])
            i += 1
    return tokens


def encode_smiles(stoi, smi):
    unk = stoi.get('<UNK>')
    return [stoi.get(t, unk) for t in tokenize_smiles(smi)]


_MF_RE = re.compile(r'([A-Z][a-z]*)(\d{1,2})?')


def encode_MF(stoi_MF, mf):
    tokens = []
    for element, count in _MF_RE.findall(mf):
        tokens.append(element)
        if count:
            tokens.append(count)
    unk = stoi_MF.get('<UNK>')
    return [stoi_MF.get(t, unk) for t in tokens]


def tensor_to_smiles(tensor, itos):
    """Cut each column at the first <EOS>; one device->host copy instead of one .item() per token."""
    eos = next(int(k) for k, v in itos.items() if v == "<EOS>")
    table = [itos[str(i)] for i in range(len(itos))]
    t = tensor.detach().to("cpu").tolist()
    if tensor.dim() > 1:
        out = []
        for i in range(tensor.shape[1]):
            seq = []
            for row in t:
                tok = int(row[i])
                if tok == eos:
                    break
                seq.append(table[tok])
            out.append("".join(seq))
        return out
    seq = []
    for tok in t:
        if int(tok) == eos:
            break
        seq.append(table[int(tok)])
    return "".join(seq)


def tensor_to_smiles_and_prob_2(tensor, token_prob, itos):
    """helper_functions_pl_v15_4.py:390-419."""
    if tensor.dim() > 1 and token_prob.dim() > 1:
        smiles = tensor_to_smiles(tensor, itos)
        lens = []
        eos = next(int(k) for k, v in itos.items() if v == "<EOS>")
        cols = tensor.detach().to("cpu").t().tolist()
        for col in cols:
            n = 0
            for tok in col:
                if int(tok) == eos:
                    break
                n += 1
            lens.append(n)
        return smiles, [token_prob[:n, i] for i, n in enumerate(lens)]
    smi = tensor_to_smiles(tensor, itos)
    n = sum(1 for _ in _upto_eos(tensor, itos))
    return smi, token_prob[:n]


def _upto_eos(tensor, itos):
    eos = next(int(k) for k, v in itos.items() if v == "<EOS>")
    for tok in tensor.detach().to("cpu").tolist():
        if int(tok) == eos:
            return
        yield tok


def zero_pad(data, pad_length, dimensions=1):
    """dataloaders_pl_v15_4.py:267-299, including the 1-D >= pad_length case whose mask stays all ones."""
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


def collate_fn(batch):
    """dataloaders_pl_v15_4.py:665-712 (same keys and dtypes)."""
    from torch.nn.utils.rnn import pad_sequence
    out = {}
    for st in ('1H', '13C', 'HSQC', 'COSY', 'IR', 'HSQC_', 'COSY_'):
        for kind in ('src_', 'mask_'):
            key = kind + st
            data = [item[key] for item in batch if key in item]
            if data:
                out[key] = pad_sequence(data, batch_first=True).float()
    out['trg_MW'] = torch.stack([item['trg_MW'] for item in batch]).float()
    trg = [item['trg_enc_SMI'] for item in batch if 'trg_enc_SMI' in item]
    if trg:
        padded = pad_sequence(trg, batch_first=True).long()
        out['trg_enc_SMI'] = F.pad(padded, (0, 64 - padded.size(1)))
    out['trg_FP'] = pad_sequence([item['trg_FP'] for item in batch], batch_first=True).float()
    if all('src_MS' in item for item in batch):
        out['src_MS'] = pad_sequence([item['src_MS'] for item in batch], batch_first=True).long()
        out['mask_MS'] = pad_sequence([item['mask_MS'] for item in batch], batch_first=True).long()
    out['src_MF'] = pad_sequence([item['src_MF'] for item in batch], batch_first=True).long()
    out['mask_MF'] = pad_sequence([item['mask_MF'] for item in batch], batch_first=True).long()
    return out
