"""Synthetic workload of the benchmark and the measurement tools: a collate_fn-shaped batch of seeded random
spectra (shapes, dtypes and value ranges of utils_MMT/dataloaders_pl_v15_4.py:665-712; peak counts per modality
drawn from the ranges of the reference's CSV fixtures) and the architecture defaults of utils_MMT/config_V8.json.
There is no network for datasets or checkpoints, so throughput is measured on this data with random-init weights.
"""
import torch


def synthetic_batch(B, seed, mode="1H_HSQC_COSY_IR_MF_MW", float_masks=True):
    """A collate_fn-shaped batch dict of synthetic spectra.  Masks of MF/MS are
    float32 (same 0/1 values as the reference's int64) because torch >= 2
    refuses integer key-padding masks."""
    import numpy as np
    rng = np.random.default_rng(seed)
    b = {}

    def spectrum2d(lo, hi):
        x = np.zeros((B, 64, 2), np.float32)
        m = np.ones((B, 64), np.float32)
        for i in range(B):
            n = int(rng.integers(lo, hi + 1))
            x[i, :n] = rng.uniform(0, 1, (n, 2))
            m[i, :n] = 0
        return torch.from_numpy(x), torch.from_numpy(m)

    b["src_1H"], b["mask_1H"] = spectrum2d(2, 64)
    b["src_1H"][:, :, 1] = torch.round(b["src_1H"][:, :, 1] * 3) * (b["mask_1H"] == 0)
    b["src_HSQC"], b["mask_HSQC"] = spectrum2d(2, 25)
    b["src_COSY"], b["mask_COSY"] = spectrum2d(2, 54)
    x = np.zeros((B, 64), np.float32)
    m = np.ones((B, 64), np.float32)
    for i in range(B):
        n = int(rng.integers(8, 22))
        x[i, :n] = rng.uniform(0, 1, n)
        m[i, :n] = 0
    b["src_13C"], b["mask_13C"] = torch.from_numpy(x), torch.from_numpy(m)
    b["src_IR"] = torch.from_numpy(rng.uniform(0, 1, (B, 1000)).astype(np.float32))
    b["mask_IR"] = torch.zeros(B, 1000)
    mf = np.zeros((B, 64), np.int64)
    mfm = np.ones((B, 64), np.int64)
    ms = np.zeros((B, 64), np.int64)
    msm = np.ones((B, 64), np.int64)
    for i in range(B):
        k = int(rng.integers(4, 13))
        mf[i, 0] = 3
        mf[i, 1:1 + k] = rng.integers(5, 212, k)
        mf[i, 1 + k] = 2
        mfm[i, :k + 2] = 0
        ell = int(rng.integers(25, 61))
        ms[i, 0] = 3
        ms[i, 1:1 + ell] = rng.integers(5, 43, ell)
        ms[i, 1 + ell] = 2
        msm[i, :ell + 2] = 0
    b["src_MF"] = torch.from_numpy(mf)
    b["src_MS"] = torch.from_numpy(ms)
    if float_masks:
        b["mask_MF"] = torch.from_numpy(mfm).float()
        b["mask_MS"] = torch.from_numpy(msm).float()
    else:
        b["mask_MF"] = torch.from_numpy(mfm)
        b["mask_MS"] = torch.from_numpy(msm)
    b["trg_MW"] = torch.from_numpy(rng.uniform(150, 500, B).astype(np.float32))
    b["trg_enc_SMI"] = b["src_MS"].clone()
    b["trg_FP"] = torch.from_numpy((rng.uniform(0, 1, (B, 512)) < 0.1).astype(np.float32))
    return b


def default_config(**over):
    cfg = dict(input_dim_1H=2, input_dim_13C=1, input_dim_HSQC=2, input_dim_COSY=2, input_dim_IR=1000,
               MF_vocab_size=212, MS_vocab_size=43, in_size=43, hidden_size=128, out_size=43,
               num_encoder_layers=6, num_decoder_layers=6, num_heads=16, max_len=128, drop_out=0.1,
               fingerprint_size=512, use_real_data=False, training_mode="1H_HSQC_COSY_IR_MF_MW",
               temperature=1, lr_pretraining=1e-4)
    cfg.update(over)
    return cfg
