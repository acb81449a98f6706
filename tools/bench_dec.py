"""Time of the decoder forward (teacher-forced, train mode, dropout on) and of the decoder backward of the full-size
model on 64 molecules of synthetic spectra, alone on the GPU: tools/bench_dec.py [batch] [reps]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from multimodalspectraltransformer_amd import synthetic as O
from multimodalspectraltransformer_amd.models_MMT import MultimodalTransformer

MODE = "1H_HSQC_COSY_IR_MF_MW"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = O.default_config(training_mode=MODE)
cfg.update(device="cuda:0")
config = argparse.Namespace(**cfg)
torch.manual_seed(0)
model = MultimodalTransformer(config, src_pad_idx=0).to("cuda:0").train()
batch = {k: v.cuda() for k, v in O.synthetic_batch(B, 5, MODE).items()}
eng = model._engine
trg = batch["trg_enc_SMI"]
T = trg.shape[1] - 1
with torch.no_grad():
    pl = model.run_encoder_half(batch, T=T)
    pl.cv.view(pl.tok, model.max_len + 1, pl.B)[:T].copy_(trg.t()[:-1])
    for _ in range(3):
        eng.decode(pl, True, T=T)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        eng.decode(pl, True, T=T)
    e1.record()
    torch.cuda.synchronize()
    print("decoder forward (embed + %d layers + fc_out), %d molecules x %d positions: %.1f us" %
          (model.num_decoder_layers, B, T, e0.elapsed_time(e1) * 1e3 / REPS), flush=True)

if os.environ.get("DEC_STAMPS"):
    import numpy as np
    l = 2
    h = pl.cv.view(pl.dec_acts[l]["h"], pl.B * T * 2048 + 64)[pl.B * T * 2048:].view(torch.int64).cpu().numpy()
    print("mid stamps (cycles from entry):", [int(x - h[0]) for x in h[:7]])
    q = pl.cv.view(pl.dec_acts[l]["qkv"], pl.B * T * 384 + 256)[pl.B * T * 384:].view(torch.int64).cpu().numpy()
    print("head stamps:", [int(x - q[16]) for x in q[16:24]])
