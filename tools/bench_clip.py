"""Time of one CLIP contrastive step (encoder half + projection heads + loss + backward) at the reference's
CLIP_batch_size (128), config_V8 architecture, synthetic inputs, given ChemBERTa features.  usage: bench_clip.py [B]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from multimodalspectraltransformer_amd import synthetic as S
from multimodalspectraltransformer_amd.models_MMT import MultimodalTransformer
from multimodalspectraltransformer_amd.models_CLIP import CLIPModel

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
mode = "1H_HSQC_COSY_IR_MF_MW"
cfg = S.default_config(training_mode=mode)
cfg.update(device="cuda", training_setup="pretraining", model_save_dir="/tmp/mmst_clip", CB_embedding=1024, MT_embedding=512,
           projection_dim=512, dropout=0.1, CLIP_temperature=1)
config = argparse.Namespace(**cfg)
torch.manual_seed(0)
mt = MultimodalTransformer(config).to("cuda")
clip = CLIPModel(None, mt, None, None, config)
clip.train()
batch = {k: v.cuda() for k, v in S.synthetic_batch(B, 3, mode).items()}
cb = torch.randn(B, 1024, device="cuda")
eng = mt._engine
eng.ensure("cuda")
for _ in range(3):
    eng.arena.G.zero_()
    loss, losses, dcb = clip.loss_and_backward(batch, cb)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    eng.arena.G.zero_()
    loss, losses, dcb = clip.loss_and_backward(batch, cb)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("CLIP step, %d molecules: %.2f ms -> %.0f molecules/s (loss %.3f)" % (B, dt * 1e3, B / dt, float(loss)))
