import sys; sys.path.insert(0,'.')
import torch, numpy as np
from tests.test_model_gpu import build, call
from tests.util import load_golden
from oracle import mmt_oracle as O
from multimodalspectraltransformer_amd.engine import D
name, mode = "fwd_c2_L2", "1H_HSQC_COSY_IR_MF_MW"
g = load_golden(name)
model, config, cfg, sd, batch, db = build(g, mode)
model.eval()
trg = db["trg_enc_SMI"].t()
with torch.no_grad():
    logits, fp, memory, mask = call(model, mode, db, trg[:-1, :])
    ologits, ofp, omem, omask = O.forward(sd, cfg, batch, batch["trg_enc_SMI"].t()[:-1])
err = (memory.cpu() - omem).abs()   # (L,B,128)
rowerr = err.amax(dim=2)            # (L,B)
print("max err", err.max().item())
bad = (rowerr > 0.1).nonzero()
print("bad rows (s,b):", bad.tolist()[:50], len(bad))
pl = model._last_plan if hasattr(model,'_last_plan') else list(model._engine.plans.values())[0]
B, Lm = pl.B, pl.Lmem
mem_in = pl.cv.view(pl.mem_in, B, Lm, D).float().cpu()
# oracle modal outputs
mode_ = cfg["training_mode"]
print("mods", pl.mods, "Lmem", Lm)
# compute oracle pre-cross memory
import types
def oracle_pre_cross():
    # replicate forward up to concatenation
    memory=[]
    nl, nh = cfg["num_encoder_layers"], cfg["num_heads"]
    mw = batch["trg_MW"].unsqueeze(1)
    e_MF,m_MF = O.embed_tokens(sd,"MF",batch["src_MF"],batch["mask_MF"])
    e_MW,m_MW = O.embed_MW(sd,mw); e_MW=e_MW.unsqueeze(0)
    outs={}
    for nme in ("1H","HSQC","COSY"):
        e,m = O.embed_spectrum(sd,nme,batch["src_"+nme],batch["mask_"+nme])
        x=torch.cat([e,e_MF,e_MW],0); mm=torch.cat([m,m_MF,m_MW],1)
        outs[nme]=(x, O.encoder(x,sd,nme,nl,nh,mm))
    e,m=O.embed_IR(sd,batch["src_IR"]); e=e.unsqueeze(0); m=m.unsqueeze(-1)
    x=torch.cat([e,e_MF,e_MW],0); mm=torch.cat([m,m_MF,m_MW],1)
    outs["IR"]=(x,O.encoder(x,sd,"IR",nl,nh,mm))
    return outs
with torch.no_grad():
    outs = oracle_pre_cross()
for nme,S,off in pl.mods:
    x_or, y_or = outs[nme]
    xs = pl.cv.view(pl.xseq[nme], B, S, D).float().cpu().transpose(0,1)
    print(nme, "embed err", (xs-x_or).abs().max().item(), "enc out err", (mem_in[:,off:off+S].transpose(0,1)-y_or).abs().amax(dim=2).max().item())
    e2 = (mem_in[:,off:off+S].transpose(0,1)-y_or).abs().amax(dim=2)
    print("   worst rows", (e2>0.1).nonzero().tolist()[:10])
