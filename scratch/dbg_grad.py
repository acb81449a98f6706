import sys; sys.path.insert(0,'.')
import torch, numpy as np
from tests.test_model_gpu import build, call
from tests.util import load_golden
from oracle import mmt_oracle as O
name, mode = sys.argv[1], sys.argv[2]
g = load_golden(name)
model, config, cfg, sd, batch, db = build(g, mode)
model.eval()
eng = model._engine; eng.ensure("cuda"); eng.arena.G.zero_()
loss = model.loss_and_backward(db)
sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
lo = O.training_forward(sdo, cfg, batch)[0]; lo.backward()
print("loss", float(loss), float(lo))
rows=[]
for k,p in sdo.items():
    if p.grad is None: continue
    a = eng.arena.grad_view(k).cpu().reshape(-1).double(); b = p.grad.reshape(-1).double()
    if b.norm()==0: continue
    cos = float((a@b)/(a.norm()*b.norm()+1e-30)); rel = float((a-b).norm()/b.norm())
    rows.append((cos, rel, k))
rows.sort()
for r in rows[:25]: print("%.5f %.4f %s"%r)
print("median cos", np.median([r[0] for r in rows]), "n", len(rows))
