"""Per-workgroup start / end of the feed-forward kernels in a training step: per launch the spread of the workgroups' durations, the
compute units that hold two of them, and which weight-gradient workgroups share a unit with the slowest ones.
Needs the library built by tools/wglog.sh build (kinds: 1 feed-forward backward, 2 forward, 3 weight gradients, 4 / 5 attention backward, 6 / 7 / 8 attention forward / dQ,
9 the in_proj-backward GEMM with its LayerNorm-backward epilogue)."""
import argparse, os, sys, ctypes as C
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from multimodalspectraltransformer_amd import synthetic as O, _lib as L
from multimodalspectraltransformer_amd.train import TransformerMultiGPU
MODE = "1H_HSQC_COSY_IR_MF_MW"
cfg = O.default_config(training_mode=MODE)
cfg.update(device="cuda:0", training_setup="pretraining", lr_finetuning=5e-5, model_save_dir="/tmp/mmst_bench",
           gen_SMI_sequence=False, sampling_method="greedy", batch_size=64, save_model=False)
config = argparse.Namespace(**cfg)
torch.manual_seed(1234)
tm = TransformerMultiGPU(config).to("cuda:0")
tm.configure_optimizers()
b = {k: v.cuda() for k, v in O.synthetic_batch(64, 1000, MODE).items()}
for i in range(4):
    tm.training_step(b, i)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (4 * 32768))()
n = C.c_uint(0)
assert L.lib().mmst_debug_wglog(buf, C.byref(n)) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).copy()
cnt = min(int(n.value), 32768)
a = a[:cnt]
for fn in ("mmst_debug_wglog2", "mmst_debug_wglog3", "mmst_debug_wglog4"):
    buf2 = (C.c_ulonglong * (4 * 32768))()
    assert getattr(L.lib(), fn)(buf2, C.byref(n)) == 0
    a2 = np.frombuffer(buf2, dtype=np.uint64).reshape(-1, 4)[:min(int(n.value), 32768)].copy()
    a = np.concatenate([a, a2])
cnt = len(a)
a = a[np.argsort(a[:, 0])]
# launches: a gap of > 5 us between consecutive starts of the same kind separates them
recs = [(int(r[0]), int(r[1]), int(r[2] >> 32), int(r[2] & 0xFFFFFFFF), int(r[3] >> 32) & 0xF, int(r[3] & 0xFFFFFFFF)) for r in a]
launches = []
for r in recs:
    same = [l for l in launches[-3:] if l[0][2] == r[2]]
    if same and r[0] - same[-1][-1][0] < 3000 and r[0] < same[-1][0][0] + 30000:
        same[-1].append(r)
    else:
        launches.append([r])
print("records", cnt, "launches", len(launches))
for ln in launches[-64:]:
    t0 = min(r[0] for r in ln); t1 = max(r[1] for r in ln)
    st = np.array([(r[0] - t0) / 100.0 for r in ln]); en = np.array([(r[1] - t0) / 100.0 for r in ln]); du = en - st
    cu = {}
    for r in ln:
        hw = r[5]; key = (r[4], (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)
        cu[key] = cu.get(key, 0) + 1
    print("t %7.1f kind %d wgs %3d span %6.1f us | start p50 %5.1f max %5.1f | dur p10 %5.1f p50 %5.1f p90 %5.1f max %5.1f | end p50 %5.1f p90 %5.1f | CUs %d, with 2+: %d"
          % ((t0 - launches[-64][0][0]) / 100.0, ln[0][2], len(ln), (t1 - t0) / 100.0, np.median(st), st.max(), np.percentile(du, 10), np.median(du), np.percentile(du, 90), du.max(),
             np.median(en), np.percentile(en, 90), len(cu), sum(1 for v in cu.values() if v > 1)))
print("slowest workgroups of the last backward launches (block, xcc, se, cu, start, dur):")
for ln in [l for l in launches if l[0][2] == 1][-8:]:
    t0 = min(r[0] for r in ln)
    top = sorted(ln, key=lambda r: r[0] - r[1])[:6]
    print(len(ln), [(r[3], r[4], (r[5] >> 13) & 7, (r[5] >> 8) & 15, round((r[0] - t0) / 100.0, 1), round((r[1] - r[0]) / 100.0, 1)) for r in top])
print("co-residents of the slowest backward workgroups (modal launches):")
k1 = [l for l in launches if l[0][2] == 1 and len(l) == 228][-4:]
k3 = [l for l in launches if l[0][2] == 3]
for ln in k1:
    t0 = min(r[0] for r in ln)
    w = [l for l in k3 if abs(l[0][0] - t0) < 3000]
    if not w: continue
    w = w[0]
    key = lambda r: (r[4], (r[5] >> 8) & 0xFF)
    on = {}
    for r in w: on.setdefault(key(r), []).append(r)
    top = sorted(ln, key=lambda r: r[0] - r[1])
    def desc(r):
        o = on.get(key(r), [])
        return (r[3], round((r[1] - r[0]) / 100.0, 1), [(x[3], round((x[0] - t0) / 100.0, 1), round((x[1] - x[0]) / 100.0, 1)) for x in o])
    print(" slow:", [desc(r) for r in top[:5]])
    print(" mid: ", [desc(r) for r in top[110:114]])
    print(" fast:", [desc(r) for r in top[-4:]])

print("attention backward launches: resident workgroups every 10 us; durations of the dK/dV blocks (first half of the grid) and dQ blocks")
for ln in [l for l in launches if l[0][2] in (4, 5)][-12:]:
    t0 = min(r[0] for r in ln); t1 = max(r[1] for r in ln)
    nb = max(r[3] for r in ln) + 1
    kv = np.array([(r[1] - r[0]) / 100.0 for r in ln if r[3] < nb // 2]); qq = np.array([(r[1] - r[0]) / 100.0 for r in ln if r[3] >= nb // 2])
    occ = [sum(1 for r in ln if r[0] <= t0 + 1000 * k < r[1]) for k in range(int((t1 - t0) / 1000) + 1)]
    print("kind %d wgs %4d span %6.1f us | kv p50 %5.1f max %5.1f | q p50 %5.1f max %5.1f | resident %s" % (ln[0][2], len(ln), (t1 - t0) / 100.0, np.median(kv), kv.max(), np.median(qq), qq.max(), occ))
print("attention workgroups that start within 10 us of their launch: how many sit on a CU that holds a weight-gradient workgroup at that moment")
allw = [r for l in launches if l[0][2] == 3 for r in l]
allb = [r for l in launches if l[0][2] == 1 for r in l]
key = lambda r: (r[4], (r[5] >> 8) & 0xFF)
for ln in [l for l in launches if l[0][2] in (4, 5)][-8:]:
    t0 = min(r[0] for r in ln)
    early = [r for r in ln if r[0] < t0 + 1000]
    wres = {}
    for w in allw:
        if w[0] <= t0 + 500 < w[1]: wres[key(w)] = wres.get(key(w), 0) + 1
    bres = {}
    for w in allb:
        if w[0] <= t0 + 500 < w[1]: bres[key(w)] = bres.get(key(w), 0) + 1
    cus = {}
    for r in early: cus[key(r)] = cus.get(key(r), 0) + 1
    print("kind %d: %d early workgroups on %d CUs; weight-gradient workgroups resident on %d CUs, feed-forward backward on %d; early ones on a CU with a weight-gradient workgroup: %d"
          % (ln[0][2], len(early), len(cus), len(wres), len(bres), sum(1 for r in early if key(r) in wres)))

# (the attention log holds 32768 records, less than a step's attention launches: only the launches near the end of the
# last step are complete; a forward launch listed here may have lost records to later ones)
print("attention forward launches still in the log (kind 6: head_dim 32, 7: head_dim 8): resident workgroups every 5 us")
for ln in [l for l in launches if l[0][2] in (6, 7)][-8:]:
    t0 = min(r[0] for r in ln); t1 = max(r[1] for r in ln)
    du = np.array([(r[1] - r[0]) / 100.0 for r in ln])
    occ = [sum(1 for r in ln if r[0] <= t0 + 500 * k < r[1]) for k in range(int((t1 - t0) / 500) + 1)]
    print("kind %d wgs %4d span %6.1f us | dur p10 %5.1f p50 %5.1f p90 %5.1f max %5.1f | resident %s" % (ln[0][2], len(ln), (t1 - t0) / 100.0, np.percentile(du, 10), np.median(du), np.percentile(du, 90), du.max(), occ))
