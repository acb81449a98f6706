"""Time of the cached greedy decode of the full-size model (64 molecules x max_len - 1 positions, no early stop), alone
on the GPU, for the forms of the token step: tools/bench_gen.py [batch] [reps]
  unsplit: one launch per position, every workgroup (= sequence) streams all weights
  split:   per layer one per-sequence launch and one token-batched feed-forward launch
each with plain launches and with one hipGraph per position (MMST_DECODE_GRAPH=1)."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from multimodalspectraltransformer_amd import synthetic as O
from multimodalspectraltransformer_amd import generate as gen
from multimodalspectraltransformer_amd import _lib as L
from multimodalspectraltransformer_amd.models_MMT import MultimodalTransformer

MODE = "1H_HSQC_COSY_IR_MF_MW"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ONLY = sys.argv[3].split(",") if len(sys.argv) > 3 else None
cfg = O.default_config(training_mode=MODE)
cfg.update(device="cuda:0")
config = argparse.Namespace(**cfg)
torch.manual_seed(0)
model = MultimodalTransformer(config, src_pad_idx=0).to("cuda:0").eval()
batch = {k: v.cuda() for k, v in O.synthetic_batch(B, 5, MODE).items()}
T = model.max_len - 1
ref = None
with torch.no_grad():
    pl = model.run_encoder_half(batch, T=T)
    for name, split, graph in (("unsplit", 0, "0"), ("split", 1, "0"), ("unsplit+graph", 0, "1"), ("split+graph", 1, "1")):
        if ONLY and name not in ONLY:
            continue
        os.environ["MMST_DECODE_GRAPH"] = graph
        L.lib().mmst_decoder_stack_config(split)
        if hasattr(pl, "_step_state"):
            pl._step_state.graphs = {}
        for _ in range(2):
            toks, probs = gen._decode_plan(model, pl, T, 1.0, 3, early_stop=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(REPS):
            toks, probs = gen._decode_plan(model, pl, T, 1.0, 3, early_stop=False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / REPS
        same = "" if ref is None else (" tokens equal: %s" % bool(torch.equal(toks, ref)))
        if ref is None:
            ref = toks.clone()
        if os.environ.get("GEN_STAMPS") and split:
            # timing build of the library (-DMMST_DEC_STAMP): room for the stamps behind the workspace, one more decode
            stt = pl._step_state
            n = stt.stack_ws.numel()
            stt.stack_ws = torch.zeros(n + 64, dtype=torch.float32, device="cuda")
            stt.stack.ffn_ws = stt.stack_ws.data_ptr()
            gen._decode_plan(model, pl, T, 1.0, 3, early_stop=False)
            torch.cuda.synchronize()
            st = stt.stack_ws[n:n + 20].view(torch.int64).cpu().numpy()
            print("stamps of the layer-2 launch at the last position (cycles from entry):", [int(x - st[0]) for x in st[:10]])
        print("%-14s %d molecules x %d positions: %.2f ms (%.1f us per position)%s" % (name, B, T, dt * 1e3, dt * 1e6 / T, same),
              flush=True)
