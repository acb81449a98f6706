"""Benchmark of the headline metric: molecules/s of one TRAINING step (forward in train mode with
dropout 0.1, cross-entropy on every position, backward, gradient all-reduce, AdamW) of the 4-modality
MultimodalTransformer (config_V8.json architecture, training_mode 1H_HSQC_COSY_IR_MF_MW, 64 molecules per
GPU), on N MI355X of one node, one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N --steps K --warmup W        (starts the N ranks itself, one process per GPU)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

--gpus N has the meaning of the reference's Trainer(gpus=N, accelerator='ddp') (MT_functions_v15_4.py:139-151): one
call, N processes, per-rank batch = --batch.  A rank count that differs from --gpus is an error, never a silent
single-GPU measurement.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MODE = "1H_HSQC_COSY_IR_MF_MW"
FLOP_PER_MOL_STEP = 28.08e9   # closed-form algorithmic work of one training step per molecule (3 x forward)


def host_cores():
    """Host cores this process may use: its CPU affinity, capped by the thread budget the node's launcher exports
    (a one-GPU share of a large host is a fraction of what os.cpu_count() shows)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for var in ("OMP_NUM_THREADS", "MKL_NUM_THREADS"):
        v = os.environ.get(var)
        if v and v.isdigit() and int(v) > 0:
            n = min(n, int(v))
    return max(1, n)


def cpu_baseline(sample_b=64, steps=3):
    """The CPU oracle (port of the reference arithmetic: the same torch CPU operators the reference runs, fp32)
    timed on the host cores on a bounded sample: training steps (forward + CE + backward + AdamW) on `sample_b`
    molecules of the same workload, one untimed warm-up step first."""
    from oracle import mmt_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    cfg = O.default_cfg(training_mode=MODE)
    sd = O.make_state_dict(cfg, 1)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    m = {k: torch.zeros_like(v) for k, v in sd.items()}
    v = {k: torch.zeros_like(x) for k, x in sd.items()}
    batch = O.synthetic_batch(sample_b, 5, MODE)
    t0 = None
    for step in range(0, steps + 1):
        if step == 1:
            t0 = time.time()
        for p in params.values():
            p.grad = None
        loss = O.training_forward(params, cfg, batch)[0]
        loss.backward()
        with torch.no_grad():
            O.adamw_step(params, {k: p.grad for k, p in params.items()}, m, v, step + 1, cfg["lr_pretraining"])
    dt = time.time() - t0
    return {"value": steps * sample_b / dt, "unit": "molecules/s", "cores": cores, "kind": "port",
            "sample": f"{steps} training steps (fwd+CE+bwd+AdamW, fp32, dropout off) of the CPU oracle on {sample_b} "
                      f"molecules each after one warm-up step, {MODE}, 6+6 layers; {dt:.1f} s on {cores} threads "
                      f"(os.cpu_count() = {os.cpu_count()})"}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary of the same command
    (profiles/*_hbm_traffic.json, written by tools/summarize_profile.py), newest round first; None if absent."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        for k, v in d.items():
            if k.startswith(kernel):
                return v["hbm_bytes_per_launch"]
    return None


def pmc_traffic_token_step():
    """HBM bytes per decode token step from the same summaries: a token step is one call of
    mmst_decoder_stack_step, i.e. one launch of the decoder-stack kernel or, in its split form, one launch per layer
    plus the token-batched feed-forward launches and a closing launch; the bytes of all of them are summed and divided
    by the number of token steps of the counter run (the launches of the least-launched stack-kernel instance)."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        ks = {k: v for k, v in d.items() if k.startswith("decode_stack_kernel") or k.startswith("decode_ffn_kernel")}
        stack = [v["launches_in_pmc_run"] for k, v in ks.items() if k.startswith("decode_stack_kernel")]
        if stack:
            return sum(v["hbm_bytes_per_launch"] * v["launches_in_pmc_run"] for v in ks.values()) / min(stack)
    return None


def launch_ranks(n):
    """Start n ranks of this script (torch.distributed.run, one process per GPU) and pass their output through.
    Called before anything in this process has touched the GPU; the parent only waits."""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def decode_bench(args, quiet=False):
    """Secondary benchmark (BASELINE configuration 4): inference-only beam-search SMILES generation of
    validate_generate_MMT_v15_4.beam_search for `--batch` molecules x `--beams` beams x gen_len 64 positions on the
    config_V8 model (6 + 6 layers), encoder memory already resident.  One step = one whole search (replayed from a
    hipGraph).  Roofline of the decoder token step (one launch, or one launch per layer plus the token-batched
    feed-forward launches, timed as a whole): algorithmic bytes (decoder weights
    once, every molecule's projected memory keys/values once, every sequence's cached prefix once, the new cache rows
    and logits written) over the launch durations measured with events on the launch stream."""
    from multimodalspectraltransformer_amd import synthetic as O
    from multimodalspectraltransformer_amd import generate as gen
    from multimodalspectraltransformer_amd.models_MMT import MultimodalTransformer
    dev = "cuda:0"
    torch.cuda.set_device(0)
    B, W, gen_len = args.batch, args.beams, 64
    cfg = O.default_config(training_mode=MODE)
    cfg.update(device=dev, gen_len=gen_len)
    config = argparse.Namespace(**cfg)
    torch.manual_seed(1234)
    model = MultimodalTransformer(config).to(dev).eval()
    batch = {k: v.to(dev) for k, v in O.synthetic_batch(B, 2000, MODE).items()}
    batch["src_HSQC_"], batch["src_COSY_"] = batch["src_HSQC"], batch["src_COSY"]
    stoi = {"<SOS>": 3, "<EOS>": 2}
    memory, mask, *_ = gen.run_model(model, batch, config)
    # per-launch timing of the token step in plain-launch mode
    os.environ["MMST_DECODE_GRAPH"] = "0"
    gen.beam_search(model, stoi, memory, mask, config, W)
    gen.STEP_EVENTS = []
    gen.beam_search(model, stoi, memory, mask, config, W)
    torch.cuda.synchronize()
    durs = [a.elapsed_time(b) * 1e-3 for a, b in gen.STEP_EVENTS]
    gen.STEP_EVENTS = None
    os.environ["MMST_DECODE_GRAPH"] = "1"
    nl, Lm, D, FF = model.num_decoder_layers, memory.shape[0], 128, 2048
    weights = nl * 2 * (3 * D * D + D * D + D * D + D * D + 2 * D * FF) + 2 * 64 * D
    per_pos = [weights + nl * (B * Lm * 2 * D * 2 + B * W * (pos + 1) * 2 * D * 2 + B * W * 3 * D * 2) + B * W * 64 * 4
               for pos in range(gen_len)]
    for i in range(args.warmup):
        gen.beam_search(model, stoi, memory, mask, config, W)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        beams = gen.beam_search(model, stoi, memory, mask, config, W)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if not durs:   # the host-ranked search (MMST_BEAM_HOST=1) has no per-launch timing
        durs = [float("nan")]
    a = sum(per_pos) / sum(durs) / 1e9
    res = {"metric": "molecules/sec beam-search SMILES decode", "value": B * args.steps / dt, "unit": "molecules/s",
           "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000 * dt / args.steps,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": f"beam search, {B} molecules x {W} beams x {gen_len} positions, {MODE}, config_V8 "
                                  f"architecture (6+6 layers), KV-cached, bookkeeping on the device, hipGraph replay",
                      "global_batch": B, "parallelism": "dp1"},
           "tokens_per_s": B * W * gen_len * args.steps / dt,
           "roofline": {"bound": "hbm", "achieved": a, "peak": 8000.0, "unit": "GB/s", "frac": a / 8000.0,
                        "traffic": pmc_traffic_token_step(),
                        "algorithmic_bytes_per_launch": sum(per_pos) / gen_len,
                        "kernel": f"decoder token step (mmst_decoder_stack_step: decode_stack_kernel<{W}> per layer + decode_ffn_kernel)",
                        "launches": len(durs), "avg_us": 1e6 * sum(durs) / len(durs)}}
    if not quiet:
        print(json.dumps(res))
    return res


def secondary_metrics(args, tm, batches, steps=5):
    """The two secondary metrics, measured after the primary timed region on the same GPU (a few seconds):
    gen_step: the training step with the reference's default in-training generation (config_V8.json gen_SMI_sequence
    true, sampling_method "mix": models_MMT_v15_4.py:1070, :1222-1297), `steps` steps after one warm-up step;
    decode: BASELINE configuration 4, beam search for 256 molecules x 3 beams x 64 positions (`--decode`), `steps`
    searches after two warm-up searches."""
    out = {}
    cfg = tm.config
    old = (cfg.gen_SMI_sequence, cfg.sampling_method)
    cfg.gen_SMI_sequence, cfg.sampling_method = True, "mix"
    try:
        tm.training_step(batches[0], 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            tm.training_step(batches[i % len(batches)], i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out["gen_step"] = {"ms_per_step": 1000 * dt / steps, "molecules_per_s": steps * args.batch / dt, "steps": steps,
                           "workload": f"training step as above plus in-training generation, gen_SMI_sequence=True, "
                                       f"sampling_method='mix', {cfg.max_len - 1} positions, {args.batch} molecules"}
    finally:
        cfg.gen_SMI_sequence, cfg.sampling_method = old
    del tm
    torch.cuda.empty_cache()
    dargs = argparse.Namespace(batch=256, beams=3, steps=steps, warmup=2)
    d = decode_bench(dargs, quiet=True)
    out["decode"] = {"ms_per_search": d["ms_per_step"], "molecules_per_s": d["value"], "tokens_per_s": d["tokens_per_s"],
                     "steps": steps, "workload": d["config"]["workload"], "roofline": d["roofline"]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tap", action="store_true", help="do not time the dominant kernel with HIP events")
    ap.add_argument("--gen", action="store_true", help="also run the in-training greedy generation (secondary metric)")
    ap.add_argument("--decode", action="store_true", help="secondary benchmark: beam-search decode instead of training")
    ap.add_argument("--beams", type=int, default=3)
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary metrics (training step with in-training generation; beam-search decode)")
    ap.add_argument("--repeats", type=int, default=3, help="timed blocks of --steps steps (the first one is `value`)")
    args = ap.parse_args()
    if args.decode:
        if args.batch == 64 and "--batch" not in sys.argv:
            args.batch = 256
        decode_bench(args)
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); run `python bench.py --gpus "
                 f"{args.gpus}` (it starts the ranks itself) or give torch.distributed.run --nproc-per-node {args.gpus}")
    # MMST_BENCH_ONE_GPU=1 is a REHEARSAL of the multi-rank code path on a single-GPU box (all ranks share
    # cuda:0 and reduce through gloo); its numbers mean nothing.  Real runs: one rank per GPU over RCCL.
    rehearsal = os.environ.get("MMST_BENCH_ONE_GPU") == "1"
    if rehearsal:
        import faulthandler
        local_rank = 0
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        faulthandler.dump_traceback_later(100, exit=True)   # a rehearsal that hangs says where
    if not rehearsal and torch.cuda.device_count() < world:
        sys.exit(f"bench.py: {world} ranks need {world} GPUs, this node shows {torch.cuda.device_count()}")
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(dev))

    from multimodalspectraltransformer_amd import synthetic as O
    from multimodalspectraltransformer_amd import _lib as L
    from multimodalspectraltransformer_amd.train import TransformerMultiGPU

    cfg = O.default_config(training_mode=MODE)
    cfg.update(device=dev, training_setup="pretraining", lr_finetuning=5e-5, model_save_dir="/tmp/mmst_bench",
               gen_SMI_sequence=bool(args.gen), sampling_method="greedy", batch_size=args.batch, save_model=False)
    config = argparse.Namespace(**cfg)
    torch.manual_seed(1234)
    tm = TransformerMultiGPU(config).to(dev)
    tm.configure_optimizers()
    # synthetic, seeded, resident in HBM before the timed region; a few distinct batches per rank
    batches = []
    for i in range(4):
        b = O.synthetic_batch(args.batch, 1000 + 17 * rank + i, MODE)
        batches.append({k: v.to(dev) for k, v in b.items()})

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if os.environ.get("MMST_BENCH_HP_STREAM") == "1":   # experiment: critical path on a high-priority stream
        hp = torch.cuda.Stream(priority=-1)
        hp.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(hp)
    for i in range(args.warmup):
        tm.training_step(batches[i % len(batches)], i)
    sync()
    lib = L.lib()
    # time every launch of the dominant kernel (the split-K weight-gradient GEMM, see profiles/) with HIP
    # events on its own stream during the timed steps
    tap = rank == 0 and not args.no_tap
    if tap:
        # the kernel with the largest share of GPU time (profiles/): the recomputing feed-forward weight-gradient kernel,
        # or, with stored hidden gradients (MMST_FFN_WGRAD_RECOMPUTE=0), the split-K weight-gradient GEMM
        recompute = lib.mmst_ffn_wgrad_mode() >= 1
        L.check(lib.mmst_tap_config(100 if recompute else L.EPI_ATOMIC, 1, 1, 256 * args.steps), "mmst_tap_config")
    gaps = os.environ.get("MMST_BENCH_GAPS") == "1"   # diagnostic: GPU idle time between consecutive steps
    evs = []
    red_wait, red_coll, red_bytes = [], [], []

    def timed_block(first):
        """--steps training steps between two barrier + synchronize pairs; seconds (max over ranks), host enqueue time"""
        nonlocal out
        sync()
        t0 = time.perf_counter()
        for i in range(args.steps):
            if gaps and first:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            out = tm.training_step(batches[i % len(batches)], i)
            if gaps and first:
                e1.record()
                evs.append((e0, e1))
            if world > 1 and first:
                red_coll.append(tm.reducer.collectives)
                red_bytes.append(tm.reducer.bytes_reduced)
                red_wait.append((tm.reducer.wait_events, tm.reducer.wait_host_s))
        t_enq = time.perf_counter() - t0   # host time to enqueue the steps (the GPU may still be running)
        sync()
        dt_ = time.perf_counter() - t0
        t = torch.tensor([dt_], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t), t_enq

    out = None
    dt, t_enqueue = timed_block(True)
    repeats = [dt]
    for _ in range(max(args.repeats, 1) - 1):
        repeats.append(timed_block(False)[0])
    loss = float(out["loss"])
    # the tap's average covers the timed steps only: read it before the isolated steps below
    tap_read = None
    if tap:
        ms = C.c_float(0)
        cnt = C.c_int(0)
        flop = C.c_double(0)
        byts = C.c_double(0)
        name = C.create_string_buffer(128)
        lib.mmst_tap_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int]
        lib.mmst_tap_read(C.byref(ms), C.byref(cnt), C.byref(flop), C.byref(byts), name, 128)
        tap_read = (ms.value, cnt.value, flop.value, byts.value, name.value.decode())
        L.check(lib.mmst_tap_config(-1, 0, 0, 0), "mmst_tap_config")
    # host cost of enqueueing ONE step with an empty GPU queue (in the timed loop above the host runs ahead of the GPU
    # until the hardware queue is full and then waits for it, so the loop's enqueue time only mirrors the GPU time)
    iso = []
    for i in range(5):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        tm.training_step(batches[i % len(batches)], args.steps + i)
        iso.append(time.perf_counter() - t1)
    torch.cuda.synchronize()
    host_isolated_ms = 1000 * sorted(iso)[len(iso) // 2]
    replicas_identical = None
    if world > 1:
        # replicas that average their gradients every step stay bit-identical
        ar = tm.model._engine.arena
        mine = {k: float(ar.P[ar.offset[k]: ar.offset[k] + ar.numel[k]].double().sum()) for k in ar.order}
        sums = [None] * world
        dist.all_gather_object(sums, mine)
        bad = [k for k in ar.order if any(s[k] != sums[0][k] for s in sums)]
        assert not bad, f"replicas diverged in {len(bad)} of {len(ar.order)} tensors, e.g. {bad[:6]}"
        replicas_identical = True

    if rank == 0:
        mols = world * args.batch * args.steps
        res = {"metric": "molecules/sec training step (4-modality, seq<=128)", "value": mols / dt,
               "unit": "molecules/s", "n_gpus": world, "ranks": dist.get_world_size() if world > 1 else 1,
               "collective": (dist.get_backend() if world > 1 else None), "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": 1000 * dt / args.steps, "host_enqueue_ms_per_step": host_isolated_ms,
               "host_loop_ms_per_step": 1000 * t_enqueue / args.steps,
               "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": f"training step, {MODE}, config_V8 architecture (6+6 layers, d=128, ff=2048), "
                                      f"{args.batch} molecules/GPU, dropout 0.1, AdamW, "
                                      f"gen_SMI_sequence={bool(args.gen)}",
                          "global_batch": world * args.batch, "parallelism": f"dp{world}"},
               "final_loss": loss,
               "model_tflops": mols * FLOP_PER_MOL_STEP / dt / 1e12,
               # the same timed block again (--repeats in all, the first one above): one noisy box does not decide
               "ms_per_step_repeats": [1000 * r / args.steps for r in repeats],
               "ms_per_step_min": 1000 * min(repeats) / args.steps,
               "ms_per_step_median": 1000 * sorted(repeats)[len(repeats) // 2] / args.steps}
        if world > 1:
            # gradient exchange of rank 0 in the first timed block: what backward did not hide is the time finish()
            # held the compute stream up
            waits = [(w[0][0].elapsed_time(w[0][1]) if w[0] is not None else 0.0) + 1000.0 * w[1] for w in red_wait]
            res["allreduce_wait_ms_per_step"] = sum(waits) / max(len(waits), 1)
            res["collectives_per_step"] = sum(red_coll) / max(len(red_coll), 1)
            res["bytes_reduced_per_step"] = sum(red_bytes) / max(len(red_bytes), 1)
            res["replicas_identical"] = replicas_identical
        if gaps:
            res["gpu_idle_between_steps_ms"] = sum(evs[i][1].elapsed_time(evs[i + 1][0]) for i in range(len(evs) - 1)) / (len(evs) - 1)
            res["gpu_ms_inside_step"] = sum(a.elapsed_time(b) for a, b in evs) / len(evs)
        if tap_read is not None:
            class _V:   # (the tap was read right after the timed blocks)
                def __init__(self, v):
                    self.value = v
            ms, cnt, flop, byts = (_V(v) for v in tap_read[:4])
            name = _V(tap_read[4].encode())
            if cnt.value > 0:
                avg_s = ms.value / 1000.0 / cnt.value
                # the roof that binds: the larger of bytes / HBM peak and flops / MFMA peak per launch
                t_hbm, t_mfma = byts.value / cnt.value / 8.0e12, flop.value / cnt.value / 2.5e15
                bound = os.environ.get("MMST_TAP_BOUND") or ("hbm" if t_hbm >= t_mfma else "mfma")
                if bound == "hbm":
                    a = byts.value / cnt.value / avg_s / 1e9
                    res["roofline"] = {"bound": "hbm", "achieved": a, "peak": 8000.0, "unit": "GB/s",
                                       "frac": a / 8000.0, "traffic": pmc_traffic(name.value.decode()),
                                       "algorithmic_bytes_per_launch": byts.value / cnt.value,
                                       "kernel": name.value.decode(),
                                       "launches": cnt.value, "avg_us": avg_s * 1e6}
                else:
                    a = flop.value / cnt.value / avg_s / 1e12
                    res["roofline"] = {"bound": "mfma", "achieved": a, "peak": 2500.0, "unit": "TFLOP/s",
                                       "frac": a / 2500.0, "traffic": pmc_traffic(name.value.decode()),
                                       "algorithmic_flops_per_launch": flop.value / cnt.value,
                                       "algorithmic_bytes_per_launch": byts.value / cnt.value,
                                       "hbm_frac": byts.value / cnt.value / avg_s / 8.0e12,
                                       "kernel": name.value.decode(), "launches": cnt.value, "avg_us": avg_s * 1e6}
        if world == 1 and not args.no_secondary:
            res["secondary"] = secondary_metrics(args, tm, batches)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
