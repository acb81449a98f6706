"""Training harness for the MI355X path: the counterpart of TransformerMultiGPU + run_training_MMT
(utils_MMT/models_MMT_v15_4.py:985-1220, utils_MMT/MT_functions_v15_4.py:93-158) without
pytorch_lightning: one process per GPU, per-rank batch = config.batch_size, CE loss on every
position, AdamW(lr) + ReduceLROnPlateau(factor 0.5, patience 2), gradients averaged over ranks
with RCCL all-reduce on contiguous tail segments of the flat gradient buffer while the earlier
layers are still in backward.
"""
import ctypes as C
import json
import os
import time

import torch
import torch.distributed as dist

from . import _lib as L
from . import helper_functions as hf
from .models_MMT import MultimodalTransformer


def shard_indices(n, rank, world):
    """DistributedSampler(shuffle=False, drop_last=False) sharding: pad the index list to a multiple of
    `world` by wrapping around, then rank r takes r, r + world, ..."""
    idx = list(range(n))
    total = (n + world - 1) // world * world
    idx += idx[: total - n]
    return idx[rank:total:world]


class GradReducer:
    """Average a flat gradient buffer over ranks in contiguous segments, each all-reduce issued as soon as its
    segment is final (device tensors: RCCL; CPU tensors: gloo).

    Device tensors: the collective is issued asynchronously with the engine's weight-gradient stream current.
    torch's process group runs it on its own communication stream, ordered after everything enqueued so far on
    that stream (which by then has waited for the activation-gradient stream), so neither compute stream ever
    waits for a reduction until finish().  No stream of its own is created: with four hardware queues per
    process every extra stream shares a queue with a compute stream."""

    def __init__(self, flat, segments, world=None, active=None):
        """active: optional list of [lo, hi) ranges; only their intersection with a segment is reduced (parameters
        outside never receive gradients in the current training mode, e.g. an absent modality's encoder)."""
        self.flat, self.segments, self.active = flat, segments, active
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.cuda = flat.is_cuda
        self.works = []
        # what the last step sent: collectives issued, bytes reduced; exposed wait of finish() as a pair of events
        # on the stream that waited (device tensors) or as host seconds (CPU tensors, host-staged rehearsal)
        self.collectives = 0
        self.bytes_reduced = 0
        self.wait_events = None
        self.wait_host_s = 0.0

    def on_segment(self, name, also=()):
        """`also`: the engine's weight-gradient stream (its already-enqueued work writes this segment)."""
        if self.world == 1:
            return
        lo, hi = self.segments[name]
        if self.active is None:
            pieces = [(lo, hi)]
        else:
            pieces = [(max(lo, a), min(hi, b)) for a, b in self.active if max(lo, a) < min(hi, b)]
        if pieces:
            # the pieces of one segment (a segment is cut where an inactive parameter range lies inside it) go out as
            # ONE collective
            self._reduce([self.flat[a:b] for a, b in pieces], also)

    def _reduce(self, views, also):
        side = next((s for s in also if s is not None), None) if self.cuda else None
        self.collectives += 1
        self.bytes_reduced += sum(v.numel() * v.element_size() for v in views)

        def issue():
            if len(views) == 1:
                return dist.all_reduce(views[0], op=dist.ReduceOp.SUM, async_op=True)
            return dist.all_reduce_coalesced(views, op=dist.ReduceOp.SUM, async_op=True)

        if self.cuda and dist.get_backend() == "gloo":
            # single-GPU rehearsal of the multi-rank path (several ranks on one card): stage through the host,
            # synchronously -- this exercises the segment bookkeeping, not the overlap
            t0 = time.perf_counter()
            if side is not None:
                side.synchronize()
            torch.cuda.current_stream().synchronize()
            host = torch.cat([v.reshape(-1) for v in views]).cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            off = 0
            for v in views:
                v.copy_(host[off:off + v.numel()])
                off += v.numel()
            self.wait_host_s += time.perf_counter() - t0
        elif side is not None:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self.works.append(issue())
        else:
            self.works.append(issue())

    def begin_step(self):
        self.collectives, self.bytes_reduced, self.wait_events, self.wait_host_s = 0, 0, None, 0.0

    def finish(self):
        """Wait for the segment reductions (device tensors: the current stream waits, not the host); the caller
        scales by 1/world (fused into the optimizer).  The wait is bracketed by two events on the waiting stream (device
        tensors) or timed on the host: what a step spends here is the part of the reductions that backward did not hide."""
        timed = self.cuda and bool(self.works)
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        t0 = time.perf_counter()
        for i, w in enumerate(self.works):
            if os.environ.get("MMST_DEBUG_REDUCE"):
                print(f"[reduce] rank {dist.get_rank()} waits for segment work {i} of {len(self.works)}", flush=True)
            w.wait()
        if timed:
            e1.record()
            self.wait_events = (e0, e1)
        else:
            self.wait_host_s += time.perf_counter() - t0
        self.works = []
        return 1.0 / self.world

    def exposed_wait_ms(self):
        """Milliseconds the last step's finish() held the compute stream (or the host) up; call after a synchronize."""
        ms = 1000.0 * self.wait_host_s
        if self.wait_events is not None:
            ms += self.wait_events[0].elapsed_time(self.wait_events[1])
        return ms


class FusedAdamW:
    """torch.optim.AdamW(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01) on the flat parameter
    buffer: one launch over all the active parameter runs of the training mode (which also leaves their gradients
    zeroed for the next step), or one launch per run for caller-given ranges; also refreshes the bf16 weights."""

    def __init__(self, model, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        self.model, self.lr, self.betas, self.eps, self.wd = model, lr, betas, eps, weight_decay
        self.step_count = 0
        model._external_optimizer = False

    def step(self, mode, grad_scale=1.0, ranges=None):
        """ranges: [lo, hi) element ranges to update instead of the training mode's active parameters."""
        eng = self.model._engine
        ar = eng.arena
        if ar.M is None:
            ar.M = torch.zeros_like(ar.P)
            ar.V = torch.zeros_like(ar.P)
        self.step_count += 1
        ar.weights_serial += 1
        st = L.stream_ptr()
        rs = list(ranges if ranges is not None else ar.active_ranges(mode))
        if len(rs) <= 16 and ranges is None:
            # every active range in one launch, which also leaves their gradients zeroed for the next step
            lo = (C.c_long * len(rs))(*[r[0] for r in rs])
            hi = (C.c_long * len(rs))(*[r[1] for r in rs])
            L.check(L.lib().mmst_adamw_ranges(C.c_void_p(ar.P.data_ptr()), C.c_void_p(ar.G.data_ptr()),
                                              C.c_void_p(ar.M.data_ptr()), C.c_void_p(ar.V.data_ptr()),
                                              C.c_void_p(ar.Pb.data_ptr()), len(rs), lo, hi, C.c_float(self.lr),
                                              C.c_float(self.betas[0]), C.c_float(self.betas[1]), C.c_float(self.eps),
                                              C.c_float(self.wd), self.step_count, C.c_float(grad_scale), 1, st),
                    "mmst_adamw_ranges")
            ar.grads_zeroed = True
            rs = []
        for lo, hi in rs:
            L.check(L.lib().mmst_adamw(C.c_void_p(ar.P.data_ptr() + 4 * lo), C.c_void_p(ar.G.data_ptr() + 4 * lo),
                                       C.c_void_p(ar.M.data_ptr() + 4 * lo), C.c_void_p(ar.V.data_ptr() + 4 * lo),
                                       C.c_void_p(ar.Pb.data_ptr() + 2 * lo), C.c_long(hi - lo), C.c_float(self.lr),
                                       C.c_float(self.betas[0]), C.c_float(self.betas[1]), C.c_float(self.eps),
                                       C.c_float(self.wd), self.step_count, C.c_float(grad_scale), st), "mmst_adamw")
        ar.refresh_fragments()   # read by the decoder's forward pass: on this stream
        # the transposed feed-forward weights are read by the backward pass only: refresh them on the engine's second
        # stream, beside the next forward pass; Engine.backward waits for ar.transposed_ready
        main = torch.cuda.current_stream()
        ws = eng._side_stream(main)
        ws.wait_stream(main)
        with torch.cuda.stream(ws):
            ar.refresh_transposed()
            ar.transposed_ready = torch.cuda.Event()
            ar.transposed_ready.record(ws)

    def zero_grad(self):
        self.model._engine.arena.G.zero_()


class ReduceLROnPlateau:
    """mode='min', factor 0.5, patience 2 (models_MMT_v15_4.py:1215), threshold 1e-4 relative."""

    def __init__(self, opt, factor=0.5, patience=2, threshold=1e-4):
        self.opt, self.factor, self.patience, self.threshold = opt, factor, patience, threshold
        self.best, self.bad = float("inf"), 0

    def step(self, metric):
        if metric < self.best * (1 - self.threshold):
            self.best, self.bad = metric, 0
        else:
            self.bad += 1
        if self.bad > self.patience:
            self.opt.lr *= self.factor
            self.bad = 0


class TransformerMultiGPU:
    """Same role and attribute names as the reference's LightningModule wrapper."""

    def __init__(self, config):
        self.config = config
        os.makedirs(config.model_save_dir, exist_ok=True)
        self.model = MultimodalTransformer(config, src_pad_idx=0)
        self.lr = config.lr_pretraining if config.training_setup == "pretraining" else config.lr_finetuning
        self.itos = self.stoi = self.itos_MF = self.stoi_MF = None
        for name in ("itos", "stoi", "itos_MF", "stoi_MF"):
            path = getattr(config, name + "_path", None)
            if path and os.path.exists(path):
                setattr(self, name, json.load(open(path)))
        self.optimizer = None
        self.scheduler = None
        self.reducer = None
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.start_time = time.time()
        self.current_epoch = 0
        self.global_step = 0
        self.last_train_loss = None    # device tensor of the latest training_step (read only when a file is written)

    def to(self, device):
        self.model.to(device)
        return self

    def configure_optimizers(self):
        self.optimizer = FusedAdamW(self.model, lr=self.lr)
        self.scheduler = ReduceLROnPlateau(self.optimizer)
        return self.optimizer

    def forward(self, x):
        mode = self.config.training_mode
        g = lambda k, tag: x[k] if tag in mode else None  # noqa: E731
        return self.model(g('src_1H', '1H'), g('mask_1H', '1H'), g('src_13C', '13C'), g('mask_13C', '13C'),
                          g('src_HSQC', 'HSQC'), g('mask_HSQC', 'HSQC'), g('src_COSY', 'COSY'), g('mask_COSY', 'COSY'),
                          g('src_IR', 'IR'), g('mask_IR', 'IR'), g('src_MF', 'MF'), g('mask_MF', 'MF'),
                          g('src_MS', 'MS'), g('mask_MS', 'MS'), x['trg_MW'], x.get('trg_SMI_input'))

    def training_step(self, x, batch_idx=0):
        """One optimisation step: forward, CE (weight_SMI = 1, every other weight 0 in config_V8.json),
        backward, gradient all-reduce, AdamW.  Returns {'loss': device tensor}."""
        if self.optimizer is None:
            self.configure_optimizers()
        model, eng = self.model, self.model._engine
        model.train()
        eng.step_seed = (eng.step_seed * 1664525 + 1013904223) & 0xFFFFFFFF
        eng.ensure(model._dev())
        ar = eng.arena
        if self.reducer is None or self.reducer.flat is not ar.G:
            self.reducer = GradReducer(ar.G, ar.reduce_segments(), self.world,
                                       active=ar.active_ranges(self.config.training_mode))
        self.reducer.begin_step()
        if os.environ.get("MMST_C_STEP", "1") != "0" and not getattr(self.config, "use_real_data", False):
            # the whole step in one call into the library (mmst_train_step); with several ranks the optimizer runs
            # after the segment reductions have been waited for
            T = x['trg_enc_SMI'].shape[1] - 1
            first = next(x["src_" + t] for t in ("1H", "13C", "HSQC", "COSY") if t in self.config.training_mode)
            pl = eng.plan(first.shape[0], T, self.config.training_mode)
            opt = self.optimizer
            adam = None
            # MMST_REDUCE_PATH=1 takes the several-rank host path (segment callbacks, separate optimizer launch) with
            # one rank too, to time it on one GPU
            ddp_path = self.world > 1 or os.environ.get("MMST_REDUCE_PATH") == "1"
            # in-training generation decodes with the weights the forward pass used (the reference calls it between
            # forward and loss, models_MMT_v15_4.py:1064-1077): the optimizer runs behind it, as a launch of its own
            gen = bool(getattr(self.config, "gen_SMI_sequence", False))
            if not ddp_path and not gen:
                opt.step_count += 1
                ar.weights_serial += 1
                adam = (opt.step_count, opt.lr, opt.betas, opt.eps, opt.wd)
            names = {L.SEG_DECODER: "decoder", L.SEG_CROSS: "cross", L.SEG_EMBED: "embed"}
            cb = None
            if ddp_path:
                def cb(seg, user):
                    name = names.get(seg) or f"modal_l{seg - L.SEG_MODAL0}"
                    self.reducer.on_segment(name, also=(getattr(eng, "wstream", None),))
            loss = eng.train_step(pl, x, train=True, zero_grads=not ar.grads_zeroed, adam=adam, on_segment=cb)
            model._last_plan = pl
            if gen:
                gen_out = self._generate_before_optimizer(pl)
            if ddp_path or gen:
                scale = self.reducer.finish()
                self.optimizer.step(self.config.training_mode, grad_scale=scale)
        else:
            ar.G.zero_()
            loss = model.loss_and_backward(
                x, on_segment=lambda name: self.reducer.on_segment(name, also=(getattr(eng, "wstream", None),)))
            if getattr(self.config, "gen_SMI_sequence", False):
                gen_out = self._generate_before_optimizer(model._last_plan)
            scale = self.reducer.finish()
            self.optimizer.step(self.config.training_mode, grad_scale=scale)
        out = {'loss': loss.clone()}   # the plan's loss slot is rewritten by the next step
        self.last_train_loss = out['loss']
        self.global_step += 1
        if getattr(self.config, "gen_SMI_sequence", False):
            out['gen'] = gen_out
        return out

    def _generate_before_optimizer(self, pl):
        """In-training generation on the plan of the step that has just been enqueued, BEFORE its optimizer: the
        reference generates between the forward pass and the loss (models_MMT_v15_4.py:1064-1077), i.e. with the weights
        the forward pass used, on that pass's train-mode memory; the decoder layers' projections of that memory are
        still in the plan.  (Running the decode loop on a stream of its own beside the backward pass gains nothing: a
        workgroup of the token step is 1024 threads and needs a whole idle compute unit, which the backward kernels
        never leave; measured, the first token launch started when backward ended.)"""
        from .generate import generate_sequences
        out = generate_sequences(self.model, pl, self.config, self.stoi, self.itos, temperature=1, kv_ready=True)
        self.model.train()
        return out

    def validation_step(self, x, batch_idx=0):
        self.model.eval()
        with torch.no_grad():
            x = dict(x)
            trg = x['trg_enc_SMI'].to(self.model._dev()).t()
            x['trg_SMI_input'] = trg[:-1, :]
            logits, fp, memory, mask = self.forward(x)
            pl = self.model._engine.plan(trg.shape[1], trg.shape[0] - 1, self.config.training_mode)
            loss = self.model._engine.ce_loss(pl, x['trg_enc_SMI'], want_grad=False)
        return {'loss': loss.clone()}

    def _generate_sequences(self, model, config, memory, src_padding_mask, max_output_len, temperature, stoi, itos):
        """Reference signature and return tuple (models_MMT_v15_4.py:1222-1297): (gen tensor (max_output_len + 1, B)
        incl. <SOS>, SMILES list, confidence list) for config.sampling_method greedy / multinomial / mix; like the
        reference it leaves the model in train mode."""
        from .generate import _plan_for, generate_sequences
        pl = _plan_for(model, memory)
        out = generate_sequences(model, pl, config, stoi, itos, temperature=temperature, max_len=max_output_len)
        model.train()
        return out

    def on_epoch_end(self, epoch, loss):
        """Per-epoch weights file, models_MMT_v15_4.py:1190-1194."""
        path = os.path.join(self.config.model_save_dir, f"MultimodalTransformer_Epoch_{epoch}_Loss{loss:.3f}.pth")
        torch.save(self.model.state_dict(), path)
        return path

    def on_batch_end(self):
        """Wall-clock checkpoints, models_MMT_v15_4.py:1196-1210: once config.model_save_interval seconds have passed
        since the last one, write the weights (.pth) and a Lightning-layout checkpoint (.ckpt) and restart the clock.
        Returns the two paths, or None when nothing was due."""
        interval = getattr(self.config, "model_save_interval", None)
        if interval is None or time.time() - self.start_time < interval:
            return None
        loss = float(self.last_train_loss) if self.last_train_loss is not None else float("nan")
        base = os.path.join(self.config.model_save_dir, f"MultimodalTransformer_time_{str(time.time())}_Loss_{loss:.3f}")
        torch.save(self.model.state_dict(), base + ".pth")
        self.save_checkpoint(base + ".ckpt")
        self.start_time = time.time()
        return base + ".pth", base + ".ckpt"

    def state_dict(self):
        """Keys of the reference LightningModule's state_dict: the model's, prefixed 'model.'."""
        return {"model." + k: v for k, v in self.model.state_dict().items()}

    def save_checkpoint(self, path):
        """A file in the layout of pytorch_lightning's Trainer.save_checkpoint as the reference's loaders use it
        (mmt_result_test_functions_15_4.py:467-475, MT_functions_v15_4.py:121-123): a dict whose 'state_dict' maps
        'model.<key>' to tensors, with the training position and the optimizer state beside it.  Only tensors and
        plain Python values are stored, so the file loads with torch.load(weights_only=True)."""
        ckpt = {"epoch": int(self.current_epoch), "global_step": int(self.global_step),
                "pytorch-lightning_version": "0.7.3",
                "state_dict": {k: v.detach().cpu() for k, v in self.state_dict().items()}}
        opt = self.optimizer
        if opt is not None:
            ar = self.model._engine.arena
            state = {"lr": float(opt.lr), "betas": [float(b) for b in opt.betas], "eps": float(opt.eps),
                     "weight_decay": float(opt.wd), "step": int(opt.step_count)}
            if ar is not None and ar.M is not None:
                state["exp_avg"] = {k: ar.M[ar.offset[k]: ar.offset[k] + ar.numel[k]].view(ar.named[k].shape).cpu().clone()
                                    for k in ar.order}
                state["exp_avg_sq"] = {k: ar.V[ar.offset[k]: ar.offset[k] + ar.numel[k]].view(ar.named[k].shape).cpu().clone()
                                       for k in ar.order}
            ckpt["optimizer_states"] = [state]
            if self.scheduler is not None:
                ckpt["lr_schedulers"] = [{"best": float(self.scheduler.best), "num_bad_epochs": int(self.scheduler.bad),
                                          "factor": float(self.scheduler.factor), "patience": int(self.scheduler.patience)}]
        torch.save(ckpt, path)
        return path

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, config=None, resume_optimizer=False):
        """Reads the reference's .ckpt (Lightning dict, keys prefixed 'model.') or .pth (plain state_dict)
        with a loader that executes nothing from the file.  resume_optimizer=True also restores the AdamW moments,
        step count, learning rate and plateau scheduler of a checkpoint written by save_checkpoint()."""
        self = cls(config)
        sd = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
        full = sd if "state_dict" in sd else None
        if full is not None:
            sd = {k[len("model."):]: v for k, v in full["state_dict"].items() if k.startswith("model.")}
        self.model.load_state_dict(sd)
        if full is not None:
            self.current_epoch = int(full.get("epoch", 0))
            self.global_step = int(full.get("global_step", 0))
            if resume_optimizer and full.get("optimizer_states"):
                self._resume_state = (full["optimizer_states"][0], (full.get("lr_schedulers") or [None])[0])
        return self

    def restore_optimizer(self):
        """Second half of load_from_checkpoint(resume_optimizer=True): call once the model sits on its device."""
        st = getattr(self, "_resume_state", None)
        if st is None:
            return False
        ost, sst = st
        if self.optimizer is None:
            self.configure_optimizers()
        opt, eng = self.optimizer, self.model._engine
        eng.ensure(self.model._dev())
        ar = eng.arena
        opt.lr, opt.betas, opt.eps, opt.wd = ost["lr"], tuple(ost["betas"]), ost["eps"], ost["weight_decay"]
        opt.step_count = int(ost["step"])
        if "exp_avg" in ost:
            # in place when the moment buffers exist: a step plan built earlier holds their addresses
            if ar.M is None:
                ar.M, ar.V = torch.zeros_like(ar.P), torch.zeros_like(ar.P)
            else:
                ar.M.zero_()
                ar.V.zero_()
            for k in ar.order:
                ar.M[ar.offset[k]: ar.offset[k] + ar.numel[k]].copy_(ost["exp_avg"][k].reshape(-1))
                ar.V[ar.offset[k]: ar.offset[k] + ar.numel[k]].copy_(ost["exp_avg_sq"][k].reshape(-1))
        if sst is not None:
            self.scheduler.best, self.scheduler.bad = sst["best"], sst["num_bad_epochs"]
        self._resume_state = None
        return True


def run_training_MMT(config, batches, num_epochs=None, log=print):
    """Epoch loop of MT_functions_v15_4.py:93-158: `batches` is a callable epoch -> iterable of collated
    batch dicts for THIS rank."""
    tm = TransformerMultiGPU(config)
    if getattr(config, "load_model", False) and getattr(config, "checkpoint_path", None):
        tm = TransformerMultiGPU.load_from_checkpoint(config.checkpoint_path, config=config)   # MT_functions_v15_4.py:121-123
    if getattr(config, "use_real_data", False):
        # MT_functions_v15_4.py:124-133: fine-tuning on real data trains real_data_linear only
        for k, p in tm.model.named_parameters():
            p.requires_grad = k.startswith("real_data_linear.")
    tm.to(config.device)
    tm.configure_optimizers()
    for epoch in range(num_epochs or config.num_epochs):
        tm.current_epoch = epoch
        losses = []
        for i, batch in enumerate(batches(epoch)):
            losses.append(tm.training_step(batch, i)['loss'])
            if config.save_model and (not dist.is_initialized() or dist.get_rank() == 0):
                tm.on_batch_end()
        mean = float(torch.stack(losses).mean()) if losses else float("nan")
        if dist.is_initialized():
            t = torch.tensor([mean], device=config.device)
            dist.all_reduce(t)
            mean = float(t) / dist.get_world_size()
        tm.scheduler.step(mean)
        if config.save_model and (not dist.is_initialized() or dist.get_rank() == 0):
            tm.on_epoch_end(epoch, mean)
        log(f"epoch {epoch} loss {mean:.4f} lr {tm.optimizer.lr:g}")
    return tm
