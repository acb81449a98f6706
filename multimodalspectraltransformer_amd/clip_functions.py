"""Contrastive (CLIP) training loop on the accelerated encoder half.

Counterpart of CLIPMultiGPU (utils_MMT/models_CLIP_v15_4.py:435-505), CLIP_make (:46-108) and run_training_CLIP
(utils_MMT/clip_functions_v15_4.py:276-329) without pytorch_lightning: one process per GPU, the loss of
CLIPModel.forward, and the reference's optimizer: AdamW over three parameter groups, the SMILES model at
config.CB_lr, the spectra model at config.MT_lr (both without weight decay) and the two projection heads at
config.MT_projection_lr with config.weight_decay, under ReduceLROnPlateau(factor 0.5, patience 2) on the epoch's
mean training loss.

The spectra model's group is the flat-buffer AdamW kernel of this package; the heads and the SMILES model are small
torch modules and keep torch.optim.AdamW.  The SMILES branch is the caller's module (ChemBERTa weights are not part
of the reference repository); its features enter the loss detached and the gradient at the features, returned by
CLIPModel.loss_and_backward, is pushed back through the caller's autograd graph.
"""
import json
import os
import random
import time

import torch
import torch.distributed as dist

from . import _lib as L
from . import helper_functions as hf
from .models_CLIP import CLIPModel
from .models_MMT import MultimodalTransformer
from .train import FusedAdamW, GradReducer

DECODER_SIDE = ("decoder.", "fc_out.", "embed_trg.", "pe_trg.", "real_data_linear.")


def clip_active_ranges(arena, mode):
    """[lo, hi) ranges of the flat buffers that receive a gradient from the contrastive loss: the encoders and
    embeddings of the present modalities, the cross encoder and fp1; nothing on the decoder side (those parameters
    keep grad None in the reference, so its AdamW never touches them)."""
    act = set(arena.active_keys(mode)) | {"fp1.weight", "fp1.bias"}
    act = {k for k in act if not k.startswith(DECODER_SIDE) and arena.named[k].requires_grad}
    ranges = []
    for i, k in enumerate(arena.order):
        if k not in act:
            continue
        lo = arena.offset[k]
        hi = arena.offset[arena.order[i + 1]] if i + 1 < len(arena.order) else arena.total
        if ranges and ranges[-1][1] == lo:
            ranges[-1][1] = hi
        else:
            ranges.append([lo, hi])
    return [tuple(r) for r in ranges]


def CLIP_make(config, stoi, stoi_MF, itos, CB_model=None, tokenizer=None, MT_model=None):
    """(CLIP_model, optimizer) as models_CLIP_v15_4.py:46-108.  CB_model / tokenizer: the SMILES branch and its
    tokenizer (the reference loads ChemBERTa from config.model_version and config.CB_model_path, files that are not
    in its repository: pass the loaded module in; None = features are handed to every step).  The spectra model is
    built from config and, when config.MT_model_path names a file, loaded from it (:67-69).  As in the reference the
    SMILES model is frozen except its LL1 layer (:74-79)."""
    if MT_model is None:
        MT_model = MultimodalTransformer(config, src_pad_idx=0)
        path = getattr(config, "MT_model_path", None)
        if path and os.path.exists(path):
            MT_model.load_state_dict(torch.load(path, map_location="cpu", weights_only=True))
    MT_model.to(config.device)
    if CB_model is not None:
        CB_model.to(config.device)
        for p in CB_model.parameters():
            p.requires_grad = False
        if hasattr(CB_model, "LL1"):
            for p in CB_model.LL1.parameters():
                p.requires_grad = True
    model = CLIPModel(CB_model, MT_model, tokenizer, itos, config)
    return model, CLIPOptimizer(model, config)


class CLIPOptimizer:
    """torch.optim.AdamW([CB @ CB_lr, MT @ MT_lr, heads @ MT_projection_lr, weight_decay], weight_decay=0.)
    (models_CLIP_v15_4.py:99-106).  `lr` is the plateau scheduler's handle: a factor applied to all three groups."""

    def __init__(self, model, config):
        self.model, self.config = model, config
        self.mt = FusedAdamW(model.MT_model, lr=float(config.MT_lr), weight_decay=0.0)
        groups = []
        cb = [p for p in model.CB_model.parameters() if p.requires_grad] if model.CB_model is not None else []
        if cb:
            groups.append({"params": cb, "lr": float(config.CB_lr)})
        heads = list(model.CB_projection.parameters()) + list(model.MT_projection.parameters())
        groups.append({"params": heads, "lr": float(config.MT_projection_lr), "weight_decay": float(config.weight_decay)})
        self.torch_opt = torch.optim.AdamW(groups, weight_decay=0.0)
        self._lr_scale = 1.0

    @property
    def lr(self):
        return self._lr_scale

    @lr.setter
    def lr(self, scale):
        ratio = scale / self._lr_scale
        self._lr_scale = scale
        self.mt.lr *= ratio
        for g in self.torch_opt.param_groups:
            g["lr"] *= ratio

    def zero_grad(self):
        eng = self.model.MT_model._engine
        if eng.arena is not None:
            eng.arena.G.zero_()
        self.torch_opt.zero_grad(set_to_none=True)

    def step(self, grad_scale=1.0):
        mtm = self.model.MT_model
        ar = mtm._engine.arena
        if grad_scale != 1.0:
            for g in self.torch_opt.param_groups:
                for p in g["params"]:
                    if p.grad is not None:
                        p.grad.mul_(grad_scale)
        self.mt.step(self.config.training_mode, grad_scale=grad_scale,
                     ranges=clip_active_ranges(ar, self.config.training_mode))
        self.torch_opt.step()


class CLIPMultiGPU:
    """Same role and attribute names as the reference's LightningModule wrapper (models_CLIP_v15_4.py:435-505)."""

    def __init__(self, config, CB_model=None, tokenizer=None, MT_model=None):
        self.start_time = time.time()
        self.config = config
        load = lambda p: json.load(open(p)) if p and os.path.exists(p) else None  # noqa: E731
        itos, stoi, stoi_MF = (load(getattr(config, k, None)) for k in ("itos_path", "stoi_path", "stoi_MF_path"))
        self.CLIP_model, self.optimizer = CLIP_make(config, stoi, stoi_MF, itos, CB_model, tokenizer, MT_model)
        self.scheduler = None
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.reducer = None
        self.current_epoch = 0
        self.global_step = 0
        self.last_train_loss = None

    def configure_optimizers(self):
        from .train import ReduceLROnPlateau
        self.scheduler = ReduceLROnPlateau(self.optimizer)    # mode='min', factor=0.5, patience=2 (:498)
        return self.optimizer

    def forward(self, x, CB_features=None):
        loss, _ = self.CLIP_model(x, CB_features)
        return loss

    def _features(self, x, CB_features):
        """SMILES-branch features with autograd history when that branch has trainable parameters."""
        if CB_features is not None:
            return CB_features
        if "CB_features" in x:
            return x["CB_features"]
        if self.CLIP_model.CB_model is None:
            raise L.MMSTError("no SMILES model was given: pass CB_features or put them into the batch")
        return self.CLIP_model._cb_features(x)

    def training_step(self, batch, batch_idx=0, CB_features=None, deterministic=False):
        """One optimisation step: zero gradients, loss and backward (models_CLIP_v15_4.py:456-461), gradient
        averaging over ranks, optimizer step.  Returns {'loss': device tensor}.  deterministic=True runs the step in
        eval mode (no dropout anywhere), for reproducible comparisons."""
        if self.scheduler is None:
            self.configure_optimizers()
        clip = self.CLIP_model
        mt = clip.MT_model
        eng = mt._engine
        clip.train(not deterministic)
        eng.step_seed = (eng.step_seed * 1664525 + 1013904223) & 0xFFFFFFFF
        eng.ensure(mt._dev())
        ar = eng.arena
        self.optimizer.zero_grad()
        feats = self._features(batch, CB_features)
        if self.reducer is None or self.reducer.flat is not ar.G:
            self.reducer = GradReducer(ar.G, ar.reduce_segments(), self.world,
                                       active=clip_active_ranges(ar, self.config.training_mode))
        # Small gradients outside the segments of the flat buffer travel packed, asynchronously: the two projection
        # heads and fp1 are final before the encoder backward starts, so their all-reduce is issued at its first
        # segment callback and runs beside it; what the caller's SMILES model receives follows after its backward.
        pending = []

        def reduce_packed(tensors):
            tensors = [t for t in tensors if t is not None]
            if not tensors:
                return
            flat = torch.cat([t.reshape(-1) for t in tensors])
            if flat.is_cuda and dist.get_backend() == "gloo":   # several ranks on one card (rehearsal): through the host
                host = flat.cpu()
                dist.all_reduce(host)
                flat.copy_(host)
                pending.append((None, flat, tensors))
            else:
                pending.append((dist.all_reduce(flat, async_op=True), flat, tensors))

        head_params = [q for m in (clip.CB_projection, clip.MT_projection) for q in m.parameters()]
        head_ids = {id(q) for q in head_params}
        on_segment = None
        if self.world > 1:
            def on_segment(name):
                if name == "decoder":   # nothing on the decoder side has a gradient: the first callback of the backward
                    reduce_packed([ar.grad_view("fp1.weight"), ar.grad_view("fp1.bias")] + [q.grad for q in head_params])
                else:
                    self.reducer.on_segment(name, also=(getattr(eng, "wstream", None),))
        loss, losses, dfeat = clip.loss_and_backward(batch, feats.detach(), on_segment=on_segment)
        if feats.requires_grad:
            feats.backward(dfeat)      # the caller's SMILES model: gradient of the mean loss at its output
        scale = 1.0
        if self.world > 1:
            reduce_packed([q.grad for g in self.optimizer.torch_opt.param_groups for q in g["params"]
                           if id(q) not in head_ids])
            scale = self.reducer.finish()
            for work, flat, tensors in pending:
                if work is not None:
                    work.wait()
                off = 0
                for t in tensors:
                    t.copy_(flat[off:off + t.numel()].view_as(t))
                    off += t.numel()
        self.optimizer.step(grad_scale=scale)
        self.global_step += 1
        self.last_train_loss = loss.clone()
        return {"loss": self.last_train_loss, "losses": losses}

    def validation_step(self, batch, batch_idx=0, CB_features=None):
        """models_CLIP_v15_4.py:464-469: the same loss on a held-out batch.  The reference's LightningModule runs it
        in eval mode without gradients."""
        self.CLIP_model.eval()
        with torch.no_grad():
            loss = self.forward(batch, self._features(batch, CB_features))
        return {"loss": loss.clone()}

    def _save(self, stem, loss):
        d = self.config.CLIP_model_save_dir
        os.makedirs(d, exist_ok=True)
        base = os.path.join(d, f"{stem}{loss:.3f}")
        sd = {k: v.detach().cpu() for k, v in self.CLIP_model.state_dict().items()}
        torch.save(sd, base + ".pth")                                            # :478, :491
        torch.save({"epoch": int(self.current_epoch), "global_step": int(self.global_step),
                    "state_dict": {"CLIP_model." + k: v for k, v in sd.items()}}, base + ".ckpt")   # Lightning layout
        return base + ".pth", base + ".ckpt"

    def on_epoch_end(self, epoch=None, loss=None):
        """Weights (.pth) and checkpoint (.ckpt) per epoch, models_CLIP_v15_4.py:472-478."""
        epoch = self.current_epoch if epoch is None else epoch
        loss = float(self.last_train_loss) if loss is None else loss
        return self._save(f"MultimodalCLIP_Epoch_{epoch}_Loss", loss)

    def on_batch_end(self):
        """Wall-clock saves every config.model_save_interval seconds, models_CLIP_v15_4.py:480-494."""
        interval = getattr(self.config, "model_save_interval", None)
        if interval is None or time.time() - self.start_time < interval:
            return None
        loss = float(self.last_train_loss) if self.last_train_loss is not None else float("nan")
        out = self._save(f"MultimodalCLIP__time_{str(time.time())}_Loss_", loss)
        self.start_time = time.time()
        return out

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, config=None, **kw):
        """.ckpt ('CLIP_model.'-prefixed state_dict, as Lightning writes the wrapper) or .pth (CLIPModel.state_dict());
        read with a loader that executes nothing from the file."""
        self = cls(config, **kw)
        sd = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
        if "state_dict" in sd:
            self.current_epoch = int(sd.get("epoch", 0))
            self.global_step = int(sd.get("global_step", 0))
            sd = {k[len("CLIP_model."):]: v for k, v in sd["state_dict"].items() if k.startswith("CLIP_model.")}
        self.CLIP_model.load_state_dict(sd)
        return self


def create_CLIP_batches(dataset, config, rank=0, world=1, rng=None, mask_float=True):
    """One epoch of collated batches of similar molecular weight (create_CLIP_dataloaders,
    clip_functions_v15_4.py:197-250: WeightSortedBatchSampler(batch_size=config.CLIP_batch_size, drop_last=False));
    with several ranks, rank r takes batches r, r + world, ... of the epoch's batch list."""
    from .dataloaders import weight_sorted_batches
    mw = list(dataset.ref_data["MW"]) if hasattr(dataset, "ref_data") else list(dataset.mol_weights)
    for i, idx in enumerate(weight_sorted_batches(mw, int(config.CLIP_batch_size), False, None, rng or random)):
        if i % world != rank:
            continue
        batch = hf.collate_fn([dataset[j] for j in idx])
        if mask_float:
            for k in ("mask_MF", "mask_MS"):
                if k in batch:
                    batch[k] = batch[k].float()
        yield batch


def run_training_CLIP(train_batches, test_batches, config, CB_model=None, tokenizer=None, MT_model=None,
                      num_epochs=None, log=print):
    """Epoch loop of clip_functions_v15_4.py:276-329.  train_batches / test_batches: callables epoch -> iterable of
    collated batch dicts of THIS rank (test_batches may be None).  Writes the configuration next to the models (:281-287),
    resumes from config.CLIP_model_path when config.CLIP_continue_training (:304-306), saves per epoch and on the
    wall-clock interval, steps the plateau scheduler on the epoch's mean training loss (:498-504), and like the
    reference keeps a last_backup_checkpoint when training raises (:325-329)."""
    os.makedirs(config.CLIP_model_save_dir, exist_ok=True)
    lead = not dist.is_initialized() or dist.get_rank() == 0
    if lead:
        plain = {k: v for k, v in vars(config).items() if isinstance(v, (int, float, str, bool, list, type(None)))}
        with open(os.path.join(config.CLIP_model_save_dir, f"config_{random.randint(0, 10000)}.json"), "w") as f:
            json.dump(plain, f, indent=4)
    path = getattr(config, "CLIP_model_path", None)
    if getattr(config, "CLIP_continue_training", False) and path and os.path.exists(path):
        tm = CLIPMultiGPU.load_from_checkpoint(path, config=config, CB_model=CB_model, tokenizer=tokenizer, MT_model=MT_model)
    else:
        tm = CLIPMultiGPU(config, CB_model, tokenizer, MT_model)
    tm.configure_optimizers()
    try:
        for epoch in range(num_epochs or int(config.CLIP_NUM_EPOCHS)):
            tm.current_epoch = epoch
            losses = []
            for i, batch in enumerate(train_batches(epoch)):
                losses.append(tm.training_step(batch, i)["loss"])
                if lead:
                    tm.on_batch_end()
            mean = float(torch.stack(losses).mean()) if losses else float("nan")
            if dist.is_initialized():
                t = torch.tensor([mean], device=config.device)
                dist.all_reduce(t)
                mean = float(t) / dist.get_world_size()
            test = [float(tm.validation_step(b, i)["loss"]) for i, b in enumerate(test_batches(epoch))] if test_batches else []
            tm.scheduler.step(mean)
            if lead:
                tm.on_epoch_end(epoch, mean)
            test_txt = f" test_loss {sum(test) / len(test):.4f}" if test else ""
            log(f"epoch {epoch} train_loss {mean:.4f}{test_txt} lr x{tm.optimizer.lr:g}")
    except Exception as e:   # noqa: BLE001 -- the reference's behaviour: report, keep a backup, go on
        log(f"Error occurred: {e}")
        if lead:
            sd = {"CLIP_model." + k: v.detach().cpu() for k, v in tm.CLIP_model.state_dict().items()}
            torch.save({"epoch": tm.current_epoch, "global_step": tm.global_step, "state_dict": sd},
                       os.path.join(config.CLIP_model_save_dir, "last_backup_checkpoint.ckpt"))
            log("Model saved.")
    return tm
