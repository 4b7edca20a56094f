"""The thin training loop that stands where `pl.Trainer.fit` stands in the reference's entry scripts
(GPT_train.py:99-131, GPT_VAE_train.py:166-205): epochs over the module's own train / val dataloaders, the module's
training_step / validation_step, AdamW with the reference's grouping, data parallelism as one process per GPU
(RANK / WORLD_SIZE from the launcher; gradients exchanged by dp.DataParallel over RCCL), Lightning-format
checkpoints (`{"state_dict", "epoch", "global_step", ...}`: last + best by validation loss).  TensorBoard image /
audio logging callbacks of the reference are outside the hot path and not reproduced."""
from __future__ import annotations

import os
import time

import torch
import torch.distributed as dist


def _to_device(batch, device):
    return {k: (v.to(device, non_blocking=True) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}


class Fit:
    def __init__(self, module, args, device=None, fused_optimizer=True, log=print):
        self.module, self.args, self.log = module, args, log
        self.device = torch.device(device or args.device)
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.history = {"train_loss": [], "val_loss": [], "steps": 0, "epochs": 0}
        self.global_step = 0
        self.best_val = float("inf")
        self.dp = None
        module.to(self.device)
        # what is trained = what the reference's configure_optimizers walks: `self.transformer` of a Lit_minGPT (a loaded
        # first_stage_model stays frozen, minGPT.py:632), the whole GPT_VAE (Lit_GPT_VAE.py:908).  The flat store, the
        # gradient exchange and the optimizer all cover exactly this sub-tree, at every world size.
        target = self.target = module.transformer if hasattr(module, "transformer") else module
        if self.world > 1:
            from .dp import DataParallel

            assert dist.is_initialized(), "initialise torch.distributed (backend nccl = RCCL) before Fit under WORLD_SIZE > 1"
            if not (fused_optimizer and self.device.type == "cuda"):
                raise RuntimeError("data-parallel training uses the fused optimizer (it folds the 1/world averaging in)")
            self.dp = DataParallel(target)
            self.dp.broadcast_parameters(0)     # C1 of SURVEY 2b: ranks start from rank 0's weights, whatever their seeds
            if getattr(module, "data", None) is not None:
                module.data.rank, module.data.world = self.rank, self.world
        # fp16 flavour of the library: a 5-bit exponent - the mean cross entropy's dlogits (1e-7 .. 3e-5 at 128 x 265
        # tokens) are subnormal or zero in fp16 - so the loss is scaled before backward and the optimizer's grad_scale
        # takes the factor out again; a step whose gradient is not finite is skipped and the scale halved (dynamic loss
        # scaling).  bf16 / f32 lanes: scale 1, no check.
        from . import _ffi
        self.fp16 = getattr(args, "dtype", "f32") == "fp16" and _ffi.HALF == "fp16"
        self.loss_scale = 4096.0 if self.fp16 else 1.0
        self._good_steps = 0
        self.skipped_steps = 0
        if fused_optimizer and self.device.type == "cuda":
            from .optim import FusedAdamW

            self.opt = FusedAdamW(target, lr=args.learning_rate, betas=(0.9, 0.95), weight_decay=0.01)
            self.opt.grad_scale = 1.0 / (self.world * self.loss_scale)
        else:
            if self.fp16:
                raise RuntimeError("the fp16 lane needs the fused optimizer (loss scaling lives in its grad_scale)")
            self.opt = module.configure_optimizers()

    # ------------------------------------------------------------------------------------------------ one epoch
    def train_epoch(self, epoch, max_steps=None):
        m = self.module
        m.train()
        if getattr(m, "data", None) is not None:
            m.data.set_epoch(epoch)
        losses = []
        t0 = time.perf_counter()
        for i, batch in enumerate(m.train_dataloader()):
            if max_steps is not None and i >= max_steps:
                break
            batch = _to_device(batch, self.device)
            loss = m.training_step(batch, i)
            self.opt.zero_grad()
            (loss * self.loss_scale if self.loss_scale != 1.0 else loss).backward()
            if self.dp is not None:
                self.dp.finish()
            if self._step_ok():
                self.opt.step()
            self._grow_loss_scale()
            self.global_step += 1
            losses.append(loss.detach())
            freq = int(getattr(self.args, "logging_frequency", 200) or 200)
            if self.global_step % freq == 0:
                # the step's logged scalars (Lit_GPT_VAE.py:310-313: four `self.log(..., sync_dist=True)` = four
                # all-reduces per step under DDP) as ONE all-reduce, and only on logging steps
                names = list(getattr(m, "last_metrics", {}) or {"train/loss": losses[-1]})
                vals = [getattr(m, "last_metrics", {}).get(k, losses[-1]) for k in names]
                if self.dp is not None:
                    vals = self.dp.reduce_metrics(*vals)
                if self.rank == 0:
                    txt = "  ".join(f"{k} {float(v):.4f}" for k, v in zip(names, vals))
                    self.log(f"epoch {epoch} step {self.global_step}  {txt}  "
                             f"({(time.perf_counter() - t0) / (i + 1):.3f} s/step)")
        out = torch.stack(losses).float() if losses else torch.zeros(0)
        self.history["train_loss"].append([float(v) for v in out.cpu()])
        self.history["steps"] = self.global_step
        return out

    def _step_ok(self):
        """fp16 lane only: is the (scaled, reduced) gradient finite?  One fused sum over the flat buffer and one host
        read per step; an overflow skips the step and halves the scale at once (the step is not taken, so nothing sees the
        new factor with the old gradient); 200 clean steps double it (up to 65 536) - but only AFTER `opt.step()` has
        consumed this backward's gradient with the grad_scale that matches the scale it was produced at
        (`_grow_loss_scale`)."""
        if not self.fp16:
            return True
        from . import ops
        from .flat import ensure_flat

        fp = ensure_flat(self.target)
        fp.zero_missing_grads()
        ok = bool(torch.isfinite(ops.sum_f32(fp.grad)).item())
        if ok:
            self._good_steps += 1
        else:
            self.skipped_steps += 1
            self._good_steps = 0
            self.loss_scale = max(1.0, self.loss_scale / 2.0)
            self.opt.grad_scale = 1.0 / (self.world * self.loss_scale)
        return ok

    def _grow_loss_scale(self):
        """fp16 lane, after the optimizer step: 200 clean steps in a row double the loss scale for the NEXT backward."""
        if self.fp16 and self._good_steps >= 200 and self.loss_scale < 65536.0:
            self.loss_scale *= 2.0
            self._good_steps = 0
            self.opt.grad_scale = 1.0 / (self.world * self.loss_scale)

    @torch.no_grad()
    def validate(self):
        m = self.module
        loader = m.val_dataloader() if hasattr(m, "val_dataloader") else None
        if loader is None:
            return None
        m.eval()
        outs = []
        for i, batch in enumerate(loader):
            outs.append(m.validation_step(_to_device(batch, self.device), i))
        if not outs:
            return None
        if isinstance(outs[0], dict):          # GPT_VAE: summed ELBO per batch (Lit_GPT_VAE.py:361)
            m.validation_epoch_end(outs)
            val = float(m.test_loss)
        else:
            val = float(torch.stack([o.detach().float() for o in outs]).mean())
        if self.dp is not None:
            (v,) = self.dp.reduce_metrics(val)
            val = float(v)
        self.history["val_loss"].append(val)
        return val

    # ------------------------------------------------------------------------------------------------ checkpoints
    def checkpoint_dir(self):
        a = self.args
        return os.path.join(getattr(a, "log_root", "lightning_logs"), f"{a.experiment}-{a.dataset}", "checkpoints")

    def save(self, name, epoch):
        if self.dp is not None:
            self.dp.check()      # every rank: the last step's has-gradient verdict is drained BEFORE a checkpoint is written
        if self.rank != 0:
            return None
        d = self.checkpoint_dir()
        os.makedirs(d, exist_ok=True)
        ck = {"state_dict": {k: v.detach().cpu().clone() for k, v in self.module.state_dict().items()}, "epoch": epoch,
              "global_step": self.global_step, "best_val_loss": self.best_val}
        if hasattr(self.opt, "state_dict"):
            # torch.optim.AdamW's layout from both optimizers (FusedAdamW.state_dict writes it per parameter, in the
            # reference's group order): what Lightning stores under `optimizer_states`
            sd = self.opt.state_dict()
            sd["state"] = {i: {k: (v.detach().cpu() if isinstance(v, torch.Tensor) else v) for k, v in st.items()}
                           for i, st in sd["state"].items()}
            ck["optimizer_states"] = [sd]
        if hasattr(self.module, "on_save_checkpoint"):
            self.module.on_save_checkpoint(ck)
        path = os.path.join(d, name)
        torch.save(ck, path)
        return path

    def resume(self, path):
        ck = torch.load(path, map_location="cpu", weights_only=False)
        res = self.module.load_state_dict(ck["state_dict"], strict=False)
        # a checkpoint that does not fit must not resume from random weights: only the constant `attn.mask` buffers
        # (checkpoint ABI, never read) and a frozen first-stage VQ-VAE / vocoder the file did not carry may be absent
        optional = ("first_stage_model.", "vocoder.")
        missing = [k for k in res.missing_keys if not k.endswith("attn.mask") and not k.startswith(optional)]
        unexpected = [k for k in res.unexpected_keys if not k.startswith(optional)]
        if missing or unexpected:
            raise RuntimeError(f"{path} does not fit {type(self.module).__name__}: missing {missing[:6]} "
                               f"({len(missing)}), unexpected {unexpected[:6]} ({len(unexpected)})")
        self.global_step = int(ck.get("global_step", 0))
        self.best_val = float(ck.get("best_val_loss", float("inf")))
        st = ck.get("optimizer_states")
        if st:
            self.opt.load_state_dict(st[0])     # torch.optim.AdamW's layout (ours and the reference's), or the old flat one
        if self.dp is not None:
            self.dp.fp.generation += 1          # load_state_dict wrote the parameters in place: refresh shadow and caches
        if hasattr(self.module, "on_load_checkpoint") and "kl_weight" in ck:
            self.module.on_load_checkpoint(ck)
        return int(ck.get("epoch", -1)) + 1

    # ------------------------------------------------------------------------------------------------ fit
    def fit(self, epochs=None, max_steps_per_epoch=None, ckpt_path=None):
        first = self.resume(ckpt_path) if ckpt_path else 0
        epochs = int(epochs if epochs is not None else self.args.epochs)
        for epoch in range(first, epochs):
            self.train_epoch(epoch, max_steps_per_epoch)
            val = self.validate()
            self.history["epochs"] = epoch + 1
            self.save("last.ckpt", epoch)
            if val is not None and val < self.best_val:
                self.best_val = val
                self.save(f"{self.args.dataset}-model-epoch={epoch:02d}-loss={val:.2f}.ckpt", epoch)
            if self.rank == 0:
                tl = self.history["train_loss"][-1]
                self.log(f"epoch {epoch}: train/loss {sum(tl) / max(len(tl), 1):.4f}"
                         + (f"  val/loss {val:.4f}" if val is not None else ""))
        return self.history
