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
        if self.world > 1:
            from .dp import DataParallel

            assert dist.is_initialized(), "initialise torch.distributed (backend nccl = RCCL) before Fit under WORLD_SIZE > 1"
            self.dp = DataParallel(module)
            if getattr(module, "data", None) is not None:
                module.data.rank, module.data.world = self.rank, self.world
        target = module.transformer if hasattr(module, "transformer") else module
        if fused_optimizer and self.device.type == "cuda":
            from .optim import FusedAdamW

            self.opt = FusedAdamW(target, lr=args.learning_rate, betas=(0.9, 0.95), weight_decay=0.01)
            self.opt.grad_scale = 1.0 / self.world
        else:
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
            loss.backward()
            if self.dp is not None:
                self.dp.finish()
            self.opt.step()
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
        if self.rank != 0:
            return None
        d = self.checkpoint_dir()
        os.makedirs(d, exist_ok=True)
        ck = {"state_dict": {k: v.detach().cpu().clone() for k, v in self.module.state_dict().items()}, "epoch": epoch,
              "global_step": self.global_step, "best_val_loss": self.best_val}
        if hasattr(self.opt, "state_dict"):
            sd = self.opt.state_dict()
            ck["optimizer_states"] = [{k: (v.detach().cpu().clone() if isinstance(v, torch.Tensor) else v)
                                       for k, v in sd.items()}] if "exp_avg" in sd else [sd]
        if hasattr(self.module, "on_save_checkpoint"):
            self.module.on_save_checkpoint(ck)
        path = os.path.join(d, name)
        torch.save(ck, path)
        return path

    def resume(self, path):
        ck = torch.load(path, map_location="cpu", weights_only=False)
        self.module.load_state_dict(ck["state_dict"], strict=False)
        self.global_step = int(ck.get("global_step", 0))
        self.best_val = float(ck.get("best_val_loss", float("inf")))
        st = ck.get("optimizer_states")
        if st and "exp_avg" in st[0] and hasattr(self.opt, "_state"):
            self.opt._state()
            self.opt.load_state_dict({k: (v.to(self.device) if isinstance(v, torch.Tensor) else v) for k, v in st[0].items()})
        elif st:
            self.opt.load_state_dict(st[0])
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
