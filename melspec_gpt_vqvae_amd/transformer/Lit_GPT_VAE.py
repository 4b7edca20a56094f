"""GPT-VAE training logic on the HIP kernels - the hot-path half of the reference's transformer/Lit_GPT_VAE.py
(GPT_VAE.__init__ :25-89, encode/decode, loss :176-195, get_input :229-240, training_step :246-315,
GPT_configure_optimizers :895-943, checkpoint extras :959-971).  The MI / AU analytics, TensorBoard logging and
dataset plumbing of that file are outside the path."""
from __future__ import annotations

import numpy as np
import torch

from .. import ops
from .decoders import GPTDecoder
from .encoders import GPTEncoder
from .minGPT import _LitBase, make_adamw, pl


class GPT_VAE(_LitBase):
    def __init__(self, args):
        super().__init__()
        self.args = args
        self.len_train_data = getattr(args, "len_train_data", 0)
        self.data = None
        self.datamodule_loader()          # reference :33-37: the anneal rate below needs len(train data)
        self.encoder = GPTEncoder(args, n_unmasked=self.args.block_size, last_linear=self.args.n_embd * 2)
        self.decoder = GPTDecoder(args, embd_pdrop=args.embd_pdrop, resid_pdrop=args.resid_pdrop,
                                  attn_pdrop=args.attn_pdrop, block_size=self.args.block_size + 1)
        self.ns = 2
        self.best_loss = 1e4
        self.pre_mi = 0
        self.kl_weight = self.args.kl_start
        if getattr(self.args, "warm_up", 0) > 0 and self.len_train_data > 0:
            self.anneal_rate = (1.0 - self.args.kl_start) / (self.args.warm_up * (self.len_train_data / self.args.batch_size))
        else:
            self.anneal_rate = 0
        self.dim_target_kl = getattr(self.args, "target_kl", 0.0) / float(self.args.n_embd)
        self.forward_shuffle_idx, self.backward_shuffle_idx = self.make_idx(5, 53)

    def datamodule_loader(self):
        """reference :945-957 (same DataModule as Lit_minGPT); absent data path -> len_train_data comes from args."""
        if getattr(self.args, "spec_dir_path", None) and getattr(self.args, "load_data", True):
            from ..datasets import DataModule

            a = self.args
            kw = {"splits_path": a.splits_dir} if hasattr(a, "splits_dir") and "vggsound" in a.spec_dir_path else \
                ({"splits_dir": a.splits_dir} if hasattr(a, "splits_dir") else {})
            if hasattr(a, "meta_path"):
                kw["meta_path"] = a.meta_path
            self.data = DataModule(batch_size=a.batch_size, spec_dir_path=a.spec_dir_path, mel_num=80, spec_len=860,
                                   spec_crop_len=848, random_crop=False, num_workers=getattr(a, "workers", 0), **kw)
            self.data.setup()
            self.len_train_data = len(self.data.train_dataset)

    def train_dataloader(self):
        return self.data.train_dataloader() if self.data is not None else None

    def val_dataloader(self):
        return self.data.val_dataloader() if self.data is not None else None

    def test_dataloader(self):
        return self.data.test_dataloader() if self.data is not None and self.data.test_dataset is not None else None

    def encode(self, x, nsamples=1, eps=None):
        return self.encoder.encode(x, nsamples, eps=eps)

    def loss(self, x, kl_weight, nsamples=1, eps=None):
        """-> (total (B,), reconstruction (B,), KL (B,)) - reference :176-195."""
        z, KL = self.encode(x, nsamples, eps=eps)
        reconstruct_err = self.decoder.reconstruct_error(x, z).mean(dim=1)
        return reconstruct_err + kl_weight * KL, reconstruct_err, KL

    def make_idx(self, H, W):
        idx = np.arange(H * W).reshape(H, W).T
        idx = torch.tensor(idx.ravel())
        return idx, torch.argsort(idx)

    def get_input(self, batch):
        """(B,5,53) codes -> (B,265) time-major - reference :229-240."""
        x = batch['codes'].to(self.args.device)
        return ops.codes_permute(x, x.shape[1], x.shape[2])

    def training_step(self, batch, batch_idx, eps=None):
        """reference :246-315 (fb in {0,1,2,3}); returns the batch-mean loss.  `eps` (B, nsamples, nz) replaces the
        in-kernel N(0,1) draw of the reparameterisation (parity tests replay the reference's noise through it)."""
        x = self.get_input(batch)
        a = self.args
        if a.beta == 0:
            self.kl_weight = a.beta
        else:
            self.kl_weight = min(1.0, self.kl_weight + self.anneal_rate)
        fb = getattr(a, "fb", 0)
        if a.beta == 0 or fb == 0:
            loss, loss_rc, loss_kl = self.loss(x, self.kl_weight, nsamples=a.nsamples, eps=eps)
        elif fb == 1:      # free bits per sequence: the KL term counts only where it exceeds target_kl
            loss, loss_rc, loss_kl = self.loss(x, self.kl_weight, nsamples=a.nsamples, eps=eps)
            kl_mask = (loss_kl > a.target_kl).float()
            loss = loss_rc + kl_mask * self.kl_weight * loss_kl
        elif fb == 2:      # free bits per latent dimension
            mu, logvar, _ = self.encoder(x)
            z = self.encoder.reparameterize(mu, logvar, a.nsamples, eps=eps)
            loss_kl = 0.5 * (mu.pow(2) + logvar.exp() - logvar - 1)
            kl_mask = (loss_kl > self.dim_target_kl).float()
            fake_loss_kl = (kl_mask * loss_kl).sum(dim=1)
            loss_rc = self.decoder.reconstruct_error(x, z).mean(dim=1)
            loss = loss_rc + self.kl_weight * fake_loss_kl
        else:              # fb == 3: one switch for the whole batch, on the batch-mean KL
            loss, loss_rc, loss_kl = self.loss(x, self.kl_weight, nsamples=a.nsamples, eps=eps)
            kl_mask = (loss_kl.mean() > a.target_kl).float()
            loss = loss_rc + kl_mask * self.kl_weight * loss_kl
        loss = loss.mean(dim=-1)
        B = x.size(0)
        # the four scalars the reference logs with sync_dist=True (:310-313) - kept as device tensors (no .item()
        # sync in the step); under data parallelism dp.DataParallel.reduce_metrics averages them in ONE all-reduce
        rc_per_sent, kl_per_sent = loss_rc.detach().sum() / B, loss_kl.detach().sum() / B
        report = loss.detach() if a.beta == 0 else rc_per_sent + kl_per_sent
        self.last_metrics = {"train/loss": report, "train/loss_rc": rc_per_sent, "train/loss_kl": kl_per_sent,
                             "train/kl_weight": self.kl_weight}
        if pl is not None:
            for k, v in self.last_metrics.items():
                self.log(k, v, prog_bar=True, logger=True, on_step=True, on_epoch=True, sync_dist=True)
        return loss

    @torch.no_grad()
    def validation_step(self, batch, batch_idx, eps=None):
        """reference :321-361: the ELBO at KL weight 1.0 (the annealed weight only when beta == 0), SUMMED over the
        batch; returns the dict validation_epoch_end accumulates."""
        x = self.get_input(batch)
        a = self.args
        B, T = x.size()
        kl_w = self.kl_weight if a.beta == 0 else 1.0
        loss, loss_rc, loss_kl = self.loss(x, kl_w, nsamples=a.nsamples, eps=eps)
        loss, loss_rc, loss_kl = loss.sum(), loss_rc.sum(), loss_kl.sum()
        if a.beta != 0 and getattr(a, "warm_up", 0) == 0 and a.kl_start < 1e-6:
            loss = loss_rc       # pure autoencoder runs are selected on the reconstruction term alone (:346-348)
        report = loss / B
        self.last_metrics = {"loss": report, "val/loss": report, "val/loss_rc": loss_rc / B, "val/loss_kl": loss_kl / B}
        if pl is not None:
            for k, v in self.last_metrics.items():
                self.log(k, v, prog_bar=k != "loss", logger=k != "loss", on_step=True, on_epoch=True, sync_dist=True)
        return {"val_loss": loss, "val_loss_rc": loss_rc, "val_loss_kl": loss_kl, "report_num_words": (T - 1) * B,
                "report_num_sents": B}

    def validation_epoch_end(self, validation_step_outputs):
        """reference :363-386: epoch totals -> test_loss, nll, kl_loss, rec_loss, ppl."""
        tot = {k: 0 for k in ("val_loss", "val_loss_rc", "val_loss_kl", "report_num_words", "report_num_sents")}
        for out in validation_step_outputs:
            for k in tot:
                tot[k] = tot[k] + out[k]
        n_sents, n_words = tot["report_num_sents"], tot["report_num_words"]
        self.test_loss = tot["val_loss"] / n_sents
        self.kl_loss = tot["val_loss_kl"] / n_sents
        self.rec_loss = tot["val_loss_rc"] / n_sents
        self.nll = self.kl_loss + self.rec_loss
        self.ppl = torch.exp(torch.as_tensor(self.nll * n_sents / n_words))

    def GPT_configure_optimizers(self):
        return make_adamw(self, self.args.learning_rate)

    def configure_optimizers(self):
        return self.GPT_configure_optimizers()

    def on_save_checkpoint(self, checkpoint):
        checkpoint["kl_weight"] = self.kl_weight
        checkpoint["best_loss"] = self.best_loss
        checkpoint["pre_mi"] = self.pre_mi

    def on_load_checkpoint(self, checkpoint):
        self.kl_weight = checkpoint["kl_weight"]
        self.best_loss = checkpoint["best_loss"]
        self.pre_mi = checkpoint["pre_mi"]
