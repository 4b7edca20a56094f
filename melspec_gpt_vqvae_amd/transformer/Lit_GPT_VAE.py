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

    def training_step(self, batch, batch_idx):
        """reference :246-315 (fb in {0,1,2,3}); returns the batch-mean loss."""
        x = self.get_input(batch)
        a = self.args
        if a.beta == 0:
            self.kl_weight = a.beta
        else:
            self.kl_weight = min(1.0, self.kl_weight + self.anneal_rate)
        fb = getattr(a, "fb", 0)
        if a.beta == 0 or fb == 0:
            loss, loss_rc, loss_kl = self.loss(x, self.kl_weight, nsamples=a.nsamples)
        elif fb == 1:
            loss, loss_rc, loss_kl = self.loss(x, self.kl_weight, nsamples=a.nsamples)
            kl_mask = (loss_kl > a.target_kl).float()
            loss = loss_rc + kl_mask * self.kl_weight * loss_kl
        elif fb == 2:
            mu, logvar, _ = self.encoder(x)
            z = self.encoder.reparameterize(mu, logvar, a.nsamples)
            loss_kl = 0.5 * (mu.pow(2) + logvar.exp() - logvar - 1)
            kl_mask = (loss_kl > self.dim_target_kl).float()
            fake_loss_kl = (kl_mask * loss_kl).sum(dim=1)
            loss_rc = self.decoder.reconstruct_error(x, z).mean(dim=1)
            loss = loss_rc + self.kl_weight * fake_loss_kl
        else:
            loss, loss_rc, loss_kl = self.loss(x, self.kl_weight, nsamples=a.nsamples)
            kl_mask = (loss_kl.mean() > a.target_kl).float()
            loss = loss_rc + kl_mask * self.kl_weight * loss_kl
        loss = loss.mean(dim=-1)
        if pl is not None:
            self.log("train/loss", loss, prog_bar=True, on_step=True, on_epoch=True, sync_dist=True)
            self.log("train/kl_weight", self.kl_weight, prog_bar=True, on_step=True, on_epoch=True, sync_dist=True)
        return loss

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
