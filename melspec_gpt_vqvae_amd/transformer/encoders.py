"""GPT encoder of the GPT-VAE on the HIP kernels - mirror of the reference's transformer/encoders.py
(GPTEncoder :11-104): an unmasked GPT whose last position yields (mu, logvar); reparameterisation and the KL term
run in one kernel (melgpt_vae_reparam_fwd / _bwd)."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .. import _ffi
from .minGPT import GPT, _Seeds


class _ReparamFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, stats, eps, nsamples, seed):
        B, two_nz = stats.shape
        nz = two_nz // 2
        st = stats.float().contiguous()
        gen = eps is None
        e = torch.empty(B, nsamples, nz, dtype=torch.float32, device=stats.device) if gen else eps.float().contiguous()
        z = torch.empty(B, nsamples, nz, dtype=torch.float32, device=stats.device)
        kl = torch.empty(B, dtype=torch.float32, device=stats.device)
        _ffi.call("melgpt_vae_reparam_fwd", _ffi.ptr(st), _ffi.ptr(e), int(gen), int(seed), B, nsamples, nz, _ffi.ptr(z),
                  _ffi.ptr(kl), _ffi.stream())
        ctx.save_for_backward(st, e)
        ctx.dims = (B, nsamples, nz)
        return z, kl

    @staticmethod
    def backward(ctx, dz, dkl):
        st, e = ctx.saved_tensors
        B, ns, nz = ctx.dims
        d = torch.empty_like(st)
        dz = dz.float().contiguous() if dz is not None else None
        dkl = dkl.float().contiguous() if dkl is not None else None
        _ffi.call("melgpt_vae_reparam_bwd", _ffi.ptr(st), _ffi.ptr(e), _ffi.ptr(dz), _ffi.ptr(dkl), B, ns, nz, _ffi.ptr(d),
                  _ffi.stream())
        return d, None, None, None


class GPTEncoder(nn.Module):
    """GPT encoder with constant-length data (reference :11-19)."""

    def __init__(self, args, embd_pdrop=0., resid_pdrop=0., attn_pdrop=0., n_unmasked=0, last_linear=None,
                 block_size=None):
        super().__init__()
        self.args = args
        self.transformer = GPT(self.args, embd_pdrop=embd_pdrop, resid_pdrop=resid_pdrop, attn_pdrop=attn_pdrop,
                               n_unmasked=n_unmasked, last_linear=last_linear, block_size=block_size)

    def _stats(self, input):
        logits, _, att = self.transformer.forward(input)
        return logits[:, -1, :], att

    def forward(self, input):
        """-> (mean (B,nz), logvar (B,nz), attention) - reference :21-42."""
        last_state, att = self._stats(input)
        mean, logvar = last_state.chunk(2, -1)
        if getattr(self.args, "fix_var", 0) > 0:
            logvar = mean.new_tensor([[[math.log(self.args.fix_var)]]]).expand_as(mean)
        return mean, logvar, att

    def encode_stats(self, x):
        return self.forward(x)

    def reparameterize(self, mu, logvar, nsamples=1, eps=None):
        """z = mu + eps*exp(logvar/2), eps ~ N(0,1) drawn in-kernel unless given (reference :81-104)."""
        z, _ = _ReparamFn.apply(torch.cat((mu, logvar), -1), eps, nsamples, _Seeds.next())
        return z

    def sample(self, input, nsamples):
        mu, logvar, att = self.forward(input)
        return self.reparameterize(mu, logvar, nsamples), (mu, logvar), att

    def encode(self, input, nsamples, eps=None):
        """-> (z (B,nsamples,nz), KL (B,)) - reference :62-79."""
        last_state, _ = self._stats(input)
        if getattr(self.args, "fix_var", 0) > 0:
            mu, _ = last_state.chunk(2, -1)
            last_state = torch.cat((mu, mu.new_full(mu.shape, math.log(self.args.fix_var))), -1)
        return _ReparamFn.apply(last_state, eps, nsamples, _Seeds.next())
