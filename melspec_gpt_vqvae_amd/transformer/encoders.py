"""GPT encoder of the GPT-VAE on the HIP kernels - mirror of the reference's transformer/encoders.py
(GPTEncoder :11-170): an unmasked GPT whose last position yields (mu, logvar); reparameterisation and the KL term
run in one kernel (melgpt_vae_reparam_fwd / _bwd); the eval-time analytics log q(z|x) and the mutual-information
estimate are one launch each (melgpt_gauss_log_density, melgpt_vae_calc_mi)."""
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

    @staticmethod
    def _f32(v):
        return v.detach().float().contiguous()

    @torch.no_grad()
    def eval_inference_dist(self, x, z, param=None):
        """log q(z|x) of z (batch, nsamples, nz) under the row's own posterior -> (batch, nsamples); `param` =
        (mu, logvar) skips the encoder pass - reference :106-134."""
        if not param:
            mu, logvar, _ = self.forward(x)
        else:
            mu, logvar = param
        mu, logvar, z = self._f32(mu), self._f32(logvar), self._f32(z)
        B, S, nz = z.shape
        assert mu.shape == (B, nz) and logvar.shape == (B, nz)
        out = torch.empty(B, S, dtype=torch.float32, device=z.device)
        _ffi.call("melgpt_gauss_log_density", _ffi.ptr(z), _ffi.ptr(mu), _ffi.ptr(logvar), nz, B, S, nz, 0, _ffi.ptr(out),
                  _ffi.stream())
        return out

    @torch.no_grad()
    def calc_mi(self, x, eps=None):
        """I(x, z) ~ E_x E_q(z|x) log q(z|x) - E_x E_q(z|x) log q(z) -> Python float - reference :136-170.  The one
        reparameterised draw per row uses `eps` (batch, 1, nz) when given, else noise drawn in-kernel."""
        mu, logvar, _ = self.forward(x)
        mu, logvar = self._f32(mu), self._f32(logvar)
        B, nz = mu.shape
        gen = eps is None
        e = torch.empty(B, nz, dtype=torch.float32, device=mu.device) if gen else self._f32(eps).reshape(B, nz)
        ws = torch.empty(B * nz + B, dtype=torch.float32, device=mu.device)
        mi = torch.empty(1, dtype=torch.float32, device=mu.device)
        _ffi.call("melgpt_vae_calc_mi", _ffi.ptr(mu), _ffi.ptr(logvar), nz, _ffi.ptr(e), int(gen), int(_Seeds.next()), B,
                  nz, _ffi.ptr(ws), _ffi.ptr(mi), _ffi.stream())
        return mi.item()
