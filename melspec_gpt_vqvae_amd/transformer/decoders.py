"""GPT decoder of the GPT-VAE on the HIP kernels - mirror of the reference's transformer/decoders.py
(GPTDecoder :10-123): a GPT whose first position is the latent z; per-token cross entropy summed per sequence."""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from .. import ops
from .minGPT import GPT, _Seeds, sequence_cross_entropy


class GPTDecoder(nn.Module):
    def __init__(self, args, embd_pdrop=0., resid_pdrop=0., attn_pdrop=0., n_unmasked=0, last_linear=None,
                 block_size=None):
        super().__init__()
        self.args = args
        self.transformer = GPT(self.args, embd_pdrop=embd_pdrop, resid_pdrop=resid_pdrop, attn_pdrop=attn_pdrop,
                               n_unmasked=n_unmasked, last_linear=last_linear, block_size=block_size)

    def forward(self, x, c=None):
        """reference :23-38: logits for p(x_i | x_<i, z); c = z (B, n_cond, C)."""
        with self.transformer.discard_att():
            logits, _, _ = self.transformer(x[:, :-1], c)
        cond_size = c.size(-2)
        return logits[:, cond_size - 1:], x

    def reconstruct_error(self, x, z):
        """-> (batch, n_sample) summed token NLL - reference :40-68 (n_sample == 1 on this path: z.size(1) is the
        number of conditioning positions)."""
        batch_size, seq_len = x.size()
        n_sample = z.size(1)
        if n_sample != 1:
            raise NotImplementedError("n_sample > 1 feeds z as several conditioning positions in the reference; "
                                      "the GPT-VAE configs use nsamples = 1")
        output_logits, tgt = self(x, z)
        loss = sequence_cross_entropy(output_logits, tgt)
        return loss.view(batch_size, n_sample)

    def log_probability(self, x, z):
        return -self.reconstruct_error(x, z)

    def top_k_logits(self, logits, k):
        v, ix = torch.topk(logits, k)
        out = logits.clone()
        out[out < v[..., [-1]]] = -float('Inf')
        return out

    @torch.no_grad()
    def sample(self, x, c, steps, temperature=1.0, sample=False, top_k=None, callback=lambda k: None, kv_cache=True):
        """reference :89-123.  kv_cache=True (default): one position per step through GPT.decode_step (the latent
        `c` (B,1,C) is the first position); kv_cache=False: the reference's full re-forward per step."""
        block_size = self.transformer.get_block_size()
        assert not self.transformer.training
        seed = _Seeds.next()
        att = None
        if kv_cache and steps > 0 and self.transformer.kv_cacheable():
            tr = self.transformer
            cond_size = c.size(-2)
            assert cond_size == 1 and x.size(1) + cond_size + steps - 1 <= block_size
            cache = tr.decode_begin(x.size(0))
            logits = tr.decode_step(cache, embeddings=c)
            for j in range(x.size(1)):
                logits = tr.decode_step(cache, idx=x[:, j:j + 1])
            n_prompt = x.size(1)
            use_graph = steps >= 4 and os.environ.get('MELGPT_DECODE_GRAPH', '1') != '0'
            for k in range(steps):
                callback(k)
                ix = ops.sample_logits(logits, temperature=temperature, top_k=top_k, sample=sample, seed=seed, step=k)
                x = torch.cat((x, ix), dim=1)
                if use_graph:
                    # the remaining steps run as one captured HIP graph replayed per token (no host round trips)
                    for kk in range(k + 1, steps):
                        callback(kk)
                    rest = tr.decode_sample_graph(cache, ix, steps - 1 - k, temperature=temperature, top_k=top_k,
                                                   sample=sample, seed=seed, n_prompt=n_prompt)
                    x = torch.cat((x, rest), dim=1)
                    break
                if k + 1 < steps:
                    logits = tr.decode_step(cache, idx=ix)
            _, _, att = self.transformer(x[:, :-1], c)
            return x, att.detach().cpu()
        for k in range(steps):
            callback(k)
            cond_size = c.size(-2)
            assert x.size(1) + cond_size <= block_size
            logits, _, att = self.transformer(x, c)
            ix = ops.sample_logits(logits[:, -1, :], temperature=temperature, top_k=top_k, sample=sample, seed=seed,
                                   step=k)
            x = torch.cat((x, ix), dim=1)
        return x, att.detach().cpu()
