"""Oracle: minGPT forward as pure functions over a state_dict (torch CPU fp32).

TEST INFRASTRUCTURE - see oracle/__init__.py.  Functional restatement of
/root/reference/transformer/minGPT.py:45-212 (CausalSelfAttention, Block, GPT, GPTClass),
transformer/encoders.py:21-104, transformer/decoders.py:23-68 and the step logic of
Lit_minGPT (minGPT.py:260-285, 387-456) / GPT_VAE.loss (Lit_GPT_VAE.py:176-195).
Gradients come from torch autograd over these same fp32 CPU ops.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def as_torch_sd(sd, requires_grad=False):
    out = {}
    for k, v in sd.items():
        tv = torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v
        tv = tv.detach().clone().float()
        if requires_grad:
            tv.requires_grad_(True)
        out[k] = tv
    return out


def attention_mask(T, n_unmasked=0, device=None):
    """minGPT.py:65-69 - lower-triangular, with a fully visible n_unmasked x n_unmasked corner."""
    m = torch.tril(torch.ones(T, T, device=device))
    m[:n_unmasked, :n_unmasked] = 1
    return m


def self_attention(sd, prefix, x, n_head, n_unmasked=0, attn_pdrop=0.0, resid_pdrop=0.0, train=False):
    """minGPT.py:72-90.  Returns (y, att) with att = post-softmax, pre-dropout probabilities."""
    B, T, C = x.shape
    hs = C // n_head

    def lin(nm, v):
        return F.linear(v, sd[f"{prefix}{nm}.weight"], sd[f"{prefix}{nm}.bias"])

    def heads(v):
        return v.view(B, T, n_head, hs).transpose(1, 2)

    k, q, v = heads(lin("key", x)), heads(lin("query", x)), heads(lin("value", x))
    att = (q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(hs))
    att = att.masked_fill(attention_mask(T, n_unmasked, x.device)[None, None] == 0, float("-inf"))
    att = F.softmax(att, dim=-1)
    y = F.dropout(att, attn_pdrop, train) @ v
    y = y.transpose(1, 2).contiguous().view(B, T, C)
    y = F.dropout(lin("proj", y), resid_pdrop, train)
    return y, att


def block(sd, prefix, x, n_head, n_unmasked=0, pdrop=(0.0, 0.0), train=False):
    """minGPT.py:107-119 (pre-LN block; exact-erf GELU MLP)."""
    C = x.shape[-1]
    h = F.layer_norm(x, (C,), sd[prefix + "ln1.weight"], sd[prefix + "ln1.bias"], 1e-5)
    a, att = self_attention(sd, prefix + "attn.", h, n_head, n_unmasked, pdrop[0], pdrop[1], train)
    x = x + a
    h = F.layer_norm(x, (C,), sd[prefix + "ln2.weight"], sd[prefix + "ln2.bias"], 1e-5)
    h = F.linear(h, sd[prefix + "mlp.0.weight"], sd[prefix + "mlp.0.bias"])
    h = F.gelu(h)
    h = F.linear(h, sd[prefix + "mlp.2.weight"], sd[prefix + "mlp.2.bias"])
    x = x + F.dropout(h, pdrop[1], train)
    return x, att


def gpt_forward(sd, idx, n_layer, n_head, embeddings=None, n_unmasked=0, targets=None,
                pdrop=(0.0, 0.0, 0.0), train=False, prefix=""):
    """minGPT.py:168-199: (logits, loss|None, att_of_last_block).  pdrop = (embd, attn, resid)."""
    tok = F.embedding(idx, sd[prefix + "tok_emb.weight"])
    if embeddings is not None:
        tok = torch.cat((embeddings, tok), dim=1)
    t = tok.shape[1]
    assert t <= sd[prefix + "pos_emb"].shape[1], "Cannot forward, model block size is exhausted."
    x = F.dropout(tok + sd[prefix + "pos_emb"][:, :t, :], pdrop[0], train)
    att = None
    for i in range(n_layer):
        x, att = block(sd, f"{prefix}blocks.{i}.", x, n_head, n_unmasked, (pdrop[1], pdrop[2]), train)
    C = x.shape[-1]
    x = F.layer_norm(x, (C,), sd[prefix + "ln_f.weight"], sd[prefix + "ln_f.bias"], 1e-5)
    logits = F.linear(x, sd[prefix + "head.weight"])
    loss = None
    if targets is not None:
        loss = F.cross_entropy(logits.view(-1, logits.size(-1)), targets.view(-1))
    return logits, loss, att


def gptclass_forward(sd, idx, token, n_layer, n_head, **kw):
    """minGPT.py:209-212: prepend embedder(token)."""
    prefix = kw.get("prefix", "")
    emb = F.embedding(token, sd[prefix + "embedder.weight"])
    return gpt_forward(sd, idx, n_layer, n_head, embeddings=emb, **kw)


# ---------------------------------------------------------------- Lit_minGPT step logic
def codes_to_sequence(codes):
    """get_x (minGPT.py:387-394): (B,5,53) row-major -> (B,265) time-major, p = w*5 + h."""
    return torch.flatten(codes.permute(0, 2, 1), start_dim=1)


def make_idx(H, W):
    """minGPT.py:431-435."""
    idx = np.arange(H * W).reshape(H, W).T.ravel()
    return idx, np.argsort(idx)


def class_gpt_loss(sd, x, c, n_layer, n_head, **kw):
    """Lit_minGPT.forward + shared_step (minGPT.py:260-285, 413-417).  x (B,265) int64, c (B,1)."""
    logits, _, att = gptclass_forward(sd, x[:, :-1], c, n_layer, n_head, **kw)
    cond = c.size(-1)
    logits = logits[:, cond - 1:]
    loss = F.cross_entropy(logits.reshape(-1, logits.size(-1)), x.reshape(-1))
    return loss, logits, att


def top_k_logits(logits, k):
    """minGPT.py:287-291."""
    v, _ = torch.topk(logits, k)
    out = logits.clone()
    out[out < v[..., [-1]]] = -float("inf")
    return out


@torch.no_grad()
def sample_class_gpt(sd, x, c, steps, n_layer, n_head, temperature=1.0, top_k=None, **kw):
    """Greedy branch of Lit_minGPT.sample (minGPT.py:331-358): full re-forward per step."""
    att = None
    for _ in range(steps):
        logits, _, att = gptclass_forward(sd, x, c, n_layer, n_head, **kw)
        logits = logits[:, -1, :] / temperature
        if top_k is not None:
            logits = top_k_logits(logits, top_k)
        probs = F.softmax(logits, dim=-1)
        _, ix = torch.topk(probs, k=1, dim=-1)
        x = torch.cat((x, ix), dim=1)
    return x, att


# ------------------------------------------------------------------------- GPT-VAE
def vae_encode_stats(sd, x, n_layer, n_head, block_size, prefix=""):
    """GPTEncoder.forward (encoders.py:21-42): fully unmasked GPT, last position -> (mu, logvar)."""
    logits, _, att = gpt_forward(sd, x, n_layer, n_head, n_unmasked=block_size, prefix=prefix)
    mu, logvar = logits[:, -1, :].chunk(2, -1)
    return mu, logvar, att


def vae_loss(enc_sd, dec_sd, x, eps, kl_weight, n_layer, n_head, block_size):
    """GPTEncoder.encode + GPTDecoder.reconstruct_error + GPT_VAE.loss
    (encoders.py:62-104, decoders.py:23-68, Lit_GPT_VAE.py:176-195), nsamples = eps.shape[1]."""
    mu, logvar, _ = vae_encode_stats(enc_sd, x, n_layer, n_head, block_size)
    z = mu.unsqueeze(1) + eps * (0.5 * logvar).exp().unsqueeze(1)
    KL = 0.5 * (mu.pow(2) + logvar.exp() - logvar - 1).sum(dim=1)
    logits, _, _ = gpt_forward(dec_sd, x[:, :-1], n_layer, n_head, embeddings=z)
    cond = z.size(-2)
    logits = logits[:, cond - 1:]
    ce = F.cross_entropy(logits.reshape(-1, logits.size(-1)), x.reshape(-1), reduction="none")
    rec = ce.view(x.size(0), z.size(1), -1).sum(-1)
    loss = (rec.mean(dim=1) + kl_weight * KL).mean()
    return loss, rec, KL, mu, logvar


def vae_anneal(kl_weight, kl_start, warm_up, len_train_data, batch_size, beta=1.0):
    """KL-weight schedule of GPT_VAE (Lit_GPT_VAE.py:70-73, 253-256): one increment per training step."""
    if beta == 0:
        return beta
    rate = (1.0 - kl_start) / (warm_up * (len_train_data / batch_size)) if warm_up > 0 else 0
    return min(1.0, kl_weight + rate)


def vae_training_step(enc_sd, dec_sd, x, eps, kl_weight, n_layer, n_head, block_size, fb=0, target_kl=0.0, beta=1.0):
    """GPT_VAE.training_step after the anneal update (Lit_GPT_VAE.py:265-293): the free-bits variants.
    fb 0: rec + w*KL;  fb 1: KL counted per sequence where KL > target_kl;  fb 2: per latent dimension where the
    dimension's KL > target_kl / nz;  fb 3: all-or-nothing on the batch-mean KL.  Returns the batch-mean loss."""
    mu, logvar, _ = vae_encode_stats(enc_sd, x, n_layer, n_head, block_size)
    z = mu.unsqueeze(1) + eps * (0.5 * logvar).exp().unsqueeze(1)
    kl_dim = 0.5 * (mu.pow(2) + logvar.exp() - logvar - 1)
    KL = kl_dim.sum(dim=1)
    logits, _, _ = gpt_forward(dec_sd, x[:, :-1], n_layer, n_head, embeddings=z)
    logits = logits[:, z.size(-2) - 1:]
    ce = F.cross_entropy(logits.reshape(-1, logits.size(-1)), x.reshape(-1), reduction="none")
    rec = ce.view(x.size(0), z.size(1), -1).sum(-1).mean(dim=1)
    if beta == 0 or fb == 0:
        loss = rec + kl_weight * KL
    elif fb == 1:
        loss = rec + (KL > target_kl).float() * kl_weight * KL
    elif fb == 2:
        nz = mu.size(1)
        loss = rec + kl_weight * ((kl_dim > target_kl / float(nz)).float() * kl_dim).sum(dim=1)
    else:
        loss = rec + (KL.mean() > target_kl).float() * kl_weight * KL
    return loss.mean(dim=-1), rec, KL


def vae_validation_step(enc_sd, dec_sd, x, eps, n_layer, n_head, block_size):
    """GPT_VAE.validation_step for beta != 0 (Lit_GPT_VAE.py:321-361): ELBO at KL weight 1.0, summed over the batch."""
    _, rec, KL, _, _ = vae_loss(enc_sd, dec_sd, x, eps, 1.0, n_layer, n_head, block_size)
    rec = rec.mean(dim=1)
    return {"val_loss": (rec + KL).sum(), "val_loss_rc": rec.sum(), "val_loss_kl": KL.sum(),
            "report_num_words": (x.size(1) - 1) * x.size(0), "report_num_sents": x.size(0)}


def log_sum_exp(value, dim):
    """utils.log_sum_exp (transformer/utils.py:6-19), the `dim is not None, keepdim=False` branch the encoder uses."""
    m, _ = torch.max(value, dim=dim, keepdim=True)
    return m.squeeze(dim) + torch.log(torch.sum(torch.exp(value - m), dim=dim))


def vae_eval_inference_dist(mu, logvar, z):
    """GPTEncoder.eval_inference_dist (encoders.py:106-134): log q(z|x) of z (B, S, nz) under N(mu, exp(logvar)) of
    its own row -> (B, S)."""
    nz = z.size(2)
    mu, logvar = mu.unsqueeze(1), logvar.unsqueeze(1)
    dev = z - mu
    return -0.5 * ((dev ** 2) / logvar.exp()).sum(dim=-1) - 0.5 * (nz * math.log(2 * math.pi) + logvar.sum(-1))


def vae_calc_mi(mu, logvar, eps):
    """GPTEncoder.calc_mi (encoders.py:136-170): I(x, z) ~ E log q(z|x) - E log q(z), one reparameterised draw per x
    (eps (B, 1, nz) = that draw's noise), the aggregate posterior as the batch mixture -> Python float."""
    x_batch, nz = mu.size()
    neg_entropy = (-0.5 * nz * math.log(2 * math.pi) - 0.5 * (1 + logvar).sum(-1)).mean()
    z_samples = mu.unsqueeze(1) + eps * (0.5 * logvar).exp().unsqueeze(1)
    mu, logvar = mu.unsqueeze(0), logvar.unsqueeze(0)
    dev = z_samples - mu
    log_density = -0.5 * ((dev ** 2) / logvar.exp()).sum(dim=-1) - 0.5 * (nz * math.log(2 * math.pi) + logvar.sum(-1))
    log_qz = log_sum_exp(log_density, dim=1) - math.log(x_batch)
    return (neg_entropy - log_qz.mean(-1)).item()


def optimizer_groups(param_names):
    """configure_optimizers' partition (minGPT.py:618-665) expressed on parameter NAMES:
    decay = weights of Linear layers; no_decay = biases, LayerNorm/Embedding weights, pos_emb."""
    decay, no_decay = [], []
    for n in param_names:
        leaf = n.split(".")[-1]
        parent = n.split(".")[-2] if "." in n else ""
        if leaf == "bias" or n.endswith("pos_emb"):
            no_decay.append(n)
        elif parent in ("ln1", "ln2", "ln_f", "tok_emb", "embedder"):
            no_decay.append(n)
        else:
            decay.append(n)
    return sorted(decay), sorted(no_decay)
