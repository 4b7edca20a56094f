#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz by importing and RUNNING the real
reference (karchkha/MelSpec_GPT_VQVAE, mounted read-only at /root/reference) on CPU.

Run from the repo root in the build container:  python tests/golden/make_golden.py
The reference cannot travel to the GPU box, so only these small input/output vectors
(and this script) are committed.  Nothing here is imported by the product package.

Import recipe: pytorch_lightning / torchvision / albumentations are not
installed and the reference's `datasets/` dir (no __init__.py) is shadowed by the
HuggingFace package, so tiny `sys.modules` stand-ins are installed *for the import only*;
none of them touches the arithmetic being recorded.
"""
from __future__ import annotations

import importlib.machinery
import math
import os
import sys
import types

sys.dont_write_bytecode = True  # never leave __pycache__ inside /root/reference
REF = os.environ.get("MELGPT_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np
import torch
import torch.nn as nn

import synth  # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Sequential(object):
    """what torchvision.transforms.Compose does with a list of callables (the package is not installed): apply
    them in order; `.transforms` is the list, as the reference's inv_transforms reads it (:157)."""

    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, x):
        for f in self.transforms:
            x = f(x)
        return x


def import_reference():
    class LightningModule(nn.Module):
        def log(self, *a, **k):
            pass

        def print(self, *a, **k):
            pass

    _stub("pytorch_lightning", LightningModule=LightningModule, LightningDataModule=object,
          Callback=object, seed_everything=lambda *a, **k: None)
    tv = _stub("torchvision")
    tv.transforms = _stub("torchvision.transforms", Compose=_Sequential)
    tv.utils = _stub("torchvision.utils")
    _stub("albumentations")
    ds = _stub("datasets")
    ds.__path__ = []
    ds.datamodule = _stub("datasets.datamodule", DataModule=object)
    sys.path.insert(0, REF)
    import transformer.minGPT as ref_gpt
    import transformer.encoders as ref_enc
    import transformer.decoders as ref_dec
    import vqvae.big_model_attn_gan as ref_vq
    return ref_gpt, ref_enc, ref_dec, ref_vq


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def load_sd(module, sd_np, strict=True):
    sd = {k: t(v) for k, v in sd_np.items()}
    missing = module.load_state_dict(sd, strict=False)
    # only the constant `mask` buffers may be missing from a synthetic state_dict
    bad = [k for k in missing.missing_keys if not k.endswith("mask")]
    assert not bad and not missing.unexpected_keys, (bad, missing.unexpected_keys)


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"  wrote {name}.npz  ({os.path.getsize(path) / 1024:.0f} KiB)")


# ------------------------------------------------------------------------------ VQ
def top2_gap_ulps(flat, emb):
    """distance formula of big_model_attn_gan.py:28-30 in fp32, then gap between the two
    smallest distances in units of ulp(min distance)."""
    d = (torch.sum(flat ** 2, dim=1, keepdim=True) + torch.sum(emb ** 2, dim=1)
         - 2 * torch.matmul(flat, emb.t()))
    v, _ = torch.topk(d, 2, dim=1, largest=False)
    ulp = np.spacing(np.abs(v[:, 0].numpy()).astype(np.float32))
    return ((v[:, 1] - v[:, 0]).numpy() / ulp).astype(np.float32), torch.topk(d, 2, dim=1, largest=False)[1].numpy()


def gen_vq(ref_vq):
    print("VQ")
    for tag, cb in (("normal", "normal"), ("default", "default")):
        vq = ref_vq.VectorQuantizer(128, 256, 0.25)
        if cb == "normal":
            E = synth.normal(11, (128, 256))
        else:
            E = synth.uniform(12, (128, 256), -1 / 128, 1 / 128)
        vq._embedding.weight.data.copy_(t(E))
        scale = 1.0 if cb == "normal" else 0.02
        z = synth.normal(10, (2, 256, 5, 53)) * np.float32(scale)
        zt = t(z).requires_grad_(True)
        loss, q, (perp, enc, idx) = vq(zt)
        # gradient semantics (STE + the two MSE terms)
        g = synth.normal(13, q.shape)
        (loss * 3.0 + (q * t(g)).sum()).backward()
        flat = t(z).permute(0, 2, 3, 1).reshape(-1, 256)
        gap, top2 = top2_gap_ulps(flat, vq._embedding.weight.data)
        save(f"vq_small_{tag}", z_seed=10, z_scale=scale, codebook_seed=(11 if cb == "normal" else 12),
             upstream_seed=13, loss=loss.item(), quantized=q.detach().numpy(),
             perplexity=perp.item(), indices=idx.numpy().astype(np.int16), gap_ulps=gap, top2=top2.astype(np.int16),
             enc_rowsum=enc.sum(1).numpy(), dz=zt.grad.numpy(),
             dcodebook=vq._embedding.weight.grad.numpy())
        gcb = vq.get_codebook_entry(idx.squeeze(1), (2, 5, 53, 256))
        assert torch.equal(gcb, q.detach() * 0 + gcb)
        save(f"vq_gather_{tag}", indices=idx.numpy().astype(np.int16), out=gcb.detach().numpy())

    # B=64 (config 2): only seeds + indices are stored
    vq = ref_vq.VectorQuantizer(128, 256, 0.25)
    E = synth.normal(21, (128, 256))
    vq._embedding.weight.data.copy_(t(E))
    z = synth.normal(20, (64, 256, 5, 53))
    with torch.no_grad():
        loss, q, (perp, enc, idx) = vq(t(z))
    flat = t(z).permute(0, 2, 3, 1).reshape(-1, 256)
    gap, top2 = top2_gap_ulps(flat, t(E))
    save("vq_b64", z_seed=20, codebook_seed=21, indices=idx.numpy().astype(np.int16).ravel(),
         gap_ulps=gap, top2=top2.astype(np.int16), loss=loss.item(), perplexity=perp.item(),
         q_checksum=float(q.double().sum()), q_abs_checksum=float(q.double().abs().sum()))

    # exact ties: duplicated codebook rows -> lowest index must win (torch.argmin semantics)
    E = synth.normal(31, (128, 256))
    E[77] = E[5]
    E[100] = E[5]
    E[64] = E[63]
    vq._embedding.weight.data.copy_(t(E))
    z = synth.normal(30, (1, 256, 5, 53))
    zz = z.reshape(256, 265)
    zz[:, 0] = E[5]       # exactly on a triplicated code
    zz[:, 1] = E[63]      # exactly on a duplicated code
    zz[:, 2] = 0.0        # all-zero vector
    z = zz.reshape(1, 256, 5, 53)
    with torch.no_grad():
        _, _, (_, _, idx) = vq(t(z))
    assert idx[0, 0] == 5 and idx[1, 0] == 63
    save("vq_ties", z=z, codebook=E, indices=idx.numpy().astype(np.int16).ravel())


# ----------------------------------------------------------------------- attention
def gen_attention(ref_gpt):
    print("CausalSelfAttention / Block")
    for n_unmasked in (0, 265):
        cfg = ref_gpt.GPTConfig(128, 265, n_embd=128, n_head=2, attn_pdrop=0.0, resid_pdrop=0.0,
                                n_unmasked=n_unmasked)
        att = ref_gpt.CausalSelfAttention(cfg)
        sd = {}
        for j, nm in enumerate(("key", "query", "value", "proj")):
            sd[f"{nm}.weight"] = synth.normal(100 + j, (128, 128), 0.08)
            sd[f"{nm}.bias"] = synth.normal(110 + j, (128,), 0.05)
        load_sd(att, sd)
        x = synth.normal(120, (2, 265, 128))
        xt = t(x).requires_grad_(True)
        y, a = att(xt)
        gy = synth.normal(121, y.shape)
        (y * t(gy)).sum().backward()
        out = dict(x=x, gy=gy, y=y.detach().numpy(), att=a.detach().numpy()[:1].astype(np.float32),
                   att_rowsum=a.detach().sum(-1).numpy(), dx=xt.grad.numpy())
        for nm, p in att.named_parameters():
            out["w." + nm] = sd[nm]
            out["g." + nm] = p.grad.numpy()
        save(f"attn_u{n_unmasked}", **out)

    # one Block, tuple-in/tuple-out (minGPT.py:107-119)
    cfg = ref_gpt.GPTConfig(128, 265, n_embd=128, n_head=2, attn_pdrop=0.0, resid_pdrop=0.0, n_unmasked=0)
    blk = ref_gpt.Block(cfg)
    args = synth.gpt_args(n_layer=1, n_head=2, n_embd=128, block_size=265)
    full = synth.gpt_state_dict(args, 7)
    sd = {k[len("blocks.0."):]: v for k, v in full.items() if k.startswith("blocks.0.")}
    load_sd(blk, sd)
    x = synth.normal(130, (2, 265, 128))
    xt = t(x).requires_grad_(True)
    y, a = blk((xt, None))
    gy = synth.normal(131, y.shape)
    (y * t(gy)).sum().backward()
    out = dict(x=x, gy=gy, y=y.detach().numpy(), dx=xt.grad.numpy(), sd_seed=7)
    for nm, p in blk.named_parameters():
        out["g." + nm] = p.grad.numpy()
    save("block", **out)


# ------------------------------------------------------------------------------ GPT
def gen_gpt(ref_gpt):
    print("GPT / GPTClass")
    # (a) small 2-layer class-conditioned GPT: everything stored
    args = synth.gpt_args(n_layer=2, n_head=4, n_embd=256)
    m = ref_gpt.GPTClass(args)
    sd = synth.gpt_state_dict(args, 1)
    load_sd(m, sd)
    m.eval()
    x = synth.randint(200, 0, 128, (2, 265))
    c = synth.randint(201, 0, 8, (2, 1))
    logits, _, att = m(t(x)[:, :-1], t(c))
    loss = nn.functional.cross_entropy(logits.reshape(-1, 128), t(x).reshape(-1))
    loss.backward()
    out = dict(x=x, c=c, logits=logits.detach().numpy(), loss=loss.item(),
               att=att.detach().numpy()[:1, :2], sd_seed=1)
    for nm, p in m.named_parameters():
        out["gnorm." + nm] = float(p.grad.double().norm())
    for nm in ("tok_emb.weight", "embedder.weight", "pos_emb", "head.weight", "ln_f.weight",
               "blocks.0.attn.key.bias", "blocks.1.mlp.0.bias", "blocks.1.ln2.weight"):
        out["g." + nm] = dict(m.named_parameters())[nm].grad.numpy()
    out["g.blocks.0.mlp.2.weight"] = dict(m.named_parameters())["blocks.0.mlp.2.weight"].grad.numpy()
    save("gptclass_small", **out)

    # (b) GPT.forward with targets, unmasked (= GPTEncoder's transformer) + last_linear
    args = synth.gpt_args(n_layer=2, n_head=4, n_embd=256, block_size=265, fix_var=0)
    m = ref_gpt.GPT(args, n_unmasked=265, last_linear=512, block_size=265)
    sd = synth.gpt_state_dict(args, 2, block_size=265, with_embedder=False, out_features=512)
    load_sd(m, sd)
    x = synth.randint(210, 0, 128, (2, 265))
    logits, _, att = m(t(x))
    save("gpt_unmasked_small", x=x, logits_last=logits.detach().numpy()[:, -1],
         logits_sum=float(logits.double().sum()), att_last_row=att.detach().numpy()[:, :, -1], sd_seed=2)

    # (c) VAS-size (C=1024, H=16) 2-layer model: logits + loss + grad norms
    args = synth.gpt_args(n_layer=2, n_head=16, n_embd=1024)
    m = ref_gpt.GPTClass(args)
    sd = synth.gpt_state_dict(args, 3)
    load_sd(m, sd)
    x = synth.randint(220, 0, 128, (2, 265))
    c = synth.randint(221, 0, 8, (2, 1))
    logits, _, att = m(t(x)[:, :-1], t(c))
    loss = nn.functional.cross_entropy(logits.reshape(-1, 128), t(x).reshape(-1))
    loss.backward()
    out = dict(x=x, c=c, logits=logits.detach().numpy(), loss=loss.item(), sd_seed=3,
               att_b0h3=att.detach().numpy()[0, 3])
    for nm, p in m.named_parameters():
        out["gnorm." + nm] = float(p.grad.double().norm())
    save("gptclass_vas2", **out)

    # (d) full 24-layer VAS model: loss only (weights regenerated from the seed)
    args = synth.gpt_args(n_layer=24, n_head=16, n_embd=1024)
    m = ref_gpt.GPTClass(args)
    sd = synth.gpt_state_dict(args, 4)
    load_sd(m, sd)
    del sd
    x = synth.randint(230, 0, 128, (2, 265))
    c = synth.randint(231, 0, 8, (2, 1))
    with torch.no_grad():
        logits, _, _ = m(t(x)[:, :-1], t(c))
        loss = nn.functional.cross_entropy(logits.reshape(-1, 128), t(x).reshape(-1))
    save("gptclass_vas24", x=x, c=c, loss=loss.item(), logits_b0_t17=logits[0, 17].numpy(),
         logits_b1_last=logits[1, -1].numpy(), sd_seed=4)

    # (e) reference init statistics + optimizer grouping (minGPT.py:159-166, 618-665)
    torch.manual_seed(0)
    args = synth.gpt_args(n_layer=24, n_head=16, n_embd=1024)
    m = ref_gpt.GPTClass(args)
    decay, no_decay = set(), set()
    for mn, mod in m.named_modules():
        for pn, p in mod.named_parameters():
            fpn = "%s.%s" % (mn, pn) if mn else pn
            if pn.endswith("bias"):
                no_decay.add(fpn)
            elif pn.endswith("weight") and isinstance(mod, nn.Linear):
                decay.add(fpn)
            elif pn.endswith("weight") and isinstance(mod, (nn.LayerNorm, nn.Embedding)):
                no_decay.add(fpn)
    no_decay.add("pos_emb")
    save("gpt_optim_groups", decay=np.array(sorted(decay)), no_decay=np.array(sorted(no_decay)),
         n_params=sum(p.numel() for p in m.parameters()),
         embedder_std=float(m.embedder.weight.std()), tok_emb_std=float(m.tok_emb.weight.std()),
         keys=np.array(list(m.state_dict().keys())))


def gen_lit(ref_gpt):
    """Lit_minGPT.forward/shared_step/sample/get_x/make_idx (minGPT.py:260-456)."""
    print("Lit_minGPT")

    class Lit(ref_gpt.Lit_minGPT):
        def datamodule_loader(self):
            self.data = None

    args = synth.gpt_args(n_layer=2, n_head=4, n_embd=256, reconstruct_spec="", device="cpu", batch_size=2)
    lit = Lit(args)
    load_sd(lit.transformer, synth.gpt_state_dict(args, 1))
    lit.eval()
    codes = synth.randint(300, 0, 128, (2, 5, 53))
    target = synth.randint(301, 0, 8, (2,))
    batch = {"codes": t(codes), "target": t(target)}
    x = lit.get_x(batch)
    loss = lit.shared_step(batch, 0)
    fwd, bwd = lit.make_idx(5, 53)
    assert torch.equal(lit.code_reader(t(codes).reshape(2, 265)), x)
    x0 = x[:, :9]
    xs, att = lit.sample(x0, lit.get_c(batch), steps=16, sample=False)
    xk, _ = lit.sample(x0, lit.get_c(batch), steps=4, sample=False, top_k=5, temperature=0.7)
    save("lit_mingpt", codes=codes, target=target, x=x.numpy(), loss=loss.item(), fwd_idx=fwd.numpy(),
         bwd_idx=bwd.numpy(), greedy16=xs.numpy(), greedy4_topk=xk.numpy(), att_shape=np.array(att.shape),
         att_last=att.numpy()[:, :, -1], sd_seed=1)


def gen_vae(ref_gpt, ref_enc, ref_dec):
    """GPTEncoder / GPTDecoder and the ELBO pieces of GPT_VAE.loss (Lit_GPT_VAE.py:176-195)."""
    print("GPT-VAE")
    args = synth.gpt_args(n_layer=2, n_head=4, n_embd=256, block_size=265, fix_var=0)
    enc = ref_enc.GPTEncoder(args, n_unmasked=265, last_linear=512)
    dec = ref_dec.GPTDecoder(args, block_size=266)
    load_sd(enc.transformer, synth.gpt_state_dict(args, 5, block_size=265, with_embedder=False, out_features=512))
    load_sd(dec.transformer, synth.gpt_state_dict(args, 6, block_size=266, with_embedder=False))
    x = synth.randint(400, 0, 128, (2, 265))
    mu, logvar, _ = enc(t(x))
    eps = synth.normal(401, (2, 1, 256))
    z = mu.unsqueeze(1) + t(eps) * (0.5 * logvar).exp().unsqueeze(1)
    KL = 0.5 * (mu.pow(2) + logvar.exp() - logvar - 1).sum(dim=1)
    rec = dec.reconstruct_error(t(x), z)
    kl_w = 0.3
    loss = (rec.mean(dim=1) + kl_w * KL).mean()
    loss.backward()
    out = dict(x=x, eps=eps, mu=mu.detach().numpy(), logvar=logvar.detach().numpy(), KL=KL.detach().numpy(),
               rec=rec.detach().numpy(), loss=loss.item(), kl_weight=kl_w, enc_seed=5, dec_seed=6)
    for nm, p in enc.transformer.named_parameters():
        out["enc.gnorm." + nm] = float(p.grad.double().norm())
    for nm, p in dec.transformer.named_parameters():
        out["dec.gnorm." + nm] = float(p.grad.double().norm())
    dec.eval()
    with torch.no_grad():
        xs, _ = dec.sample(t(x)[:, :3], z.detach(), steps=8, sample=False)
    out["dec_greedy8"] = xs.numpy()
    save("gpt_vae_small", **out)


def gen_vae_mi(ref_enc):
    """GPTEncoder.eval_inference_dist and calc_mi (encoders.py:106-170; utils.log_sum_exp :6-19) of the real encoder on
    a batch of 5: log q(z|x) at recorded z points (own statistics, and `param=` given), and the mutual-information
    estimate with the reparameterisation noise of its one draw recorded (the encoder runs without dropout, so the
    only consumer of the generator between the seed and `normal_()` is that draw)."""
    print("GPT-VAE eval_inference_dist / calc_mi")
    args = synth.gpt_args(n_layer=2, n_head=4, n_embd=256, block_size=265, fix_var=0)
    enc = ref_enc.GPTEncoder(args, n_unmasked=265, last_linear=512)
    load_sd(enc.transformer, synth.gpt_state_dict(args, 5, block_size=265, with_embedder=False, out_features=512))
    enc.eval()
    B, S, nz = 5, 3, 256
    x = synth.randint(410, 0, 128, (B, 265))
    with torch.no_grad():
        mu, logvar, _ = enc(t(x))
        z = mu.unsqueeze(1) + t(synth.normal(411, (B, S, nz))) * (0.5 * logvar).exp().unsqueeze(1) * 1.5
        logq = enc.eval_inference_dist(t(x), z)
        mu_p, logvar_p = t(synth.normal(412, (B, nz), 0.5)), t(synth.normal(413, (B, nz), 0.3))
        logq_p = enc.eval_inference_dist(t(x), z, param=(mu_p, logvar_p))
        torch.manual_seed(4141)
        mi = enc.calc_mi(t(x))
        torch.manual_seed(4141)
        eps = torch.zeros(B, 1, nz).normal_()
        # the recorded noise really is the draw calc_mi made: its lines restated on (mu, logvar, eps) give the same number
        zs = mu.unsqueeze(1) + eps * (0.5 * logvar).exp().unsqueeze(1)
        dev = zs - mu.unsqueeze(0)
        ld = -0.5 * ((dev ** 2) / logvar.exp().unsqueeze(0)).sum(-1) - 0.5 * (nz * math.log(2 * math.pi) + logvar.unsqueeze(0).sum(-1))
        m = ld.max(1, keepdim=True)[0]
        lqz = (m.squeeze(1) + torch.log(torch.exp(ld - m).sum(1))) - math.log(B)
        ne = (-0.5 * nz * math.log(2 * math.pi) - 0.5 * (1 + logvar).sum(-1)).mean()
        assert abs((ne - lqz.mean(-1)).item() - mi) <= 1e-6 * max(1.0, abs(mi)), ((ne - lqz.mean(-1)).item(), mi)
    save("gpt_vae_mi", x=x, z=z.numpy(), mu=mu.numpy(), logvar=logvar.numpy(), logq=logq.numpy(), mu_p=mu_p.numpy(),
         logvar_p=logvar_p.numpy(), logq_p=logq_p.numpy(), mi=np.float64(mi), mi_eps=eps.numpy(), enc_seed=5)


# ------------------------------------------------------------------ GPT_VAE.training_step / validation_step
def _import_ref_gpt_vae():
    """the reference's real LightningModule (transformer/Lit_GPT_VAE.py:23-89); its data loading
    (datamodule_loader / *_dataloader, :945-957) is overridden as SURVEY 8c describes - nothing on the recorded
    arithmetic path is touched."""
    sys.path.insert(0, REF)
    import transformer.Lit_GPT_VAE as ref_lit

    class RefVAE(ref_lit.GPT_VAE):
        def datamodule_loader(self):
            self.len_train_data = self.args.len_train_data

        def train_dataloader(self):
            return None

        def val_dataloader(self):
            return None

        def test_dataloader(self):
            return None

    return RefVAE


def _vae_args(**kw):
    d = dict(n_layer=2, n_head=4, n_embd=256, block_size=265, fix_var=0, kl_start=0.1, warm_up=2, batch_size=2,
             target_kl=8.0, beta=1.0, nsamples=1, fb=0, device="cpu", learning_rate=1e-6, len_train_data=12,
             iw_train_nsamples=-1)
    d.update(kw)
    return synth.gpt_args(**d)


def gen_vae_steps():
    """One training_step per free-bits branch fb in {0,1,2,3} (Lit_GPT_VAE.py:246-315) with the reparameterisation
    noise captured (the only random draw at dropout 0: `zeros_like(std).normal_()` right after manual_seed), the
    KL-weight anneal over 3 consecutive steps (:70-73,253-256), the beta = 0 branch, and validation_step (:321-361)."""
    print("GPT_VAE.training_step / validation_step")
    RefVAE = _import_ref_gpt_vae()
    codes = synth.randint(500, 0, 128, (2, 5, 53))
    batch = {"codes": t(codes)}
    out = dict(codes=codes, enc_seed=5, dec_seed=6, len_train_data=12, warm_up=2, kl_start=0.1, batch_size=2)
    # thresholds chosen from a probe run so that every mask has both outcomes somewhere (see the probe values stored)
    probe = None
    for fb in (0, 1, 2, 3):
        for variant in ("a", "b"):
            if fb == 0 and variant == "b":
                continue
            args = _vae_args(fb=fb)
            m = RefVAE(args)
            load_sd(m.encoder.transformer, synth.gpt_state_dict(args, 5, block_size=265, with_embedder=False,
                                                                out_features=512))
            load_sd(m.decoder.transformer, synth.gpt_state_dict(args, 6, block_size=266, with_embedder=False))
            m.train()
            if probe is None:
                with torch.no_grad():
                    mu, logvar, _ = m.encoder(m.get_input(batch))
                    kl = 0.5 * (mu.pow(2) + logvar.exp() - logvar - 1)
                probe = dict(kl_per_seq=kl.sum(1).numpy(), kl_dim_median=float(kl.median()))
                out["probe_kl_per_seq"] = probe["kl_per_seq"]
            kls = probe["kl_per_seq"]
            if fb == 1:      # per-sequence mask: (a) one of two sequences above the threshold, (b) none
                args.target_kl = float(kls.mean()) if variant == "a" else float(kls.max() * 2)
            elif fb == 2:    # per-dimension mask: threshold at the median dimension (a) / tiny (b: all kept)
                args.target_kl = probe["kl_dim_median"] * 256 if variant == "a" else 1e-9
                m.dim_target_kl = args.target_kl / float(args.n_embd)
            elif fb == 3:    # batch-mean mask: (a) below the mean -> on, (b) above -> off
                args.target_kl = float(kls.mean()) * (0.5 if variant == "a" else 2.0)
            tag = f"fb{fb}{variant}"
            seed = 600 + 10 * fb + (variant == "b")
            torch.manual_seed(seed)
            eps = torch.zeros(2, 1, 256).normal_()
            torch.manual_seed(seed)
            m.zero_grad()
            loss = m.training_step(batch, 0)
            loss.backward()
            out[tag + ".eps"] = eps.numpy()
            out[tag + ".target_kl"] = args.target_kl
            out[tag + ".loss"] = loss.item()
            out[tag + ".kl_weight"] = m.kl_weight
            for nm in ("encoder.transformer.head.weight", "encoder.transformer.blocks.0.attn.query.weight",
                       "decoder.transformer.tok_emb.weight", "decoder.transformer.blocks.1.mlp.2.weight"):
                out[tag + ".gnorm." + nm] = float(dict(m.named_parameters())[nm].grad.double().norm())
    # anneal: three consecutive steps of one module, fb = 0
    args = _vae_args(fb=0)
    m = RefVAE(args)
    load_sd(m.encoder.transformer, synth.gpt_state_dict(args, 5, block_size=265, with_embedder=False, out_features=512))
    load_sd(m.decoder.transformer, synth.gpt_state_dict(args, 6, block_size=266, with_embedder=False))
    m.train()
    out["anneal_rate"] = m.anneal_rate
    ws, ls, es = [], [], []
    for k in range(3):
        torch.manual_seed(700 + k)
        es.append(torch.zeros(2, 1, 256).normal_().numpy())
        torch.manual_seed(700 + k)
        ls.append(m.training_step(batch, k).item())
        ws.append(m.kl_weight)
    out.update(anneal_kl_weights=np.array(ws), anneal_losses=np.array(ls), anneal_eps=np.stack(es))
    # beta == 0: kl_weight pinned to 0 (plain autoencoder objective)
    args = _vae_args(beta=0.0)
    m = RefVAE(args)
    load_sd(m.encoder.transformer, synth.gpt_state_dict(args, 5, block_size=265, with_embedder=False, out_features=512))
    load_sd(m.decoder.transformer, synth.gpt_state_dict(args, 6, block_size=266, with_embedder=False))
    m.train()
    torch.manual_seed(800)
    out["beta0.eps"] = torch.zeros(2, 1, 256).normal_().numpy()
    torch.manual_seed(800)
    out["beta0.loss"] = m.training_step(batch, 0).item()
    out["beta0.kl_weight"] = float(m.kl_weight)
    # validation_step: kl weight 1.0 whatever the anneal state; sums, not means
    args = _vae_args()
    m = RefVAE(args)
    load_sd(m.encoder.transformer, synth.gpt_state_dict(args, 5, block_size=265, with_embedder=False, out_features=512))
    load_sd(m.decoder.transformer, synth.gpt_state_dict(args, 6, block_size=266, with_embedder=False))
    m.eval()
    torch.manual_seed(900)
    out["val.eps"] = torch.zeros(2, 1, 256).normal_().numpy()
    torch.manual_seed(900)
    with torch.no_grad():
        r = m.validation_step(batch, 0)
    out.update({"val.val_loss": float(r["val_loss"]), "val.val_loss_rc": float(r["val_loss_rc"]),
                "val.val_loss_kl": float(r["val_loss_kl"]), "val.report_num_words": r["report_num_words"],
                "val.report_num_sents": r["report_num_sents"]})
    save("gpt_vae_steps", **out)


def gen_vae_xl(ref_enc, ref_dec):
    """2-layer GPT-VAE at the XL WIDTH of BASELINE configs[3] (config/config_GPT_VAE_vggsound.py:43-58: C = 1472,
    23 heads, vocab 1024, block 265): mu / logvar / rec / KL / loss / per-parameter gradient norms."""
    print("GPT-VAE, XL width")
    args = synth.gpt_args(vocab_size=1024, n_layer=2, n_head=23, n_embd=1472, block_size=265, fix_var=0)
    enc = ref_enc.GPTEncoder(args, n_unmasked=265, last_linear=2944)
    dec = ref_dec.GPTDecoder(args, block_size=266)
    load_sd(enc.transformer, synth.gpt_state_dict(args, 8, block_size=265, with_embedder=False, out_features=2944))
    load_sd(dec.transformer, synth.gpt_state_dict(args, 9, block_size=266, with_embedder=False))
    x = synth.randint(410, 0, 1024, (2, 265))
    mu, logvar, att = enc(t(x))
    eps = synth.normal(411, (2, 1, 1472))
    z = mu.unsqueeze(1) + t(eps) * (0.5 * logvar).exp().unsqueeze(1)
    KL = 0.5 * (mu.pow(2) + logvar.exp() - logvar - 1).sum(dim=1)
    logits, _ = dec(t(x), z)
    rec = dec.reconstruct_error(t(x), z)
    kl_w = 0.5
    loss = (rec.mean(dim=1) + kl_w * KL).mean()
    loss.backward()
    out = dict(x=x, eps=eps, mu=mu.detach().numpy(), logvar=logvar.detach().numpy(), KL=KL.detach().numpy(),
               rec=rec.detach().numpy(), loss=loss.item(), kl_weight=kl_w, enc_seed=8, dec_seed=9,
               dec_logits_b0_t100=logits[0, 100].detach().numpy(), dec_logits_b1_last=logits[1, -1].detach().numpy(),
               enc_att_h22_row264=att[:, 22, 264].detach().numpy())
    for nm, p in enc.transformer.named_parameters():
        out["enc.gnorm." + nm] = float(p.grad.double().norm())
    for nm, p in dec.transformer.named_parameters():
        out["dec.gnorm." + nm] = float(p.grad.double().norm())
    save("gpt_vae_xl2", **out)


# ---------------------------------------------------------------------------- VQVAE
def gen_vqvae(ref_vq):
    print("VQVAE encoder / decoder")
    # (a) narrow model (ch=32), odd-ish geometry 80x848 is kept: the attention level must be 5x53
    hp = dict(ch=32, ch_mult=(1, 1, 2, 2, 4), num_res_blocks=2, z_channels=64)
    enc = ref_vq.Encoder(ch=32, out_ch=1, ch_mult=(1, 1, 2, 2, 4), num_res_blocks=2, attn_resolutions=[53],
                         in_channels=1, resolution=848, z_channels=64, double_z=False)
    dec = ref_vq.Decoder(ch=32, out_ch=1, ch_mult=(1, 1, 2, 2, 4), num_res_blocks=2, attn_resolutions=[53],
                         in_channels=1, resolution=848, z_channels=64, double_z=False)
    load_sd(enc, synth.encoder_state_dict(40, **hp))
    load_sd(dec, synth.decoder_state_dict(40, **hp))
    x = 2 * synth.mel_tiles(41, 1)[:, None, :, 6:854] - 1
    with torch.no_grad():
        h = enc(t(x))
        taps = {}
        hh = enc.conv_in(t(x))
        taps["conv_in_sum"] = float(hh.double().sum())
        hh = enc.down[0].block[0](hh, None)
        taps["d0b0_sum"] = float(hh.double().sum())
        taps["d0b0_patch"] = hh[0, :, 10:12, 100:104].numpy()
        zq = synth.normal(42, (1, 64, 5, 53))
        y = dec(t(zq))
    save("vqvae_narrow", x=x, enc_out=h.numpy(), dec_in=zq, dec_out=y.numpy(), seed=40, **taps)

    # (b) full-size (ch=128): one tile through encode + VQ, decoder on a quantised latent
    m = ref_vq.LitVQVAE(num_embeddings=128, embedding_dim=256)
    sd = synth.vqvae_state_dict(50)
    missing = m.load_state_dict({k: t(v) for k, v in sd.items()}, strict=False)
    assert all(k.startswith("discriminator.") for k in missing.missing_keys), missing.missing_keys[:5]
    assert not missing.unexpected_keys
    m.eval()
    x = 2 * synth.mel_tiles(51, 2)[:, None, :, 6:854] - 1
    with torch.no_grad():
        z = m.encode(t(x))
        loss, q, (perp, enc1h, idx) = m._vq_vae(z)
        flat = z.permute(0, 2, 3, 1).reshape(-1, 256)
        gap, top2 = top2_gap_ulps(flat, m._vq_vae._embedding.weight.data)
        rec = m.decode(q[:1])
    save("vqvae_full", x=x, z=z.numpy(), indices=idx.numpy().astype(np.int16).ravel(), gap_ulps=gap,
         top2=top2.astype(np.int16), vq_loss=loss.item(), perplexity=perp.item(),
         rec=rec.numpy(), seed=50,
         sd_keys=np.array([k for k in m.state_dict().keys()]))


def gen_resblock_grad(ref_vq):
    """Gradients of the REAL ResnetBlock (vqvae/big_model_attn_gan.py:75-135: norm1 -> swish -> conv1 -> norm2 -> swish ->
    conv2, + x) by torch autograd on CPU: the pin of melgpt_conv3x3_bwd_{data,weight} / melgpt_groupnorm_swish_bwd."""
    print("ResnetBlock gradients")
    C, B, H, W = 64, 2, 10, 22
    blk = ref_vq.ResnetBlock(in_channels=C, out_channels=C, dropout=0.0, temb_channels=0)
    load_sd(blk, synth.resblock_state_dict(60, C))
    x = t(synth.normal(61, (B, C, H, W), 1.2, 0.3)).requires_grad_(True)
    g = t(synth.normal(62, (B, C, H, W), 0.7))
    y = blk(x, None)
    y.backward(g)
    grads = {("d_" + k.replace(".", "_")): p.grad.numpy() for k, p in blk.named_parameters()}
    save("resblock_grad", x=x.detach().numpy(), g=g.numpy(), y=y.detach().numpy(), dx=x.grad.numpy(), seed=60, channels=C, **grads)


NARROW = dict(ch=32, z_channels=64)   # module globals of big_model_attn_gan.py patched for the narrow end-to-end model


def grad_probe(name, g, k=16):
    """(norm, k sampled entries at seeded positions) of one gradient tensor: small enough to commit for every parameter"""
    flat = np.asarray(g, dtype=np.float64).ravel()
    pos = synth.randint(sum(map(ord, name)), 0, flat.size, (k,))
    return float(np.sqrt((flat * flat).sum())), flat[pos].astype(np.float32), pos.astype(np.int64)


def gen_vqvae_grad(ref_vq):
    """Gradients of the REAL LitVQVAE.forward (vqvae/big_model_attn_gan.py:622-634) - encoder, quant convs, VectorQuantizer with
    its straight-through estimator, decoder - for a narrow model (module globals ch = 32, z_channels = 64; one 80 x 848 tile, so
    that the AttnBlocks sit at 5 x 53): L = vq_loss + sum(x_recon * g).  Per parameter: gradient norm + 16 sampled entries."""
    print("LitVQVAE gradients (narrow)")
    saved = {k: getattr(ref_vq, k) for k in NARROW}
    try:
        for k, v in NARROW.items():
            setattr(ref_vq, k, v)
        m = ref_vq.LitVQVAE(num_embeddings=128, embedding_dim=256)
    finally:
        for k, v in saved.items():
            setattr(ref_vq, k, v)
    sd = synth.vqvae_state_dict(70, num_embeddings=128, embedding_dim=256, ch=32, z_channels=64)
    missing = m.load_state_dict({k: t(v) for k, v in sd.items()}, strict=False)
    assert all(k.startswith("discriminator.") for k in missing.missing_keys) and not missing.unexpected_keys
    m.train()
    x = t(2 * synth.mel_tiles(71, 1)[:, None, :, 6:854] - 1)
    g = t(synth.normal(72, (1, 1, 80, 848), 0.01))
    loss, x_recon, info = m(x)
    L = loss + (x_recon * g).sum()
    L.backward()
    # (x = 2 * synth.mel_tiles(71, 1)[:, None, :, 6:854] - 1 and g = synth.normal(72, (1,1,80,848), 0.01) are regenerated from their seeds)
    out = dict(vq_loss=float(loss), L=float(L), indices=info[2].numpy().astype(np.int16).ravel(),
               rec_sum=float(x_recon.double().sum()), rec_patch=x_recon.detach()[0, 0, 30:34, 400:408].numpy(), seed=70)
    names = []
    for k, p in m.named_parameters():
        if k.startswith("discriminator.") or p.grad is None:
            continue
        nrm, samp, pos = grad_probe(k, p.grad.numpy())
        key = k.replace(".", "__")
        out["n__" + key], out["s__" + key], out["p__" + key] = nrm, samp, pos
        names.append(k)
    out["names"] = np.array(names)
    save("vqvae_grad", **out)


def gen_melgan():
    """MelGAN generator (vocoder/modules.py:38-79): the module imports librosa at the top only for a filter-bank
    helper that the Generator never calls, so a stand-in satisfies the import."""
    lib = _stub("librosa")
    lib.filters = _stub("librosa.filters", mel=lambda *a, **k: None)
    sys.path.insert(0, REF)
    import vocoder.modules as ref_voc

    hp = dict(input_size=80, ngf=8, n_residual_layers=3)
    g = ref_voc.Generator(hp["input_size"], hp["ngf"], hp["n_residual_layers"]).eval()
    sd = synth.melgan_state_dict(61, **hp)
    assert [k for k in g.state_dict().keys()] == list(sd.keys()), "key names / order of the weight-normed generator"
    load_sd(g, sd)
    x = synth.normal(62, (2, 80, 12), 1.0)
    with torch.no_grad():
        y = g(t(x))
        blk = g.model[4]                                   # first ResnetBlock (64 channels at ngf=8), dilation 1
        xb = synth.normal(63, (2, hp["ngf"] * 8, 40), 1.0)
        yb = blk(t(xb))
    assert tuple(y.shape) == (2, 1, 12 * 256)
    save("melgan_small", x=x, y=y.numpy(), xb=xb, yb=yb.numpy(), seed=61, ngf=hp["ngf"],
         sd_keys=np.array(list(sd.keys())))


def gen_mel():
    """M2 + the host half of M3 from the REAL reference (feature_extraction/extract_mel_spectrogram.py).  librosa
    0.8.1 is absent, so its three entry points are stand-ins that do NOT pretend to be librosa: `filters.mel` and
    `stft` return seeded arrays (recorded), `load` returns a seeded f32 waveform.  Everything else executes the
    reference's own code: the eight small transform classes :40-127 chained by TRANSFORMS :141-151, `np.abs(.)**1`
    and `np.dot(mel_basis, spec)` of MelSpectrogram.__call__ :36-37, and get_spectrogram's pad / truncate + dtype
    rule :166-173 with the file naming of :183-187."""
    print("mel transforms / get_spectrogram host rule")
    import tempfile

    calls = {}
    basis = synth.standin_mel_basis()

    def fake_mel(**kw):
        calls["mel_kwargs"] = kw
        return basis

    def fake_stft(x, **kw):
        calls["stft_x"] = x
        calls["stft_kwargs"] = kw
        return synth.standin_stft(x, kw["hop_length"])

    def fake_load(path, sr="unset"):
        calls["load_sr"] = sr
        return synth.standin_wav(os.path.basename(path).split("_")[0]), 22050

    lib = _stub("librosa", stft=fake_stft, load=fake_load)
    lib.filters = _stub("librosa.filters", mel=fake_mel)
    sys.path.insert(0, REF)
    sys.modules.pop("feature_extraction.extract_mel_spectrogram", None)
    import feature_extraction.extract_mel_spectrogram as ref_mel

    out = {"mel_kwargs": np.array(sorted(f"{k}={v}" for k, v in calls["mel_kwargs"].items()))}
    out["chain"] = np.array([type(f).__name__ for f in ref_mel.TRANSFORMS.transforms])
    # (1) M2: the tail TRANSFORMS.transforms[1:] on synthetic mel matrices, f32 and f64, 1e-8 .. 1e3 with exact zeros
    tail = _Sequential(ref_mel.TRANSFORMS.transforms[1:])
    m64 = synth.mel_matrix()
    o64, o32 = tail(m64), tail(m64.astype(np.float32))
    out.update(mel_matrix_seed=910, out64=o64, out32=o32)
    out["out_dtypes"] = np.array([str(o64.dtype), str(o32.dtype)])
    # (2) M1's own two numpy lines + the tail, on the recorded stand-in STFT / basis (whole TRANSFORMS)
    for tag in ("short", "long", "exact"):
        y, mel = ref_mel.get_spectrogram(f"/nowhere/{tag}_clip.wav", None, 220500, save_results=False)
        wav = synth.standin_wav(tag)
        assert calls["stft_x"] is y
        out[f"{tag}.wav_len"] = len(wav)
        out[f"{tag}.y_dtype"] = str(y.dtype)
        out[f"{tag}.y_equals_wav_prefix"] = bool(np.array_equal(y[:min(len(wav), 220500)], wav[:220500]))
        out[f"{tag}.y_tail_abs_max"] = float(np.abs(y[len(wav):]).max()) if len(wav) < 220500 else 0.0
        out[f"{tag}.mel"] = mel
        out[f"{tag}.mel_dtype"] = str(mel.dtype)
    out["stft_kwargs"] = np.array(sorted(f"{k}={v}" for k, v in calls["stft_kwargs"].items()))
    out["load_sr"] = str(calls["load_sr"])
    # (3) save_results=True: the file the reference writes
    with tempfile.TemporaryDirectory() as d:
        r = ref_mel.get_spectrogram("/nowhere/short_clip.wav", os.path.join(d, "melspec_10s_22050hz"), 220500)
        assert r is None
        names = sorted(os.listdir(os.path.join(d, "melspec_10s_22050hz")))
        saved = np.load(os.path.join(d, "melspec_10s_22050hz", names[0]))
        out["saved_names"] = np.array(names)
        out["saved_shape"] = np.array(saved.shape)
        out["saved_dtype"] = str(saved.dtype)
        assert np.array_equal(saved, out["short.mel"])
    save("mel_transforms", **out)


def main():
    torch.manual_seed(synth.SEED)
    torch.set_num_threads(8)
    ref_gpt, ref_enc, ref_dec, ref_vq = import_reference()
    which = set(sys.argv[1:]) or {"vq", "attn", "gpt", "lit", "vae", "vae_mi", "vae_steps", "vae_xl", "vqvae", "resblock", "vqvae_grad", "melgan", "mel"}
    if which <= {"melgan", "mel"}:
        if "melgan" in which:
            gen_melgan()
        if "mel" in which:
            gen_mel()
        assert not any("__pycache__" in d for d, _, _ in os.walk(REF)), "bytecode leaked into reference"
        return
    if "vq" in which:
        gen_vq(ref_vq)
    if "attn" in which:
        gen_attention(ref_gpt)
    if "gpt" in which:
        gen_gpt(ref_gpt)
    if "lit" in which:
        gen_lit(ref_gpt)
    if "vae" in which:
        gen_vae(ref_gpt, ref_enc, ref_dec)
    if "vae_mi" in which:
        gen_vae_mi(ref_enc)
    if "vae_steps" in which:
        gen_vae_steps()
    if "vae_xl" in which:
        gen_vae_xl(ref_enc, ref_dec)
    if "vqvae" in which:
        gen_vqvae(ref_vq)
    if "resblock" in which:
        gen_resblock_grad(ref_vq)
    if "vqvae_grad" in which:
        gen_vqvae_grad(ref_vq)
    if "melgan" in which:
        gen_melgan()
    if "mel" in which:
        gen_mel()
    assert not any("__pycache__" in d for d, _, _ in os.walk(REF)), "bytecode leaked into reference"


if __name__ == "__main__":
    main()
