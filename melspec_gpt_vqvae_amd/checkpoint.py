"""Checkpoint key conventions of the reference: the build's modules keep the reference's parameter
names, so a reference checkpoint loads by name; what differs between files is the wrapping -
  * a bare `state_dict` saved with torch.save (the VQ-VAE codebook file read by extract_codes.py:77-78,
    Lit_minGPT / GPT_VAE constructors loading `args.vqvae_ckpt`),
  * a Lightning checkpoint: {'state_dict': {...}, 'epoch': ..., ...} whose keys carry the attribute path of the
    LightningModule (`transformer.blocks.0...`, `first_stage_model._encoder...`, `encoder.transformer...`),
  * the stage-2 warm start of GPT_VAE_train.py:131-144: only the keys containing "encoder" are taken from a
    Lightning checkpoint and loaded with strict=False.
Nothing here touches the GPU."""
from __future__ import annotations

import torch


def read_state_dict(path_or_dict, map_location="cpu"):
    """-> flat {name: tensor} from a file or an already loaded object (bare state_dict or Lightning checkpoint)."""
    obj = path_or_dict
    if not isinstance(obj, dict):
        obj = torch.load(obj, map_location=map_location, weights_only=False)
    if isinstance(obj, dict) and "state_dict" in obj and isinstance(obj["state_dict"], dict):
        obj = obj["state_dict"]
    return obj


def strip_prefix(sd, prefix):
    """keys under `prefix` (e.g. 'transformer.' or 'first_stage_model.') with the prefix removed."""
    if prefix and not prefix.endswith("."):
        prefix += "."
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def load_state_dict_any(module, path_or_dict, prefix=None, strict=True):
    """load a reference checkpoint into `module`.  prefix=None tries the keys as they are, then every attribute-path
    prefix found in the file whose stripped keys fit the module best (Lightning wrapping).  The persistent
    `attn.mask` buffers (checkpoint ABI only) may be absent from older files; they are never required."""
    sd = read_state_dict(path_or_dict)
    own = set(module.state_dict().keys())
    if prefix is not None:
        cand = strip_prefix(sd, prefix)
    else:
        cand, best = sd, len(own & set(sd.keys()))
        prefixes = sorted({k[:i + 1] for k in sd for i, ch in enumerate(k) if ch == "."})
        for pre in prefixes:
            c = strip_prefix(sd, pre)
            hit = len(own & set(c.keys()))
            if hit > best:
                cand, best = c, hit
    res = module.load_state_dict(cand, strict=False)
    missing = [k for k in res.missing_keys if not k.endswith("attn.mask")]
    if strict and (missing or res.unexpected_keys):
        raise RuntimeError(f"checkpoint does not fit {type(module).__name__}: missing {missing[:6]} "
                           f"unexpected {list(res.unexpected_keys)[:6]}")
    return res


def warm_start_encoder(vae, path_or_dict):
    """GPT_VAE_train.py:131-144: take every key containing "encoder" from a Lightning checkpoint's state_dict and load
    it non-strictly into the GPT-VAE (its `encoder.*` sub-module)."""
    sd = read_state_dict(path_or_dict)
    enc = {k: v for k, v in sd.items() if "encoder" in k}
    return vae.load_state_dict(enc, strict=False)
