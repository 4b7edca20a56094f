"""Hyper-parameter sets of the reference's config/ directory (data, not code): `params(name)` returns a fresh dict for
"GPT_vas" (config/config_GPT_vas.py:1-18), "GPT_VAE_vas" (config_GPT_VAE_vas.py:1-17) and "GPT_VAE_vggsound"
(config_GPT_VAE_vggsound.py:43-58, the active GPT-XL block; the commented-out medium / large variants are
"GPT_VAE_vggsound_medium" / "_large")."""
from __future__ import annotations

_COMMON = dict(learning_rate=1e-6, sample_rate=22050, n_unmasked=0, last_linear=None)

_SETS = {
    "GPT_vas": dict(vocab_size=128, block_size=266, n_layer=24, n_head=16, n_embd=1024, class_size=8, epochs=300,
                    batch_size=8, spec_dir_path="./data/vas/features/*/melspec_10s_22050hz", embd_pdrop=0.5,
                    resid_pdrop=0.5, attn_pdrop=0.5),
    "GPT_VAE_vas": dict(vocab_size=128, block_size=265, n_layer=24, n_head=16, n_embd=1024, epochs=10000, batch_size=24,
                        spec_dir_path="./data/vas/features/*/melspec_10s_22050hz", embd_pdrop=0.3, resid_pdrop=0.3,
                        attn_pdrop=0.3),
    "GPT_VAE_vggsound": dict(vocab_size=1024, block_size=265, n_layer=40, n_head=23, n_embd=1472, epochs=10000,
                             batch_size=1, spec_dir_path="./data/vggsound/melspec_10s_22050hz/", embd_pdrop=0.0,
                             resid_pdrop=0.0, attn_pdrop=0.0),
    "GPT_VAE_vggsound_large": dict(vocab_size=1024, block_size=265, n_layer=36, n_head=20, n_embd=1280, epochs=10000,
                                   batch_size=7, spec_dir_path="./data/vggsound/melspec_10s_22050hz/", embd_pdrop=0.0,
                                   resid_pdrop=0.0, attn_pdrop=0.0),
    "GPT_VAE_vggsound_medium": dict(vocab_size=1024, block_size=265, n_layer=24, n_head=16, n_embd=1024, epochs=10000,
                                    batch_size=32, spec_dir_path="./data/vggsound/melspec_10s_22050hz/", embd_pdrop=0.0,
                                    resid_pdrop=0.0, attn_pdrop=0.0),
}


def params(name: str) -> dict:
    if name not in _SETS:
        raise KeyError(f"no config set {name!r}; have {sorted(_SETS)}")
    return {**_COMMON, **_SETS[name]}
