#!/usr/bin/env python3
"""`python tools/train.py gpt --dataset vas --experiment x --train 1 ...` / `python tools/train.py gpt_vae ...`:
command-line front of melspec_gpt_vqvae_amd.GPT_train / GPT_VAE_train (the reference's two entry scripts)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else ""
    if which == "gpt":
        from melspec_gpt_vqvae_amd import GPT_train as entry
    elif which == "gpt_vae":
        from melspec_gpt_vqvae_amd import GPT_VAE_train as entry
    else:
        raise SystemExit(__doc__)
    entry.main(entry.init_config(sys.argv[2:]))
