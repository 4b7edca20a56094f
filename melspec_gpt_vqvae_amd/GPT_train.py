"""Entry point with the flags of the reference's GPT_train.py (:25-68): `--dataset vas --experiment NAME --train 1
[--resume CKPT] [--workers N] [--logging_frequency N] [--reconstruct_spec VQVAE.ckpt] [--vocoder CKPT]`.  The config
set `GPT_<dataset>` supplies the model / data parameters (config/), the seed is the reference's 783435 (:56-61).
`pl.Trainer.fit` is replaced by trainer.Fit (no Lightning in the loop); extra flags - all optional - override config
entries for smoke runs (`--epochs`, `--batch_size`, `--n_layer`, `--spec_dir_path`, `--splits_dir`, `--dtype`).
Launch N ranks with `python -m torch.distributed.run --nproc-per-node N -m melspec_gpt_vqvae_amd.GPT_train ...`."""
from __future__ import annotations

import argparse
import os

import numpy as np
import torch

from . import config as _config

SEED = 783435


REQUIRED = object()
# (flag, type, default, help) - names and defaults are the reference's command line (GPT_train.py:25-50); the ints that
# act as switches (train / eval / test) default to False there, kept
REFERENCE_FLAGS = [
    ("dataset", str, REQUIRED, "config set GPT_<dataset> / GPT_VAE_<dataset>"),
    ("experiment", str, REQUIRED, "run name: logs and checkpoints go under <log_root>/<experiment>-<dataset>"),
    ("train", int, False, "1 = run the training loop"),
    ("resume", str, None, "checkpoint to continue from"),
    ("workers", int, 1, "DataLoader worker processes"),
    ("eval", int, False, "1 = run validation and print val/loss"),
    ("test", int, False, "kept for command-line compatibility"),
    ("logging_frequency", int, 200, "steps between text log lines"),
    ("test_interpolation", int, False, "kept for command-line compatibility"),
    ("reconstruct_spec", str, "", "VQ-VAE checkpoint (LitVQVAE state_dict) for decode_to_img"),
    ("vocoder", str, "", "MelGAN checkpoint for audio reconstruction"),
]
# not in the reference: overrides of config entries, where split lists / logs live, numerics lane, smoke-run cap
LOCAL_FLAGS = [(n, t, None, "override of the config entry") for n, t in
               (("epochs", int), ("batch_size", int), ("n_layer", int), ("n_head", int), ("n_embd", int),
                ("spec_dir_path", str), ("learning_rate", float))] + [
    ("splits_dir", str, "./data", "directory of the vas_*/vggsound_* split lists"),
    ("log_root", str, "lightning_logs", "root of logs and checkpoints"),
    ("max_steps_per_epoch", int, None, "cap on steps per epoch (smoke runs)"),
]


def add_flags(parser, table):
    for name, typ, default, text in table:
        kw = {"required": True} if default is REQUIRED else {"default": default}
        parser.add_argument("--" + name, type=typ, help=text, **kw)


def _common_flags(parser):
    add_flags(parser, REFERENCE_FLAGS)
    add_flags(parser, LOCAL_FLAGS)
    parser.add_argument("--dtype", choices=["f32", "bf16", "fp16"], default="f32", help="kernel numerics lane")


def merge_config(args, set_name):
    """argparse namespace + config set -> one namespace (GPT_train.py:63-66); explicit overrides win."""
    over = {k: v for k, v in vars(args).items() if v is not None}
    base = _config.params(set_name)
    merged = {**{k: v for k, v in vars(args).items() if k not in base}, **base}
    merged.update({k: v for k, v in over.items() if k in base})
    return argparse.Namespace(**merged)


def select_half(dtype):
    """`--dtype bf16|fp16`: the 16-bit format is a property of the LIBRARY flavour a process loads (_ffi.HALF, fixed at
    import from MELGPT_HALF).  Called by init_config before anything imports `_ffi`: sets MELGPT_HALF from the flag, or
    refuses when the process already holds the other flavour (silently training in the wrong format is not an option)."""
    import sys

    if dtype not in ("bf16", "fp16"):
        return
    loaded = sys.modules.get(__package__ + "._ffi")
    if loaded is None:
        os.environ["MELGPT_HALF"] = dtype
    elif loaded.HALF != dtype:
        raise SystemExit(f"--dtype {dtype}: this process already loaded the {loaded.HALF} flavour of the library "
                         f"(MELGPT_HALF={loaded.HALF}); start a fresh process or export MELGPT_HALF={dtype}")


def seed_all(seed):
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)


def init_config(argv=None):
    parser = argparse.ArgumentParser(description='GPT transformer for VQVAE_spec')
    _common_flags(parser)
    args = parser.parse_args(argv)
    select_half(args.dtype)
    args.cuda = torch.cuda.is_available()
    args.seed = SEED
    seed_all(args.seed)
    return merge_config(args, "GPT_%s" % args.dataset)


def init_distributed(args):
    """one process per GPU: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the launcher"""
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.cuda:
        gpus = getattr(args, "gpus", None)          # GPT_VAE_train's `--gpus 0 1 ...`: rank r drives the r-th listed GPU
        if isinstance(gpus, (list, tuple)) and world > 1 and len(gpus) == world:
            local = int(gpus[local])
        torch.cuda.set_device(local)
        args.device = f"cuda:{local}"
    else:
        args.device = "cpu"
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.cuda:
            from .dp import pin_rccl_channels

            pin_rccl_channels()
        dist.init_process_group("nccl" if args.cuda else "gloo")
    return world


def main(args):
    from . import _ffi
    from .trainer import Fit
    from .transformer.minGPT import Lit_minGPT, set_compute_dtype

    init_distributed(args)
    if not args.cuda:
        raise SystemExit("melspec_gpt_vqvae_amd runs on an MI355X only (there is no CPU path)")
    gpt = Lit_minGPT(args)
    set_compute_dtype(gpt.transformer, _ffi.HALF_DTYPE if args.dtype in ("bf16", "fp16") else torch.float32)
    fit = Fit(gpt, args)
    hist = None
    if args.train:
        hist = fit.fit(ckpt_path=args.resume, max_steps_per_epoch=args.max_steps_per_epoch)
    if args.eval == 1:
        if args.resume and not args.train:
            fit.resume(args.resume)
        val = fit.validate()
        print(f"val/loss {val}")
    return fit, hist


if __name__ == '__main__':
    main(init_config())
