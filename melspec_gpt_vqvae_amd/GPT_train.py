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


def _common_flags(parser):
    parser.add_argument('--dataset', type=str, required=True, help='dataset to use')
    parser.add_argument('--experiment', type=str, required=True, help='experiment name')
    parser.add_argument('--train', type=int, default=False, help='start training process')
    parser.add_argument('--resume', type=str, default=None, help='resume_from the checkpoint')
    parser.add_argument('--workers', type=int, default=1, help='number of workers for data')
    parser.add_argument('--eval', type=int, default=False, help='evaluate model')
    parser.add_argument('--test', type=int, default=False, help='test model')
    parser.add_argument('--logging_frequency', type=int, default=200, help='number of steps for text logging')
    parser.add_argument('--test_interpolation', type=int, default=False)
    parser.add_argument('--reconstruct_spec', type=str, default='', help="model ckpt for mel-spectrograms reconstuction")
    parser.add_argument('--vocoder', type=str, default='', help="model ckpt for vocoder for audio reconstuction")
    # not in the reference: overrides of config entries + where the split lists and logs live
    for name, typ in (("epochs", int), ("batch_size", int), ("n_layer", int), ("n_head", int), ("n_embd", int),
                      ("spec_dir_path", str), ("learning_rate", float)):
        parser.add_argument("--" + name, type=typ, default=None)
    parser.add_argument('--splits_dir', type=str, default='./data')
    parser.add_argument('--log_root', type=str, default='lightning_logs')
    parser.add_argument('--dtype', choices=["f32", "bf16", "fp16"], default="f32", help="kernel numerics lane")
    parser.add_argument('--max_steps_per_epoch', type=int, default=None)


def merge_config(args, set_name):
    """argparse namespace + config set -> one namespace (GPT_train.py:63-66); explicit overrides win."""
    over = {k: v for k, v in vars(args).items() if v is not None}
    base = _config.params(set_name)
    merged = {**{k: v for k, v in vars(args).items() if k not in base}, **base}
    merged.update({k: v for k, v in over.items() if k in base})
    return argparse.Namespace(**merged)


def seed_all(seed):
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)


def init_config(argv=None):
    parser = argparse.ArgumentParser(description='GPT transformer for VQVAE_spec')
    _common_flags(parser)
    args = parser.parse_args(argv)
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
        torch.cuda.set_device(local)
        args.device = f"cuda:{local}"
    else:
        args.device = "cpu"
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
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
