"""Entry point with the flags of the reference's GPT_VAE_train.py (:28-116): the GPT-VAE (GPTEncoder + GPTDecoder) on
VQ-code sequences, the one script the reference runs under DDP (:166-190, strategy "ddp_find_unused_parameters_false").
Here: one process per GPU started by torch.distributed.run, gradients exchanged by dp.DataParallel over RCCL.
`--load_path CKPT` warm-starts the encoder from a stage-1 checkpoint (the `"encoder" in k` filter of :131-144)."""
from __future__ import annotations

import argparse

import torch

from .GPT_train import _common_flags, init_distributed, merge_config, seed_all


def init_config(argv=None):
    parser = argparse.ArgumentParser(description='VAE mode collapse study')
    _common_flags(parser)
    parser.add_argument('--gpus', nargs='+', type=int, default=[0], help='GPU device IDs (the launcher decides here)')
    parser.add_argument('--num_nodes', type=int, default=1)
    parser.add_argument('--momentum', type=float, default=0)
    parser.add_argument('--opt', type=str, choices=["sgd", "adam"], default="sgd")
    parser.add_argument('--lr', type=float, default=1.0)
    parser.add_argument('--nsamples', type=int, default=1, help='number of iw samples for training')
    parser.add_argument('--iw_train_nsamples', type=int, default=-1)
    parser.add_argument('--iw_train_ns', type=int, default=1)
    parser.add_argument('--iw_nsamples', type=int, default=500)
    parser.add_argument('--load_path', type=str, default='')
    parser.add_argument('--reconstruct_from', type=str, default='')
    parser.add_argument('--reconstruct_to', type=str, default="decoding.txt")
    parser.add_argument('--decoding_strategy', type=str, choices=["greedy", "beam", "sample"], default="greedy")
    parser.add_argument('--warm_up', type=int, default=10, help="number of annealing epochs")
    parser.add_argument('--kl_start', type=float, default=1.0, help="starting KL weight")
    parser.add_argument('--seed', type=int, default=783435)
    parser.add_argument("--save_latent", type=int, default=0)
    parser.add_argument("--fix_var", type=float, default=-1)
    parser.add_argument("--freeze_epoch", type=int, default=-1)
    parser.add_argument("--beta", type=float, default=1.0, help="0 = plain autoencoder")
    parser.add_argument("--fb", type=int, default=0, help="0: no fb; 1: fb; 2: max(target_kl, kl) for each dimension")
    parser.add_argument("--target_kl", type=float, default=-1, help="target kl of the free bits trick")
    parser.set_defaults(logging_frequency=500)
    args = parser.parse_args(argv)
    args.cuda = torch.cuda.is_available()
    seed_all(args.seed)
    args = merge_config(args, "GPT_VAE_%s" % args.dataset)
    args.label = getattr(args, "label", False)
    return args


def main(args):
    from . import _ffi
    from .checkpoint import warm_start_encoder
    from .trainer import Fit
    from .transformer.Lit_GPT_VAE import GPT_VAE
    from .transformer.minGPT import set_compute_dtype

    init_distributed(args)
    if not args.cuda:
        raise SystemExit("melspec_gpt_vqvae_amd runs on an MI355X only (there is no CPU path)")
    vae = GPT_VAE(args)
    if args.load_path:
        warm_start_encoder(vae, args.load_path)
    set_compute_dtype(vae, _ffi.HALF_DTYPE if args.dtype in ("bf16", "fp16") else torch.float32)
    fit = Fit(vae, args)
    hist = None
    if args.train:
        hist = fit.fit(ckpt_path=args.resume, max_steps_per_epoch=args.max_steps_per_epoch)
    if args.eval == 1:
        if args.resume and not args.train:
            fit.resume(args.resume)
        print(f"val/loss {fit.validate()}")
    return fit, hist


if __name__ == '__main__':
    main(init_config())
