"""Entry point with the flags of the reference's GPT_VAE_train.py (:28-116): the GPT-VAE (GPTEncoder + GPTDecoder) on
VQ-code sequences, the one script the reference runs under DDP (:166-190, strategy "ddp_find_unused_parameters_false").
Here: one process per GPU - `--gpus 0 1 2 3` starts them itself (launch.spawn_ranks, where Lightning's DDP launcher
stands), or torch.distributed.run does - and gradients are exchanged by dp.DataParallel over RCCL.
`--load_path CKPT` warm-starts the encoder from a stage-1 checkpoint (the `"encoder" in k` filter of :131-144)."""
from __future__ import annotations

import argparse

import torch

from .GPT_train import _common_flags, add_flags, init_distributed, merge_config, seed_all, select_half


# (flag, type, default, help): the reference's extra command line for the VAE (GPT_VAE_train.py:33-97)
VAE_FLAGS = [
    ("num_nodes", int, 1, "kept for command-line compatibility (one node here)"),
    ("momentum", float, 0, "unused by AdamW; kept"),
    ("lr", float, 1.0, "unused (learning_rate comes from the config set); kept"),
    ("nsamples", int, 1, "latent samples per sequence in training"),
    ("iw_train_nsamples", int, -1, "importance-weighted training samples (-1: off)"),
    ("iw_train_ns", int, 1, "kept"),
    ("iw_nsamples", int, 500, "kept"),
    ("load_path", str, "", "stage-1 checkpoint whose encoder weights warm-start this model"),
    ("reconstruct_from", str, "", "kept"),
    ("reconstruct_to", str, "decoding.txt", "kept"),
    ("warm_up", int, 10, "epochs over which the KL weight anneals to 1"),
    ("kl_start", float, 1.0, "initial KL weight"),
    ("seed", int, 783435, "random seed"),
    ("save_latent", int, 0, "kept"),
    ("fix_var", float, -1, "fixed posterior variance (> 0 enables)"),
    ("freeze_epoch", int, -1, "kept"),
    ("beta", float, 1.0, "0 = plain autoencoder objective"),
    ("fb", int, 0, "free bits: 0 off, 1 per sequence, 2 per latent dimension, 3 on the batch mean"),
    ("target_kl", float, -1, "free-bits threshold"),
]


def init_config(argv=None):
    parser = argparse.ArgumentParser(description='VAE mode collapse study')
    _common_flags(parser)
    add_flags(parser, VAE_FLAGS)
    parser.add_argument('--gpus', nargs='+', type=int, default=[0], help='kept (the launcher decides which GPUs)')
    parser.add_argument('--opt', choices=['sgd', 'adam'], default='sgd', help='kept (AdamW is used, as in the reference)')
    parser.add_argument('--decoding_strategy', choices=['greedy', 'beam', 'sample'], default='greedy')
    parser.set_defaults(logging_frequency=500)
    args = parser.parse_args(argv)
    from .launch import launched_by_a_launcher, spawn_ranks
    if len(args.gpus) > 1 and not launched_by_a_launcher():
        # `devices=args.gpus` of the reference's pl.Trainer (:172): one fresh rank per listed GPU, started before this
        # process touches a GPU; rank r drives args.gpus[r] (init_distributed)
        import sys
        raise SystemExit(spawn_ranks(["-m", __spec__.name] + list(sys.argv[1:] if argv is None else argv), len(args.gpus)))
    select_half(args.dtype)
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
