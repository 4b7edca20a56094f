"""Models / batches shared by tests/test_dp_gpu.py and its child ranks (test infrastructure)."""
import torch

import synth
from util import t

BATCH = 4


def build(which, dev):
    if which == "gptclass_vas16":
        # 2 layers at the VAS width in the 16-bit lane, 12 sequences = 3 180 rows: fc1 / fc2 / their dgrads are 208-tile
        # problems, i.e. they run on the persistent 256 x 256 kernel (csrc/gemm.hip pick_tile) - the one that draws claimed
        # tiles while an all-reduce can be in flight
        from melspec_gpt_vqvae_amd import _ffi
        from melspec_gpt_vqvae_amd.transformer.minGPT import GPTClass, cross_entropy, set_compute_dtype

        args = synth.gpt_args(n_layer=2, n_head=16, n_embd=1024)
        m = GPTClass(args)
        m.load_state_dict({k: t(v) for k, v in synth.gpt_state_dict(args, 3).items()}, strict=False)
        m.to(dev).train()
        set_compute_dtype(m, _ffi.HALF_DTYPE)
        batch = {"x": t(synth.randint(920, 0, 128, (12, 265)), dev), "c": t(synth.randint(921, 0, 8, (12, 1)), dev)}

        def loss_fn(model, b):
            with model.discard_att():
                logits, _, _ = model(b["x"][:, :-1], b["c"])
            return cross_entropy(logits.reshape(-1, logits.size(-1)), b["x"].reshape(-1))

        return m, batch, loss_fn
    if which == "gptclass":
        from melspec_gpt_vqvae_amd.transformer.minGPT import GPTClass, cross_entropy

        args = synth.gpt_args(n_layer=2, n_head=4, n_embd=256)
        m = GPTClass(args)
        m.load_state_dict({k: t(v) for k, v in synth.gpt_state_dict(args, 1).items()}, strict=False)
        m.to(dev).train()
        batch = {"x": t(synth.randint(900, 0, 128, (BATCH, 265)), dev), "c": t(synth.randint(901, 0, 8, (BATCH, 1)), dev)}

        def loss_fn(model, b):
            with model.discard_att():
                logits, _, _ = model(b["x"][:, :-1], b["c"])
            return cross_entropy(logits.reshape(-1, logits.size(-1)), b["x"].reshape(-1))

        return m, batch, loss_fn
    from melspec_gpt_vqvae_amd.transformer.Lit_GPT_VAE import GPT_VAE

    args = synth.gpt_args(n_layer=2, n_head=4, n_embd=256, block_size=265, fix_var=0, kl_start=0.3, warm_up=0,
                          batch_size=BATCH, target_kl=0.0, beta=1.0, nsamples=1, fb=0, device=dev, learning_rate=1e-6)
    m = GPT_VAE(args)
    m.encoder.transformer.load_state_dict({k: t(v) for k, v in synth.gpt_state_dict(
        args, 5, block_size=265, with_embedder=False, out_features=512).items()}, strict=False)
    m.decoder.transformer.load_state_dict({k: t(v) for k, v in synth.gpt_state_dict(
        args, 6, block_size=266, with_embedder=False).items()}, strict=False)
    m.to(dev).train()
    batch = {"x": t(synth.randint(910, 0, 128, (BATCH, 265)), dev), "eps": t(synth.normal(911, (BATCH, 1, 256)), dev)}

    def loss_fn(model, b):
        total, _, _ = model.loss(b["x"], 0.3, nsamples=1, eps=b["eps"])
        return total.mean()

    return m, batch, loss_fn
