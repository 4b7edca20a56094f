"""Host side of the entry points (no GPU): flag names and defaults of the reference's GPT_train.py:25-68 /
GPT_VAE_train.py:28-116, config sets of config/*.py, override rule."""
import pytest


def test_gpt_train_flags_and_config_merge():
    from melspec_gpt_vqvae_amd import GPT_train, config

    a = GPT_train.init_config(["--dataset", "vas", "--experiment", "e"])
    for k, v in dict(train=False, resume=None, workers=1, eval=False, test=False, logging_frequency=200,
                     reconstruct_spec="", vocoder="", seed=783435).items():
        assert getattr(a, k) == v, k
    p = config.params("GPT_vas")
    assert p == dict(vocab_size=128, block_size=266, n_layer=24, n_head=16, n_embd=1024, class_size=8,
                     learning_rate=1e-6, epochs=300, batch_size=8,
                     spec_dir_path="./data/vas/features/*/melspec_10s_22050hz", sample_rate=22050, embd_pdrop=0.5,
                     resid_pdrop=0.5, attn_pdrop=0.5, n_unmasked=0, last_linear=None)
    for k, v in p.items():
        assert getattr(a, k) == v, k
    b = GPT_train.init_config(["--dataset", "vas", "--experiment", "e", "--batch_size", "2", "--epochs", "1"])
    assert b.batch_size == 2 and b.epochs == 1 and b.n_layer == 24
    with pytest.raises(KeyError):
        GPT_train.init_config(["--dataset", "nope", "--experiment", "e"])


def test_gpt_vae_train_flags():
    from melspec_gpt_vqvae_amd import GPT_VAE_train

    a = GPT_VAE_train.init_config(["--dataset", "vggsound", "--experiment", "e"])
    for k, v in dict(nsamples=1, iw_train_nsamples=-1, warm_up=10, kl_start=1.0, seed=783435, fix_var=-1, beta=1.0,
                     fb=0, target_kl=-1, logging_frequency=500, load_path="", gpus=[0], num_nodes=1,
                     label=False).items():
        assert getattr(a, k) == v, k
    # config_GPT_VAE_vggsound.py:43-58 (GPT-XL)
    assert (a.vocab_size, a.block_size, a.n_layer, a.n_head, a.n_embd, a.batch_size) == (1024, 265, 40, 23, 1472, 1)
    assert (a.embd_pdrop, a.resid_pdrop, a.attn_pdrop, a.learning_rate) == (0.0, 0.0, 0.0, 1e-6)
    v = GPT_VAE_train.init_config(["--dataset", "vas", "--experiment", "e", "--fb", "2", "--target_kl", "8"])
    assert (v.n_layer, v.n_embd, v.batch_size, v.embd_pdrop, v.fb, v.target_kl) == (24, 1024, 24, 0.3, 2, 8.0)


def test_fp16_loss_scale_grows_only_after_the_step_that_used_the_old_scale(monkeypatch):
    """Dynamic loss scaling (trainer.Fit): the optimizer step that follows the 200th clean backward must still see the
    grad_scale of the scale that backward ran at; the doubled scale applies from the next backward on.  An overflow
    halves the scale at once (that step is skipped)."""
    import types

    import torch

    from melspec_gpt_vqvae_amd import flat, ops, trainer

    class _FP:
        grad = torch.zeros(4)

        def zero_missing_grads(self):
            pass

    finite = {"v": True}
    monkeypatch.setattr(flat, "ensure_flat", lambda m: _FP())
    monkeypatch.setattr(ops, "sum_f32", lambda g: torch.tensor(0.0 if finite["v"] else float("inf")))
    f = trainer.Fit.__new__(trainer.Fit)
    f.fp16, f.world, f.loss_scale, f._good_steps, f.skipped_steps, f.target = True, 2, 4096.0, 198, 0, None
    f.opt = types.SimpleNamespace(grad_scale=1.0 / (2 * 4096.0))
    seen = []
    for _ in range(3):                       # clean steps 199, 200, 201
        scale_at_backward = f.loss_scale
        assert f._step_ok()
        seen.append((scale_at_backward, f.opt.grad_scale))   # what opt.step() would use
        f._grow_loss_scale()
    assert seen == [(4096.0, 1 / 8192.0), (4096.0, 1 / 8192.0), (8192.0, 1 / 16384.0)]
    assert f.loss_scale == 8192.0 and f._good_steps == 1
    finite["v"] = False
    assert not f._step_ok()
    assert f.loss_scale == 4096.0 and f.opt.grad_scale == 1 / 8192.0 and f.skipped_steps == 1 and f._good_steps == 0
    f._grow_loss_scale()
    assert f.loss_scale == 4096.0
