"""BASELINE configs[0] - "VAS config_GPT_vas.py, batch=2, 1 epoch with precomputed mel features (plumbing)" - end to
end through the entry point that mirrors the reference's GPT_train.py (:25-131): synthetic VAS tree on disk (8 classes
x 2 clips: `*_mel.npy` (80, 860) = clip(N(mu_f, sigma_f), 0, 1), `*_mel_code.npy` (5, 53) int64, split lists in the
`class/video` format) -> DataModule -> Lit_minGPT.training_step -> AdamW, same flag names as the reference.
Pass: logits (2, 265, 128), loss finite, evaluation loss lower after the epoch; checkpoint written in the
Lightning layout and resumable.  The codes of class k are uniform in [16k, 16k + 16) so that one epoch of 8 steps
has something to learn (uniform [0, 128) codes carry no signal: the loss floor is ln 128 = the initial loss)."""
import os

import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu
CLASSES = ["baby", "cough", "dog", "drum", "fireworks", "gun", "hammer", "sneeze"]


def make_vas_tree(root):
    rng = np.random.default_rng(783435)
    mel = synth.mel_tiles(783435, 16)
    lines = []
    for ci, cls in enumerate(CLASSES):
        md = os.path.join(root, "vas", "features", cls, "melspec_10s_22050hz")
        cd = os.path.join(root, "vas", "features", cls, "codes_10s")
        os.makedirs(md)
        os.makedirs(cd)
        for v in range(2):
            np.save(os.path.join(md, f"video_{v:05d}_mel.npy"), mel[2 * ci + v])
            np.save(os.path.join(cd, f"video_{v:05d}_mel_code.npy"), rng.integers(16 * ci, 16 * ci + 16, (5, 53)))
            lines.append(f"{cls}/video_{v:05d}")
    sp = os.path.join(root, "splits")
    os.makedirs(sp)
    for name in ("train", "valid"):
        with open(os.path.join(sp, f"vas_{name}.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
    return os.path.join(root, "vas", "features", "*", "melspec_10s_22050hz"), sp


def test_config1_vas_plumbing_one_epoch(tmp_path):
    from melspec_gpt_vqvae_amd import GPT_train
    from melspec_gpt_vqvae_amd.trainer import Fit
    from melspec_gpt_vqvae_amd.transformer.minGPT import Lit_minGPT

    spec_dir, splits = make_vas_tree(str(tmp_path))
    argv = ["--dataset", "vas", "--experiment", "plumbing", "--train", "1", "--workers", "0", "--epochs", "1",
            "--batch_size", "2", "--n_layer", "2", "--learning_rate", "1e-4", "--spec_dir_path", spec_dir,
            "--splits_dir", splits, "--log_root", os.path.join(str(tmp_path), "lightning_logs")]
    args = GPT_train.init_config(argv)
    # everything else is the reference's VAS config (config/config_GPT_vas.py)
    assert (args.vocab_size, args.block_size, args.n_head, args.n_embd, args.class_size) == (128, 266, 16, 1024, 8)
    assert (args.embd_pdrop, args.resid_pdrop, args.attn_pdrop) == (0.5, 0.5, 0.5) and args.seed == 783435
    args.device = "cuda:0"
    lit = Lit_minGPT(args)
    assert len(lit.data.train_dataset) == 16 and lit.data.train_dataset.label2target == {c: i for i, c in enumerate(CLASSES)}
    fit = Fit(lit, args)
    batch = next(iter(lit.train_dataloader()))
    assert batch["image"].shape == (2, 80, 848) and batch["codes"].shape == (2, 5, 53) and batch["target"].shape == (2,)
    lit.eval()
    dev_batch = {k: (v.to("cuda:0") if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    x, c = lit.get_xc(dev_batch)
    logits, target = lit(x, c)
    assert logits.shape == (2, 265, 128) and target.shape == (2, 265)
    val0 = fit.validate()
    hist = fit.fit(epochs=1)
    assert hist["steps"] == 8 and len(hist["train_loss"][0]) == 8 and np.isfinite(hist["train_loss"][0]).all()
    val1 = hist["val_loss"][-1]
    assert np.isfinite(val1) and val1 < val0 - 0.5, (val0, val1)       # CPU oracle with the same recipe: 5.15 -> 3.4
    ck_dir = fit.checkpoint_dir()
    files = sorted(os.listdir(ck_dir))
    assert "last.ckpt" in files and any(f.startswith("vas-model-epoch=00-loss=") for f in files)
    ck = torch.load(os.path.join(ck_dir, "last.ckpt"), map_location="cpu", weights_only=False)
    assert ck["epoch"] == 0 and ck["global_step"] == 8
    assert "transformer.blocks.1.attn.key.weight" in ck["state_dict"] and "transformer.embedder.weight" in ck["state_dict"]
    # resume: a fresh module picks up weights, optimizer moments and the step count; its evaluation loss is val1
    torch.manual_seed(1)
    lit2 = Lit_minGPT(args)
    fit2 = Fit(lit2, args)
    assert fit2.resume(os.path.join(ck_dir, "last.ckpt")) == 1 and fit2.global_step == 8
    assert abs(fit2.validate() - val1) < 1e-5
    assert torch.equal(fit2.opt._m, fit.opt._m) and torch.equal(fit2.opt._v, fit.opt._v) and fit2.opt.step_count == 8
    # `optimizer_states` is torch.optim.AdamW's layout (what Lightning stores for the reference): it loads into the
    # reference-style optimizer of configure_optimizers as it stands
    ost = ck["optimizer_states"][0]
    assert set(ost) >= {"state", "param_groups"} and len(ost["param_groups"]) == 2
    topt = lit2.configure_optimizers()
    topt.load_state_dict(ost)
    name0 = ost["param_names"][0]
    p0 = dict(lit2.transformer.named_parameters())[name0]
    assert torch.equal(topt.state[p0]["exp_avg"].cpu(), ost["state"][0]["exp_avg"])
    # a checkpoint that does not fit is refused instead of resuming from random weights
    bad = dict(ck, state_dict={k: v for k, v in ck["state_dict"].items() if "blocks.1." not in k})
    torch.save(bad, os.path.join(str(tmp_path), "bad.ckpt"))
    with pytest.raises(RuntimeError, match="does not fit"):
        Fit(Lit_minGPT(args), args).resume(os.path.join(str(tmp_path), "bad.ckpt"))


def test_entry_point_main_runs_bf16(tmp_path):
    """the module's main(): flags -> model -> loop, on the bf16 lane"""
    from melspec_gpt_vqvae_amd import GPT_train

    spec_dir, splits = make_vas_tree(str(tmp_path))
    args = GPT_train.init_config(["--dataset", "vas", "--experiment", "bf16", "--train", "1", "--workers", "0",
                                  "--epochs", "1", "--batch_size", "2", "--n_layer", "2", "--dtype", "bf16",
                                  "--learning_rate", "1e-4", "--spec_dir_path", spec_dir, "--splits_dir", splits,
                                  "--max_steps_per_epoch", "3",
                                  "--log_root", os.path.join(str(tmp_path), "lightning_logs")])
    fit, hist = GPT_train.main(args)
    assert hist["steps"] == 3 and np.isfinite(hist["train_loss"][0]).all() and np.isfinite(hist["val_loss"][-1])
