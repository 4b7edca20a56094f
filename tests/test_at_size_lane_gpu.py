"""The bench lane AT BENCH SIZE, end to end.  Kernels are checked one by one at M = 33 920 (tests/test_gemm8p_gpu.py) and
models against the reference's goldens at 2-4 sequences - but a batch of 2 runs other kernels (128 x 128 GEMM tiles, other
tile lists, single-round launches) than the 128-sequence batch of bench.py / BASELINE configs[2].  Here the full 24-layer
VAS class-GPT runs at B = 128 in the 16-bit lane, training mode (the fused training epilogues), and sequences 0-1 are
compared with the SAME sequences run at B = 2 and with the f32 oracle (oracle/gpt.py; whose mean over the two is the
reference's own golden value, tests/golden/gptclass_vas24.npz).  Dropout: the masks are counter-based and keyed by
(seed, site, row, column) - rows 0..529 are sequences 0-1 at either batch size - so the dropout-1/2 lane is compared between
the two batch sizes as well.  Last: two complete forward + backward passes at size must give the same bits (loss and the
whole flat gradient) - the at-size path's reproducibility (a race in a backward GEMM does not show in a loss).
Reference: transformer/minGPT.py:168-212,413-417."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import synth
from util import golden, report, t

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
B_BIG = 128


_SD = {}


def _model(pdrop):
    from melspec_gpt_vqvae_amd import _ffi
    from melspec_gpt_vqvae_amd.transformer.minGPT import GPTClass, set_compute_dtype

    g = golden("gptclass_vas24")
    args = synth.gpt_args(n_layer=24, n_head=16, n_embd=1024, embd_pdrop=pdrop, resid_pdrop=pdrop, attn_pdrop=pdrop)
    if "sd" not in _SD:                       # 302.85 M parameters regenerated from the golden's seed, once per process
        _SD["sd"] = synth.gpt_state_dict(args, int(g["sd_seed"]))
    sd = _SD["sd"]
    m = GPTClass(args)
    res = m.load_state_dict({k: t(v) for k, v in sd.items()}, strict=False)
    assert not res.unexpected_keys
    m.to(DEV).train()
    set_compute_dtype(m, _ffi.HALF_DTYPE)
    x2, c2 = g["x"].astype(np.int64), g["c"].astype(np.int64)
    x = np.concatenate([x2, synth.randint(4242, 0, 128, (B_BIG - 2, 265))], 0)
    c = np.concatenate([c2, synth.randint(4243, 0, 8, (B_BIG - 2, 1))], 0)
    return m, sd, args, g, t(x, DEV), t(c, DEV)


def _per_sequence_loss(m, x, c, counter):
    from melspec_gpt_vqvae_amd.transformer import minGPT

    minGPT._Seeds.counter = counter          # the same dropout key for every call that is compared
    with m.discard_att():
        logits, _, _ = m(x[:, :-1], c)
    per_tok = minGPT.cross_entropy(logits.reshape(-1, logits.size(-1)), x.reshape(-1), reduction="none")
    return per_tok.reshape(x.shape[0], -1).mean(1), logits


def test_bench_size_batch_gives_the_small_batch_and_oracle_losses_for_the_same_sequences():
    from oracle import gpt as ogpt

    m, sd, args, g, x, c = _model(0.0)
    with torch.no_grad():
        big, logits_big = _per_sequence_loss(m, x, c, 100)
        small, logits_small = _per_sequence_loss(m, x[:2], c[:2], 100)
    osd = ogpt.as_torch_sd(sd)
    xo, co = x[:2].cpu(), c[:2].cpu()
    with torch.no_grad():
        lo, _, _ = ogpt.gptclass_forward(osd, xo[:, :-1], co, 24, 16)
        ref = F.cross_entropy(lo.reshape(-1, lo.size(-1)), xo.reshape(-1), reduction="none").reshape(2, -1).mean(1)
    assert abs(float(ref.mean()) - float(g["loss"])) < 1e-4, "the oracle call reproduces the reference's golden loss"
    d_small = (big[:2] - small).abs().max().item()
    d_ref = (big[:2].cpu() - ref).abs().max().item()
    d_logits = (logits_big[:2].float() - logits_small.float()).abs().max().item() / logits_small.float().abs().max().item()
    report("at_size_lane_b128_vs_b2_vs_f32_oracle", per_sequence_loss_b128=[float(v) for v in big[:2]],
           per_sequence_loss_b2=[float(v) for v in small], per_sequence_loss_oracle_f32=[float(v) for v in ref],
           max_abs_diff_vs_b2=d_small, max_abs_diff_vs_oracle=d_ref, logits_rel_to_max_diff_vs_b2=d_logits)
    assert torch.isfinite(big).all()
    assert d_small <= 2e-2 and d_ref <= 2e-2, (d_small, d_ref)
    assert d_logits <= 3e-2


def test_bench_size_batch_with_dropout_one_half_matches_the_small_batch_under_the_same_masks():
    m, _, _, _, x, c = _model(0.5)
    with torch.no_grad():
        big, _ = _per_sequence_loss(m, x, c, 200)
        small, _ = _per_sequence_loss(m, x[:2], c[:2], 200)
        other, _ = _per_sequence_loss(m, x[:2], c[:2], 201)
    d = (big[:2] - small).abs().max().item()
    report("at_size_lane_dropout_half_b128_vs_b2", per_sequence_loss_b128=[float(v) for v in big[:2]],
           per_sequence_loss_b2=[float(v) for v in small], max_abs_diff=d,
           other_seed_diff=(other - small).abs().max().item())
    assert d <= 2e-2, d
    assert (other - small).abs().max().item() > 5 * max(d, 1e-3), "another key gives other masks (the comparison is not vacuous)"


def test_bench_size_step_is_bit_reproducible():
    from melspec_gpt_vqvae_amd.flat import ensure_flat
    from melspec_gpt_vqvae_amd.transformer import minGPT

    m, _, _, _, x, c = _model(0.5)
    fp = ensure_flat(m)
    runs = []
    for _ in range(3):
        for p in m.parameters():
            p.grad = None
        minGPT._Seeds.counter = 300
        with m.discard_att():
            logits, _, _ = m(x[:, :-1], c)
        loss = minGPT.cross_entropy(logits.reshape(-1, logits.size(-1)), x.reshape(-1))
        loss.backward()
        runs.append((loss.detach().clone(), fp.grad.clone()))
    for loss, grad in runs[1:]:
        assert torch.equal(loss, runs[0][0])
        assert torch.equal(grad, runs[0][1]), int((grad != runs[0][1]).sum())
    assert torch.isfinite(runs[0][1]).all() and float(runs[0][1].abs().max()) > 0
