"""CPU-side checks of the host logic: checkpoint ABI (state_dict keys), initialisation, optimizer grouping, the
no-CPU-fallback rule, and the data-parallel engine on 2 gloo processes.  No GPU, no compute through the C ABI."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import synth
from util import golden


def test_gptclass_checkpoint_abi_init_and_optimizer_groups():
    from melspec_gpt_vqvae_amd._ffi import MelgptError
    from melspec_gpt_vqvae_amd.transformer.minGPT import GPTClass, decay_groups

    g = golden("gpt_optim_groups")
    torch.manual_seed(0)
    m = GPTClass(synth.gpt_args(n_layer=24, n_head=16, n_embd=1024))
    assert list(m.state_dict().keys()) == [str(k) for k in g["keys"]], "state_dict keys / order are the checkpoint ABI"
    assert sum(p.numel() for p in m.parameters()) == int(g["n_params"]) == 302854144
    assert m.blocks[0].attn.mask.shape == (1, 1, 266, 266) and m.get_block_size() == 266
    decay, no_decay = decay_groups(m)
    assert decay == [str(s) for s in g["decay"]] and no_decay == [str(s) for s in g["no_decay"]]
    assert len(decay) == 145 and len(no_decay) == 245
    # reference init: N(0, 0.02) Linear/Embedding, zero biases, unit LayerNorm; embedder keeps N(0, 1)
    assert abs(float(m.tok_emb.weight.std()) - 0.02) < 1e-3 and abs(float(m.embedder.weight.std()) - 1.0) < 0.05
    assert float(m.blocks[3].mlp[0].bias.abs().max()) == 0.0 and float(m.ln_f.weight.min()) == 1.0
    assert float(m.pos_emb.abs().max()) == 0.0
    with pytest.raises(MelgptError):
        m(torch.zeros(1, 264, dtype=torch.int64), torch.zeros(1, 1, dtype=torch.int64))
    with pytest.raises(AssertionError):
        from melspec_gpt_vqvae_amd.transformer.minGPT import CausalSelfAttention, GPTConfig

        CausalSelfAttention(GPTConfig(128, 266, n_embd=100, n_head=16, attn_pdrop=0, resid_pdrop=0))


def test_vqvae_checkpoint_abi():
    from melspec_gpt_vqvae_amd.vqvae.big_model_attn_gan import LitVQVAE

    g = golden("vqvae_full")
    m = LitVQVAE(num_embeddings=128, embedding_dim=256)
    keys = list(m.state_dict().keys())
    assert keys == [str(k) for k in g["sd_keys"]]
    assert sum(1 for k in keys if k.startswith("discriminator.")) == 22
    assert m._vq_vae._embedding.weight.shape == (128, 256)
    assert float(m._vq_vae._embedding.weight.abs().max()) <= 1.0 / 128 + 1e-9
    assert sum(p.numel() for p in m._encoder.parameters()) > 29000000


def test_gpt_vae_structure():
    from melspec_gpt_vqvae_amd.transformer.Lit_GPT_VAE import GPT_VAE

    a = synth.gpt_args(n_layer=2, n_head=4, n_embd=256, block_size=265, fix_var=0, kl_start=0.1, warm_up=10,
                       batch_size=4, target_kl=8.0, beta=1.0, nsamples=1, fb=0, device="cpu", learning_rate=1e-6,
                       len_train_data=400)
    v = GPT_VAE(a)
    assert v.encoder.transformer.head.weight.shape == (512, 256)           # last_linear = 2C
    assert v.encoder.transformer.blocks[0].attn.n_unmasked == 265          # fully visible
    assert v.decoder.transformer.get_block_size() == 266                   # block_size + 1
    assert abs(v.anneal_rate - (1 - 0.1) / (10 * 100)) < 1e-12
    keys = list(v.state_dict().keys())
    assert "encoder.transformer.blocks.1.attn.key.weight" in keys and "decoder.transformer.pos_emb" in keys
    opt = v.GPT_configure_optimizers()
    assert len(opt.param_groups) == 2


def test_lit_optimizer_leaves_the_first_stage_model_frozen_and_kv_cache_guard():
    """reference minGPT.py:632,652: configure_optimizers walks self.transformer ONLY - with a VQ-VAE attached (the
    documented training command sets reconstruct_spec) its Conv2d / GroupNorm weights must neither trip the
    'not separated' assertion nor be trained."""
    from melspec_gpt_vqvae_amd.transformer.minGPT import GPT, Lit_minGPT
    from melspec_gpt_vqvae_amd.vqvae.big_model_attn_gan import LitVQVAE

    a = synth.gpt_args(n_layer=2, n_head=4, n_embd=64, reconstruct_spec="", device="cpu", learning_rate=1e-6)
    lit = Lit_minGPT(a)
    lit.first_stage_model = LitVQVAE(num_embeddings=128, embedding_dim=256)
    opt = lit.configure_optimizers()
    trained = {id(p) for g in opt.param_groups for p in g["params"]}
    assert trained == {id(p) for p in lit.transformer.parameters()}
    assert not trained & {id(p) for p in lit.first_stage_model.parameters()}
    assert [g["weight_decay"] for g in opt.param_groups] == [0.01, 0.0] and opt.param_groups[0]["betas"] == (0.9, 0.95)
    # KV-cached decoding is only equivalent to the reference's re-forward loop for strictly causal blocks
    assert lit.transformer.kv_cacheable()
    enc = GPT(synth.gpt_args(n_layer=1, n_head=4, n_embd=64), n_unmasked=5).eval()
    assert not enc.kv_cacheable()
    with pytest.raises(AssertionError, match="n_unmasked"):
        enc.decode_begin(1)


def test_distributed_shard_is_the_distributed_sampler_rule():
    from torch.utils.data.distributed import DistributedSampler

    from melspec_gpt_vqvae_amd.dp import distributed_shard

    for n, world in ((11773, 8), (768, 8), (10, 4), (7, 2)):
        for rank in range(world):
            for epoch in (0, 3):
                s = DistributedSampler(range(n), num_replicas=world, rank=rank, shuffle=True, seed=5)
                s.set_epoch(epoch)
                assert list(s) == distributed_shard(n, rank, world, seed=5, epoch=epoch)
            s = DistributedSampler(range(n), num_replicas=world, rank=rank, shuffle=False, drop_last=True)
            assert list(s) == distributed_shard(n, rank, world, shuffle=False, drop_last=True)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from melspec_gpt_vqvae_amd.dp import GradientExchange

    n = 100_003
    g = torch.arange(n, dtype=torch.float32) * (rank + 1)
    ex = GradientExchange(g, max_bucket_elems=4096)
    # two "blocks" announce their slices early and out of order; finish() covers the rest exactly once
    ex.launch(50_000, 70_000)
    ex.launch(1_000, 9_000)
    ex.finish()
    expect = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
    ok = torch.equal(g, expect)
    # a second step reuses the engine
    g.copy_(torch.ones(n) * (rank + 1))
    ex.finish()
    ok = ok and torch.equal(g, torch.ones(n) * sum(r + 1 for r in range(world)))
    # a segment announced twice in one step (two backward passes through one block) is refused, not double-summed
    ex.launch(10, 20)
    try:
        ex.launch(10, 20)
        ok = False
    except RuntimeError:
        pass
    ex.finish()
    # 16-bit wire format (`--grad-dtype bf16`): slices cast to bf16, reduced in bf16, converted back into
    # the f32 buffer.  bf16 keeps 8 significant bits: each rounding is <= 2^-9 = 1.95e-3 relative, and an element meets
    # three of them on two ranks (two casts, one sum) - measured here 4.0e-3 of the gradient's maximum, 2.5e-3 rms (the
    # 1e-3 the review asked for is below one bf16 rounding; an f16 wire would meet it but underflows on 1e-7 gradients).
    # The f32 buffer keeps its dtype and every element of it is written exactly once.
    parts = [torch.randn(n, generator=torch.Generator().manual_seed(7 + r)) * 3e-3 for r in range(world)]
    g16 = parts[rank].clone()
    ex16 = GradientExchange(g16, max_bucket_elems=4096, wire_dtype=torch.bfloat16)
    assert ex16.bytes_per_step == 2 * n and ex.bytes_per_step == 4 * n
    ex16.launch(20_000, 30_000)
    ex16.launch(0, 5_000)
    ex16.finish()
    want = sum(parts)
    ok = ok and g16.dtype == torch.float32 and float((g16 - want).abs().max() / want.abs().max()) < 6e-3
    ok = ok and float(((g16 - want) ** 2).mean().sqrt() / (want ** 2).mean().sqrt()) < 3e-3
    untouched = torch.ones(n, dtype=torch.bool)
    ok = ok and bool((g16 != parts[rank])[untouched].float().mean() > 0.9)     # every slice was exchanged, not only the launched two
    ok = ok and not ex16._wired and not ex16._works
    q.put((rank, ok))
    dist.destroy_process_group()


def test_gradient_exchange_two_process_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


# ------------------------------------------------------------------------------------------ launch.spawn_ranks
def _run_launcher(extra):
    import subprocess
    import sys as _sys

    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys; sys.path.insert(0, %r); from melspec_gpt_vqvae_amd.launch import spawn_ranks; "
            "raise SystemExit(spawn_ranks([%r] + %r, 2, share_gpu=True, timeout=120))"
            % (os.path.dirname(here), os.path.join(here, "launch_worker.py"), extra))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    return subprocess.run([_sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)


def test_spawn_ranks_starts_one_process_per_rank_and_relays_rank0():
    """what `python bench.py --gpus N` (no launcher around it) does: N fresh ranks with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set, rank 0's stdout passed through, exit status 0."""
    r = _run_launcher([])
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip().splitlines()[-1] == "LAUNCH_OK world=2 local_rank=0 sum=3"   # gloo may print a line first


def test_spawn_ranks_propagates_a_failing_rank_instead_of_hanging():
    r = _run_launcher(["--fail-rank", "1"])
    assert r.returncode == 7, (r.returncode, r.stderr)


def test_c_abi_host_paths_under_address_sanitizer(tmp_path):
    """SURVEY 5 (sanitizer builds of the native layer): tools/asan_host.sh compiles every csrc/*.hip with
    -fsanitize=address for the HOST code (GPU ASan is refused on this pool), links a scratch library outside the tree and
    runs tools/asan_host_driver.c over the entry points' host paths - switches, workspace queries, argument refusals -
    without a GPU.  ~1.5 min of hipcc; MELGPT_SKIP_ASAN=1 skips it."""
    import shutil
    import subprocess

    if os.environ.get("MELGPT_SKIP_ASAN") == "1":
        pytest.skip("MELGPT_SKIP_ASAN=1")
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not found")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "tools", "asan_host.sh")):
        pytest.skip("tools/asan_host.sh is not shipped to the GPU box (.gpurunignore: sanitizer builds are refused there)")
    env = dict(os.environ, MELGPT_ASAN_DIR=str(tmp_path))
    r = subprocess.run(["bash", os.path.join(root, "tools", "asan_host.sh"), "6"], env=env, capture_output=True, text=True,
                       timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "asan host driver: ok" in r.stdout


def test_rccl_channel_pin_respects_the_users_environment(monkeypatch):
    """dp.pin_rccl_channels (opt-in, MELGPT_RCCL_CHANNELS=n): a known RCCL footprint (NCCL_MAX_NCHANNELS) together with that
    many reserved CUs (MELGPT_RESERVE_CUS, a multiple of 8); off by default; a value the user exported wins."""
    from melspec_gpt_vqvae_amd import dp

    for k in ("NCCL_MAX_NCHANNELS", "NCCL_MIN_NCHANNELS", "MELGPT_RCCL_CHANNELS", "MELGPT_RESERVE_CUS"):
        monkeypatch.delenv(k, raising=False)
    assert dp.pinned_rccl_channels() == 0
    assert dp.pin_rccl_channels() == 0 and "NCCL_MAX_NCHANNELS" not in os.environ and "MELGPT_RESERVE_CUS" not in os.environ
    monkeypatch.setenv("MELGPT_RCCL_CHANNELS", "12")
    monkeypatch.setenv("NCCL_MIN_NCHANNELS", "64")
    assert dp.pin_rccl_channels() == 12 and os.environ["NCCL_MAX_NCHANNELS"] == "12"
    assert os.environ["NCCL_MIN_NCHANNELS"] == "12" and os.environ["MELGPT_RESERVE_CUS"] == "16"
    monkeypatch.setenv("NCCL_MAX_NCHANNELS", "24")
    monkeypatch.setenv("MELGPT_RESERVE_CUS", "8")
    assert dp.pin_rccl_channels(16) == 24 and os.environ["MELGPT_RESERVE_CUS"] == "8", "what the user exported is respected"
    for k in ("NCCL_MAX_NCHANNELS", "NCCL_MIN_NCHANNELS", "MELGPT_RESERVE_CUS"):
        monkeypatch.delenv(k, raising=False)


def test_kernel_timer_counts_shared_time_once():
    """bench.py's roofline time is the UNION of the timed launches' intervals on the device clock (a Block's weight
    gradients run on a second stream beside the input-gradient chain: the plain sum would count shared time twice), per
    tag it is the plain sum.  Driven with stand-in events (start offsets in milliseconds); no GPU involved."""
    from melspec_gpt_vqvae_amd import ops

    class Ev:
        def __init__(self, t):
            self.t = t

        def elapsed_time(self, other):
            return other.t - self.t

    tm = ops.KernelTimer()
    assert tm.summary() == dict(launches=0, total_ms=0.0, serial_ms=0.0, flops=0)
    #            main stream: [0, 2) [2, 3)        [6, 7)      side stream: [1, 4)   (recorded out of start order on purpose)
    for a, b, fl, tag in ((0.0, 2.0, 10.0, "x"), (6.0, 7.0, 5.0, "y"), (1.0, 4.0, 30.0, "w"), (2.0, 3.0, 10.0, "x")):
        tm.records.append((Ev(a), Ev(b), fl, tag))
    s = tm.summary()
    assert s["launches"] == 4 and s["flops"] == 55.0
    assert s["serial_ms"] == pytest.approx(7.0) and s["total_ms"] == pytest.approx(5.0)   # [0, 4) and [6, 7)
    rows = {t: (n, ms, fl) for t, n, ms, fl in tm.by_tag()}
    assert rows == {"x": (2, pytest.approx(3.0), 20.0), "w": (1, pytest.approx(3.0), 30.0), "y": (1, pytest.approx(1.0), 5.0)}
    assert tm.by_tag()[-1][0] == "y"                      # slowest tag first
    tm.aux.append((Ev(0.0), Ev(0.5), 100.0, 7.0, "attn"))
    tm.aux.append((Ev(1.0), Ev(1.5), 100.0, 7.0, "attn"))
    assert tm.aux_by_tag() == {"attn": (2, pytest.approx(1.0), 200.0, 14.0)}
