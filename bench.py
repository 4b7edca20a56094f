#!/usr/bin/env python3
"""bench.py - mel-token sequences / second for one training step of the hot path
(VQ-encode + class-conditioned minGPT forward/backward + AdamW), BASELINE.json's metric.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by the driver as  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One step = one pass of the hot path over one batch of synthetic input that is already resident in HBM:
  mel tiles (B,1,80,848) in [-1,1]  --LitVQVAE.encode + 128-code L2 argmin-->  codes (B,5,53)
  --time-major permute-->  (B,265) tokens  --GPTClass (VAS: 24 L, 1024, 16 H, block 266, dropout 0.5)-->
  logits -> cross entropy -> backward -> [N>1: RCCL all-reduce of the flat gradient] -> fused AdamW.
Every tensor op is a hand-written HIP kernel behind the C ABI (include/melgpt.h); bf16 storage / MFMA operands,
f32 accumulation, f32 master weights and gradients.  Random-init weights of the real architecture, synthetic data.
Prints ONE JSON line on rank 0 (contract in the task statement), including `roofline` for the dominant kernel
(the bf16 MFMA GEMM / implicit-GEMM conv, timed live with HIP events on the launch stream) and `cpu_baseline`
(the CPU oracle's restatement of the same step timed on this box's host cores, bounded sample, rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch

PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
SEED = 783435              # the reference's fixed seed (GPT_train.py:56-61)


def vas_args(**kw):
    from types import SimpleNamespace

    d = dict(vocab_size=128, block_size=266, n_layer=24, n_head=16, n_embd=1024, class_size=8, embd_pdrop=0.5,
             resid_pdrop=0.5, attn_pdrop=0.5, n_unmasked=0, last_linear=None, learning_rate=1e-6)
    d.update(kw)
    return SimpleNamespace(**d)  # config/config_GPT_vas.py:1-18


def balanced_rows(lat, k=128, iters=400, seed=SEED):
    """Pick k rows of `lat` (N, D) f32 CPU - every row stays an actual sample - so that nearest-row assignment of `lat`
    spreads over the rows: start from k random samples (already mass-proportional), then `iters` times replace the
    least-used row by a random member of the most-used row's cell.  Bench-side set-up arithmetic (host torch), not
    part of the path.  -> (k, D) tensor, perplexity of the assignment on `lat`."""
    g = torch.Generator().manual_seed(seed)
    E = lat[torch.randperm(lat.shape[0], generator=g)[:k]].clone()
    l2 = (lat * lat).sum(1, keepdim=True)

    def assign():
        return (l2 + (E * E).sum(1)[None] - 2.0 * lat @ E.t()).argmin(1)

    for _ in range(iters):
        idx = assign()
        cnt = torch.bincount(idx, minlength=k)
        members = torch.nonzero(idx == int(cnt.argmax())).flatten()
        E[int(cnt.argmin())] = lat[members[int(torch.randint(0, len(members), (1,), generator=g))]]
    p = torch.bincount(assign(), minlength=k).double() / lat.shape[0]
    return E, float(torch.exp(-(p * torch.log(p + 1e-10)).sum()))


def spread_codebook(vqvae, device, tiles=32):
    """A frozen VQ-VAE whose codes SPREAD over the 128 entries, as a trained one's do and as BASELINE.md 4 asks of the
    GPT leg's tokens: the codebook's rows are the encoder's own latents (LitVQVAE.encode of `tiles` synthetic mel tiles
    drawn from another seed than any step's batch), chosen by `balanced_rows`.  A random encoder maps these tiles to a
    tight cluster far from a N(0,1) codebook: that pairing gives 8-11 distinct codes, perplexity 3.2 (--codebook normal)."""
    x, _ = synthetic_batch(tiles, 2000, device)
    with torch.no_grad():
        z = vqvae.encode(x)                                                    # (tiles,256,5,53): quant_conv(encoder(x))
        lat = z.float().permute(0, 2, 3, 1).reshape(-1, z.shape[1]).cpu()
        rows, perp = balanced_rows(lat)
        vqvae._vq_vae._embedding.weight.copy_(rows.to(device))
    return perp


def code_perplexity(codes, k=128):
    """exp(entropy) of the code histogram of one batch (VectorQuantizer.forward's `perplexity`, big_model_attn_gan.py:50-51)."""
    p = torch.bincount(codes.reshape(-1).cpu(), minlength=k).double()
    p = p / p.sum()
    return {"perplexity": round(float(torch.exp(-(p * torch.log(p + 1e-10)).sum())), 2), "distinct": int((p > 0).sum()),
            "top_share": round(float(p.max()), 4)}


def build_models(device, dtype, args, codebook="latents"):
    from melspec_gpt_vqvae_amd.transformer import minGPT
    from melspec_gpt_vqvae_amd.vqvae import big_model_attn_gan as vq

    torch.manual_seed(SEED)  # identical initial weights on every rank (no start-up broadcast needed)
    gpt = minGPT.GPTClass(args)
    vqvae = vq.LitVQVAE(num_embeddings=128, embedding_dim=256)
    if codebook == "normal":  # rounds 1-5: N(0,1) rows - with a random encoder the step's tokens collapse onto a few codes
        with torch.no_grad():
            vqvae._vq_vae._embedding.weight.normal_(0.0, 1.0)
    gpt.to(device).train()
    vqvae.to(device).eval()
    minGPT.set_compute_dtype(gpt, dtype)
    vq.set_compute_dtype(vqvae, dtype)
    if codebook == "latents":  # deterministic given the seed: every rank computes the same rows
        spread_codebook(vqvae, device)
    return gpt, vqvae


def synthetic_batch(batch, rank, device):
    import synth

    mel = torch.from_numpy(synth.mel_tiles(SEED + 17 * rank, batch))          # (B,80,860) in [0,1]
    x = (2 * mel[:, :, 6:854] - 1).unsqueeze(1).contiguous().to(device)       # CenterCrop 848, 2x-1
    c = torch.from_numpy(synth.randint(SEED + 1 + rank, 0, 8, (batch, 1))).to(device)
    return x, c


def vq_encode_b64(job, device, reps=5):
    """BASELINE configs[1] beside the metric (outside the timed region, a few launches' worth of time): VQ-encode + 128-code
    argmin of 64 mel tiles on the step's own frozen VQ-VAE - milliseconds per batch with a device synchronisation around
    each repetition, median of `reps`; the encoder's convolutions are 142.57 GFLOP per tile."""
    x, _ = synthetic_batch(64, 1000, device)
    with torch.no_grad():
        job.vqvae.encode_to_codes(x)
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            codes = job.vqvae.encode_to_codes(x)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        # the codebook lookup alone (north_star's ">= 60 % of HBM at batch 64" kernel): the encoder's output is kept and
        # the fused lookup launched back to back, timed with HIP events on the launch stream (torch's current stream)
        from melspec_gpt_vqvae_amd.vqvae.big_model_attn_gan import _as_nchw
        h = _as_nchw(job.vqvae._encoder._nhwc(x))
        n_vec = h.shape[0] * h.shape[2] * h.shape[3]
        job.vqvae._vq_vae.encode_indices_fused(h, job.vqvae.quant_conv)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n_l = 50
        e0.record()
        for _ in range(n_l):
            job.vqvae._vq_vae.encode_indices_fused(h, job.vqvae.quant_conv)
        e1.record()
        e1.synchronize()
        lookup_us = 1e3 * e0.elapsed_time(e1) / n_l
        sweep = lookup_sweep(job, h)
    lookup_bytes = n_vec * 520 + 66048      # 512 B bf16 latent + 8 B index per vector, + the prepared image
    ms = 1e3 * sorted(ts)[len(ts) // 2]
    return {"workload": "VQ-encode + argmin, 64 tiles (1,80,848), 16-bit lane", "ms": round(ms, 3),
            "code_stats": code_perplexity(codes), "lookup_sweep": sweep,
            "tiles_per_s": round(64 / (ms * 1e-3), 1), "encoder_tflops": round(64 * 142.57e9 / (ms * 1e-3) / 1e12, 1),
            "codes_shape": list(codes.shape), "lookup_us": round(lookup_us, 2), "lookup_vectors": n_vec,
            "lookup_bytes": lookup_bytes, "lookup_frac_hbm": round(lookup_bytes / (lookup_us * 1e-6) / 8e12, 4),
            "lookup_note": "in-stream average of 50 back-to-back launches incl. the Python launch path; 8.9 MB is 1.1 us "
                           "at 8 TB/s - below any stand-alone launch (DESIGN 4)"}


def lookup_sweep(job, h64, batches=(64, 256, 1024, 4096), n_l=50):
    """the codebook lookup alone over a batch sweep - latents of B tiles (the B = 64 encoder
    output tiled along the batch axis: real latents, so the codes are the step's codes), the fused lookup launched `n_l`
    times back to back, HIP events on the launch stream; bytes = 520 per vector + the prepared image.
    The Python launch path costs ~18 us per call - more than the small launches run - so the stream is first given ~3.5 ms
    of filler work (30 lookups of the 4096-tile tensor): the host queues the timed launches while the GPU is still busy,
    and the two events bracket GPU time only (back-to-back launches, their launch gaps included)."""
    vq, qc = job.vqvae._vq_vae, job.vqvae.quant_conv
    hs = {b: (h64 if b == 64 else h64.permute(0, 2, 3, 1).repeat(b // 64, 1, 1, 1).permute(0, 3, 1, 2)) for b in batches}  # channels-last strides kept
    filler = hs[max(batches)]
    for h in hs.values():
        vq.encode_indices_fused(h, qc)
    torch.cuda.synchronize()
    out = {}
    for b in batches:
        h = hs[b]
        n_vec = h.shape[0] * h.shape[2] * h.shape[3]
        for _ in range(30):
            vq.encode_indices_fused(filler, qc)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n_l):
            vq.encode_indices_fused(h, qc)
        e1.record()
        e1.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / n_l
        nbytes = n_vec * 520 + 66048
        out[f"B{b}"] = {"vectors": n_vec, "us": round(us, 2), "GBps": round(nbytes / (us * 1e-6) / 1e9, 1),
                        "frac_hbm": round(nbytes / (us * 1e-6) / 8e12, 4)}
    return out


def gpt_vae_xl_rank(a, device, dtype, steps=3):
    """BASELINE configs[3] beside the metric: ONE rank's work of the 8-GPU GPT-VAE XL job (2.09 B parameters, batch 128
    per GPU), `steps` timed steps after one warm-up, outside the metric's timed region (the class-GPT job is freed
    first: this one peaks at 168 GB)."""
    from types import SimpleNamespace

    from melspec_gpt_vqvae_amd import ops
    job = GPTVAEXLStep(SimpleNamespace(layers=40, batch=128), device, dtype, 0, 1)
    job.step(time.perf_counter)
    torch.cuda.synchronize()
    ops.TIMER = ops.KernelTimer()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, _ = job.step(time.perf_counter)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    timer, ops.TIMER = ops.TIMER, None
    ks = timer.summary()
    out = {"workload": job.workload, "steps": steps, "ms_per_step": round(1e3 * dt, 2), "seq_per_s": round(128 / dt, 2),
           "gemm_family_tflops": round(ks["flops"] / (ks["total_ms"] * 1e-3) / 1e12, 1) if ks["total_ms"] > 0 else None,
           "step_tflops_3416_gflop_per_seq": round(128 * 3.416e12 / dt / 1e12, 1), "final_loss": round(float(loss.detach()), 4),
           "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1),
           "note": "one rank of the dp8 job; the 8.37 GB f32 gradient exchange of the other seven ranks is not in it"}
    del job
    return out


def e2e_fp16_child(timeout=420):
    """BASELINE configs[4] beside the metric: the end-to-end chain in the library's fp16 flavour - a CHILD process
    (MELGPT_HALF is a property of the process; started after this one's GPU work is done, never replacing it): batch 1
    latency percentiles, batch 64 and batch 128 clips/s with per-stage milliseconds (tools/bench_e2e.py)."""
    import subprocess

    cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_e2e.py"), "--dtype", "fp16", "--batches", "1,64,128"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MELGPT_HALF")}
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired:
        return {"error": f"tools/bench_e2e.py did not finish in {timeout} s"}
    rows = []
    for ln in r.stdout.splitlines():
        if ln.startswith("{"):
            try:
                rows.append(json.loads(ln))
            except ValueError:
                pass
    if r.returncode != 0 or len(rows) < 2:
        return {"error": f"rc {r.returncode}", "stderr_tail": r.stderr[-400:]}
    b1, b64 = rows[0], rows[1]
    b128 = rows[2] if len(rows) > 2 else None
    out = {"workload": "wav -> HIP STFT/mel -> VQ encode -> GPT sample 265 (KV-cached) -> VQ decode -> MelGAN, fp16 flavour, one GPU",
            "batch1_latency_ms": b1["latency_ms"], "batch1_stage_ms": b1["stage_ms_median"],
            "batch64_clips_per_s": b64["clips_per_s"], "batch64_latency_ms": b64["latency_ms"],
            "batch64_stage_ms": b64["stage_ms_median"], "x_realtime_batch64": b64["x_realtime"]}
    if b128:
        out.update({"batch128_clips_per_s": b128["clips_per_s"], "batch128_latency_ms": b128["latency_ms"],
                    "batch128_stage_ms": b128["stage_ms_median"]})

    # HBM fractions of the two HBM-bound stages (peak 8 TB/s).  Decode: per sampled token the step streams the 16-bit
    # weights once (302.6 M parameters) and, per sequence, the K and V rows cached so far in all 24 layers (mean position
    # 133 of 265: 24 x 2 x 133 x 1024 x 2 bytes).  Mel frontend: 220 500 f32 PCM samples in, one 80 x 848 16-bit tile out.
    def decode_frac(batch, ms):
        bytes_per_token = 2 * 302.6e6 + batch * 24 * 2 * 133 * 1024 * 2
        return round(bytes_per_token / (ms * 1e-3 / 265) / 8e12, 4)

    def mel_frac(batch, ms):
        return round(batch * (220500 * 4 + 80 * 848 * 2) / (ms * 1e-3) / 8e12, 4)

    out["decode_frac_hbm"] = {"batch1": decode_frac(1, b1["stage_ms_median"]["gpt_sample_265"]),
                              "batch64": decode_frac(64, b64["stage_ms_median"]["gpt_sample_265"])}
    out["mel_frac_hbm"] = {"batch1": mel_frac(1, b1["stage_ms_median"]["mel_frontend"]),
                           "batch64": mel_frac(64, b64["stage_ms_median"]["mel_frontend"])}
    if b128:
        out["decode_frac_hbm"]["batch128"] = decode_frac(128, b128["stage_ms_median"]["gpt_sample_265"])
        out["mel_frac_hbm"]["batch128"] = mel_frac(128, b128["stage_ms_median"]["mel_frontend"])
    out["ms_per_token"] = {"batch1": round(b1["stage_ms_median"]["gpt_sample_265"] / 265, 4),
                           "batch64": round(b64["stage_ms_median"]["gpt_sample_265"] / 265, 4)}
    return out


def pmc_summary(workload="class_gpt"):
    """The newest committed rocprofv3 PMC summary OF THIS WORKLOAD's training step (profiles/*_summary.json written by
    tools/profile_round.sh; bench.py itself cannot collect PMC counters): HBM bytes per step of the GEMM-family kernels
    (separate FETCH_SIZE / WRITE_SIZE passes, gfx950 corrections of the guide applied there) and, when the MFMA pass
    was collected, the matrix-pipe busy fraction of that family.  Files without the GEMM-family keys (e.g. the VQ
    lookup's own summary) are skipped, and said so on stderr."""
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    best = None
    for f in sorted(glob.glob(os.path.join(here, "profiles", "*_summary.json"))):
        try:
            j = json.load(open(f))
        except (OSError, ValueError) as e:
            print(f"bench.py: {os.path.basename(f)} unreadable ({e})", file=sys.stderr)
            continue
        if "gemm_family_hbm_read_MB" not in j or j.get("workload", "class_gpt") != workload:
            print(f"bench.py: {os.path.basename(f)} is not a {workload} training-step summary - skipped", file=sys.stderr)
            continue
        best = (f, j)            # sorted(): the last matching file is the newest round / letter
    if best is None:
        return {}
    f, j = best
    steps = max(int(j.get("steps_profiled", 1)), 1)
    out = {"traffic": round((j["gemm_family_hbm_read_MB"] + j["gemm_family_hbm_write_MB"]) * 1e6 / steps),
           "traffic_source": "profiles/" + os.path.basename(f)}
    if j.get("gemm_family_mfma_busy") is not None:
        out["mfma_busy"] = j["gemm_family_mfma_busy"]
        out["mfma_busy_by_kernel"] = j.get("mfma_busy_by_kernel")
    if j.get("attention_hbm_MB_per_launch"):
        out["attention_hbm_MB_per_launch"] = j["attention_hbm_MB_per_launch"]
    if j.get("in_kernel_clock_GHz"):      # the shader clock the chip HELD inside the big kernels (stamped diagnostic build)
        out["in_kernel_clock_GHz"] = {k: v["GHz"] for k, v in j["in_kernel_clock_GHz"].items()}
    return out


def usable_cores():
    """CPU cores this process may actually use: the affinity mask capped by the cgroup CPU quota (a GPU box hands a
    one-GPU job a share of the host - os.cpu_count() there is the whole machine and oversubscribing it is slow)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(batch=8, reps=3):
    """The CPU oracle (kind "port": torch-CPU fp32 restatement pinned to the reference by tests/golden) running the
    same step - VQ-encode + class-GPT fwd/bwd + AdamW - on a bounded sample."""
    import synth
    from oracle import gpt as ogpt
    from oracle import vqvae as ovq

    threads = usable_cores()
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(1)
    a = vas_args()
    C, L, V = a.n_embd, a.n_layer, a.vocab_size

    def rn(*s, std=0.02):
        return (torch.randn(*s, generator=g) * std).requires_grad_(True)

    sd = {"pos_emb": rn(1, 266, C), "tok_emb.weight": rn(V, C), "ln_f.weight": torch.ones(C, requires_grad=True),
          "ln_f.bias": torch.zeros(C, requires_grad=True), "head.weight": rn(V, C), "embedder.weight": rn(8, C, std=1.0)}
    for i in range(L):
        p = f"blocks.{i}."
        for ln in ("ln1", "ln2"):
            sd[p + ln + ".weight"] = torch.ones(C, requires_grad=True)
            sd[p + ln + ".bias"] = torch.zeros(C, requires_grad=True)
        for nm in ("key", "query", "value", "proj"):
            sd[p + f"attn.{nm}.weight"] = rn(C, C)
            sd[p + f"attn.{nm}.bias"] = torch.zeros(C, requires_grad=True)
        sd[p + "mlp.0.weight"], sd[p + "mlp.0.bias"] = rn(4 * C, C), torch.zeros(4 * C, requires_grad=True)
        sd[p + "mlp.2.weight"], sd[p + "mlp.2.bias"] = rn(C, 4 * C), torch.zeros(C, requires_grad=True)
    vsd = {k: torch.from_numpy(v) for k, v in synth.vqvae_state_dict(50).items()}
    mel = torch.from_numpy(synth.mel_tiles(3, batch))
    c = torch.from_numpy(synth.randint(4, 0, 8, (batch, 1)))
    opt = torch.optim.AdamW(list(sd.values()), lr=1e-6, betas=(0.9, 0.95))

    def step():
        with torch.no_grad():
            codes, _ = ovq.mel_to_codes(vsd, mel)
        x = ogpt.codes_to_sequence(codes)
        loss, _, _ = ogpt.class_gpt_loss(sd, x, c, L, a.n_head, pdrop=(0.5, 0.5, 0.5), train=True)
        opt.zero_grad()
        loss.backward()
        opt.step()

    t0 = time.perf_counter()
    step()
    warm = time.perf_counter() - t0
    print(f"bench.py: cpu_baseline warm-up step {warm:.1f} s on {threads} threads", file=sys.stderr, flush=True)
    if warm > 40.0:        # keep the sample bounded (~2 min of CPU work at most): a slow host gets one timed step
        reps = 1
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
        print(f"bench.py: cpu_baseline step {times[-1]:.1f} s", file=sys.stderr, flush=True)
    dt = sorted(times)[len(times) // 2]
    return {"value": round(batch / dt, 4), "unit": "seq/s", "cores": threads, "cpu_model": cpu_model(), "kind": "port",
            "sample": f"batch {batch}, median of {reps} steps after 1 warm-up: oracle VQ-encode + class-GPT VAS fwd/bwd + "
                      f"AdamW, fp32, torch threads = usable cores (affinity / cgroup quota) = {threads} of os.cpu_count() = {os.cpu_count()}, {dt:.2f} s/step"}


def torch_baseline_child(batch, steps=3):
    """The body of `torch_gpu_baseline`, run in a CHILD process (stock torch-ROCm's allocator, MIOpen and rocBLAS/hipBLASLt
    state never share a process with the measured path): the oracle's functional restatement of the step - the same
    functions `cpu_baseline` times, pinned to the reference by tests/golden - on cuda:0 under bf16 autocast, f32
    parameters, torch.optim.AdamW with the reference's two parameter groups.  One JSON line per stage on stdout."""
    import synth
    from oracle import gpt as ogpt
    from oracle import vqvae as ovq

    dev = torch.device("cuda", 0)
    a = vas_args()
    C, L, V = a.n_embd, a.n_layer, a.vocab_size
    g = torch.Generator().manual_seed(1)

    def rn(*s, std=0.02):
        return (torch.randn(*s, generator=g) * std).to(dev).requires_grad_(True)

    def const(n, v):
        return torch.full((n,), v, device=dev).requires_grad_(True)

    sd = {"pos_emb": rn(1, 266, C), "tok_emb.weight": rn(V, C), "ln_f.weight": const(C, 1.0), "ln_f.bias": const(C, 0.0),
          "head.weight": rn(V, C), "embedder.weight": rn(8, C, std=1.0)}
    for i in range(L):
        p = f"blocks.{i}."
        for ln in ("ln1", "ln2"):
            sd[p + ln + ".weight"], sd[p + ln + ".bias"] = const(C, 1.0), const(C, 0.0)
        for nm in ("key", "query", "value", "proj"):
            sd[p + f"attn.{nm}.weight"], sd[p + f"attn.{nm}.bias"] = rn(C, C), const(C, 0.0)
        sd[p + "mlp.0.weight"], sd[p + "mlp.0.bias"] = rn(4 * C, C), const(4 * C, 0.0)
        sd[p + "mlp.2.weight"], sd[p + "mlp.2.bias"] = rn(C, 4 * C), const(C, 0.0)
    decay = [v for k, v in sd.items() if k.endswith(".weight") and v.dim() == 2 and "emb" not in k]   # nn.Linear weights
    decay_ids = {id(v) for v in decay}
    opt = torch.optim.AdamW([{"params": decay, "weight_decay": 0.01},
                             {"params": [v for v in sd.values() if id(v) not in decay_ids], "weight_decay": 0.0}],
                            lr=1e-6, betas=(0.9, 0.95))                       # minGPT.py:618-665
    vsd = {k: torch.from_numpy(v).to(dev) for k, v in synth.vqvae_state_dict(50).items()}
    mel = torch.from_numpy(synth.mel_tiles(3, batch)).to(dev)
    c = torch.from_numpy(synth.randint(4, 0, 8, (batch, 1))).to(dev)
    # the codebook spread as in the measured job (rows = the encoder's own latents)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        z = ovq.vqvae_encode(vsd, (2 * torch.from_numpy(synth.mel_tiles(2000, 32)).to(dev)[:, :, 6:854] - 1).unsqueeze(1))
    rows, _ = balanced_rows(z.float().permute(0, 2, 3, 1).reshape(-1, 256).cpu())
    vsd["_vq_vae._embedding.weight"] = rows.to(dev)

    def encode():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            x = (2 * mel[:, :, 6:854] - 1).unsqueeze(1)
            zq = ovq.vqvae_encode(vsd, x).float()
            flat = zq.permute(0, 2, 3, 1).reshape(-1, 256)
            idx = torch.argmin(ovq.vq_distances(flat, vsd["_vq_vae._embedding.weight"]), dim=1)   # big_model_attn_gan.py:28-33
        return idx.view(batch, 5, 53)

    def gpt_step(codes):
        x = ogpt.codes_to_sequence(codes)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss, _, _ = ogpt.class_gpt_loss(sd, x, c, L, a.n_head, pdrop=(0.5, 0.5, 0.5), train=True)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    def timed(fn, *args):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn(*args)
        torch.cuda.synchronize()
        return r, time.perf_counter() - t0

    codes, warm_e = timed(encode)
    print(json.dumps({"stage": "encode_warmup", "s": round(warm_e, 2)}), flush=True)
    te = sorted(timed(encode)[1] for _ in range(steps))[steps // 2]
    print(json.dumps({"stage": "encode", "ms": round(1e3 * te, 2), "code_stats": code_perplexity(codes)}), flush=True)
    _, warm_g = timed(gpt_step, codes)
    print(json.dumps({"stage": "gpt_warmup", "s": round(warm_g, 2)}), flush=True)
    ts, loss = [], None
    for _ in range(steps):
        loss, dt = timed(gpt_step, codes)
        ts.append(dt)
    tg = sorted(ts)[steps // 2]
    print(json.dumps({"stage": "gpt_step", "ms": round(1e3 * tg, 2), "loss": round(float(loss), 4),
                      "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1)}), flush=True)


def torch_gpu_baseline(batch=128, timeout=420):
    """What a user of the reference gets from stock PyTorch-ROCm on THIS box, beside `cpu_baseline`: the same functional
    step (oracle/gpt.py + oracle/vqvae.py = the reference's modules restated) on cuda through torch's own kernels
    (rocBLAS / hipBLASLt / MIOpen / ATen), bf16 autocast, batch 128, 1 warm-up + median of 3, in a child process started
    after this process's GPU work is done.  A baseline, not a target; the measured path never imports any of it."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--torch-baseline-child", "--batch", str(batch)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MELGPT_HALF")}
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
        stdout, rc, err = r.stdout, r.returncode, r.stderr[-300:]
    except subprocess.TimeoutExpired as e:
        stdout = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
        rc, err = "timeout", f"child did not finish in {timeout} s"
    st = {}
    for ln in stdout.splitlines():
        if ln.startswith("{"):
            try:
                j = json.loads(ln)
                st[j.pop("stage")] = j
            except (ValueError, KeyError):
                pass
    out = {"kind": "stock PyTorch-ROCm (torch %s) eager, the oracle's functional step on cuda:0, bf16 autocast, f32 master "
                   "weights, torch.optim.AdamW; random-init weights, same synthetic tiles" % torch.__version__,
           "batch": batch, "sample": "1 warm-up + median of 3 per stage; VQ-encode and GPT step timed separately, value = batch / their sum"}
    if "encode" in st and "gpt_step" in st:
        ms = st["encode"]["ms"] + st["gpt_step"]["ms"]
        out.update({"value": round(batch / (ms * 1e-3), 2), "unit": "seq/s", "ms_per_step": round(ms, 2),
                    "encode_ms": st["encode"]["ms"], "gpt_step_ms": st["gpt_step"]["ms"],
                    "code_stats": st["encode"].get("code_stats"), "final_loss": st["gpt_step"].get("loss"),
                    "peak_mem_GB": st["gpt_step"].get("peak_mem_GB"),
                    "warmup_s": {"encode": st.get("encode_warmup", {}).get("s"), "gpt": st.get("gpt_warmup", {}).get("s")}})
    else:
        out.update({"error": f"rc {rc}", "stages_done": sorted(st), "stderr_tail": err})
    return out


def xl_args(**kw):
    from types import SimpleNamespace

    d = dict(vocab_size=1024, block_size=265, n_layer=40, n_head=23, n_embd=1472, embd_pdrop=0.0, resid_pdrop=0.0,
             attn_pdrop=0.0, n_unmasked=0, last_linear=None, learning_rate=1e-6, fix_var=0, kl_start=0.3, warm_up=0,
             batch_size=128, target_kl=0.0, beta=1.0, nsamples=1, fb=0, iw_train_nsamples=-1)
    d.update(kw)
    return SimpleNamespace(**d)  # config/config_GPT_VAE_vggsound.py:43-58 + GPT_VAE_train.py flag defaults


# debug aid (1-GPU boxes): MELGPT_BENCH_FORCE_DP=1 runs the one rank through dp.DataParallel over a real RCCL process group
# of size 1 - the exchange's launches, streams and the reserved CUs as an N-GPU run has them, minus the wire
FORCE_DP = os.environ.get("MELGPT_BENCH_FORCE_DP") == "1"
if FORCE_DP:
    os.environ.setdefault("MELGPT_DP_FORCE_EXCHANGE", "1")  # dp.GradientExchange: launch the all-reduces of a 1-rank group


class ClassGPTStep:
    """BASELINE configs[2] (+ the VQ-encode of configs[1]): the default workload, the one `metric` is quoted on."""
    name = "class_gpt"
    seq_len = 265

    def __init__(self, a, device, dtype, rank, world):
        from melspec_gpt_vqvae_amd.dp import DataParallel
        from melspec_gpt_vqvae_amd.optim import FusedAdamW

        self.a = a
        gargs = vas_args(n_layer=a.layers)
        self.gpt, self.vqvae = build_models(device, dtype, gargs, getattr(a, "codebook", "latents"))
        self.x_mel, self.c = synthetic_batch(a.batch, rank, device)
        self.codes = None                                        # the last step's codes (code_perplexity of the line)
        # --pipeline-encode (NOT the metric's configuration): the frozen VQ-VAE's encode of the NEXT batch on a second stream
        # while this batch trains - what a prefetching data pipeline does with extract_codes.py's work; a step then consumes
        # codes produced during the step before.  Reported as a side measurement only.
        self.pipe = bool(getattr(a, "pipeline_encode", False))
        self._side = torch.cuda.Stream(device=device) if self.pipe else None
        self._next = None
        self.opt = FusedAdamW(self.gpt, lr=gargs.learning_rate, betas=(0.9, 0.95), weight_decay=0.01)
        self.opt.grad_scale = 1.0 / world
        self.dp = DataParallel(self.gpt, grad_dtype=a.grad_dtype) if world > 1 or FORCE_DP else None
        self.full = a.layers == 24
        self._one = torch.ones((), dtype=torch.float32, device=device)
        self.workload = ("VQ-encode (LitVQVAE encoder + 128-code L2 argmin on 80x848 mel tiles) + class-GPT VAS "
                         f"({a.layers} L, 1024, 16 H, T=265, V=128, dropout 0.5) fwd/bwd + AdamW")
        self.metric = "mel-token seqs/sec training step (VQ-encode + GPT fwd/bwd)"

    def step(self, mark):
        from melspec_gpt_vqvae_amd import ops
        from melspec_gpt_vqvae_amd.transformer.minGPT import cross_entropy

        t0 = mark()
        if self.pipe:
            main = torch.cuda.current_stream()
            if self._next is None:
                self._side.wait_stream(main)
                with torch.cuda.stream(self._side), torch.no_grad():
                    c0 = self.vqvae.encode_to_codes(self.x_mel)
                    self._next = (c0, ops.codes_permute(c0, 5, 53))
            main.wait_stream(self._side)                         # this batch's codes (encoded during the step before)
            codes, seq = self._next
            codes.record_stream(main)
            seq.record_stream(main)
            with torch.cuda.stream(self._side), torch.no_grad():  # the next batch's, beside this batch's training
                c1 = self.vqvae.encode_to_codes(self.x_mel)
                self._next = (c1, ops.codes_permute(c1, 5, 53))
        else:
            with torch.no_grad():
                codes = self.vqvae.encode_to_codes(self.x_mel)       # (B,5,53) int64
                seq = ops.codes_permute(codes, 5, 53)                # (B,265) time-major (get_x)
        self.codes = codes
        t1 = mark()
        with self.gpt.discard_att():                             # as Lit_minGPT.forward does: the (B,H,T,T) map it
            logits, _, _ = self.gpt(seq[:, :-1], self.c)         # ignores (`logits, _, _ = transformer(...)`) is not written
        loss = cross_entropy(logits.reshape(-1, logits.size(-1)), seq.reshape(-1))
        t2 = mark()
        self.opt.zero_grad()
        loss.backward(self._one)                                 # (a resident 1.0: autograd's own root gradient is a torch fill launch)
        if self.dp is not None:
            self.dp.finish()
        t3 = mark()
        self.opt.step()
        t4 = mark()
        return loss, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)


class GPTVAEXLStep:
    """BASELINE configs[3]: GPT-VAE XL (config_GPT_VAE_vggsound.py:43-58; the model GPT_VAE_train.py:166-190 trains
    under DDP) - encoder GPT (bidirectional, last_linear 2C) + decoder GPT, 2.09 B parameters, batch 128 per GPU,
    loss = rec + kl_weight * KL, ONE flat gradient buffer over both transformers exchanged by dp.DataParallel."""
    name = "gpt_vae_xl"
    seq_len = 265

    def __init__(self, a, device, dtype, rank, world):
        from melspec_gpt_vqvae_amd.dp import DataParallel
        from melspec_gpt_vqvae_amd.optim import FusedAdamW
        from melspec_gpt_vqvae_amd.transformer.Lit_GPT_VAE import GPT_VAE
        from melspec_gpt_vqvae_amd.transformer.minGPT import set_compute_dtype

        self.a = a
        layers = a.layers if a.layers != 24 else 40
        args = xl_args(n_layer=layers, batch_size=a.batch, device=str(device))
        torch.manual_seed(SEED)
        self.vae = GPT_VAE(args).to(device).train()
        set_compute_dtype(self.vae, dtype)
        g = torch.Generator().manual_seed(SEED + 17 * rank)
        self.x = torch.randint(0, 1024, (a.batch, 265), generator=g).to(device)
        self.opt = FusedAdamW(self.vae, lr=args.learning_rate, betas=(0.9, 0.95), weight_decay=0.01)
        self.opt.grad_scale = 1.0 / world
        self.dp = DataParallel(self.vae, grad_dtype=a.grad_dtype) if world > 1 or FORCE_DP else None
        self.full = layers == 40
        self._one = torch.ones((), dtype=torch.float32, device=device)
        self.n_params = sum(p.numel() for p in self.vae.parameters())
        self.workload = (f"GPT-VAE XL (encoder + decoder GPT, {layers}+{layers} L, 1472, 23 H, T=265, V=1024, "
                         f"{self.n_params / 1e9:.2f} B parameters) training step: ELBO fwd/bwd + AdamW, per-GPU batch "
                         f"{a.batch}, f32 gradient exchange {4 * self.n_params / 1e9:.2f} GB per step")
        self.metric = "mel-token seqs/sec training step (GPT-VAE XL fwd/bwd, BASELINE configs[3])"

    def step(self, mark):
        t0 = mark()
        total, rec, kl = self.vae.loss(self.x, 0.5, nsamples=1)
        loss = total.mean()
        t2 = mark()
        self.opt.zero_grad()
        loss.backward(self._one)
        if self.dp is not None:
            self.dp.finish()
        t3 = mark()
        self.opt.step()
        t4 = mark()
        return loss, (0.0, t2 - t0, t3 - t2, t4 - t3)


def _gemm_loops():
    import ctypes

    from melspec_gpt_vqvae_amd import _ffi

    r, p = ctypes.c_longlong(0), ctypes.c_longlong(0)
    _ffi.check(_ffi.lib().melgpt_gemm_loop_launches(ctypes.byref(r), ctypes.byref(p)), "melgpt_gemm_loop_launches")
    return r.value, p.value


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="class_gpt", choices=["class_gpt", "gpt_vae_xl"],
                    help="class_gpt = BASELINE configs[1]+[2] (the metric's configuration); gpt_vae_xl = configs[3]")
    ap.add_argument("--batch", type=int, default=128, help="sequences per GPU per step (BASELINE configs 3 and 4: 128)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "f32"],
                    help="bf16 (default, the metric's lane) | fp16 (the library's IEEE-half flavour) | f32 (parity lane)")
    ap.add_argument("--grad-dtype", default="f32", choices=["f32", "bf16"],
                    help="wire format of the data-parallel gradient exchange (N > 1): f32 (default) | bf16 = cast slice -> "
                         "all-reduce -> back into the f32 buffer (half the bytes: 8.37 -> 4.18 GB per step for GPT-VAE XL)")
    ap.add_argument("--codebook", default="latents", choices=["latents", "normal"],
                    help="frozen VQ-VAE codebook of the class_gpt workload: latents (default) = rows sampled from the encoder's "
                         "own outputs, usage-balanced -> the step's tokens spread over the 128 codes (BASELINE.md 4 asks for "
                         "uniform codes); normal = N(0,1) rows as in rounds 1-5 (tokens collapse: perplexity ~3)")
    ap.add_argument("--timer-every", type=int, default=5,
                    help="HIP-event timing of the GEMM family / attention launches on every n-th step of the timed region (1: every step)")
    ap.add_argument("--no-reference-steps", action="store_true",
                    help="skip the single-stream reference steps behind the timed region (profiled runs: every kernel of the "
                         "trace then belongs to a default step)")
    ap.add_argument("--pipeline-encode", action="store_true",
                    help="NOT the metric's configuration: encode the next batch on a second stream while this one trains (flagged in the line)")
    ap.add_argument("--no-torch-baseline", action="store_true", help="skip torch_gpu_baseline (stock PyTorch-ROCm, same box)")
    ap.add_argument("--torch-baseline-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--layers", type=int, default=24, help="debug only; anything but the configuration's depth is flagged")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip config2_vq_encode / config4_gpt_vae_xl_rank / config5_e2e_fp16 (BASELINE configs[1], [3], [4] "
                         "beside the metric): the profiled runs of tools/profile_round.sh, so that every kernel of the "
                         "trace belongs to a training step")
    ap.add_argument("--breakdown", action="store_true", help="print per-phase timings to stderr")
    a = ap.parse_args()

    if a.torch_baseline_child:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py --torch-baseline-child needs a GPU")
        torch_baseline_child(a.batch)
        return
    if a.dtype == "fp16":  # the 16-bit format is a property of the library flavour: choose it before the package loads
        os.environ["MELGPT_HALF"] = "fp16"
    share = os.environ.get("MELGPT_BENCH_SHARE_GPU") == "1"
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N` (no torch.distributed.run around it): start the N ranks here, BEFORE this
        # process makes any GPU call; rank 0 prints the JSON line on the inherited stdout, the exit status is the ranks'
        from melspec_gpt_vqvae_amd.launch import spawn_ranks

        raise SystemExit(spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], a.gpus, share_gpu=share))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # debug aid (1-GPU boxes): MELGPT_BENCH_SHARE_GPU=1 lets several ranks share cuda:0 over gloo to exercise the
    # multi-rank control flow; the numbers of such a run mean nothing and the JSON line says so
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    import torch.distributed as dist

    if world > 1 or FORCE_DP:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if FORCE_DP and world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if share:
            dist.init_process_group("gloo")
        else:
            from melspec_gpt_vqvae_amd.dp import pin_rccl_channels

            pin_rccl_channels()      # opt-in (MELGPT_RCCL_CHANNELS=n): a known RCCL footprint + that many reserved CUs (dp.py)
            dist.init_process_group("nccl", device_id=device)
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"

    from melspec_gpt_vqvae_amd import ops

    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "f32": torch.float32}[a.dtype]
    job = (ClassGPTStep if a.workload == "class_gpt" else GPTVAEXLStep)(a, device, dtype, rank, world)

    phases = {"encode": 0.0, "fwd": 0.0, "bwd": 0.0, "opt": 0.0}

    def mark():
        if a.breakdown:
            torch.cuda.synchronize()
        return time.perf_counter()

    def step():
        loss, dts = job.step(mark)
        if a.breakdown:
            for k, v in zip(phases, dts):
                phases[k] += v
        return loss

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    for k in phases:
        phases[k] = 0.0
    if job.dp is not None:
        job.dp.ex.time_events = True   # two event records per step around finish()'s waits -> exposed_comm_ms
    # Live HIP-event timing of the family's launches INSIDE the timed region, on every `--timer-every`-th of its steps
    # (default 5: steps 0, 5, 10, 15 of 20).  An event record is a marker packet in the launch queue: two around each of
    # the ~380 timed launches of a step cost 1.7-1.8 ms of a 90 ms step when every step carries them (measured A/B,
    # profiles/r06_w_event_overhead_ab.jsonl) - instrumentation the training step does not have.  `value` is the
    # throughput of all K steps, marker-carrying ones included.
    timer = ops.KernelTimer()
    every = max(1, a.timer_every)
    tsteps = len(range(0, a.steps, every))
    loops0 = _gemm_loops()
    fence()
    t0 = time.perf_counter()
    loss = None
    for i in range(a.steps):
        ops.TIMER = timer if i % every == 0 else None
        loss = step()
    ops.TIMER = None
    fence()
    elapsed = time.perf_counter() - t0
    loops1 = _gemm_loops()
    loss_val = float(loss.detach())

    # Kernel-quality reference OUTSIDE the timed region (rank 0, one GPU): more steps with everything on ONE stream (the median of five).  In
    # the timed region a Block's weight gradients run on a second stream beside the attention / LayerNorm backward and the
    # input-gradient GEMMs (transformer/minGPT.py _wgrad): the step is faster, but a launch that shares the chip lasts
    # longer, so per-launch rates and the family's time read lower there than the kernels run on their own.
    single = None
    if rank == 0 and world == 1 and not FORCE_DP and not a.no_reference_steps:
        from melspec_gpt_vqvae_amd.transformer import minGPT as _mg

        if _mg.WGRAD_SIDE:
            _mg.WGRAD_SIDE = False
            try:
                # two untimed steps (the main stream's allocator pool takes the weight-gradient partials that lived in the
                # side stream's), then five steps fenced one by one, and the MEDIAN step is the reference: a one-off host
                # stall inside one step (seen once in round 6: 21 ms inside a weight-gradient bracket) does not reach the table
                step()
                step()
                laps = []
                for _ in range(5):
                    ops.TIMER = ops.KernelTimer()
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    step()
                    torch.cuda.synchronize()
                    laps.append((time.perf_counter() - t1, ops.TIMER))
                laps.sort(key=lambda r: r[0])
                single = (laps[2][1], 1, laps[2][0])
            finally:
                ops.TIMER = None
                _mg.WGRAD_SIDE = True

    # the part of the gradient exchange the backward pass did not hide (dp.GradientExchange.exposed_ms), per step
    exposed = job.dp.ex.exposed_ms() if job.dp is not None else []
    exposed_ms = sum(exposed) / len(exposed) if exposed else 0.0
    dp_info = job.dp.describe() if job.dp is not None else None
    tmax = torch.tensor([elapsed, exposed_ms], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed, exposed_ms = float(tmax[0].item()), float(tmax[1].item())
    ms_per_step = 1e3 * elapsed / a.steps
    value = world * a.batch * a.steps / elapsed

    ks = timer.summary()
    achieved = ks["flops"] / (ks["total_ms"] * 1e-3) / 1e12 if ks["total_ms"] > 0 else 0.0
    if rank == 0:
        pmc = pmc_summary(job.name)
        # attention (HBM-bound at T = 265, hs = 64: DESIGN 4): per launch form, live HIP-event time against the algorithmic
        # bytes (ops.attn_fwd / attn_bwd) and, when the committed PMC pass has them, against the counted bytes
        aux = timer.aux_by_tag()
        attn_flops = sum(v[3] for v in aux.values())          # (FLOPs of the event-carrying steps, for step_frac)
        aux_steps = tsteps
        if single is not None:      # per-launch times from the single-stream reference steps (the backward kernel shares the
            aux, aux_steps = single[0].aux_by_tag(), single[1]   # chip with a weight-gradient GEMM in the timed region)
        attn = {}
        for tag, (n, ms, nbytes, fl) in sorted(aux.items()):
            us = 1e3 * ms / n
            row = {"launches_per_step": round(n / aux_steps, 2), "us": round(us, 2), "algorithmic_MB": round(nbytes / n / 1e6, 1),
                   "frac_hbm": round(nbytes / n / (us * 1e-6) / 8e12, 4), "tflops": round(fl / n / (us * 1e-6) / 1e12, 1)}
            counted = (pmc.get("attention_hbm_MB_per_launch") or {}).get("fwd" if " fwd " in tag else "bwd")
            if counted and " full" not in tag and job.name == "class_gpt":
                row["counted_MB"] = counted
                row["frac_hbm_counted"] = round(counted * 1e6 / (us * 1e-6) / 8e12, 4)
            attn[tag] = row
        step_flops = (ks["flops"] + attn_flops) / max(tsteps, 1)
        # per-instantiation table: one row per (layout, shape, epilogue) of the GEMM family, slowest rate first among
        # the rows that matter (>= 0.5 % of the family's time) - names the shape the family's `frac` is held down by
        # (per-shape rates from the single-stream reference steps when they exist: launches that do not share the chip)
        ptimer, psteps = (single[0], single[1]) if single is not None else (timer, tsteps)
        pks = ptimer.summary()
        rows = [{"shape": tag, "calls_per_step": round(n / psteps, 2), "ms_per_step": round(ms / psteps, 3),
                 "tflop_per_step": round(fl / psteps / 1e12, 3), "tflops": round(fl / ms / 1e9, 1) if ms > 0 else 0.0}
                for tag, n, ms, fl in ptimer.by_tag()]
        major = [r for r in rows if r["ms_per_step"] * psteps >= 0.005 * pks["total_ms"]]
        worst = min(major, key=lambda r: r["tflops"]) if major else None
        out = {
            "metric": job.metric,
            "value": round(value, 3), "unit": "seq/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "config": {
                "workload": job.workload,
                "batch_per_gpu": a.batch, "global_batch": a.batch * world, "seq_len": job.seq_len,
                "parallelism": f"dp{world}" if world > 1 else "single",
                "final_loss": round(loss_val, 4),
                "codebook": getattr(a, "codebook", None) if job.name == "class_gpt" else None,
                "code_perplexity": code_perplexity(job.codes) if getattr(job, "codes", None) is not None else None,
                # which K loop the persistent GEMM's launches of the timed region took (csrc/gemm256.hip / gemm8p.hip)
                "gemm_launches_per_step": {"ring": (loops1[0] - loops0[0]) // max(a.steps, 1),
                                           "pingpong": (loops1[1] - loops0[1]) // max(a.steps, 1)},
            },
            "roofline": {
                "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
                # the same peak over the WHOLE timed step: every algorithmic FLOP of the step (GEMM family + attention on
                # the full T x T square, backward = 2 x forward: BASELINE.md 3) / ms_per_step - what `frac` leaves out is
                # the time of everything that is not a family launch
                "step_frac": round(step_flops / (ms_per_step * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                "step_algorithmic_tflop": round(step_flops / 1e12, 3),
                "attn_frac_hbm": attn,
                "traffic": pmc.get("traffic"),
                "traffic_unit": "bytes/step", "traffic_source": pmc.get("traffic_source"),
                "mfma_busy": pmc.get("mfma_busy"), "mfma_busy_by_kernel": pmc.get("mfma_busy_by_kernel"),
                "in_kernel_clock_GHz": pmc.get("in_kernel_clock_GHz"),
                "kernel": "MFMA GEMM family (gemm8p_kernel / gemm256_kernel persistent 256x256 / gemm_kernel 128x128 + implicit-GEMM conv / "
                          "conv3x3_gn_ws_kernel with fused GroupNorm+swish), every launch of the event-carrying steps of the timed region",
                "launches_per_step": ks["launches"] // max(tsteps, 1),
                "event_timed_steps": f"{tsteps} of the {a.steps} timed steps" + (f" (every {every}th)" if every > 1 else ""),
                # time with at least one family launch running (the Block's weight gradients run on a second stream beside
                # the input-gradient chain since round 6: union of the launch intervals on the device clock) / plain sum
                "kernel_ms_per_step": round(ks["total_ms"] / max(tsteps, 1), 3),
                "kernel_serial_ms_per_step": round(ks.get("serial_ms", ks["total_ms"]) / max(tsteps, 1), 3),
                # the same family on ONE stream (the median of five reference steps outside the timed region): what the kernels do when
                # no launch shares the chip - `per_shape` / `worst_shape` below are from these steps
                "frac_single_stream": round(pks["flops"] / (pks["total_ms"] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4) if single is not None and pks["total_ms"] > 0 else None,
                "ms_per_step_single_stream": round(1e3 * single[2], 3) if single is not None else None,
                "single_stream_note": "the reference steps carry event markers on EVERY launch (+1.7-1.8 ms per step) and a device synchronisation per step: "
                                      "compare with ms_per_step only through --timer-every 1" if single is not None else None,
                "algorithmic_tflop_per_step": round(ks["flops"] / max(tsteps, 1) / 1e12, 3),
                "worst_shape": worst, "per_shape": major,
            },
        }
        if dp_info is not None:
            # a scaling record explains itself: what was exchanged, in which format, under which persistent-kernel
            # switches, and how much of it the backward pass did not hide (max over ranks)
            out["config"].update(dp_info)
            out["exposed_comm_ms"] = round(exposed_ms, 3)
        if getattr(a, "pipeline_encode", False):
            out["config"]["NOT_THE_METRIC_pipelined_encode"] = True
        if not job.full:
            out["config"]["INVALID_debug_layers"] = a.layers
        if share:
            out["config"]["INVALID_debug_shared_gpu"] = True
        if a.breakdown:
            print({k: round(1e3 * v / a.steps, 2) for k, v in phases.items()}, file=sys.stderr)
            for r in rows[:40]:
                print(f"  {r['shape']:44s} n/step={r['calls_per_step']:6.1f} ms/step={r['ms_per_step']:8.3f} "
                      f"TFLOP/s={r['tflops']:7.1f}", file=sys.stderr)
        # Side measurements (BASELINE configs[1], [3], [4] and the CPU baseline), outside the timed region.  The metric
        # above is already computed: a failure in any of them (the XL rank step peaks at 168 GB) is recorded in its own
        # field and never costs the line.
        import gc

        def side(fn, *args):
            try:
                return fn(*args)
            except Exception as e:  # noqa: BLE001 - recorded, not swallowed
                return {"error": f"{type(e).__name__}: {e}"[:400]}
            finally:
                gc.collect()
                torch.cuda.empty_cache()

        job_name, job_full = job.name, job.full
        if world == 1 and job_name == "class_gpt" and not a.no_extras:
            out["config2_vq_encode"] = side(vq_encode_b64, job, device)
        extras = world == 1 and job_name == "class_gpt" and job_full and a.batch == 128 and not a.no_extras
        if extras:
            # the class-GPT job is freed first (the XL rank step needs 168 GB)
            del job
            gc.collect()
            torch.cuda.empty_cache()
            out["config4_gpt_vae_xl_rank"] = side(gpt_vae_xl_rank, a, device, dtype)
            out["config5_e2e_fp16"] = side(e2e_fp16_child)
        if world == 1 and not a.no_cpu_baseline and out["metric"].startswith("mel-token seqs/sec training step (VQ"):
            out["cpu_baseline"] = side(cpu_baseline)
        if extras and not a.no_torch_baseline:
            tb = out["torch_gpu_baseline"] = side(torch_gpu_baseline)
            if tb.get("value"):
                tb["this_over_torch"] = round(out["value"] / tb["value"], 2)
        # Scalars a reader of the record needs without the nested tables (a driver that keeps only flat fields of
        # `roofline` / `config` still holds them), and one compact `summary` object as the LAST key of the line.
        rf, cfg = out["roofline"], out["config"]
        cp = cfg.get("code_perplexity")
        if isinstance(cp, dict):
            cfg["code_perplexity"], cfg["code_distinct"], cfg["code_top_share"] = cp["perplexity"], cp["distinct"], cp["top_share"]
        for tag, row in attn.items():
            key = "attn_" + ("fwd" if " fwd " in tag else "bwd") + ("_full" if " full" in tag else "")
            rf[key + "_us"], rf[key + "_frac_hbm"] = row["us"], row.get("frac_hbm_counted", row["frac_hbm"])
        c2 = out.get("config2_vq_encode") or {}
        for k, v in (c2.get("lookup_sweep") or {}).items():
            rf[f"vq_lookup_frac_hbm_{k}"] = v["frac_hbm"]
        if (out.get("torch_gpu_baseline") or {}).get("value"):
            rf["torch_rocm_same_box_seq_per_s"] = out["torch_gpu_baseline"]["value"]
            rf["this_over_torch_rocm"] = out["torch_gpu_baseline"]["this_over_torch"]
        c5 = out.get("config5_e2e_fp16") or {}
        out["summary"] = {
            "seq_per_s": out["value"], "ms_per_step": out["ms_per_step"], "frac": rf["frac"], "step_frac": rf["step_frac"],
            "frac_single_stream": rf.get("frac_single_stream"), "ms_per_step_single_stream": rf.get("ms_per_step_single_stream"),
            "code_perplexity": cfg.get("code_perplexity"),
            "attn": {k: [v["us"], v.get("frac_hbm_counted", v["frac_hbm"])] for k, v in attn.items()},
            "vq_lookup_frac_hbm": {k: v["frac_hbm"] for k, v in (c2.get("lookup_sweep") or {}).items()},
            "config2_ms": c2.get("ms"), "config4_rank_ms": (out.get("config4_gpt_vae_xl_rank") or {}).get("ms_per_step"),
            "config5": {"b1_p50_ms": (c5.get("batch1_latency_ms") or {}).get("p50"), "b64_clips_per_s": c5.get("batch64_clips_per_s"),
                        "ms_per_token": c5.get("ms_per_token"), "decode_frac_hbm": c5.get("decode_frac_hbm")},
            "torch_rocm_seq_per_s": (out.get("torch_gpu_baseline") or {}).get("value"),
            "cpu_seq_per_s": (out.get("cpu_baseline") or {}).get("value"),
            "per_shape_tflops": {r["shape"]: r["tflops"] for r in major[:14]},
        }
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
