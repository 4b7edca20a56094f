"""CPU oracle for the mel -> VQ -> GPT hot path.  TEST INFRASTRUCTURE ONLY.

A plain PyTorch-CPU / numpy / C restatement (fp32) of the reference algorithm
(karchkha/MelSpec_GPT_VQVAE).  Every function cites the reference file:line it follows.
Only `tests/`, `__graft_entry__.smoke()` and the baseline legs of `bench.py` (`cpu_baseline`: these
functions on the host cores; `torch_gpu_baseline`: the same functions on `cuda` through stock
PyTorch-ROCm, in a child process - what a user of the reference gets on the same box) may
import or execute anything in this directory, and only as the checker / the baseline beside the
measurement - never as the thing measured or shipped.  The product package (`melspec_gpt_vqvae_amd/`) must never
import it; `tests/test_layout.py::test_product_never_imports_oracle` enforces that.

Pinning status
  * gpt.py, vqvae.py (VQ, encoder, decoder), ordering, optimizer groups:  PINNED against
    outputs of the real reference run in the build container - tests/golden/*.npz, made
    by tests/golden/make_golden.py (tests/test_oracle_vs_golden.py, runs on CPU).
  * vq_argmin.c: PINNED through the same VQ fixtures (indices bit-exact outside the
    near-tie list stored with each fixture).
  * mel.py: PARITY UNPINNED.  The arithmetic lives in librosa 0.8.1 (requirements.txt:2),
    which is neither vendored in the reference nor installed here, and the reference has
    no test or golden vector at this boundary.  It is cross-checked against torch.stft and
    analytic known-answer tests only (tests/test_mel_oracle.py).
"""
