"""melspec_gpt_vqvae_amd - MI355X-native (gfx950) implementation of the mel-spectrogram ->
VQ-codebook -> GPT hot path of karchkha/MelSpec_GPT_VQVAE.

Layout (only what the path needs):
  csrc/                 hand-written HIP kernels + the C ABI (include/melgpt.h) -> lib/libmelgpt_hip.so
  _ffi.py               ctypes binding (raw device pointers, current HIP stream)
  vqvae/                host-side mirror of the reference's vqvae/big_model_attn_gan.py
  transformer/          host-side mirror of the reference's transformer/{minGPT,encoders,decoders}.py
  feature_extraction/   host-side mirror of feature_extraction/extract_mel_spectrogram.py
"""
__version__ = "0.1.0"
