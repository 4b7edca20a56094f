"""wav -> log-mel spectrogram on the MI355X - host-side mirror of the reference's
feature_extraction/extract_mel_spectrogram.py (MelSpectrogram :15-38, the small transform classes :40-138,
TRANSFORMS :141-151, get_spectrogram :166-190).

The whole chain  |STFT| -> mel filterbank -> max(1e-5) -> log10 -> *20 -20 +100 /100 -> clip[0,1] -> [:, :860]
runs as ONE HIP kernel (csrc/mel.hip, melgpt_mel_frontend_fwd); `TRANSFORMS` validates that its stages are the
reference's chain and passes their constants to the kernel.  The per-stage classes keep the reference's names and
constructor signatures (so TRANSFORMS.transforms can be inspected the same way), but they only carry parameters:
there is no NumPy data path in this package.  The inverse direction (mel_to_stft + Griffin-Lim, :29-34) is not on
the hot path - the reference itself uses the MelGAN vocoder instead - and raises.

librosa (0.8.1) is not a dependency: the Slaney mel filterbank is built here from its published definition
(htk=False, norm='slaney', float32), and the STFT semantics (n_fft = win_length, periodic Hann, center=True,
reflect padding, frames = 1 + len//hop) are implemented by the kernel.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from .. import _ffi


# ------------------------------------------------------------------------------ Slaney mel filterbank (host setup)
def _hz_to_mel(f):
    f = np.atleast_1d(np.asarray(f, dtype=np.float64))
    mel = f / (200.0 / 3)
    log_t = f >= 1000.0
    mel[log_t] = 15.0 + np.log(f[log_t] / 1000.0) / (np.log(6.4) / 27.0)
    return mel


def _mel_to_hz(m):
    m = np.atleast_1d(np.asarray(m, dtype=np.float64))
    f = (200.0 / 3) * m
    log_t = m >= 15.0
    f[log_t] = 1000.0 * np.exp((np.log(6.4) / 27.0) * (m[log_t] - 15.0))
    return f


def mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """Triangular, area-normalised filters on the Slaney scale -> (n_mels, 1 + n_fft//2) float32."""
    bins = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    edges = _mel_to_hz(np.linspace(_hz_to_mel(fmin)[0], _hz_to_mel(fmax)[0], n_mels + 2))
    width = np.diff(edges)
    ramps = edges[:, None] - bins[None, :]
    fb = np.zeros((n_mels, bins.size), dtype=np.float32)
    for i in range(n_mels):
        fb[i] = np.maximum(0.0, np.minimum(-ramps[i] / width[i], ramps[i + 2] / width[i + 1]))
    fb *= (2.0 / (edges[2:] - edges[:-2]))[:, None]
    return fb


class MelSpectrogram(object):
    def __init__(self, sr, nfft, fmin, fmax, nmels, hoplen, spec_power, inverse=False):
        self.sr, self.nfft, self.fmin, self.fmax = sr, nfft, fmin, fmax
        self.nmels, self.hoplen, self.spec_power, self.inverse = nmels, hoplen, spec_power, inverse
        self.mel_basis = mel_filterbank(sr, nfft, nmels, fmin, fmax)

    def __call__(self, x):
        if self.inverse:
            raise NotImplementedError("mel -> waveform (Griffin-Lim) is outside the mel->VQ->GPT hot path")
        raise _ffi.MelgptError("MelSpectrogram is a stage of the fused HIP kernel: call TRANSFORMS(y) / wav_to_mel(y)")


class _Stage(object):
    def __call__(self, x):
        raise _ffi.MelgptError(f"{type(self).__name__} is a stage of the fused HIP kernel: call TRANSFORMS(y)")


class LowerThresh(_Stage):
    def __init__(self, min_val, inverse=False):
        self.min_val, self.inverse = min_val, inverse


class Add(_Stage):
    def __init__(self, val, inverse=False):
        self.val, self.inverse = val, inverse


class Subtract(Add):
    pass


class Multiply(_Stage):
    def __init__(self, val, inverse=False):
        self.val, self.inverse = val, inverse


class Divide(Multiply):
    pass


class Log10(_Stage):
    def __init__(self, inverse=False):
        self.inverse = inverse


class Clip(_Stage):
    def __init__(self, min_val, max_val, inverse=False):
        self.min_val, self.max_val, self.inverse = min_val, max_val, inverse


class TrimSpec(_Stage):
    def __init__(self, max_len, inverse=False):
        self.max_len, self.inverse = max_len, inverse


class FusedTransforms(object):
    """Stand-in for torchvision.transforms.Compose([...]) over the reference's chain (:141-151)."""

    def __init__(self, transforms):
        self.transforms = list(transforms)
        kinds = [type(t) for t in self.transforms]
        if kinds != [MelSpectrogram, LowerThresh, Log10, Multiply, Subtract, Add, Divide, Clip, TrimSpec]:
            raise _ffi.MelgptError("only the reference's mel chain is fused: MelSpectrogram, LowerThresh, Log10, "
                                   "Multiply, Subtract, Add, Divide, Clip, TrimSpec")
        m = self.transforms[0]
        if m.spec_power != 1 or m.nfft != 1024:
            raise _ffi.MelgptError("the kernel is built for |STFT|**1 with n_fft = 1024")
        self._dev_tables = {}

    def _tables(self, device):
        key = str(device)
        if key not in self._dev_tables:
            fb = self.transforms[0].mel_basis
            nz = fb > 0
            lo = np.where(nz.any(1), nz.argmax(1), 0).astype(np.int32)
            hi = np.where(nz.any(1), fb.shape[1] - 1 - nz[:, ::-1].argmax(1), -1).astype(np.int32)
            self._dev_tables[key] = (torch.from_numpy(fb).to(device), torch.from_numpy(lo).to(device),
                                     torch.from_numpy(hi).to(device))
        return self._dev_tables[key]

    def run(self, wav, *, want_mel=True, tile_dtype=None, crop0=6, crop_len=848):
        """wav (n_clips, L) f32 CUDA tensor -> (mel (n,80,860) f32 | None, tile (n,1,80,848) | None)."""
        if not wav.is_cuda:
            raise _ffi.MelgptError("the mel frontend runs on the GPU only (no CPU fallback)")
        m, lt, _, mul, sub, add, div, clip, trim = self.transforms
        wav = wav.contiguous()
        n, L = wav.shape
        fb, lo, hi = self._tables(wav.device)
        mel = torch.empty(n, m.nmels, trim.max_len, dtype=torch.float32, device=wav.device) if want_mel else None
        tile = torch.empty(n, 1, m.nmels, crop_len, dtype=tile_dtype, device=wav.device) if tile_dtype is not None else None
        _ffi.call("melgpt_mel_frontend_fwd", _ffi.ptr(wav), n, L, m.nfft, m.hoplen, _ffi.ptr(fb), _ffi.ptr(lo),
                  _ffi.ptr(hi), m.nmels, float(lt.min_val), float(mul.val), float(sub.val), float(add.val),
                  float(div.val), float(clip.min_val), float(clip.max_val), _ffi.ptr(mel), trim.max_len, _ffi.ptr(tile),
                  _ffi.dtype_code(tile_dtype) if tile_dtype is not None else 0, crop0, crop_len, _ffi.stream())
        return mel, tile

    def tail(self, mel_mag, *, tile_dtype=None, crop0=6, crop_len=848):
        """`Compose(TRANSFORMS.transforms[1:])` of the reference (:143-150) on mel magnitudes (n, n_mels, n_frames) f32
        that are already on the GPU -> (mel (n, n_mels, max_len) f32, tile | None): the kernel's own last stage."""
        if not mel_mag.is_cuda:
            raise _ffi.MelgptError("the mel frontend runs on the GPU only (no CPU fallback)")
        _, lt, _, mul, sub, add, div, clip, trim = self.transforms
        mel_mag = mel_mag.contiguous().float()
        n, nm, nf = mel_mag.shape
        keep = min(trim.max_len, nf)
        mel = torch.empty(n, nm, keep, dtype=torch.float32, device=mel_mag.device)
        tile = torch.empty(n, 1, nm, crop_len, dtype=tile_dtype, device=mel_mag.device) if tile_dtype is not None else None
        _ffi.call("melgpt_mel_transforms_fwd", _ffi.ptr(mel_mag), n, nm, nf, float(lt.min_val), float(mul.val),
                  float(sub.val), float(add.val), float(div.val), float(clip.min_val), float(clip.max_val), _ffi.ptr(mel),
                  keep, _ffi.ptr(tile), _ffi.dtype_code(tile_dtype) if tile_dtype is not None else 0, crop0, crop_len,
                  _ffi.stream())
        return mel, tile

    def __call__(self, y):
        """y: 1-D waveform (numpy or torch) -> (80, 860) numpy array, like the reference's TRANSFORMS(y)."""
        yt = torch.as_tensor(np.asarray(y, dtype=np.float32) if not torch.is_tensor(y) else y.float())
        mel, _ = self.run(yt.reshape(1, -1).to("cuda"))
        return mel[0].cpu().numpy()


TRANSFORMS = FusedTransforms([
    MelSpectrogram(sr=22050, nfft=1024, fmin=125, fmax=7600, nmels=80, hoplen=1024 // 4, spec_power=1),
    LowerThresh(1e-5),
    Log10(),
    Multiply(20),
    Subtract(20),
    Add(100),
    Divide(100),
    Clip(0, 1.0),
    TrimSpec(860)
])


def inv_transforms(x, folder_name='melspec_10s_22050hz'):
    raise NotImplementedError("inverse transforms (Griffin-Lim) are outside the mel->VQ->GPT hot path")


def fit_length(wav, length):
    """zero-pad or truncate to `length` samples (get_spectrogram :169-173)."""
    wav = np.asarray(wav)
    y = np.zeros(length, dtype=np.float32)
    n = min(len(wav), length)
    y[:n] = wav[:n]
    return y


def wav_to_mel(wavs, device="cuda", tile_dtype=None):
    """batch API: list/array of equal-length waveforms -> (mel (n,80,860) f32, VQ-VAE input tile or None), on device."""
    w = torch.as_tensor(np.stack([np.asarray(x, dtype=np.float32) for x in wavs])) if not torch.is_tensor(wavs) else wavs
    return TRANSFORMS.run(w.to(device).float(), tile_dtype=tile_dtype)


def get_spectrogram(audio_path, save_dir, length, folder_name='melspec_10s_22050hz', save_results=True):
    """reference :166-190.  Decoding the audio file is host I/O (the reference uses librosa.load(sr=None)); here the
    waveform is read with the standard-library `wave` module (16-bit PCM, mono or first channel)."""
    import wave

    with wave.open(audio_path, "rb") as f:
        n, ch, sw = f.getnframes(), f.getnchannels(), f.getsampwidth()
        raw = f.readframes(n)
    if sw != 2:
        raise ValueError("only 16-bit PCM wav files are supported")
    wav = np.frombuffer(raw, dtype="<i2").reshape(-1, ch).astype(np.float32).mean(axis=1) / 32768.0
    y = fit_length(wav, length)
    if folder_name != 'melspec_10s_22050hz':
        raise NotImplementedError
    mel_spec = TRANSFORMS(y)
    if len(wav) < length:
        # reference :169-171: a zero-padded clip is a float64 array and so is everything computed from it (the saved
        # `_mel.npy` is float64); a truncated clip keeps librosa.load's float32.  The kernel computes in f32 either way.
        y, mel_spec = y.astype(np.float64), mel_spec.astype(np.float64)
    if save_results:
        os.makedirs(save_dir, exist_ok=True)
        audio_name = os.path.basename(audio_path).split('.')[0]
        np.save(os.path.join(save_dir, audio_name + '_mel.npy'), mel_spec)
    else:
        return y, mel_spec
