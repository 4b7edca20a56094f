"""Oracle: wav -> log-mel tile, numpy restatement.  TEST INFRASTRUCTURE - see oracle/__init__.py.

PINNED (tests/golden/mel_transforms.npz, recorded from the real reference by make_golden.gen_mel): the transform
tail `transform_tail` (= TRANSFORMS.transforms[1:], :143-150), the two numpy lines of MelSpectrogram.__call__
(`mel_project`, :36-37), get_spectrogram's pad / truncate + dtype rule (`fit_length`, :169-173), the keyword arguments
the reference hands to librosa, and the saved file's name / shape / dtype - bit for bit, f32 and f64.
PARITY UNPINNED: `stft_mag` and `mel_filterbank`, i.e. the arithmetic INSIDE librosa==0.8.1 (pinned in the
reference's requirements.txt:2 / MSGVenv.yml:84) which is neither vendored under
/root/reference nor installed here, and the reference has no test at this boundary.
These two restate the *published* librosa-0.8.1 algorithm for the exact call sites
  feature_extraction/extract_mel_spectrogram.py:26   librosa.filters.mel(sr, n_fft, fmin, fmax, n_mels)
  feature_extraction/extract_mel_spectrogram.py:36   np.abs(librosa.stft(x, n_fft, hop_length)) ** power
  feature_extraction/extract_mel_spectrogram.py:37   np.dot(mel_basis, spec)
  feature_extraction/extract_mel_spectrogram.py:141-151  TRANSFORMS
  feature_extraction/extract_mel_spectrogram.py:166-190  get_spectrogram (pad / truncate to `length`)
and is cross-checked against torch.stft and analytic known answers in tests/test_mel_oracle.py.

librosa 0.8.1 defaults that matter: win_length = n_fft, window = scipy.signal.get_window('hann',
n_fft, fftbins=True) (periodic Hann), center=True with np.pad(mode='reflect') of n_fft//2 each side,
n_frames = 1 + len(y)//hop; filters.mel: htk=False (Slaney scale), norm='slaney', dtype float32.
"""
from __future__ import annotations

import numpy as np

SR, N_FFT, HOP, N_MELS, FMIN, FMAX = 22050, 1024, 256, 80, 125, 7600  # :142
SPEC_LEN, CROP_LEN = 860, 848                                           # :150; extract_codes crop


def hz_to_mel_slaney(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    if f.ndim:
        m = f >= min_log_hz
        mels[m] = min_log_mel + np.log(f[m] / min_log_hz) / logstep
    elif f >= min_log_hz:
        mels = min_log_mel + np.log(f / min_log_hz) / logstep
    return mels


def mel_to_hz_slaney(mels):
    mels = np.asanyarray(mels, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * mels
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    m = mels >= min_log_mel
    freqs[m] = min_log_hz * np.exp(logstep * (mels[m] - min_log_mel))
    return freqs


def mel_filterbank(sr=SR, n_fft=N_FFT, n_mels=N_MELS, fmin=FMIN, fmax=FMAX):
    """librosa.filters.mel (0.8.1): triangular filters on the Slaney mel scale, area-normalised,
    returned as float32 (n_mels, 1 + n_fft//2)."""
    weights = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float32)
    fftfreqs = np.linspace(0, float(sr) / 2, 1 + n_fft // 2, endpoint=True)
    mel_f = mel_to_hz_slaney(np.linspace(hz_to_mel_slaney(fmin), hz_to_mel_slaney(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights


def hann_periodic(n):
    """scipy.signal.get_window('hann', n, fftbins=True)."""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def stft_mag(y, n_fft=N_FFT, hop=HOP):
    """|librosa.stft(y, n_fft, hop)|: reflect-pad n_fft//2, frame, periodic Hann, rfft.  (513, 1+len//hop)."""
    y = np.asarray(y)
    yp = np.pad(y.astype(np.float64), n_fft // 2, mode="reflect")
    n_frames = 1 + (len(yp) - n_fft) // hop
    idx = np.arange(n_fft)[:, None] + hop * np.arange(n_frames)[None, :]
    frames = yp[idx] * hann_periodic(n_fft)[:, None]
    spec = np.fft.rfft(frames, axis=0)
    if y.dtype == np.float32:  # librosa stores complex64 for float32 input (util.dtype_r2c)
        spec = spec.astype(np.complex64)
    return np.abs(spec)


def fit_length(wav, length=220500):
    """get_spectrogram :169-173: zero-pad (-> float64) or truncate (keeps the wav dtype)."""
    wav = np.asarray(wav)
    if wav.shape[0] < length:
        y = np.zeros(length)
        y[:len(wav)] = wav
        return y
    return wav[:length]


def mel_project(mel_basis, spec, power=1):
    """MelSpectrogram.__call__ :36-37 around librosa.stft's result: np.abs(spec) ** spec_power, then np.dot."""
    return np.dot(mel_basis, np.abs(spec) ** power)


def transform_tail(m):
    """TRANSFORMS.transforms[1:] :143-150, stage by stage in the reference's order and dtype (f32 stays f32):
    LowerThresh(1e-5), Log10, Multiply(20), Subtract(20), Add(100), Divide(100), Clip(0, 1), TrimSpec(860)."""
    m = np.maximum(1e-5, m)
    m = np.log10(m)
    m = m * 20
    m = m - 20
    m = m + 100
    m = m / 100
    m = np.clip(m, 0, 1.0)
    return m[:, :SPEC_LEN]


def log_mel(y, mel_basis=None, stft=None):
    """TRANSFORMS :141-151 on a waveform that already has its final length -> (80, 860) in [0,1].  `stft` (complex
    spectrum of y) and `mel_basis` default to the restatements of librosa above."""
    if mel_basis is None:
        mel_basis = mel_filterbank()
    mag = stft_mag(y) if stft is None else np.abs(stft(y))
    return transform_tail(np.dot(mel_basis, mag))


def crop_and_scale(mel):
    """extract_codes.py:42-43 / datasets/vas.py:81: CenterCrop(80,848) -> columns [6:854]; 2x-1."""
    x1 = (mel.shape[-1] - CROP_LEN) // 2
    return 2 * mel[..., x1:x1 + CROP_LEN] - 1
