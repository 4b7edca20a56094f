"""The mel-frontend oracle (oracle/mel.py) is PARITY-UNPINNED against the reference (librosa 0.8.1 is neither
vendored nor installed, and the reference has no test at this boundary).  These CPU tests pin it to what CAN be
checked here: torch.stft with librosa's framing conventions, and analytic known answers."""
import numpy as np
import torch

import synth
from oracle import mel as om


def test_stft_matches_torch_stft_with_librosa_conventions():
    y = synth.waveform(1, n=22050)
    mag = om.stft_mag(y)
    assert mag.shape == (513, 1 + 22050 // 256)
    ref = torch.stft(torch.from_numpy(y), 1024, 256, window=torch.hann_window(1024, periodic=True, dtype=torch.float64),
                     center=True, pad_mode="reflect", return_complex=True).abs().numpy()
    assert np.abs(mag - ref).max() < 1e-9


def test_frame_count_and_trim_for_a_10s_clip():
    y = synth.waveform(2)
    assert len(y) == 220500
    m = om.log_mel(y)
    assert om.stft_mag(y).shape == (513, 862) and m.shape == (80, 860)
    assert m.min() >= 0.0 and m.max() <= 1.0
    x = om.crop_and_scale(m)
    assert x.shape == (80, 848) and np.array_equal(x, 2 * m[:, 6:854] - 1)


def test_silence_maps_to_exactly_zero():
    m = om.log_mel(np.zeros(220500))
    assert np.all(m == 0.0)   # max(1e-5, 0) -> log10 = -5 -> (-100 - 20 + 100)/100 = -0.2 -> clip -> 0


def test_filterbank_is_slaney_triangles():
    fb = om.mel_filterbank()
    assert fb.shape == (80, 513) and fb.dtype == np.float32 and fb.min() >= 0
    freqs = np.linspace(0, 11025, 513)
    centers = freqs[fb.argmax(1)]
    assert np.all(np.diff(centers) > 0) and centers[0] > 125 and centers[-1] < 7600
    # each row: rises to one peak then falls (triangle), zero outside
    for r in fb:
        nz = np.nonzero(r)[0]
        assert np.all(np.diff(nz) == 1)
        k = r.argmax()
        assert np.all(np.diff(r[nz[0]:k + 1]) >= 0) and np.all(np.diff(r[k:nz[-1] + 1]) <= 0)
    # slaney area normalisation: integral of each filter over Hz is ~1 (exactly 1 for the continuous triangle)
    area = (fb * (11025 / 512)).sum(1)
    assert np.all(np.abs(area - 1.0) < 0.2)
    # scale: linear below 1 kHz (200/3 Hz per mel), logarithmic above (step ln(6.4)/27)
    assert abs(om.hz_to_mel_slaney(1000.0) - 15.0) < 1e-12
    assert abs(om.mel_to_hz_slaney(np.array([15.0 + 27.0]))[0] - 6400.0) < 1e-6


def test_sine_peaks_in_the_nearest_band():
    t = np.arange(220500) / 22050.0
    m = om.log_mel(0.5 * np.sin(2 * np.pi * 1000.0 * t))
    fb = om.mel_filterbank()
    freqs = np.linspace(0, 11025, 513)
    centers = freqs[fb.argmax(1)]
    band = int(np.abs(centers - 1000.0).argmin())
    assert abs(int(m[:, 100:700].mean(1).argmax()) - band) <= 1


def test_fit_length_dtype_rule():
    short = np.ones(1000, dtype=np.float32)
    long = np.ones(300000, dtype=np.float32)
    assert om.fit_length(short).dtype == np.float64 and len(om.fit_length(short)) == 220500
    assert om.fit_length(long).dtype == np.float32 and len(om.fit_length(long)) == 220500
