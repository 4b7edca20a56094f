"""GPU parity of the fused wav -> log-mel kernel (csrc/mel.hip) against the CPU oracle (oracle/mel.py; parity
UNPINNED w.r.t. librosa, see its header).  Tolerance 1e-4 absolute on the [0,1] log-mel (north star)."""
import numpy as np
import pytest
import torch

import synth
from oracle import mel as om

pytestmark = pytest.mark.gpu


def test_wav_to_mel_matches_oracle_1e4():
    from melspec_gpt_vqvae_amd.feature_extraction.extract_mel_spectrogram import TRANSFORMS, wav_to_mel

    wavs = [synth.waveform(10 + i).astype(np.float32) for i in range(3)]
    mel, tile = wav_to_mel(wavs, tile_dtype=torch.float32)
    assert mel.shape == (3, 80, 860) and tile.shape == (3, 1, 80, 848)
    for i, w in enumerate(wavs):
        ref = om.log_mel(w)
        err = np.abs(mel[i].cpu().numpy() - ref).max()
        assert err < 1e-4, err
        assert np.abs(tile[i, 0].cpu().numpy() - om.crop_and_scale(ref)).max() < 2e-4
    one = TRANSFORMS(wavs[0])
    assert isinstance(one, np.ndarray) and one.shape == (80, 860)
    assert np.array_equal(one, mel[0].cpu().numpy())
    _, tb = wav_to_mel(wavs, tile_dtype=torch.bfloat16)
    assert np.abs(tb.float().cpu().numpy() - tile.cpu().numpy()).max() < 8e-3


def test_known_answers_silence_and_sine_and_short_clip():
    from melspec_gpt_vqvae_amd.feature_extraction.extract_mel_spectrogram import fit_length, wav_to_mel

    mel, _ = wav_to_mel([np.zeros(220500, dtype=np.float32)])
    assert float(mel.abs().max()) == 0.0
    t = np.arange(220500) / 22050.0
    s = (0.5 * np.sin(2 * np.pi * 1000.0 * t)).astype(np.float32)
    mel, _ = wav_to_mel([s])
    ref = om.log_mel(s)
    assert np.abs(mel[0].cpu().numpy() - ref).max() < 1e-4
    assert mel.min() >= 0 and mel.max() <= 1
    short = fit_length(synth.waveform(3, n=50000), 220500)   # zero-padded like get_spectrogram
    mel, _ = wav_to_mel([short])
    assert np.abs(mel[0].cpu().numpy() - om.log_mel(om.fit_length(synth.waveform(3, n=50000)))).max() < 1e-4
    assert float(mel[0, :, 400:].abs().max()) == 0.0          # the padded tail is silence -> exactly 0
