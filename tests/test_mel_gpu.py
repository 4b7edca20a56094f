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


def test_many_clips_odd_length_and_wide_filter_fallback():
    """More blocks than the device holds workgroups (the persistent loop iterates), a clip length that is odd and not
    a multiple of the hop, and a 16-filter bank over the whole spectrum whose dense blocks do not fit the kernel's LDS
    budget (the banded fallback)."""
    from melspec_gpt_vqvae_amd.feature_extraction import extract_mel_spectrogram as fe

    n, L = 24, 220500
    base = [synth.waveform(40 + i).astype(np.float32) for i in range(3)]
    wavs = [np.roll(base[i % 3], 977 * i) for i in range(n)]
    mel, _ = fe.wav_to_mel(wavs)
    for i in (0, 7, 23):
        assert np.abs(mel[i].cpu().numpy() - om.log_mel(wavs[i])).max() < 1e-4
    odd = synth.waveform(44, n=33333).astype(np.float32)                 # 131 frames; frames 129, 130 reflect at the end
    wide = fe.FusedTransforms([fe.MelSpectrogram(sr=22050, nfft=1024, fmin=0, fmax=11025, nmels=16, hoplen=256, spec_power=1),
                               fe.LowerThresh(1e-5), fe.Log10(), fe.Multiply(20), fe.Subtract(20), fe.Add(100),
                               fe.Divide(100), fe.Clip(0, 1.0), fe.TrimSpec(128)])
    for tr, basis in ((fe.TRANSFORMS, None), (wide, om.mel_filterbank(n_mels=16, fmin=0.0, fmax=11025.0))):
        t2 = fe.FusedTransforms(list(tr.transforms[:-1]) + [fe.TrimSpec(128)])
        got, _ = t2.run(torch.from_numpy(odd).reshape(1, -1).cuda())
        m = np.dot(om.mel_filterbank() if basis is None else basis, om.stft_mag(odd))
        ref = np.clip((np.log10(np.maximum(1e-5, m)) * 20 - 20 + 100) / 100, 0, 1.0)[:, :128]
        assert got.shape[1:] == ref.shape
        assert np.abs(got[0].cpu().numpy() - ref).max() < 1e-4


def test_transform_tail_against_the_real_reference_1e4():
    """M2 pinned: the kernel's last stage (melgpt_mel_transforms_fwd = the device function the fused kernel ends with)
    against TRANSFORMS.transforms[1:] of the REAL reference recorded in tests/golden/mel_transforms.npz."""
    from melspec_gpt_vqvae_amd.feature_extraction.extract_mel_spectrogram import TRANSFORMS
    from util import golden

    g = golden("mel_transforms")
    assert [type(f).__name__ for f in TRANSFORMS.transforms] == list(g["chain"])
    m32 = synth.mel_matrix(int(g["mel_matrix_seed"])).astype(np.float32)
    x = torch.from_numpy(np.stack([m32, m32[::-1].copy()])).cuda()
    mel, tile = TRANSFORMS.tail(x, tile_dtype=torch.float32)
    assert mel.shape == (2, 80, 860) and tile.shape == (2, 1, 80, 848)
    got = mel.cpu().numpy()
    assert np.abs(got[0] - g["out32"]).max() < 1e-4 and np.abs(got[1] - g["out32"][::-1]).max() < 1e-4
    assert np.abs(got[0] - g["out64"]).max() < 1e-4
    assert np.all(got[0][g["out32"] == 0.0] == 0.0) and np.all(got[0][g["out32"] == 1.0] == 1.0)   # clip edges exact
    assert np.abs(tile[0, 0].cpu().numpy() - (2 * g["out32"][:, 6:854] - 1)).max() < 2e-4


def test_get_spectrogram_host_rule_matches_the_reference(tmp_path):
    """M3's host half pinned: pad / truncate, dtypes, file name and shape as the real reference produced them."""
    import wave

    from melspec_gpt_vqvae_amd.feature_extraction import extract_mel_spectrogram as fe
    from util import golden

    g = golden("mel_transforms")
    for tag in ("short", "long"):
        wav = synth.standin_wav(tag)
        pcm = np.clip(np.round(wav * 32768.0), -32768, 32767).astype("<i2")
        path = str(tmp_path / f"{tag}_clip.wav")
        with wave.open(path, "wb") as f:
            f.setnchannels(1); f.setsampwidth(2); f.setframerate(22050); f.writeframes(pcm.tobytes())
        y, mel = fe.get_spectrogram(path, None, 220500, save_results=False)
        assert str(y.dtype) == str(g[f"{tag}.y_dtype"]) and str(mel.dtype) == str(g[f"{tag}.mel_dtype"])
        n = min(len(wav), 220500)
        assert len(y) == 220500 and np.array_equal(y[:n], (pcm[:n].astype(np.float32) / 32768.0))
        assert np.all(y[n:] == 0.0) and mel.shape == (80, 860)
    out = tmp_path / "melspec_10s_22050hz"
    assert fe.get_spectrogram(str(tmp_path / "short_clip.wav"), str(out), 220500) is None
    import os
    assert sorted(os.listdir(out)) == list(g["saved_names"])
    saved = np.load(out / "short_clip_mel.npy")
    assert list(saved.shape) == list(g["saved_shape"]) and str(saved.dtype) == str(g["saved_dtype"])
