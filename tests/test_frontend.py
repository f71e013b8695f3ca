"""Host front-end (numpy restatement of torchaudio Resample + MFCC).  PARITY UNPINNED against torchaudio itself (not
installed anywhere we can run); checked here against an independent float64 evaluation of the same published formulae
and closed-form cases."""
import math
import os

import numpy as np
import pytest

from talkshow_amd import frontend as fe

REF_AUDIO = "/root/reference/demo_audio"


def _mfcc_f64(wave, sr, n_fft=2048, hop=734, n_mels=256, n_mfcc=64):
    x = np.pad(wave.astype(np.float64), (n_fft // 2, n_fft // 2), mode="reflect")
    T = 1 + (len(x) - n_fft) // hop
    w = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)
    k = np.arange(n_fft // 2 + 1)[:, None] * np.arange(n_fft)[None, :]
    E = np.exp(-2j * np.pi * k / n_fft)                                   # explicit DFT matrix (no FFT library)
    P = np.stack([np.abs(E @ (x[t * hop:t * hop + n_fft] * w)) ** 2 for t in range(T)])
    fb = fe.melscale_fbanks(n_fft // 2 + 1, 0.0, sr // 2, n_mels, sr).astype(np.float64)
    db = 10 * np.log10(np.maximum(P @ fb, 1e-10))
    db = np.maximum(db, db.max() - 80.0)
    return (db @ fe.create_dct(n_mfcc, n_mels).astype(np.float64)).T


def test_mfcc_against_float64_dft():
    rng = np.random.default_rng(0)
    sr = 22000
    t = np.arange(6000) / sr
    wave = (0.3 * np.sin(2 * np.pi * 440 * t) + 0.05 * rng.standard_normal(t.size)).astype(np.float32)
    got = fe.mfcc(wave, sr)
    ref = _mfcc_f64(wave, sr)
    assert got.shape == ref.shape == (64, 6000 // 734 + 1)
    np.testing.assert_allclose(got, ref, atol=2e-2, rtol=1e-4)          # coefficients are O(100); fp32 pipeline


def test_dct_and_filterbank_properties():
    d = fe.create_dct(64, 256).astype(np.float64)
    np.testing.assert_allclose(d.T @ d, np.eye(64), atol=1e-6)           # orthonormal DCT-II rows
    fb = fe.melscale_fbanks(1025, 0.0, 11000.0, 256, 22000)
    assert fb.shape == (1025, 256) and fb.min() >= 0 and fb.max() <= 1.0 + 1e-6
    peaks = fb.argmax(0)
    assert np.all(np.diff(peaks) >= 0)                                   # centre frequencies increase


def test_resample_preserves_a_tone_and_length():
    sr0, sr1 = 16000, 22000
    t = np.arange(16000) / sr0
    x = np.sin(2 * np.pi * 1000 * t).astype(np.float32)[None]
    y = fe.resample_sinc_hann(x, sr0, sr1)
    assert y.shape == (1, math.ceil(22000 * 16000 / 16000))
    t1 = np.arange(y.shape[1]) / sr1
    ref = np.sin(2 * np.pi * 1000 * t1)
    np.testing.assert_allclose(y[0, 200:-200], ref[200:-200], atol=2e-3)  # band-limited interpolation of an in-band tone
    assert fe.resample_sinc_hann(x, 16000, 16000) is x


def test_length_rule_matches_survey():
    # T = floor(N_22k / 734) + 1
    for n in (734 * 3, 220000, 12345):
        assert fe.mfcc(np.zeros(n, np.float32) + 1e-3, 22000).shape[1] == n // 734 + 1


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_AUDIO, "style.wav")), reason="reference demo audio not present")
def test_reference_demo_wavs_give_the_documented_frame_counts():
    # style.wav is 22 kHz stereo, exactly 10.0 s -> 300 frames (+1); 1st-page.wav 16 kHz, 12.816 s -> 385
    f = fe.get_mfcc_ta(os.path.join(REF_AUDIO, "style.wav"), sr=22000, fps=30)
    assert f.shape == (301, 64) or f.shape == (300, 64)
    f2 = fe.get_mfcc_ta(os.path.join(REF_AUDIO, "1st-page.wav"), sr=22000, fps=30)
    assert f2.shape[1] == 64 and abs(f2.shape[0] - 385) <= 1
    feat, gap = fe.get_mfcc_sepa(os.path.join(REF_AUDIO, "style.wav"), sr=22000, fps=30)
    assert gap == 44000 // 734 + 1 and feat.shape[1] == 64
    w = fe.get_wav16(os.path.join(REF_AUDIO, "1st-page.wav"))
    assert w.ndim == 2 and w.shape[1] == 1 and np.abs(w).max() <= 1.0
