"""Pin the face oracle (oracle/face_oracle.py) to goldens produced by the reference `s2g_face` wrapper running over the
installed transformers wav2vec2 module (tests/golden/make_golden.py, case `face_full`).  CPU only."""
import numpy as np

from oracle import face_oracle as FO
from talkshow_amd import synth


def test_face_generator_matches_reference(golden):
    g = golden("face_full")
    sd = synth.face_state_dict(seed=7)
    frame = g["out"].shape[1]
    hs, inter = FO.wav2vec2_forward(g["wav"], sd, frame, return_intermediates=True)
    np.testing.assert_allclose(hs, g["hidden"], atol=5e-5, rtol=0)
    out = FO.face_generator(g["wav"], g["ids"], sd, frame)
    assert out.shape == g["out"].shape == (2, 60, 103)
    np.testing.assert_allclose(out, g["out"], atol=1e-4, rtol=0)
    # smplx_face.TrainWrapper.generate uses the all-zero id vector (smplx_face.py:232-233)
    out0 = FO.face_generator(g["wav"], np.zeros_like(g["ids"]), sd, frame)
    np.testing.assert_allclose(out0, g["generate_zero_id"], atol=1e-4, rtol=0)
    assert np.abs(out0[1] - out[1]).max() > 1e-3          # the one-hot id of clip 1 matters


def test_legacy_weight_norm_keys_are_equivalent():
    a = synth.face_state_dict(seed=3, n_layers=1)
    b = synth.face_state_dict(seed=3, n_layers=1, legacy_weight_norm_keys=True)
    p = "audio_encoder.encoder.pos_conv_embed.conv"
    np.testing.assert_array_equal(FO.pos_conv_weight(a, p), FO.pos_conv_weight(b, p))


def test_linear_interpolation_endpoints():
    x = np.arange(499, dtype=np.float32)[None, :, None]
    y = FO.linear_interpolation(x, 300)[0, :, 0]
    assert y.shape == (300,) and abs(y[0] - 0.3316667) < 1e-4 and y[-1] <= 498 and np.all(np.diff(y) > 0)


def test_face_generator_on_a_recording(golden):
    """Real speech (tests/golden/real_audio_face.npz: the reference wrapper on its own demo recordings): french.wav, 9.6 s at
    16 kHz as the host kaiser_best twin produces it (tests/golden/audio/french.wav.wav16.npy, a build-time file; skipped without it)."""
    import hashlib
    import os
    import pytest
    g = golden("real_audio_face")
    side = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "audio", "french.wav.wav16.npy")
    if not os.path.exists(side):
        pytest.skip("tests/golden/audio/french.wav.wav16.npy is absent (__graft_entry__.build() writes it where /root/reference exists)")
    wav = np.load(side)
    if hashlib.sha256(wav.tobytes()).hexdigest() != str(g["french_wav16_sha256"]):
        pytest.skip("the side file is not the golden's input")
    N, frame, spk = (int(v) for v in g["french_n"])
    sd = synth.face_state_dict(seed=7)
    hs = FO.wav2vec2_forward(wav[None], sd, frame)
    np.testing.assert_allclose(hs[0, ::6], g["french_hidden_6"], atol=5e-5, rtol=0)
    ids = np.zeros((1, 4), np.float32)
    np.testing.assert_allclose(FO.face_generator(wav[None], ids, sd, frame)[0], g["french_out_zero_id"], atol=1e-4, rtol=0)
    ids[0, spk] = 1.0
    np.testing.assert_allclose(FO.face_generator(wav[None], ids, sd, frame)[0], g["french_out_one_hot"], atol=1e-4, rtol=0)
