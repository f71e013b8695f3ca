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
