"""Pin the CPU oracle (oracle/talkshow_oracle.py) to the golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import talkshow_oracle as O
from talkshow_amd import synth

TOL = 2e-5   # oracle (numpy BLAS) vs reference (torch/oneDNN): summation-order noise only


def _vq_sd(cfg):
    in_dim, emb, n_emb, hid, layers, salt, seed = [int(v) for v in cfg]
    return synth.vqvae_state_dict(seed=seed, in_dim=in_dim, embedding_dim=emb, num_embeddings=n_emb,
                                  num_hiddens=hid, num_residual_layers=layers, salt=salt)


@pytest.mark.parametrize("name", ["vq_small", "vq_full_body", "vq_full_hand"])
def test_vqvae_encode_decode(golden, name):
    g = golden(name)
    sd = _vq_sd(g["cfg"])
    z, e, idx = O.vqvae_encode(g["poses"], sd)
    np.testing.assert_allclose(z, g["z"], atol=TOL, rtol=0)
    np.testing.assert_array_equal(idx, g["idx"])
    np.testing.assert_array_equal(e, g["quantized"])
    recon = O.vqvae_decode(idx, sd)
    np.testing.assert_allclose(recon, g["recon"], atol=TOL, rtol=0)
    e2, recon2, _ = O.vqvae_forward(g["poses"], sd)
    np.testing.assert_array_equal(recon2, recon)


def test_audio_encoder(golden):
    g = golden("audioenc_full")
    out = O.audio_encoder(np.ascontiguousarray(g["mfcc"].transpose(0, 2, 1)), synth.audioencoder_state_dict(seed=7))
    np.testing.assert_allclose(out, g["out"], atol=TOL, rtol=0)


@pytest.mark.parametrize("name", ["pix_small", "pix_full"])
def test_pixelcnn_greedy(golden, name):
    g = golden(name)
    input_dim, dim, n_layers, n_cls, seed = [int(v) for v in g["cfg"]]
    sd = synth.pixelcnn_state_dict(seed=seed, input_dim=input_dim, dim=dim, n_layers=n_layers, n_classes=n_cls)
    aud = np.repeat(g["aud"].transpose(0, 2, 1)[:, :, :, None], 2, axis=3)
    H = g["codes"].shape[1]
    # teacher-forced full grid first (no error propagation), then the free-running greedy loop
    full = O.pixelcnn_forward(g["codes"], g["label"], aud, O.causal_weights(sd, n_layers), n_layers)
    np.testing.assert_allclose(full.transpose(0, 2, 3, 1), g["full_logits"], atol=2e-4, rtol=0)
    if name == "pix_small":
        codes, logits = O.pixelcnn_generate(g["label"], aud, sd, n_layers, H, return_logits=True)
        np.testing.assert_allclose(logits, g["step_logits"], atol=2e-4, rtol=0)
    else:
        codes = O.pixelcnn_generate(g["label"], aud, sd, n_layers, H)
    np.testing.assert_array_equal(codes, g["codes"])


@pytest.mark.parametrize("tag,audio,bh", [("noaud_bh", False, True), ("aud_v", True, False), ("noaud_v", False, False)])
def test_pixelcnn_variants(golden, tag, audio, bh):
    """The other constructor variants of GatedPixelCNN (audio=False and / or bh_model=False: `gated_pixelcnn_v2.py:37-42,80-85,
    137-150`) against the reference's own module (tests/golden/make_golden.py --only pix_variants)."""
    g = golden("pix_variants")
    input_dim, dim, n_layers, n_cls, seed = [int(v) for v in g["cfg"]]
    sd = synth.pixelcnn_state_dict(seed=seed, input_dim=input_dim, dim=dim, n_layers=n_layers, n_classes=n_cls, audio=audio, bh_model=bh)
    codes = g[tag + "_codes"]
    H, W = codes.shape[1:]
    aud = np.repeat(g["aud"].transpose(0, 2, 1)[:, :, :, None], W, axis=3) if audio else None
    full = O.pixelcnn_forward(codes, g["label"], aud, O.causal_weights(sd, n_layers), n_layers, audio, bh)
    np.testing.assert_allclose(full.transpose(0, 2, 3, 1), g[tag + "_full_logits"], atol=2e-4, rtol=0)
    got, logits = O.pixelcnn_generate(g["label"], aud, sd, n_layers, H, return_logits=True, audio=audio, bh_model=bh, W=W)
    np.testing.assert_allclose(logits, g[tag + "_step_logits"], atol=2e-4, rtol=0)
    np.testing.assert_array_equal(got, codes)


def test_causality(golden):
    """Receptive-field property: logits at (i, j) do not depend on codes at or after (i, j)."""
    g = golden("pix_small")
    input_dim, dim, n_layers, n_cls, seed = [int(v) for v in g["cfg"]]
    sd = O.causal_weights(synth.pixelcnn_state_dict(seed=seed, input_dim=input_dim, dim=dim, n_layers=n_layers), n_layers)
    aud = np.repeat(g["aud"].transpose(0, 2, 1)[:, :, :, None], 2, axis=3)
    x = g["codes"].copy()
    base = O.pixelcnn_forward(x, g["label"], aud, sd, n_layers)
    i, j = 4, 1
    x2 = x.copy()
    x2[:, i, j:] = (x2[:, i, j:] + 5) % input_dim
    x2[:, i + 1:] = (x2[:, i + 1:] + 9) % input_dim
    pert = O.pixelcnn_forward(x2, g["label"], aud, sd, n_layers)
    np.testing.assert_array_equal(base[:, :, :i], pert[:, :, :i])
    np.testing.assert_array_equal(base[:, :, i, :j + 1], pert[:, :, i, :j + 1])
    assert np.abs(base[:, :, i + 1] - pert[:, :, i + 1]).max() > 1e-3


def test_continuity_prefix(golden):
    """generate(pre_latents, pre_audio) (`gated_pixelcnn_v2.py:158-165`): greedy tail behind the golden head == the golden
    tail (causality makes the prefix form equivalent to the single run)."""
    g = golden("pix_small")
    input_dim, dim, n_layers, n_cls, seed = [int(v) for v in g["cfg"]]
    sd = synth.pixelcnn_state_dict(seed=seed, input_dim=input_dim, dim=dim, n_layers=n_layers, n_classes=n_cls)
    aud = np.repeat(g["aud"].transpose(0, 2, 1)[:, :, :, None], 2, axis=3)
    H0 = 4
    tail = O.pixelcnn_generate(g["label"], aud[:, :, H0:], sd, n_layers, g["codes"].shape[1] - H0,
                               pre_latents=g["codes"][:, :H0], pre_audio=aud[:, :, :H0])
    np.testing.assert_array_equal(tail, g["codes"][:, H0:])


def test_sampler_distribution():
    rng = np.random.default_rng(0)
    logits = rng.standard_normal((1, 16)).astype(np.float32) * 2
    p = O.softmax(logits)[0]
    n = 20000
    draws = O.sample_inverse_cdf(np.repeat(logits, n, 0), rng.random(n))
    freq = np.bincount(draws, minlength=16) / n
    chi2 = (n * (freq - p) ** 2 / p).sum()
    assert chi2 < 45.0     # 15 dof, p ~ 1e-4


@pytest.mark.slow
def test_body_e2e_full(golden):
    g = golden("body_e2e_full")
    codes, poses, feat = O.body_pixel_infer(
        g["mfcc"], g["ids"], synth.audioencoder_state_dict(seed=7), synth.pixelcnn_state_dict(seed=7),
        synth.vqvae_state_dict(seed=7, in_dim=39), synth.vqvae_state_dict(seed=7, in_dim=90, salt=1))
    np.testing.assert_allclose(feat.transpose(0, 2, 1), g["aud_feat"], atol=TOL, rtol=0)
    np.testing.assert_array_equal(codes, g["codes"])
    np.testing.assert_allclose(poses, g["poses"], atol=1e-4, rtol=0)


def test_body_vq_e2e_full(golden):
    g = golden("body_vq_e2e_full")
    out, codes = O.body_vq_infer(g["poses129"], synth.vqvae_state_dict(seed=7, in_dim=39),
                                 synth.vqvae_state_dict(seed=7, in_dim=90, salt=1))
    np.testing.assert_array_equal(codes, g["codes"])
    np.testing.assert_allclose(out, g["out"], atol=1e-4, rtol=0)


def test_ae_feature_extractor(golden):
    """vqvae_1d.AE behind nets.s2g_body_ae.extract (the FGD feature space): oracle vs the reference wrapper's output."""
    g = golden("ae_full")
    z, recon = O.ae_forward(g["poses129"], synth.ae_state_dict(seed=7))
    np.testing.assert_allclose(z, g["z"], atol=TOL, rtol=0)
    np.testing.assert_allclose(z.transpose(0, 2, 1), g["feat"], atol=TOL, rtol=0)
    np.testing.assert_allclose(recon, g["recon"], atol=TOL, rtol=0)


def test_assemble_full(golden):
    """demo.py:207-229 + part2full: jaw | body (aligned to the face length) | expression with the lower-body block inserted."""
    from talkshow_amd.pose_index import lower_pose_block
    g = golden("assemble_full")
    for tag in ("longer_face", "shorter_face"):
        for stand, key in ((False, "full_"), (True, "full_stand_")):
            out = O.assemble_full(g["body"], g["face_" + tag], lower_pose_block(stand))
            assert out.shape == g[key + tag].shape
            assert np.array_equal(out, g[key + tag])


# ---- the torch-CPU port used as bench.py's cpu_baseline (oracle/torch_port.py): pinned to the same goldens ----------
def test_torch_port_vq_and_audio(golden):
    import torch
    from oracle import torch_port as TP
    g = golden("vq_full_body")
    sd = TP._t(_vq_sd(g["cfg"]))
    with torch.no_grad():
        z, e, idx = TP.vqvae_encode(torch.from_numpy(g["poses"]), sd)
        recon = TP.vqvae_decode(idx, sd)
    np.testing.assert_allclose(z.numpy(), g["z"], atol=TOL, rtol=0)
    np.testing.assert_array_equal(idx.numpy(), g["idx"])
    np.testing.assert_allclose(recon.numpy(), g["recon"], atol=TOL, rtol=0)
    ga = golden("audioenc_full")
    with torch.no_grad():
        out = TP.audio_encoder(torch.from_numpy(ga["mfcc"]).transpose(1, 2), TP._t(synth.audioencoder_state_dict(seed=7)))
    np.testing.assert_allclose(out.numpy(), ga["out"], atol=TOL, rtol=0)


@pytest.mark.parametrize("name", ["pix_small", "pix_full"])
def test_torch_port_pixelcnn_greedy(golden, name):
    import torch
    from oracle import torch_port as TP
    g = golden(name)
    input_dim, dim, n_layers, n_cls, seed = [int(v) for v in g["cfg"]]
    sd = TP._t(synth.pixelcnn_state_dict(seed=seed, input_dim=input_dim, dim=dim, n_layers=n_layers, n_classes=n_cls))
    aud = torch.from_numpy(g["aud"]).permute(0, 2, 1).unsqueeze(-1).repeat(1, 1, 1, 2)
    with torch.no_grad():
        codes = TP.pixelcnn_generate(torch.from_numpy(g["label"]), aud, sd, n_layers, g["codes"].shape[1])
    np.testing.assert_array_equal(codes.numpy(), g["codes"])


# ---- real audio: the reference's own demo recordings (tests/golden/real_audio_body.npz) ---------------------
@pytest.mark.parametrize("rec", ["style", "1st_page", "french"])
def test_audio_encoder_on_recordings(golden, rec):
    """MFCC rows of real speech reach 470 .. 660 (the synthetic rows ~80): the audio encoder restatement against the reference's
    AudioEncoder on them."""
    g = golden("real_audio_body")
    rows = g[rec + "_rows"]
    assert np.abs(rows).max() > 400
    out = O.audio_encoder(np.ascontiguousarray(rows.T[None]), synth.audioencoder_state_dict(seed=7))
    np.testing.assert_allclose(out[0].T, g[rec + "_aud_feat"], atol=TOL, rtol=0)


@pytest.mark.slow
def test_body_on_a_recording(golden):
    """french.wav (H = 72) under two of the four speaker ids: the oracle's whole path against the reference's greedy harness and
    decoders on real-speech rows — codes equal, poses within 1e-4."""
    g = golden("real_audio_body")
    rows, spk = g["french_rows"], int(g["french_pose_id"])
    ids = np.asarray([0, spk], np.int64)
    codes, poses, _ = O.body_pixel_infer(np.repeat(rows[None], 2, 0), ids, synth.audioencoder_state_dict(seed=7), synth.pixelcnn_state_dict(seed=7),
                                         synth.vqvae_state_dict(seed=7, in_dim=39), synth.vqvae_state_dict(seed=7, in_dim=90, salt=1))
    np.testing.assert_array_equal(codes, g["french_codes"][ids])
    np.testing.assert_allclose(poses[1], g["french_poses"], atol=1e-4, rtol=0)
