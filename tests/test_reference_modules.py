"""The reference's OWN nn.Modules of the body path, as lifted into oracle/_ref by oracle/build_ref_callers.py (whole files
`nets/spg/{gated_pixelcnn_v2, vqvae_modules, wav2vec, vqvae_1d}.py` compiled to code objects where /root/reference exists; they
travel to the GPU box with the snapshot), checked against the committed goldens — which the same modules produced when they were
imported from the reference tree by tests/golden/make_golden.py.  These are what bench.py's `cpu_baseline` (kind "reference") times
on the GPU box's host cores; CPU only, skipped where oracle/_ref was not built.
"""
import contextlib
import io
import os
import sys

import numpy as np
import pytest
import torch

from talkshow_amd import synth

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import build_ref_callers as BRC  # noqa: E402


@pytest.fixture(scope="module")
def ref():
    got = BRC.load_reference_modules()
    if got is None:
        pytest.skip("oracle/_ref holds no reference modules (python oracle/build_ref_callers.py where /root/reference exists)")
    return got


def test_lifted_vqvae_reproduces_the_goldens(ref, golden):
    g = golden("vq_small")
    in_dim, emb, n_emb, hid, layers, salt, seed = [int(v) for v in g["cfg"]]
    net = ref.VQVAE(in_dim, emb, n_emb, hid, layers, 512)
    net.load_state_dict(synth.to_torch(synth.vqvae_state_dict(seed=seed, in_dim=in_dim, embedding_dim=emb, num_embeddings=n_emb, num_hiddens=hid,
                                                              num_residual_layers=layers, salt=salt)), strict=True)
    net.eval()
    with torch.no_grad():
        e, lat = net.encode(gt_poses=torch.from_numpy(g["poses"]))
        rec, _ = net.decode(b=lat.shape[0], w=lat.shape[1], latents=lat)
    np.testing.assert_array_equal(lat.numpy(), g["idx"])
    np.testing.assert_allclose(rec.numpy(), g["recon"], atol=2e-5, rtol=0)


def test_lifted_pixelcnn_and_audio_encoder_reproduce_the_goldens(ref, golden):
    import bench
    g = golden("pix_small")
    input_dim, dim, n_layers, n_cls, seed = [int(v) for v in g["cfg"]]
    with contextlib.redirect_stdout(io.StringIO()):
        pix = ref.GatedPixelCNN(input_dim, dim, n_layers, 4, True, True)
    pix.load_state_dict(synth.to_torch(synth.pixelcnn_state_dict(seed=seed, input_dim=input_dim, dim=dim, n_layers=n_layers, n_classes=n_cls)), strict=True)
    pix.eval()
    aud = torch.from_numpy(g["aud"]).permute(0, 2, 1).unsqueeze(-1).repeat(1, 1, 1, 2)
    codes = bench.reference_greedy(pix, torch.from_numpy(g["label"]), aud)       # the harness the cpu_baseline leg times
    np.testing.assert_array_equal(codes.numpy(), g["codes"])
    ga = golden("audioenc_full")
    ae = ref.AudioEncoder(64, 256, 2, 256)
    ae.load_state_dict(synth.to_torch(synth.audioencoder_state_dict(seed=7)), strict=True)
    ae.eval()
    with torch.no_grad():
        out = ae(torch.from_numpy(ga["mfcc"]).transpose(1, 2))
    np.testing.assert_allclose(out.numpy(), ga["out"], atol=2e-5, rtol=0)


def test_cpu_baseline_reference_leg_runs_small(ref):
    """bench.cpu_baseline's `reference` leg end to end on a tiny budget: kind, cores, sample text, a positive rate."""
    import bench
    sds = dict(audio=synth.audioencoder_state_dict(seed=0), pix=synth.pixelcnn_state_dict(seed=0, input_dim=64, dim=32, n_layers=2),
               body=synth.vqvae_state_dict(seed=0, in_dim=39, num_embeddings=64, num_hiddens=64),
               hand=synth.vqvae_state_dict(seed=0, in_dim=90, num_embeddings=64, num_hiddens=64, salt=1))
    out = bench.cpu_baseline_reference(sds, 5, budget_s=1.0, dims=dict(input_dim=64, dim=32, n_layers=2, num_embeddings=64, num_hiddens=64), frames=40)
    assert out["kind"] == "reference" and out["value"] > 0 and out["cores"] >= 1 and "GatedPixelCNN.forward" in out["sample"]
