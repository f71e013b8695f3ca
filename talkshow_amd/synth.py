"""Seeded synthetic checkpoints in the reference's state_dict key scheme.

There is no network and the reference's checkpoints are not shipped
(SURVEY.md §0.6), so every test, `smoke()` and `bench.py` runs on weights made
here.  The generator is numpy `default_rng` only, so the same seed gives the
same bits in this container and on the GPU box; `tests/golden/make_golden.py`
loads these dicts into the reference's own modules with `strict=True`, which
is what pins the key names and shapes below to

  * `nets/spg/vqvae_1d.py:11-34,66-92,116-149,152-208`   (AudioEncoder / VQVAE)
  * `nets/spg/vqvae_modules.py:87-212,252-323`           (ConvNormRelu / stacks / VQ-EMA)
  * `nets/spg/gated_pixelcnn_v2.py:25-128`               (GatedPixelCNN)

Values are chosen so activations stay O(1) through the stacks (healthy logit
margins for the bit-exact greedy comparison, poses of SMPL-X magnitude for the
1e-4 absolute tolerance) and so nothing is degenerate: biases, BatchNorm
running statistics and class embeddings are all non-zero (the reference's own
init zero-fills conv biases, `gated_pixelcnn_v2.py:6-13`).
"""
from collections import OrderedDict

import numpy as np

F32 = np.float32


def _rng(seed, *salt):
    return np.random.default_rng([int(seed)] + [int(s) for s in salt])


class _Builder:
    def __init__(self, seed, salt):
        self.rng = _rng(seed, salt)
        self.sd = OrderedDict()

    def normal(self, key, shape, std):
        self.sd[key] = (self.rng.standard_normal(shape) * std).astype(F32)

    def uniform(self, key, shape, lo, hi):
        self.sd[key] = self.rng.uniform(lo, hi, shape).astype(F32)

    def conv(self, prefix, cout, cin, k, gain=1.0, transposed=False, bias_std=0.05):
        fan_in = cin * k
        shape = (cin, cout, k) if transposed else (cout, cin, k)
        if transposed:
            fan_in = cin * k // 2  # stride-2 transposed conv: 2 of 4 taps hit each output
        self.normal(prefix + ".weight", shape, gain / np.sqrt(fan_in))
        self.normal(prefix + ".bias", (cout,), bias_std)

    def bn(self, prefix, c):
        self.uniform(prefix + ".weight", (c,), 0.8, 1.2)
        self.normal(prefix + ".bias", (c,), 0.05)
        self.normal(prefix + ".running_mean", (c,), 0.05)
        self.uniform(prefix + ".running_var", (c,), 0.6, 1.4)
        self.sd[prefix + ".num_batches_tracked"] = np.asarray(100, dtype=np.int64)

    # vqvae_modules.ConvNormRelu (bn flavour), `vqvae_modules.py:87-172`
    def cnr(self, prefix, cin, cout, sample="none", residual=False, gain=1.0):
        k = 3 if sample == "none" else 4
        tr = sample == "up"
        if residual:
            self.conv(prefix + ".residual_layer", cout, cin, k, gain * 0.7, transposed=tr)
            gain = gain * 0.7
        self.conv(prefix + ".conv", cout, cin, k, gain, transposed=tr)
        self.bn(prefix + ".norm", cout)

    # vqvae_modules.Res_CNR_Stack, `vqvae_modules.py:175-212`
    def stack(self, prefix, c, layers):
        for i in range(layers):
            self.cnr(f"{prefix}._layers.{i}", c, c, gain=1.2)
        self.conv(prefix + ".conv", c, c, 3, 0.6)
        self.bn(prefix + ".norm", c)


def _encoder_like(b, prefix, in_dim, hid, layers, in_gain):
    # shared by vqvae_1d.AudioEncoder (:11-34) and vqvae_1d.Encoder (:66-92)
    b.cnr(prefix + "project", in_dim, hid // 4, gain=in_gain)
    b.stack(prefix + "_enc_1", hid // 4, layers)
    b.cnr(prefix + "_down_1", hid // 4, hid // 2, sample="down", residual=True)
    b.stack(prefix + "_enc_2", hid // 2, layers)
    b.cnr(prefix + "_down_2", hid // 2, hid, sample="down", residual=True)
    b.stack(prefix + "_enc_3", hid, layers)


def audioencoder_state_dict(seed=0, in_dim=64, num_hiddens=256, num_residual_layers=2, in_scale=20.0):
    """`AudioEncoder(in_dim, num_hiddens, num_residual_layers, ·)`; MFCC-scale inputs (std ≈ `in_scale`)."""
    b = _Builder(seed, 101)
    _encoder_like(b, "", in_dim, num_hiddens, num_residual_layers, 1.0 / in_scale)
    return b.sd


def vqvae_state_dict(seed=0, in_dim=39, embedding_dim=64, num_embeddings=2048, num_hiddens=1024,
                     num_residual_layers=2, salt=0, in_scale=0.3, out_scale=0.3, codebook=None):
    """`VQVAE(in_dim, embedding_dim, num_embeddings, num_hiddens, num_residual_layers, ·)` (`vqvae_1d.py:152-208`).

    `codebook=(mu, sigma)` (two `(embedding_dim,)` float32 arrays) re-draws `vq_layer.embeddings` as `mu + sigma * N(0, 1)`
    per channel — elementwise float32 arithmetic only, so every machine gets the same bits — for fixtures that want the
    encoder's outputs to land on many different entries (`tests/golden/make_golden.py::vq_encode_b32`); everything else in
    the dict is unchanged by it."""
    b = _Builder(seed, 202 + salt)
    hid = num_hiddens
    _encoder_like(b, "encoder.", in_dim, hid, num_residual_layers, 1.0 / in_scale)
    b.conv("encoder.pre_vq_conv", embedding_dim, hid, 1, 1.0)
    b.normal("vq_layer.embeddings", (num_embeddings, embedding_dim), 0.7)
    b.normal("vq_layer.ema_dw.hidden", (num_embeddings, embedding_dim), 0.7)
    b.uniform("vq_layer.ema_cluster_size.hidden", (num_embeddings,), 0.5, 2.0)
    b.conv("decoder.aft_vq_conv", hid, embedding_dim, 1, 1.0 / 0.7)
    b.stack("decoder._dec_1", hid, num_residual_layers)
    b.cnr("decoder._up_2", hid, hid // 2, sample="up", residual=True)
    b.stack("decoder._dec_2", hid // 2, num_residual_layers)
    b.cnr("decoder._up_3", hid // 2, hid // 4, sample="up", residual=True)
    b.stack("decoder._dec_3", hid // 4, num_residual_layers)
    b.conv("decoder.project", in_dim, hid // 4, 1, out_scale)
    if codebook is not None:
        mu, sigma = (np.asarray(v, F32).reshape(1, embedding_dim) for v in codebook)
        g = _rng(seed, 909 + salt).standard_normal((num_embeddings, embedding_dim)).astype(F32)
        b.sd["vq_layer.embeddings"] = (mu + sigma * g).astype(F32)
    return b.sd


def ae_state_dict(seed=0, in_dim=129, embedding_dim=64, num_hiddens=1024, num_residual_layers=2, salt=0, in_scale=0.3,
                  out_scale=0.3):
    """`vqvae_1d.AE(in_dim, embedding_dim, 0, num_hiddens, num_residual_layers, ·)` (`vqvae_1d.py:211-235`): the VQ-VAE's
    Encoder / Decoder without the quantiser, plus the `Decoder(ae=True)` extras (`frame_enc`, two GRUs, `:131-134`) that
    sit in its checkpoints but are never used by `forward`."""
    sd = vqvae_state_dict(seed, in_dim, embedding_dim, 8, num_hiddens, num_residual_layers, salt=700 + salt,
                          in_scale=in_scale, out_scale=out_scale)
    for k in [k for k in sd if k.startswith("vq_layer.")]:
        del sd[k]
    b = _Builder(seed, 777 + salt)
    q = num_hiddens // 4
    b.conv("decoder.frame_enc.proj", q, in_dim, 1, 1.0)
    b.stack("decoder.frame_enc.enc", q, 2)
    b.conv("decoder.frame_enc.proj_1", q, 4 * q, 1, 1.0)
    b.conv("decoder.frame_enc.proj_2", 2 * q, 4 * q, 1, 1.0)
    for name, h in (("gru_sl", num_hiddens // 2), ("gru_l", q)):
        for part, shape in (("weight_ih_l0", (3 * h, h)), ("weight_hh_l0", (3 * h, h)), ("bias_ih_l0", (3 * h,)),
                            ("bias_hh_l0", (3 * h,))):
            b.normal(f"decoder.{name}.{part}", shape, 1.0 / np.sqrt(h))
    # reference key order: decoder.* conv stacks, frame_enc, GRUs, then decoder.project last
    proj = {k: sd.pop(k) for k in ("decoder.project.weight", "decoder.project.bias")}
    sd.update(b.sd)
    sd.update(proj)
    return sd


def pixelcnn_state_dict(seed=0, input_dim=2048, dim=256, n_layers=15, n_classes=4, aud_dim=256, audio=True, bh_model=True):
    """`GatedPixelCNN(input_dim, dim, n_layers, n_classes, audio, bh_model)` (`gated_pixelcnn_v2.py:90-128`); the shipped
    configuration is audio=True, bh_model=True.  audio=False drops the three audio tensors; bh_model=False makes the vertical
    kernels one column wide (`:37-38`; the module still holds vert_to_horiz / horiz_stack / fusion_h, which its forward never reads)."""
    b = _Builder(seed, 303)

    def conv2d(prefix, cout, cin, kh, kw, gain, valid=None):
        fan_in = cin * (valid if valid is not None else kh * kw)
        b.normal(prefix + ".weight", (cout, cin, kh, kw), gain / np.sqrt(fan_in))
        b.normal(prefix + ".bias", (cout,), 0.1)

    if audio:
        conv2d("embedding_aud", dim, aud_dim, 1, 1, 1.0)
        conv2d("fusion_v", dim, 2 * dim, 1, 1, 1.0)
        conv2d("fusion_h", dim, 2 * dim, 1, 1, 1.0)
    b.normal("embedding.weight", (input_dim, dim), 1.0)
    kw = 3 if bh_model else 1
    for i in range(n_layers):
        kh = 4 if i == 0 else 2
        p = f"layers.{i}"
        b.normal(p + ".class_cond_embedding.weight", (n_classes, 2 * dim), 0.3)
        conv2d(p + ".vert_stack", 2 * dim, dim, kh, kw, 1.6, valid=(kh - 1 if i == 0 else kh) * (2 if bh_model else 1))
        conv2d(p + ".vert_to_horiz", 2 * dim, 2 * dim, 1, 1, 0.7)
        conv2d(p + ".horiz_stack", 2 * dim, dim, 1, 2, 1.6, valid=1 if i == 0 else 2)
        conv2d(p + ".horiz_resid", dim, dim, 1, 1, 1.5)
    conv2d("output_conv.0", 512, dim, 1, 1, 1.4)
    conv2d("output_conv.2", input_dim, 512, 1, 1, 3.0)
    return b.sd


def face_state_dict(seed=0, n_layers=12, hidden=768, heads=12, ffn=3072, conv_dim=512, num_classes=4,
                    pos_k=128, pos_groups=16, legacy_weight_norm_keys=False, identity=True, jaw_dim=3):
    """`s2g_face.Generator` (`nets/spg/s2g_face.py:142-224`) around the HF wav2vec2-base architecture
    (`nets/spg/wav2vec.py:73-143`; key names of transformers >= 4.3x; `legacy_weight_norm_keys=True` emits the
    4.22-era `weight_g` / `weight_v` names the reference's own checkpoints carry)."""
    b = _Builder(seed, 707)
    p = "audio_encoder."
    b.normal(p + "masked_spec_embed", (hidden,), 0.5)
    kern, = ((10, 3, 3, 3, 3, 2, 2),)
    for i, k in enumerate(kern):
        cin = 1 if i == 0 else conv_dim
        b.normal(f"{p}feature_extractor.conv_layers.{i}.conv.weight", (conv_dim, cin, k), 1.5 / np.sqrt(cin * k))
        if i == 0:
            b.uniform(f"{p}feature_extractor.conv_layers.0.layer_norm.weight", (conv_dim,), 0.8, 1.2)
            b.normal(f"{p}feature_extractor.conv_layers.0.layer_norm.bias", (conv_dim,), 0.1)

    def ln(prefix, c):
        b.uniform(prefix + ".weight", (c,), 0.8, 1.2)
        b.normal(prefix + ".bias", (c,), 0.05)

    def lin(prefix, cout, cin, gain=1.0):
        b.normal(prefix + ".weight", (cout, cin), gain / np.sqrt(cin))
        b.normal(prefix + ".bias", (cout,), 0.05)

    ln(p + "feature_projection.layer_norm", conv_dim)
    lin(p + "feature_projection.projection", hidden, conv_dim)
    b.normal(p + "encoder.pos_conv_embed.conv.bias", (hidden,), 0.05)
    g_key, v_key = (("weight_g", "weight_v") if legacy_weight_norm_keys
                    else ("parametrizations.weight.original0", "parametrizations.weight.original1"))
    b.uniform(p + "encoder.pos_conv_embed.conv." + g_key, (1, 1, pos_k), 1.0, 3.0)
    b.normal(p + "encoder.pos_conv_embed.conv." + v_key, (hidden, hidden // pos_groups, pos_k), 1.0)
    ln(p + "encoder.layer_norm", hidden)
    for l in range(n_layers):
        q = f"{p}encoder.layers.{l}."
        for nm in ("k_proj", "v_proj", "q_proj", "out_proj"):
            lin(q + "attention." + nm, hidden, hidden, 1.3 if nm in ("q_proj", "k_proj") else 1.0)
        ln(q + "layer_norm", hidden)
        lin(q + "feed_forward.intermediate_dense", ffn, hidden, 1.2)
        lin(q + "feed_forward.output_dense", hidden, ffn, 1.2)
        ln(q + "final_layer_norm", hidden)
    lin("audio_feature_map", 256, hidden)
    if identity:
        b.normal("audio_middle.id_mlp.weight", (64, num_classes, 1), 0.7)
        b.normal("audio_middle.id_mlp.bias", (64,), 0.1)
    cin = 320 if identity else 256      # `identity=False` (the convert_to_6d form, `s2g_face.py:107-113`): no id channels

    def conv(prefix, cout, cin, k, gain=1.3):
        b.normal(prefix + ".weight", (cout, cin, k), gain / np.sqrt(cin * k))
        b.normal(prefix + ".bias", (cout,), 0.05)

    fn = "audio_middle.first_net.conv_layers."
    if cin != 256:                      # equal widths: the residual branch is nn.Identity (`layers.py:95-96`), no keys
        conv(fn + "0.residual_layer.0", 256, cin, 3, 0.8)
    conv(fn + "0.conv", 256, cin, 3)
    ln(fn + "0.norm", 256)
    for i in (1, 2):
        conv(fn + f"{i}.conv", 256, 256, 3)
        ln(fn + f"{i}.norm", 256)
    # nn.GRU present in checkpoints, unused by forward (s2g_face.py:121,134)
    b.normal("audio_middle.grus.weight_ih_l0", (768, 256), 0.05)
    b.normal("audio_middle.grus.weight_hh_l0", (768, 256), 0.05)
    b.normal("audio_middle.grus.bias_ih_l0", (768,), 0.05)
    b.normal("audio_middle.grus.bias_hh_l0", (768,), 0.05)
    for d, (c0, c) in enumerate(((256, 64), (256, 256))):
        for i in range(3):
            conv(f"decoder.{d}.{i}.conv", c, c0 if i == 0 else c, 3)
            ln(f"decoder.{d}.{i}.norm", c)
    conv("final_out.0", jaw_dim, 64, 1, 0.3)
    conv("final_out.1", 100, 256, 1, 0.3)
    return b.sd


def wav16(seed, B, N, scale=0.1):
    """(B, N) float32 synthetic 16 kHz waveform."""
    return (_rng(seed, 606).standard_normal((B, N)) * scale).astype(F32)


# --- synthetic inputs -------------------------------------------------------

def mfcc_features(seed, B, T, scale=20.0):
    """(B, T, 64) float32 MFCC-scale features, one different clip per batch row."""
    return (_rng(seed, 404).standard_normal((B, T, 64)) * scale).astype(F32)


def gt_poses(seed, B, T, dim=129, scale=0.3):
    """(B, T, dim) float32 ground-truth poses in `c_index_3d` order for the VQ encode half."""
    r = _rng(seed, 505)
    # temporally smooth-ish so that neighbouring frames are correlated like real motion
    x = r.standard_normal((B, T + 4, dim))
    x = (x[:, :-4] + x[:, 1:-3] + x[:, 2:-2] + x[:, 3:-1] + x[:, 4:]) / np.sqrt(5.0)
    return (x * scale).astype(F32)


def speaker_ids(B):
    return (np.arange(B) % 4).astype(np.int64)


def to_torch(sd):
    import torch
    return OrderedDict((k, torch.from_numpy(np.array(v, copy=True, order='C'))) for k, v in sd.items())
