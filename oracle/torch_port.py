"""CPU baseline port: the reference's body hot path restated on torch CPU ops (ATen / oneDNN).

TEST / MEASUREMENT INFRASTRUCTURE ONLY — same rule as `talkshow_oracle.py`: nothing under `talkshow_amd/` or `nets/`
may import this file; `tests/` (pin against the goldens) and `bench.py`'s `cpu_baseline` leg do.

Why a second restatement: the numpy oracle next door is written for readability and bit-level checking; its per-tap
`np.matmul` convolutions run ~20x slower than the reference's own `nn.Conv*` modules on the same cores, which undersold
the reference as a CPU baseline.  The reference is plain PyTorch, and /root/reference does not exist on
the GPU box, so the baseline that can travel is this file: the same functions in the same order as the reference's
`forward`s, each calling the `torch.nn.functional` op the reference's `nn.Module` dispatches to.  It therefore runs at
the reference's speed (tools/time_reference_cpu.py times the real modules in the build container for comparison;
profiles/r02_reference_cpu_buildbox.json).

Pinned like the numpy oracle: tests/test_oracle_golden.py checks it against the golden vectors produced by the
reference's own modules (codes bit-exact, floats <= 1e-5).
"""
import numpy as np
import torch
import torch.nn.functional as F


def _t(sd):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items() if isinstance(v, np.ndarray)}


def _bn(x, sd, p):
    # nn.BatchNorm1d, eval mode (vqvae_modules.py:160,202)
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)


def conv_norm_relu(x, sd, p, sample="none", residual=False):
    """vqvae_modules.ConvNormRelu.forward (`vqvae_modules.py:167-172`), leaky=True, norm='bn'."""
    if sample == "up":
        out = F.conv_transpose1d(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"], 2, 1)
    elif sample == "down":
        out = F.conv1d(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"], 2, 1)
    else:
        out = F.conv1d(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"], 1, 1)
    out = _bn(out, sd, p + ".norm")
    if residual:
        if sample == "up":
            out = out + F.conv_transpose1d(x, sd[p + ".residual_layer.weight"], sd[p + ".residual_layer.bias"], 2, 1)
        else:
            out = out + F.conv1d(x, sd[p + ".residual_layer.weight"], sd[p + ".residual_layer.bias"], 2, 1)
    return F.leaky_relu(out, 0.2)


def res_cnr_stack(x, sd, p, layers=2):
    """vqvae_modules.Res_CNR_Stack.forward (`vqvae_modules.py:205-212`)."""
    h = x
    for i in range(layers):
        h = conv_norm_relu(h, sd, f"{p}._layers.{i}")
    h = _bn(F.conv1d(h, sd[p + ".conv.weight"], sd[p + ".conv.bias"], 1, 1), sd, p + ".norm")
    return F.relu(h + x)


def _encoder_trunk(x, sd, p, layers=2):
    h = conv_norm_relu(x, sd, p + "project")
    h = res_cnr_stack(h, sd, p + "_enc_1", layers)
    h = conv_norm_relu(h, sd, p + "_down_1", "down", True)
    h = res_cnr_stack(h, sd, p + "_enc_2", layers)
    h = conv_norm_relu(h, sd, p + "_down_2", "down", True)
    return res_cnr_stack(h, sd, p + "_enc_3", layers)


def audio_encoder(x, sd):
    """vqvae_1d.AudioEncoder.forward (`vqvae_1d.py:27-34`)."""
    return _encoder_trunk(x, sd, "")


def vqvae_encode(gt_poses, sd):
    """VQVAE.encode (`vqvae_1d.py:196-199`) + VectorQuantizerEMA eval (`vqvae_modules.py:274-286,311-323`)."""
    z = _encoder_trunk(gt_poses.transpose(1, 2), sd, "encoder.")
    z = F.conv1d(z, sd["encoder.pre_vq_conv.weight"], sd["encoder.pre_vq_conv.bias"])
    B, C, H = z.shape
    flat = z.permute(0, 2, 1).reshape(-1, C)
    emb = sd["vq_layer.embeddings"]
    d = torch.sum(flat ** 2, dim=1, keepdim=True) + torch.sum(emb ** 2, dim=1) - 2.0 * torch.matmul(flat, emb.t())
    idx = torch.argmin(d.float(), dim=1)
    e = F.embedding(idx, emb).reshape(B, H, C).permute(0, 2, 1)
    return z, e, idx.reshape(B, H)


def vqvae_decode(latents, sd):
    """VQVAE.decode(latents=...) + Decoder.forward (`vqvae_1d.py:139-149,201-208`)."""
    B, H = latents.shape
    e = F.embedding(latents.reshape(-1), sd["vq_layer.embeddings"]).reshape(B, H, -1).permute(0, 2, 1)
    h = F.conv1d(e, sd["decoder.aft_vq_conv.weight"], sd["decoder.aft_vq_conv.bias"])
    h = res_cnr_stack(h, sd, "decoder._dec_1")
    h = conv_norm_relu(h, sd, "decoder._up_2", "up", True)
    h = res_cnr_stack(h, sd, "decoder._dec_2")
    h = conv_norm_relu(h, sd, "decoder._up_3", "up", True)
    h = res_cnr_stack(h, sd, "decoder._dec_3")
    return F.conv1d(h, sd["decoder.project.weight"], sd["decoder.project.bias"])


def _gate(x):
    a, g = x.chunk(2, dim=1)                                    # GatedActivation (`gated_pixelcnn_v2.py:20-22`)
    return torch.tanh(a) * torch.sigmoid(g)


def _gated_layer(x_v, x_h, label, sd, p, kernel, residual):
    """GatedMaskedConv2d.forward, bh_model=True (`gated_pixelcnn_v2.py:61-87`)."""
    h = F.embedding(label, sd[p + ".class_cond_embedding.weight"])
    h_vert = F.conv2d(x_v, sd[p + ".vert_stack.weight"], sd[p + ".vert_stack.bias"], 1, (kernel // 2, 1))
    h_vert = h_vert[:, :, :x_v.size(-2), :]
    out_v = _gate(h_vert + h[:, :, None, None])
    h_horiz = F.conv2d(x_h, sd[p + ".horiz_stack.weight"], sd[p + ".horiz_stack.bias"], 1, (0, 1))
    h_horiz = h_horiz[:, :, :, :x_h.size(-1)]
    v2h = F.conv2d(h_vert, sd[p + ".vert_to_horiz.weight"], sd[p + ".vert_to_horiz.bias"])
    out = _gate(v2h + h_horiz + h[:, :, None, None])
    out_h = F.conv2d(out, sd[p + ".horiz_resid.weight"], sd[p + ".horiz_resid.bias"])
    if residual:
        out_h = out_h + x_h
    return out_v, out_h


def pixelcnn_forward(x, label, aud, sd, n_layers):
    """GatedPixelCNN.forward (`gated_pixelcnn_v2.py:130-150`); `sd` already mask-A zeroed."""
    e = F.embedding(x, sd["embedding.weight"]).permute(0, 3, 1, 2)
    xv = xh = e
    for i in range(n_layers):
        if i == 1:
            a = F.conv2d(aud, sd["embedding_aud.weight"], sd["embedding_aud.bias"])
            xv = F.conv2d(torch.cat([xv, a], 1), sd["fusion_v.weight"], sd["fusion_v.bias"])
            xh = F.conv2d(torch.cat([xh, a], 1), sd["fusion_h.weight"], sd["fusion_h.bias"])
        xv, xh = _gated_layer(xv, xh, label, sd, f"layers.{i}", 7 if i == 0 else 3, i != 0)
    y = F.relu(F.conv2d(xh, sd["output_conv.0.weight"], sd["output_conv.0.bias"]))
    return F.conv2d(y, sd["output_conv.2.weight"], sd["output_conv.2.bias"])


def pixelcnn_generate(label, aud, sd, n_layers, H, multinomial=False):
    """GatedPixelCNN.generate (`gated_pixelcnn_v2.py:152-177`): full-grid recompute for each of the 2H positions, as the
    reference does it.  Greedy harness unless `multinomial` (the reference's own draw, :173-176)."""
    sd = dict(sd)
    v = sd["layers.0.vert_stack.weight"].clone(); v[:, :, -1] = 0               # make_causal (:57-59)
    hz = sd["layers.0.horiz_stack.weight"].clone(); hz[:, :, :, -1] = 0
    sd["layers.0.vert_stack.weight"], sd["layers.0.horiz_stack.weight"] = v, hz
    x = torch.zeros((aud.shape[0], H, 2), dtype=torch.int64)
    for i in range(H):
        for j in range(2):
            lg = pixelcnn_forward(x, label, aud, sd, n_layers)[:, :, i, j]
            x[:, i, j] = F.softmax(lg, -1).multinomial(1).squeeze(-1) if multinomial else torch.argmax(lg, -1)
    return x


def body_pixel_infer(mfcc, ids, sd_audio, sd_pix, sd_body, sd_hand, n_layers=15, multinomial=False):
    """`s2g_body_pixel.TrainWrapper.infer_on_audio` after the front-end (`smplx_body_pixel.py:272-285`); numpy in/out."""
    with torch.no_grad():
        sa, sp, sb, sh = _t(sd_audio), _t(sd_pix), _t(sd_body), _t(sd_hand)
        feat = audio_encoder(torch.from_numpy(np.ascontiguousarray(mfcc)).transpose(1, 2), sa)
        aud = feat.unsqueeze(-1).repeat(1, 1, 1, 2)
        codes = pixelcnn_generate(torch.from_numpy(np.asarray(ids, np.int64)), aud, sp, n_layers, aud.shape[2], multinomial)
        body = vqvae_decode(codes[..., 0], sb)
        hand = vqvae_decode(codes[..., 1], sh)
        poses = torch.cat([body, hand], 1).transpose(1, 2)
    return codes.numpy(), poses.contiguous().numpy(), feat.numpy()


def vq_encode_pair(poses129, sd_body, sd_hand):
    """The encode half of BASELINE configs[1]: VQVAE.encode of the body (39) and hand (90) dims; numpy in/out."""
    with torch.no_grad():
        p = torch.from_numpy(np.ascontiguousarray(poses129))
        _, _, ib = vqvae_encode(p[..., :39], _t(sd_body))
        _, _, ih = vqvae_encode(p[..., 39:], _t(sd_hand))
    return torch.stack([ib, ih], -1).numpy()
