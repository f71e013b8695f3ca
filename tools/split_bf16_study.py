#!/usr/bin/env python3
"""CPU study for an opt-in split-bf16 arithmetic plan of the face generator's GEMMs.

x = hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): products of bf16 values are exact in fp32
(what v_mfma_f32_32x32x16_bf16 computes, fp32 accumulate).  `terms` = which cross products are kept:
    3: hh + hm + mh                (2 splits,  3/16 of the fp32-MFMA cost)
    6: hh + hm + mh + hl + mm + lh (3 splits,  6/16 of the fp32-MFMA cost)
Emulated here with torch CPU fp32 convolutions / matmuls of bf16-valued fp32 tensors, on the 10 s face golden clip; printed:
max |hidden - fp32|, max |out - fp32|, and both against the REFERENCE golden (tests/golden/face_10s.npz, tolerance 1e-4)."""
import os, sys, time
import numpy as np, torch
import torch.nn.functional as Fn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from talkshow_amd import synth
from oracle import face_oracle as FO

torch.set_num_threads(8)


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def split(x, n):
    parts, r = [], x
    for _ in range(n):
        p = bf(r)
        parts.append(p)
        r = r - p
    return parts


PAIRS = {0: None, 3: [(0, 0), (0, 1), (1, 0)], 6: [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)]}


def bilinear(op, a, b, terms):
    if not terms:
        return op(a, b)
    n = 2 if terms == 3 else 3
    A, B = split(a, n), split(b, n)
    out = None
    for i, j in reversed(PAIRS[terms]):          # small terms first
        t = op(A[i], B[j])
        out = t if out is None else out + t
    return out


def forward(wav, sd, frames, terms, attn_terms=0):
    T = lambda k: torch.from_numpy(np.ascontiguousarray(sd[k]))
    p = "audio_encoder."
    lin = lambda x, w, b=None: bilinear(lambda a, bb: a @ bb.t(), x, w, terms) + (b if b is not None else 0)
    h = torch.from_numpy(wav)[:, None, :]
    kern, stride = (10, 3, 3, 3, 3, 2, 2), (5, 2, 2, 2, 2, 2, 2)
    for i, (k, s) in enumerate(zip(kern, stride)):
        w = T(f"{p}feature_extractor.conv_layers.{i}.conv.weight")
        if i == 0:
            h = Fn.conv1d(h, w, None, s)                                            # layer 0 stays fp32 (VALU kernel, K = 10)
            h = Fn.group_norm(h, h.shape[1], T(f"{p}feature_extractor.conv_layers.0.layer_norm.weight"),
                              T(f"{p}feature_extractor.conv_layers.0.layer_norm.bias"), 1e-5)
        else:
            h = bilinear(lambda a, bb: Fn.conv1d(a, bb, None, s), h, w, terms)
        h = Fn.gelu(h)
    h = h.transpose(1, 2)
    h = torch.from_numpy(FO.linear_interpolation(h.numpy(), frames))
    h = Fn.layer_norm(h, (512,), T(p + "feature_projection.layer_norm.weight"), T(p + "feature_projection.layer_norm.bias"))
    h = lin(h, T(p + "feature_projection.projection.weight"), T(p + "feature_projection.projection.bias"))
    w = torch.from_numpy(FO.pos_conv_weight(sd, p + "encoder.pos_conv_embed.conv"))
    pos = bilinear(lambda a, bb: Fn.conv1d(a, bb, None, 1, 64, 1, 16), h.transpose(1, 2).contiguous(), w, terms)
    pos = pos + T(p + "encoder.pos_conv_embed.conv.bias")[None, :, None]
    pos = Fn.gelu(pos[:, :, :-1]).transpose(1, 2)
    h = Fn.layer_norm(h + pos, (768,), T(p + "encoder.layer_norm.weight"), T(p + "encoder.layer_norm.bias"))
    B, TT, C = h.shape
    for l in range(12):
        q = f"{p}encoder.layers.{l}."
        L = lambda name, x: lin(x, T(q + name + ".weight"), T(q + name + ".bias"))
        qh = L("attention.q_proj", h).reshape(B, TT, 12, 64).transpose(1, 2)
        kh = L("attention.k_proj", h).reshape(B, TT, 12, 64).transpose(1, 2)
        vh = L("attention.v_proj", h).reshape(B, TT, 12, 64).transpose(1, 2)
        att = bilinear(lambda a, bb: a @ bb.transpose(-1, -2), qh, kh, attn_terms) * 0.125
        att = torch.softmax(att, -1)
        o = bilinear(lambda a, bb: a @ bb, att, vh, attn_terms).transpose(1, 2).reshape(B, TT, C)
        h = Fn.layer_norm(h + L("attention.out_proj", o), (768,), T(q + "layer_norm.weight"), T(q + "layer_norm.bias"))
        ff = L("feed_forward.output_dense", Fn.gelu(L("feed_forward.intermediate_dense", h)))
        h = Fn.layer_norm(h + ff, (768,), T(q + "final_layer_norm.weight"), T(q + "final_layer_norm.bias"))
    hidden = h
    # heads (s2g_face.Generator after the encoder): audio_feature_map + id channels (zero id here) + LN-conv stacks
    feat = lin(h, T("audio_feature_map.weight"), T("audio_feature_map.bias")).transpose(1, 2)
    idc = torch.zeros(B, 4, TT)
    idc = Fn.conv1d(idc, T("audio_middle.id_mlp.weight"), T("audio_middle.id_mlp.bias"))
    x = torch.cat([feat, idc], 1)

    def cnr(x, pfx, residual):
        out = bilinear(lambda a, bb: Fn.conv1d(a, bb, None, 1, 1), x, T(pfx + ".conv.weight"), terms) + T(pfx + ".conv.bias")[None, :, None]
        out = Fn.layer_norm(out.transpose(1, 2), (out.shape[1],), T(pfx + ".norm.weight"), T(pfx + ".norm.bias")).transpose(1, 2)
        if residual:
            if pfx + ".residual_layer.0.weight" in sd:
                out = out + bilinear(lambda a, bb: Fn.conv1d(a, bb, None, 1, 1), x, T(pfx + ".residual_layer.0.weight"), terms) \
                    + T(pfx + ".residual_layer.0.bias")[None, :, None]
            else:
                out = out + x
        return torch.relu(out)
    for i in range(3):
        x = cnr(x, f"audio_middle.first_net.conv_layers.{i}", True)
    outs = []
    for d in range(2):
        m = x
        for i in range(3):
            m = cnr(m, f"decoder.{d}.{i}", False)
        outs.append(Fn.conv1d(m, T(f"final_out.{d}.weight"), T(f"final_out.{d}.bias")))
    return hidden.numpy(), torch.cat(outs, 1).transpose(1, 2).numpy()


if __name__ == "__main__":
    g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "face_10s.npz"))
    seed, B, N = (int(v) for v in g["wav_seed"])
    wav = synth.wav16(seed, B, N)[1:2]                    # golden clip 1: zero identity vector
    sd = synth.face_state_dict(seed=7)
    with torch.no_grad():
        t0 = time.time(); h0, o0 = forward(wav, sd, 300, 0); print(f"fp32: {time.time() - t0:.1f} s; out vs reference golden {np.abs(o0[0] - g['out'][1]).max():.2e}")
        for terms, at in ((3, 0), (6, 0), (6, 6)):
            t0 = time.time(); h, o = forward(wav, sd, 300, terms, at)
            print(f"terms {terms} (attention {at or 'fp32'}): hidden vs fp32 {np.abs(h - h0).max():.2e} (|hidden| max {np.abs(h0).max():.1f}), out vs fp32 {np.abs(o - o0).max():.2e}, "
                  f"out vs reference golden {np.abs(o[0] - g['out'][1]).max():.2e}   [{time.time() - t0:.0f} s]")
