"""CPU oracle: numpy restatement of the reference's speech -> SMPL-X body hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under `talkshow_amd/` or `nets/` may import this file;
only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg do, and there
only as the checker / the timed CPU baseline.  The product path is the HIP library.

Parity pinning: every function here is checked in `tests/test_oracle_golden.py` against
golden vectors that `tests/golden/make_golden.py` produced by running the reference's own
PyTorch modules (imported from /root/reference in the build container) on the same seeded
synthetic checkpoints.  The reference ships no tests / known-answer vectors of its own
(SURVEY.md §4), so those generated fixtures are the pin.

All tensors follow the reference's layouts (channels-first (B, C, L) / (B, C, H, W)) so each
function can be read next to the file:line it restates.  fp32 throughout, int64 codes.
"""
import numpy as np

F32 = np.float32


# ----------------------------------------------------------------------------------------------
# primitives (torch.nn.functional semantics)
# ----------------------------------------------------------------------------------------------

def conv1d(x, w, b=None, stride=1, padding=0):
    """nn.Conv1d: x (B,Cin,L), w (Cout,Cin,K) -> (B,Cout,Lout)."""
    B, C, L = x.shape
    O, C2, K = w.shape
    assert C == C2
    if padding:
        x = np.pad(x, ((0, 0), (0, 0), (padding, padding)))
    Lout = (x.shape[2] - K) // stride + 1
    out = np.zeros((B, O, Lout), F32)
    for k in range(K):
        xs = x[:, :, k:k + stride * (Lout - 1) + 1:stride]          # (B,C,Lout)
        out += np.matmul(np.ascontiguousarray(w[:, :, k])[None], np.ascontiguousarray(xs))
    if b is not None:
        out += b[None, :, None]
    return out


def conv_transpose1d(x, w, b=None, stride=2, padding=1):
    """nn.ConvTranspose1d: x (B,Cin,L), w (Cin,Cout,K) -> (B,Cout,(L-1)*stride-2*padding+K)."""
    B, C, L = x.shape
    C2, O, K = w.shape
    assert C == C2
    full = np.zeros((B, O, (L - 1) * stride + K), F32)
    for k in range(K):
        full[:, :, k:k + stride * (L - 1) + 1:stride] += np.matmul(np.ascontiguousarray(w[:, :, k].T)[None], x)
    out = full[:, :, padding:full.shape[2] - padding]
    if b is not None:
        out = out + b[None, :, None]
    return out.astype(F32)


def batchnorm_eval(x, sd, prefix, eps=1e-5):
    """nn.BatchNorm1d in eval mode."""
    g, be = sd[prefix + ".weight"], sd[prefix + ".bias"]
    m, v = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    inv = (g / np.sqrt(v + F32(eps))).astype(F32)
    return (x - m[None, :, None]) * inv[None, :, None] + be[None, :, None]


def leaky_relu(x, slope=0.2):
    return np.where(x >= 0, x, x * F32(slope)).astype(F32)


def relu(x):
    return np.maximum(x, 0).astype(F32)


def sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x))).astype(F32)


# ----------------------------------------------------------------------------------------------
# nets/spg/vqvae_modules.py
# ----------------------------------------------------------------------------------------------

def conv_norm_relu(x, sd, p, sample="none", residual=False):
    """vqvae_modules.ConvNormRelu.forward (`vqvae_modules.py:167-172`), leaky=True, norm='bn'."""
    if sample == "up":
        out = conv_transpose1d(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"], 2, 1)
    elif sample == "down":
        out = conv1d(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"], 2, 1)
    else:
        out = conv1d(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"], 1, 1)
    out = batchnorm_eval(out, sd, p + ".norm")
    if residual:
        if sample == "up":
            out = out + conv_transpose1d(x, sd[p + ".residual_layer.weight"], sd[p + ".residual_layer.bias"], 2, 1)
        else:
            out = out + conv1d(x, sd[p + ".residual_layer.weight"], sd[p + ".residual_layer.bias"], 2, 1)
    return leaky_relu(out)


def res_cnr_stack(x, sd, p, layers=2):
    """vqvae_modules.Res_CNR_Stack.forward (`vqvae_modules.py:205-212`)."""
    h = x
    for i in range(layers):
        h = conv_norm_relu(h, sd, f"{p}._layers.{i}")
    h = batchnorm_eval(conv1d(h, sd[p + ".conv.weight"], sd[p + ".conv.bias"], 1, 1), sd, p + ".norm")
    return relu(h + x)


def vq_get_code_indices(flat_x, emb):
    """VectorQuantizerEMA.get_code_indices (`vqvae_modules.py:311-319`): same three-term formula."""
    d = (np.sum(flat_x ** 2, axis=1, keepdims=True, dtype=F32)
         + np.sum(emb ** 2, axis=1, dtype=F32)[None, :]
         - F32(2.0) * np.matmul(flat_x, emb.T))
    return np.argmin(d, axis=1).astype(np.int64)          # ties -> lowest index, as torch.argmin


def vq_quantize(idx, emb):
    """VectorQuantizerEMA.quantize (`vqvae_modules.py:321-323`)."""
    return emb[idx]


# ----------------------------------------------------------------------------------------------
# nets/spg/vqvae_1d.py
# ----------------------------------------------------------------------------------------------

def _encoder_trunk(x, sd, p, layers=2):
    h = conv_norm_relu(x, sd, p + "project")
    h = res_cnr_stack(h, sd, p + "_enc_1", layers)
    h = conv_norm_relu(h, sd, p + "_down_1", "down", True)
    h = res_cnr_stack(h, sd, p + "_enc_2", layers)
    h = conv_norm_relu(h, sd, p + "_down_2", "down", True)
    h = res_cnr_stack(h, sd, p + "_enc_3", layers)
    return h


def audio_encoder(x, sd):
    """vqvae_1d.AudioEncoder.forward (`vqvae_1d.py:27-34`): x (B,64,T) -> (B,256,T//4)."""
    return _encoder_trunk(x, sd, "")


def vq_encoder(x, sd):
    """vqvae_1d.Encoder.forward (`vqvae_1d.py:84-92`): x (B,in_dim,T) -> z (B,64,T//4)."""
    h = _encoder_trunk(x, sd, "encoder.")
    return conv1d(h, sd["encoder.pre_vq_conv.weight"], sd["encoder.pre_vq_conv.bias"])


def vq_decoder(e, sd):
    """vqvae_1d.Decoder.forward (`vqvae_1d.py:139-149`): e (B,64,H) -> (B,out_dim,4H)."""
    h = conv1d(e, sd["decoder.aft_vq_conv.weight"], sd["decoder.aft_vq_conv.bias"])
    h = res_cnr_stack(h, sd, "decoder._dec_1")
    h = conv_norm_relu(h, sd, "decoder._up_2", "up", True)
    h = res_cnr_stack(h, sd, "decoder._dec_2")
    h = conv_norm_relu(h, sd, "decoder._up_3", "up", True)
    h = res_cnr_stack(h, sd, "decoder._dec_3")
    return conv1d(h, sd["decoder.project.weight"], sd["decoder.project.bias"])


def vqvae_encode(gt_poses, sd):
    """VQVAE.encode (`vqvae_1d.py:196-199`) + VectorQuantizerEMA.forward eval branch (`vqvae_modules.py:274-286`).

    gt_poses (B,T,in_dim) -> z (B,64,H), quantized e (B,64,H), latents (B,H) int64.
    """
    z = vq_encoder(np.ascontiguousarray(gt_poses.transpose(0, 2, 1)), sd)
    B, C, H = z.shape
    flat = np.ascontiguousarray(z.transpose(0, 2, 1)).reshape(-1, C)
    idx = vq_get_code_indices(flat, sd["vq_layer.embeddings"])
    e = vq_quantize(idx, sd["vq_layer.embeddings"]).reshape(B, H, C).transpose(0, 2, 1)
    return z, np.ascontiguousarray(e), idx.reshape(B, H)


def vqvae_decode(latents, sd):
    """VQVAE.decode(latents=...) (`vqvae_1d.py:201-208`): latents (B,H) -> recon (B,out_dim,4H)."""
    B, H = latents.shape
    e = vq_quantize(latents.reshape(-1), sd["vq_layer.embeddings"]).reshape(B, H, -1).transpose(0, 2, 1)
    return vq_decoder(np.ascontiguousarray(e), sd)


def vqvae_forward(gt_poses, sd):
    """VQVAE.forward eval branch (`vqvae_1d.py:184-189`): returns (e, x_recon (B,out_dim,T))."""
    _, e, idx = vqvae_encode(gt_poses, sd)
    return e, vq_decoder(e, sd), idx


def ae_forward(gt_poses, sd):
    """vqvae_1d.AE.forward eval branch / AE.encode (`vqvae_1d.py:225-235`): gt_poses (B,T,in_dim) ->
    (z (B,64,T//4), x_recon (B,in_dim,4*(T//4))).  The FGD feature of `body_ae.extract` is z transposed (`body_ae.py:151-152`)."""
    z = vq_encoder(np.ascontiguousarray(gt_poses.transpose(0, 2, 1)), sd)
    return z, vq_decoder(z, sd)


# ----------------------------------------------------------------------------------------------
# nets/spg/gated_pixelcnn_v2.py
# ----------------------------------------------------------------------------------------------

def conv2d(x, w, b, pad_h, pad_w):
    """nn.Conv2d stride 1: x (B,C,H,W), w (O,C,kh,kw) -> (B,O,H+2ph-kh+1,W+2pw-kw+1)."""
    B, C, H, W = x.shape
    O, _, kh, kw = w.shape
    xp = np.pad(x, ((0, 0), (0, 0), (pad_h, pad_h), (pad_w, pad_w)))
    Ho, Wo = H + 2 * pad_h - kh + 1, W + 2 * pad_w - kw + 1
    out = np.zeros((B, O, Ho, Wo), F32)
    for i in range(kh):
        for j in range(kw):
            xs = xp[:, :, i:i + Ho, j:j + Wo].reshape(B, C, Ho * Wo)
            out += np.matmul(np.ascontiguousarray(w[:, :, i, j])[None], xs).reshape(B, O, Ho, Wo)
    return out + b[None, :, None, None]


def gated_activation(x):
    """GatedActivation.forward (`gated_pixelcnn_v2.py:20-22`)."""
    a, g = np.split(x, 2, axis=1)
    return (np.tanh(a) * sigmoid(g)).astype(F32)


def causal_weights(sd, n_layers):
    """make_causal (`gated_pixelcnn_v2.py:57-59`): layer 0 (mask 'A') zeroes the last kernel row / column."""
    sd = dict(sd)
    v = sd["layers.0.vert_stack.weight"].copy()
    v[:, :, -1] = 0
    h = sd["layers.0.horiz_stack.weight"].copy()
    h[:, :, :, -1] = 0
    sd["layers.0.vert_stack.weight"], sd["layers.0.horiz_stack.weight"] = v, h
    return sd


def gated_masked_conv2d(x_v, x_h, label, sd, p, kernel, residual, bh_model=True):
    """GatedMaskedConv2d.forward (`gated_pixelcnn_v2.py:61-87`); bh_model=False: the `else` branch (:80-85), vertical kernels one
    column wide (:37-38), out_h = out_v."""
    h = sd[p + ".class_cond_embedding.weight"][label]                       # (B, 2*dim)
    h_vert = conv2d(x_v, sd[p + ".vert_stack.weight"], sd[p + ".vert_stack.bias"], kernel // 2, 1 if bh_model else 0)
    h_vert = h_vert[:, :, :x_v.shape[-2], :]
    out_v = gated_activation(h_vert + h[:, :, None, None])
    if not bh_model:
        out_v = conv2d(out_v, sd[p + ".horiz_resid.weight"], sd[p + ".horiz_resid.bias"], 0, 0)
        if residual:
            out_v = out_v + x_v
        return out_v, out_v
    h_horiz = conv2d(x_h, sd[p + ".horiz_stack.weight"], sd[p + ".horiz_stack.bias"], 0, 1)
    h_horiz = h_horiz[:, :, :, :x_h.shape[-1]]
    v2h = conv2d(h_vert, sd[p + ".vert_to_horiz.weight"], sd[p + ".vert_to_horiz.bias"], 0, 0)
    out = gated_activation(v2h + h_horiz + h[:, :, None, None])
    out_h = conv2d(out, sd[p + ".horiz_resid.weight"], sd[p + ".horiz_resid.bias"], 0, 0)
    if residual:
        out_h = out_h + x_h
    return out_v, out_h


def pixelcnn_forward(x, label, aud, sd, n_layers, audio=True, bh_model=True):
    """GatedPixelCNN.forward (`gated_pixelcnn_v2.py:130-150`), eval mode; audio / bh_model as the constructor's flags (the shipped
    configuration is audio=True, bh_model=True).

    x (B,H,W) int64 codes, label (B,) int64, aud (B,256,H,W) or None -> logits (B,input_dim,H,W).
    `sd` must already have gone through `causal_weights`.
    """
    e = sd["embedding.weight"][x]                                           # (B,H,W,C)
    xv = xh = np.ascontiguousarray(e.transpose(0, 3, 1, 2))
    for i in range(n_layers):
        if i == 1 and audio:
            a = conv2d(aud, sd["embedding_aud.weight"], sd["embedding_aud.bias"], 0, 0)
            xv = conv2d(np.concatenate([xv, a], 1), sd["fusion_v.weight"], sd["fusion_v.bias"], 0, 0)
            if bh_model:
                xh = conv2d(np.concatenate([xh, a], 1), sd["fusion_h.weight"], sd["fusion_h.bias"], 0, 0)
        xv, xh = gated_masked_conv2d(xv, xh, label, sd, f"layers.{i}", 7 if i == 0 else 3, i != 0, bh_model)
    y = relu(conv2d(xh if bh_model else xv, sd["output_conv.0.weight"], sd["output_conv.0.bias"], 0, 0))
    return conv2d(y, sd["output_conv.2.weight"], sd["output_conv.2.bias"], 0, 0)


def softmax(x):
    m = x.max(axis=-1, keepdims=True)
    e = np.exp(x - m)
    return (e / e.sum(axis=-1, keepdims=True)).astype(F32)


def pixelcnn_generate(label, aud, sd, n_layers, H, uniforms=None, return_logits=False, pre_latents=None, pre_audio=None,
                      audio=True, bh_model=True, W=2):
    """GatedPixelCNN.generate (`gated_pixelcnn_v2.py:152-177`) — the O(H^2) full-grid recompute, as written.

    `uniforms is None`: greedy harness of SURVEY.md §0.3 (argmax of logits[:, :, i, j], ties -> lowest
    index).  Otherwise `uniforms` (B,H,2) in [0,1) drives an inverse-CDF draw from
    softmax(logits[:, :, i, j]) — the distribution `probs.multinomial(1)` samples (`:173-176`); torch's
    RNG stream itself is not reproducible off-torch, so stochastic parity is defined on injected uniforms.
    `pre_latents` (B,H0,2) / `pre_audio` (B,256,H0,2): the continuity prefix (`:158-165`) — known codes and their
    audio rows are prepended, positions h0..h0+H-1 are generated, and only those are returned.
    """
    sd = causal_weights(sd, n_layers)
    B = len(label)
    x = np.zeros((B, H, W), np.int64)
    h0 = 0
    if pre_latents is not None:
        x = np.concatenate([np.asarray(pre_latents, np.int64), x], axis=1)
        if audio:
            aud = np.concatenate([pre_audio, aud], axis=2)
        h0 = pre_latents.shape[1]
    logs = []
    for i in range(h0, h0 + H):
        for j in range(W):
            lg = pixelcnn_forward(x, label, aud, sd, n_layers, audio, bh_model)[:, :, i, j]
            if return_logits:
                logs.append(lg.copy())
            if uniforms is None:
                x[:, i, j] = np.argmax(lg, axis=-1)
            else:
                x[:, i, j] = sample_inverse_cdf(lg, uniforms[:, i - h0, j])
    x = x[:, h0:]
    if return_logits:
        return x, np.stack(logs, 1).reshape(B, H, W, -1)
    return x


def det_expf(x):
    """exp(x) for x <= 0 exactly as the HIP sampler computes it (`csrc/vq.hip::det_expf`): fp32 multiplies and adds only, one
    IEEE rounding each, no fused multiply-add — numpy's float32 arithmetic gives the same bits on any host.  Cody-Waite
    reduction by ln 2, Cephes' degree-5 polynomial, 2^n by exponent; arguments below -86 give 0."""
    x = np.asarray(x, F32)
    n = np.rint(x * F32(1.44269504088896341)).astype(F32)
    r = (x - n * F32(0.693145751953125)).astype(F32)
    r = (r - n * F32(1.42860682030941723212e-6)).astype(F32)
    q = np.full_like(r, F32(1.9875691500e-4))
    for c in (1.3981999507e-3, 8.3334519073e-3, 4.1665795894e-2, 1.6666665459e-1, 5.0000001201e-1):
        q = (q * r + F32(c)).astype(F32)          # numpy rounds the product, then the sum: two fp32 operations
    y = (q * (r * r) + r).astype(F32)
    y = (y + F32(1.0)).astype(F32)
    out = (y * np.ldexp(F32(1.0), np.maximum(n, -126).astype(np.int32)).astype(F32)).astype(F32)
    return np.where(x < F32(-86.0), F32(0.0), out).astype(F32)


def sample_inverse_cdf(logits, u, nthreads=256):
    """Draw from softmax(logits) with a given uniform: first index whose running sum of exp(l - max) exceeds u * total.

    The reference draws with `probs.multinomial(1)` (`gated_pixelcnn_v2.py:173-176`); any exact inverse-CDF draw has that
    distribution.  The summation STRUCTURE below is the one the HIP sampler uses (so draws compare bit for bit):
    `nthreads` contiguous chunks summed left to right, chunk sums prefix-summed left to right, then a left-to-right walk
    inside the owning chunk; all in float32, the exponential included (`det_expf`).
    """
    B, V = logits.shape
    chunk = (V + nthreads - 1) // nthreads
    out = np.zeros(B, np.int64)
    for b in range(B):
        m = logits[b].max()
        e = det_expf((logits[b] - m).astype(F32))
        pre = np.zeros(nthreads + 1, F32)
        c = F32(0)
        for t in range(nthreads):
            s_ = F32(0)
            for v in range(t * chunk, min((t + 1) * chunk, V)):
                s_ = F32(s_ + e[v])
            c = F32(c + s_)
            pre[t + 1] = c
        thr = F32(F32(u[b]) * c)
        owner = None
        for t in range(nthreads):
            if pre[t] <= thr and (thr < pre[t + 1] or t == nthreads - 1):
                owner = t
                break
        v0, v1 = owner * chunk, min((owner + 1) * chunk, V)
        if v0 >= V:
            out[b] = V - 1
            continue
        k, c = v1 - 1, pre[owner]
        for v in range(v0, v1):
            c = F32(c + e[v])
            if c > thr:
                k = v
                break
        out[b] = k
    return out


def philox4x32_10(counter, key):
    """Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; Random123's philox4x32_R(10, ...)):
    4 counter words, 2 key words -> 4 output words.  Restated from the publication (the reference draws with torch's generator;
    this is the stream the HIP sampler uses instead, `csrc/vq.hip::philox4x32_10`); pinned to Random123's known-answer vectors
    by `tests/test_sampling_oracle.py`."""
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    c = [int(v) & 0xFFFFFFFF for v in counter]
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k0) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c[3] ^ k1) & 0xFFFFFFFF, p0 & 0xFFFFFFFF]
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return c


def philox_uniform(seed, clip_index, position):
    """The sampler's uniform: counter (position, clip_lo, clip_hi, 0), key (seed_lo, seed_hi); u = (word0 >> 8) * 2^-24."""
    w = philox4x32_10([position, clip_index & 0xFFFFFFFF, (clip_index >> 32) & 0xFFFFFFFF, 0],
                      [seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF])
    return np.float32((w[0] >> 8) * (1.0 / 16777216.0))


def philox_uniforms(seed, clip_index0, B, H):
    """(B,H,2) uniforms the HIP sampler draws in TS_SAMPLE_PHILOX mode: clip b -> subsequence clip_index0 + b."""
    u = np.zeros((B, H, 2), F32)
    for b in range(B):
        for r in range(H):
            for j in range(2):
                u[b, r, j] = philox_uniform(seed, clip_index0 + b, r * 2 + j)
    return u


# ----------------------------------------------------------------------------------------------
# nets/smplx_body_pixel.py / nets/smplx_body_vq.py orchestration
# ----------------------------------------------------------------------------------------------

def body_pixel_infer(mfcc, ids, sd_audio, sd_pix, sd_body, sd_hand, n_layers=15, uniforms=None):
    """`TrainWrapper.infer_on_audio` after the MFCC front-end (`smplx_body_pixel.py:272-285`).

    mfcc (B,T,64) -> (codes (B,H,2) int64, poses (B,4H,129) float32); greedy unless `uniforms`.
    """
    feat = audio_encoder(np.ascontiguousarray(mfcc.transpose(0, 2, 1)), sd_audio)          # (B,256,H)
    aud = np.repeat(feat[:, :, :, None], 2, axis=3)
    H = aud.shape[2]
    codes = pixelcnn_generate(ids, aud, sd_pix, n_layers, H, uniforms)
    body = vqvae_decode(codes[..., 0], sd_body)
    hand = vqvae_decode(codes[..., 1], sd_hand)
    poses = np.concatenate([body, hand], axis=1).transpose(0, 2, 1)
    return codes, np.ascontiguousarray(poses), feat


def body_pixel_infer_continuity(mfcc, gap, ids, sd_audio, sd_pix, sd_body, sd_hand, n_layers=15):
    """`infer_on_audio(continuity=True)` (`smplx_body_pixel.py:260-269,291-304`), greedy: the features are split at frame
    `gap` (`get_mfcc_sepa`), each part goes through the audio encoder and the decoders on its own, and the second part's
    codes are generated behind the first part's codes and audio rows as prefix.  -> poses (B, 4*(H0+H1), 129), codes."""
    def part(m, pre_codes=None, pre_aud=None):
        feat = audio_encoder(np.ascontiguousarray(m.transpose(0, 2, 1)), sd_audio)
        aud = np.repeat(feat[:, :, :, None], 2, axis=3)
        codes = pixelcnn_generate(ids, aud, sd_pix, n_layers, aud.shape[2], pre_latents=pre_codes, pre_audio=pre_aud)
        body, hand = vqvae_decode(codes[..., 0], sd_body), vqvae_decode(codes[..., 1], sd_hand)    # Decoder ignores pre_state
        return codes, aud, body, hand
    c0, a0, b0, h0 = part(mfcc[:, :gap])
    c1, _, b1, h1 = part(mfcc[:, gap:], c0, a0)
    poses = np.concatenate([np.concatenate([b0, b1], 2), np.concatenate([h0, h1], 2)], axis=1).transpose(0, 2, 1)
    return np.ascontiguousarray(poses), np.concatenate([c0, c1], 1)


def assemble_full(body, face, lower_pose33):
    """Caller-side output assembly, `scripts/demo.py:207-229` + `data_utils/lower_body.py:68-87` (`part2full`).

    body (B,Tb,129), face (B,Tf,103) = jaw(3) + expression(100) -> (B,Tf,265).  The body is padded with its last frame
    or trimmed to the face length (demo.py:207-211), concatenated as jaw | body | expression (demo.py:225), then the four
    lower-body blocks are inserted (lower_body.py:77-86).
    """
    B, Tb, _ = body.shape
    Tf = face.shape[1]
    if Tb < Tf:
        body = np.concatenate([body, np.repeat(body[:, -1:], Tf - Tb, axis=1)], axis=1)
    else:
        body = body[:, :Tf]
    p = np.concatenate([face[..., :3], body, face[..., 3:]], axis=-1)
    lp = np.broadcast_to(np.asarray(lower_pose33, np.float32), (B, Tf, 33))
    return np.concatenate([p[..., :3], lp[..., :15], p[..., 3:6], lp[..., 15:21], p[..., 6:9], lp[..., 21:27],
                           p[..., 9:12], lp[..., 27:], p[..., 12:]], axis=-1).astype(np.float32)


def body_vq_infer(poses129, sd_body, sd_hand):
    """`s2g_body_vq.TrainWrapper.infer_on_audio(initial_pose=gt)` core (`smplx_body_vq.py:254-281,293`).

    poses129 (B,T,129) in c_index order -> (out (T, B*129), codes (B,H,2)).
    """
    _, rb, ib = vqvae_forward(poses129[..., :39], sd_body)
    _, rh, ih = vqvae_forward(poses129[..., 39:], sd_hand)
    pred = np.concatenate([rb, rh], axis=1).transpose(0, 2, 1)                  # (B,T,129)
    out = np.concatenate(list(pred), axis=1)                                     # np.concatenate(output, axis=1)
    return out, np.stack([ib, ih], -1)
