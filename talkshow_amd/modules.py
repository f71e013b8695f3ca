"""Module-shaped host objects over the C-ABI handles.

The reference's wrappers expose `nn.Module`s (`.generator`, `.g_body`, `.g_hand`, `.audioencoder`) on which callers
call `.eval()`, `.state_dict()`, `.load_state_dict()`, `.parameters()` and the model-specific entry points
(`VQVAE.encode/decode/__call__`, `GatedPixelCNN.generate`, `AudioEncoder.__call__`).  These classes keep that
surface — weights live in a CPU `OrderedDict` under the reference's key names — but every compute method goes to
libtalkshow_hip.so.  There is no torch implementation behind them.

Shape conventions of the methods follow the reference (channels-first tensors in, channels-first tensors out) so
that `nets/` reads like the reference wrappers; the transposes to the library's NLC layout happen here, on device.
"""
import ctypes as C
import os
from collections import OrderedDict

import numpy as np
import torch

from . import _lib, synth


class NativeModule:
    """Weights in reference state_dict form + a lazily (re)built device handle."""

    _schema_cache = {}

    def __init__(self, schema_sd):
        self._sd = OrderedDict((k, torch.from_numpy(np.array(v))) for k, v in schema_sd.items())
        self._handle = None
        self._device = torch.device("cpu")
        self.training = False

    # --- nn.Module-like surface -------------------------------------------------------------------
    def state_dict(self):
        return OrderedDict((k, v.clone()) for k, v in self._sd.items())

    def load_state_dict(self, sd, strict=True):
        sd = OrderedDict((k.replace("module.", ""), v) for k, v in sd.items())
        missing = [k for k in self._sd if k not in sd]
        unexpected = [k for k in sd if k not in self._sd]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for {type(self).__name__}: "
                               f"missing keys {missing[:5]}{'...' if len(missing) > 5 else ''}, "
                               f"unexpected keys {unexpected[:5]}{'...' if len(unexpected) > 5 else ''}")
        for k, cur in self._sd.items():
            if k not in sd:
                continue
            v = sd[k]
            v = v.detach().cpu() if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v))
            if tuple(v.shape) != tuple(cur.shape):
                raise RuntimeError(f"size mismatch for {k}: copying a param with shape {tuple(v.shape)}, "
                                   f"the shape in current model is {tuple(cur.shape)}")
            self._sd[k] = v.to(cur.dtype).contiguous().clone()
        self._after_load()
        self._release()
        return self

    def _after_load(self):
        pass

    _BUFFER_SUFFIXES = ("running_mean", "running_var", "num_batches_tracked", "vq_layer.embeddings", "ema_dw.hidden",
                        "ema_cluster_size.hidden")

    def parameters(self):
        # nn.Module.parameters() leaves buffers out (BatchNorm statistics, the EMA codebook of VectorQuantizerEMA)
        return (v for k, v in self._sd.items() if v.dtype == torch.float32 and not k.endswith(self._BUFFER_SUFFIXES))

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("talkshow_amd is an inference path; training is out of scope (DESIGN.md)")
        return self.eval()

    def to(self, device):
        self._device = torch.device(device)
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda", device if device is not None else torch.cuda.current_device()))

    # --- native handle ---------------------------------------------------------------------------
    def _release(self):
        if self._handle is not None:
            self._destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _require_hip(self):
        if self._device.type != "cuda" or not torch.cuda.is_available():
            raise RuntimeError(f"{type(self).__name__}: device is '{self._device}' and torch.cuda.is_available() is "
                               f"{torch.cuda.is_available()}, but this implementation only runs on a HIP device "
                               "(torch device 'cuda:N' on ROCm, MI355X / gfx950). There is no CPU path.")
        return self._device.index if self._device.index is not None else torch.cuda.current_device()

    def _ctx(self):
        return _lib.context(self._require_hip())

    def handle(self):
        if self._handle is None:
            with torch.cuda.device(self._require_hip()):     # the caller's current device is left as it was
                self._handle = self._create(self._ctx())
        return self._handle

    def _dev(self):
        return torch.device("cuda", self._require_hip())


_range_seen = {}   # (id(owning tensor), view geometry, n) -> (weakref to the owner, _version): caller-owned device tensors already checked


def _check_index_range(x, n, what):
    """IndexError for indices outside [0, n) like nn.Embedding.  Host-origin indices (ints, lists, numpy, CPU tensors) are
    checked on the host before upload: no device sync.  A device tensor costs one device->host sync the first time this
    tensor OBJECT is seen at this version; the cache holds a weak reference to the object (never its address: tensors built
    inside a call get recycled addresses with _version 0, and a stale hit would skip the check)."""
    import weakref
    if isinstance(x, torch.Tensor) and x.is_cuda:
        owner = x._base if x._base is not None else x       # a view (ids[:n]) is a new object per call: key on the tensor it views
        key = (id(owner), x.storage_offset(), tuple(x.shape), tuple(x.stride()), n)
        ent = _range_seen.get(key)
        if ent is not None and ent[0]() is owner and ent[1] == x._version:
            return
        lo, hi = (int(x.min()), int(x.max())) if x.numel() else (0, 0)
        if lo >= 0 and hi < n:
            if len(_range_seen) > 256:
                _range_seen.clear()
            _range_seen[key] = (weakref.ref(owner), x._version)
    else:
        a = x.detach().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)
        lo, hi = (int(a.min()), int(a.max())) if a.size else (0, 0)
    if lo < 0 or hi >= n:
        raise IndexError(f"{what} out of range: [{lo}, {hi}] not within [0, {n})")


def _index_tensor(x, n, what, device):
    """x (int / list / numpy / tensor anywhere) -> flat contiguous int64 tensor on `device`, range-checked (see above)."""
    _check_index_range(x, n, what)
    return torch.as_tensor(x, dtype=torch.int64, device=device).reshape(-1).contiguous()


def _dev_f32(x, device):
    return torch.as_tensor(x, dtype=torch.float32, device=device).contiguous()


class AudioEncoder(NativeModule):
    """`vqvae_1d.AudioEncoder(in_dim, num_hiddens, num_residual_layers, num_residual_hiddens)` (`vqvae_1d.py:11-34`)."""

    def __init__(self, in_dim, num_hiddens, num_residual_layers, num_residual_hiddens=None):
        self.in_dim, self.num_hiddens, self.nres = in_dim, num_hiddens, num_residual_layers
        super().__init__(synth.audioencoder_state_dict(0, in_dim, num_hiddens, num_residual_layers))

    def _create(self, ctx):
        arr, n, keep = _lib.pack_state_dict(self._sd)
        h = C.c_void_p()
        _lib.check(_lib.load().ts_audioenc_create(ctx, arr, n, self.in_dim, self.num_hiddens, self.nres, C.byref(h)))
        return h

    def _destroy(self, h):
        _lib.load().ts_convnet_destroy(h)

    def forward_nlc(self, mfcc):
        """mfcc (B,T,in_dim) device tensor -> (B,T//4,num_hiddens)."""
        mfcc = _dev_f32(mfcc, self._dev())
        B, T, _ = mfcc.shape
        if T < 4:
            raise RuntimeError(f"sequence too short: {T} frames (need >= 4 for one code row)")
        out = torch.empty((B, T // 2 // 2, self.num_hiddens), dtype=torch.float32, device=mfcc.device)
        _lib.check(_lib.load().ts_audioenc_forward(self.handle(), _lib.dptr(mfcc), B, T, _lib.dptr(out), _lib.stream_ptr()))
        return out

    def __call__(self, x, frame_num=0):
        """reference call shape: x (B,in_dim,T) -> (B,num_hiddens,T//4) (`vqvae_1d.py:27-34`)."""
        x = _dev_f32(x, self._dev())
        return self.forward_nlc(x.transpose(1, 2).contiguous()).transpose(1, 2)


class VQVAE(NativeModule):
    """`vqvae_1d.VQVAE(in_dim, embedding_dim, num_embeddings, num_hiddens, num_residual_layers, ·)` (`vqvae_1d.py:152-208`)."""

    def __init__(self, in_dim, embedding_dim, num_embeddings, num_hiddens, num_residual_layers, num_residual_hiddens=None,
                 commitment_cost=0.25, decay=0.99, share=False):
        self.in_dim, self.embedding_dim, self.num_embeddings = in_dim, embedding_dim, num_embeddings
        self.num_hiddens, self.nres = num_hiddens, num_residual_layers
        super().__init__(synth.vqvae_state_dict(0, in_dim, embedding_dim, num_embeddings, num_hiddens, num_residual_layers))

    def _create(self, ctx):
        arr, n, keep = _lib.pack_state_dict(self._sd)
        h = C.c_void_p()
        _lib.check(_lib.load().ts_vqvae_create(ctx, arr, n, self.in_dim, self.embedding_dim, self.num_embeddings,
                                               self.num_hiddens, self.nres, C.byref(h)))
        return h

    def _destroy(self, h):
        _lib.load().ts_vqvae_destroy(h)

    # --- NLC device entry points ---
    def encode_nlc(self, poses, want_z=False, want_quantized=True):
        poses = _dev_f32(poses, self._dev())
        B, T, _ = poses.shape
        if T < 4:
            raise RuntimeError(f"sequence too short: {T} frames (need >= 4 for one code row)")
        H = T // 2 // 2
        lat = torch.empty((B, H), dtype=torch.int64, device=poses.device)
        z = torch.empty((B, H, self.embedding_dim), dtype=torch.float32, device=poses.device) if want_z else None
        q = torch.empty((B, H, self.embedding_dim), dtype=torch.float32, device=poses.device) if want_quantized else None
        _lib.check(_lib.load().ts_vqvae_encode(self.handle(), _lib.dptr(poses), B, T, _lib.dptr(z), _lib.dptr(lat),
                                               _lib.dptr(q), _lib.stream_ptr()))
        return z, q, lat

    def decode_nlc(self, latents, out=None, col0=0):
        latents = torch.as_tensor(latents, dtype=torch.int64, device=self._dev()).contiguous()
        B, H = latents.shape
        if out is None:
            out = torch.empty((B, 4 * H, self.in_dim), dtype=torch.float32, device=latents.device)
        _lib.check(_lib.load().ts_vqvae_decode(self.handle(), _lib.dptr(latents), B, H, _lib.dptr(out), out.shape[-1], col0,
                                               _lib.stream_ptr()))
        return out

    def decode_z_nlc(self, z):
        """Decoder.forward on CONTINUOUS latents z (B,H,embedding_dim) -> (B,4H,in_dim) (`ts_vqvae_decode_z`)."""
        z = _dev_f32(z, self._dev())
        B, H, _ = z.shape
        out = torch.empty((B, 4 * H, self.in_dim), dtype=torch.float32, device=z.device)
        _lib.check(_lib.load().ts_vqvae_decode_z(self.handle(), _lib.dptr(z), B, H, _lib.dptr(out), self.in_dim, 0,
                                                 _lib.stream_ptr()))
        return out

    def forward_nlc(self, poses, out=None, col0=0):
        poses = _dev_f32(poses, self._dev())
        B, T, _ = poses.shape
        H = T // 2 // 2
        lat = torch.empty((B, H), dtype=torch.int64, device=poses.device)
        if out is None:
            out = torch.empty((B, 4 * H, self.in_dim), dtype=torch.float32, device=poses.device)
        _lib.check(_lib.load().ts_vqvae_forward(self.handle(), _lib.dptr(poses), B, T, _lib.dptr(lat), _lib.dptr(out),
                                                out.shape[-1], col0, _lib.stream_ptr()))
        return lat, out

    # --- reference call shapes ---
    def encode(self, gt_poses, id=None):
        """`VQVAE.encode` (`vqvae_1d.py:196-199`): gt_poses (B,T,in_dim) -> (e (B,emb,H), latents (B,H))."""
        _, q, lat = self.encode_nlc(gt_poses)
        return q.transpose(1, 2), lat

    def decode(self, b, w, e=None, latents=None, pre_state=None):
        """`VQVAE.decode` (`vqvae_1d.py:201-208`): returns the reference's tuple (recon (B,in_dim,4w), None)."""
        if e is not None:      # continuous latents (B, embedding_dim, w): Decoder.forward on them as they are (`vqvae_1d.py:202-203`)
            z = _dev_f32(e, self._dev()).transpose(1, 2).contiguous()
            return self.decode_z_nlc(z).transpose(1, 2), None
        return self.decode_nlc(latents.reshape(b, w)).transpose(1, 2), None

    def __call__(self, gt_poses, id=None, pre_state=None):
        """`VQVAE.forward`, eval branch (`vqvae_1d.py:184-189`): (e, x_recon (B,in_dim,T))."""
        _, q, lat = self.encode_nlc(gt_poses)
        return q.transpose(1, 2), self.decode_nlc(lat).transpose(1, 2)


class AE(VQVAE):
    """`vqvae_1d.AE(in_dim, embedding_dim, num_embeddings, num_hiddens, num_residual_layers, ·)` (`vqvae_1d.py:211-235`):
    Encoder + Decoder without a quantiser — the FGD feature extractor behind `nets.s2g_body_ae` (`body_ae.py:145-152`)."""

    def __init__(self, in_dim, embedding_dim, num_embeddings, num_hiddens, num_residual_layers, num_residual_hiddens=None):
        self.in_dim, self.embedding_dim, self.num_embeddings = in_dim, embedding_dim, 0
        self.num_hiddens, self.nres = num_hiddens, num_residual_layers
        NativeModule.__init__(self, synth.ae_state_dict(0, in_dim, embedding_dim, num_hiddens, num_residual_layers))

    def encode_nlc(self, poses):
        """poses (B,T,in_dim) -> z (B,T//4,embedding_dim), device tensor."""
        poses = _dev_f32(poses, self._dev())
        B, T, _ = poses.shape
        if T < 4:
            raise RuntimeError(f"sequence too short: {T} frames (need >= 4 for one latent row)")
        z = torch.empty((B, T // 2 // 2, self.embedding_dim), dtype=torch.float32, device=poses.device)
        _lib.check(_lib.load().ts_vqvae_encode(self.handle(), _lib.dptr(poses), B, T, _lib.dptr(z), None, None,
                                               _lib.stream_ptr()))
        return z

    # --- reference call shapes ---
    def encode(self, gt_poses, id=None):
        """`AE.encode` (`vqvae_1d.py:233-235`): gt_poses (B,T,in_dim) -> z (B,embedding_dim,T//4)."""
        return self.encode_nlc(gt_poses).transpose(1, 2)

    def decode(self, *a, **k):
        raise NotImplementedError("AE has no code-index decode; use __call__ (encode -> decode of continuous latents)")

    def __call__(self, gt_poses, id=None, pre_state=None):
        """`AE.forward`, eval branch (`vqvae_1d.py:225-229`): (z (B,emb,H), x_recon (B,in_dim,T)); Decoder ignores pre_state."""
        z = self.encode_nlc(gt_poses)
        return z.transpose(1, 2), self.decode_z_nlc(z).transpose(1, 2)


class GatedPixelCNN(NativeModule):
    """`gated_pixelcnn_v2.GatedPixelCNN(input_dim, dim, n_layers, n_classes, audio, bh_model)`.

    audio=True, bh_model=True (config/body_pixel.json) is the tuned incremental chain (`ts_pixelcnn_*`).  The other three
    constructor variants are supported as the reference supports them, untuned: audio=False with bh_model=True runs the same
    chain with an identity fusion and zero audio rows (bit-identical to having no fusion: 1.0 * x + 0 is exact); bh_model=False
    is the single vertical stack (`ts_pixelcnn_v_*`: columns never mix, grid width any power of two)."""

    def __init__(self, input_dim=256, dim=64, n_layers=15, n_classes=10, audio=False, bh_model=False, aud_dim=256):
        self.input_dim, self.dim, self.n_layers, self.n_classes, self.aud_dim = input_dim, dim, n_layers, n_classes, aud_dim
        self.audio, self.bh_model = bool(audio), bool(bh_model)
        super().__init__(synth.pixelcnn_state_dict(0, input_dim, dim, n_layers, n_classes, aud_dim, audio=self.audio,
                                                   bh_model=self.bh_model))

    def _after_load(self):
        # the reference zeroes these taps in place on every forward of layer 0 (make_causal, gated_pixelcnn_v2.py:57-63),
        # so its state_dict() returns them zeroed after the first call; mirror that.
        self._sd["layers.0.vert_stack.weight"][:, :, -1] = 0
        self._sd["layers.0.horiz_stack.weight"][:, :, :, -1] = 0

    def _create(self, ctx):
        h = C.c_void_p()
        if not self.bh_model:
            arr, n, keep = _lib.pack_state_dict(self._sd)
            _lib.check(_lib.load().ts_pixelcnn_v_create(ctx, arr, n, self.input_dim, self.dim, self.n_layers, self.n_classes,
                                                        int(self.audio), self.aud_dim, C.byref(h)))
            return h
        sd = self._sd
        if not self.audio:   # no audio branch: an identity fusion over zero audio rows is the same arithmetic, exactly
            D = self.dim
            eye = torch.cat([torch.eye(D), torch.zeros(D, D)], 1).reshape(D, 2 * D, 1, 1)
            sd = OrderedDict(sd)
            sd["embedding_aud.weight"], sd["embedding_aud.bias"] = torch.zeros(D, self.aud_dim, 1, 1), torch.zeros(D)
            sd["fusion_v.weight"], sd["fusion_v.bias"] = eye.clone(), torch.zeros(D)
            sd["fusion_h.weight"], sd["fusion_h.bias"] = eye.clone(), torch.zeros(D)
        arr, n, keep = _lib.pack_state_dict(sd)
        _lib.check(_lib.load().ts_pixelcnn_create(ctx, arr, n, self.input_dim, self.dim, self.n_layers, self.n_classes,
                                                  self.aud_dim, C.byref(h)))
        return h

    def _destroy(self, h):
        (_lib.load().ts_pixelcnn_destroy if self.bh_model else _lib.load().ts_pixelcnn_v_destroy)(h)

    def run(self, label, aud_rows, mode=_lib.TS_SAMPLE_PHILOX, codes=None, uniforms=None, seed=0, clip_index0=0,
            want_logits=False, pre_codes=None, pre_aud=None, shape=None):
        """aud_rows (B,H,aud_dim) device (None for audio=False: pass shape=(B,H)); returns (codes (B,H,W) int64, logits
        (B,H,W,V) or None); W = 2 unless bh_model=False and shape=(B,H,W) says otherwise."""
        dev = self._dev()
        W = 2
        if aud_rows is not None:
            aud_rows = _dev_f32(aud_rows, dev)
            B, H, _ = aud_rows.shape
            if shape is not None and len(shape) == 3:
                W = int(shape[2])
        else:
            if self.audio:
                raise ValueError("this network was built with audio=True: aud_rows is required")
            B, H = int(shape[0]), int(shape[1])
            W = int(shape[2]) if len(shape) == 3 else 2
        if self.bh_model and W != 2:
            raise NotImplementedError("bh_model grids have exactly 2 columns (body, hand)")
        if self.bh_model and aud_rows is None:
            aud_rows = torch.zeros((B, H, self.aud_dim), dtype=torch.float32, device=dev)
        label = _index_tensor(label, self.n_classes, "class label", dev)
        if label.numel() == 1 and B > 1:
            label = label.repeat(B)
        if label.numel() != B:
            raise ValueError(f"label must hold 1 or B={B} class indices, got {label.numel()}")
        if mode == _lib.TS_TEACHER_FORCED:
            codes = torch.as_tensor(codes, dtype=torch.int64, device=dev).contiguous()
        else:
            codes = torch.zeros((B, H, W), dtype=torch.int64, device=dev)
        logits = torch.empty((B, H, W, self.input_dim), dtype=torch.float32, device=dev) if want_logits else None
        if uniforms is not None:
            uniforms = _dev_f32(uniforms, dev)
        H0 = 0
        if pre_codes is not None:
            pre_codes = torch.as_tensor(pre_codes, dtype=torch.int64, device=dev).contiguous()
            H0 = pre_codes.shape[1]
            if pre_aud is not None:
                pre_aud = _dev_f32(pre_aud, dev)
            elif self.bh_model:
                pre_aud = torch.zeros((B, H0, self.aud_dim), dtype=torch.float32, device=dev)
        if not self.bh_model:
            _lib.check(_lib.load().ts_pixelcnn_v_generate(
                self.handle(), _lib.dptr(label), _lib.dptr(aud_rows), B, H, W, mode, _lib.dptr(uniforms), int(seed) & (2 ** 64 - 1),
                int(clip_index0), _lib.dptr(codes), _lib.dptr(logits), _lib.dptr(pre_codes), _lib.dptr(pre_aud), H0, _lib.stream_ptr()))
            return codes, logits
        _lib.check(_lib.load().ts_pixelcnn_generate(
            self.handle(), _lib.dptr(label), _lib.dptr(aud_rows), B, H, mode, _lib.dptr(uniforms), int(seed) & (2 ** 64 - 1),
            int(clip_index0), _lib.dptr(codes), _lib.dptr(logits), _lib.dptr(pre_codes), _lib.dptr(pre_aud), H0,
            _lib.stream_ptr()))
        return codes, logits

    def prepare(self, batch_size, rows, mode=_lib.TS_SAMPLE_GREEDY):
        """Serving aid (`ts_pixelcnn_prepare`): capture and pin the whole-call hipGraph of a (batch_size, rows, mode) decode on the current
        stream now, so that the first real call of that shape is already one replay.  Without it a shape runs on chunk graphs until its
        third sighting among the stream's last 16 calls.  No-op for the untuned bh_model=False form."""
        if self.bh_model:
            _lib.check(_lib.load().ts_pixelcnn_prepare(self.handle(), int(batch_size), int(rows), int(mode), _lib.stream_ptr()))
        return self

    def graph_captures(self):
        """hipGraphs captured so far on the current stream (a serving loop checks that this stands still once it is warm)."""
        return int(_lib.load().ts_pixelcnn_graph_captures(self.handle(), _lib.stream_ptr())) if self.bh_model else 0

    def open_stream(self, label, batch_size, max_chunk_rows):
        """A generation session with a persistent row cache (`ts_pixelcnn_stream_*`): `.step(aud_rows)` continues the
        clip(s) where the previous step stopped, at a cost independent of the history length."""
        if not (self.audio and self.bh_model):
            raise NotImplementedError("streaming sessions exist for the shipped configuration (audio=True, bh_model=True)")
        return PixelCNNStream(self, label, batch_size, max_chunk_rows)

    @staticmethod
    def _audio_rows(aud):
        """(B, aud_dim, H, W) audio map of the reference call shape -> the (B, H, aud_dim) rows the C entry takes (ONE audio row per
        code row).  The reference convolves the whole map; its only caller builds it by repeating one row over the columns
        (`smplx_body_pixel.py:274`), and that is the case implemented: a map whose columns differ is refused, not silently
        truncated to its first column."""
        if aud is None:
            return None
        # an expanded view (stride 0 over the columns: `unsqueeze(-1).expand`) or a single column cannot differ: no device work.  A
        # materialised map (`.repeat(1, 1, 1, 2)`, the reference caller's form) costs one device compare + a host read per call on
        # this reference-call-shape path (`generate_batch`, the serving entry, takes rows and never comes here); a host that has
        # validated its maps switches it off with TS_AUDIO_MAP_CHECK=0.  NaNs are reported as NaNs, not as differing columns.
        if aud.shape[-1] > 1 and aud.stride(-1) != 0 and os.environ.get("TS_AUDIO_MAP_CHECK", "1") != "0":
            same = (aud == aud[..., :1]) | (aud != aud)
            if not bool(same.all()):
                raise NotImplementedError("GatedPixelCNN: the audio map's columns differ; one audio row per code row is supported "
                                          "(the reference's caller repeats a row over the columns, smplx_body_pixel.py:274)")
        return aud[..., 0].transpose(1, 2)

    # --- reference call shapes ---
    def generate(self, label, shape=(8, 8), batch_size=64, aud_feat=None, pre_latents=None, pre_audio=None,
                 mode=None, seed=None, uniforms=None):
        """`GatedPixelCNN.generate` (`gated_pixelcnn_v2.py:152-177`): aud_feat (B,aud_dim,H,2) -> codes (B,H,2).

        Default is stochastic like the reference (softmax + one multinomial draw per position), with Philox uniforms
        seeded from torch's default generator; `mode=TS_SAMPLE_GREEDY` gives the argmax harness.
        """
        rows = self._audio_rows(aud_feat)
        pre_rows = self._audio_rows(pre_audio)
        if mode is None:
            mode = _lib.TS_SAMPLE_PHILOX if uniforms is None else _lib.TS_SAMPLE_UNIFORMS
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if mode == _lib.TS_SAMPLE_PHILOX else 0
        codes, _ = self.run(label, rows, mode=mode, uniforms=uniforms, seed=seed, pre_codes=pre_latents, pre_aud=pre_rows,
                            shape=(batch_size, shape[0], shape[1]))
        return codes

    def __call__(self, x, label, aud=None):
        """`GatedPixelCNN.forward` (`gated_pixelcnn_v2.py:130-150`): x (B,H,2) codes -> logits (B,input_dim,H,2)."""
        rows = self._audio_rows(aud)
        _, logits = self.run(label, rows, mode=_lib.TS_TEACHER_FORCED, codes=x, want_logits=True, shape=tuple(x.shape))
        return logits.permute(0, 3, 1, 2)


class PixelCNNStream:
    """Host handle of `ts_pixelcnn_stream`: label (B,) or (1,) int64 fixed for the session."""

    def __init__(self, net, label, batch_size, max_chunk_rows):
        dev = net._dev()
        label = _index_tensor(label, net.n_classes, "class label", dev)
        if label.numel() == 1 and batch_size > 1:
            label = label.repeat(batch_size)
        if label.numel() != batch_size:
            raise ValueError(f"label must hold 1 or B={batch_size} class indices, got {label.numel()}")
        self.net, self.B, self.max_rows = net, int(batch_size), int(max_chunk_rows)
        h = C.c_void_p()
        with torch.cuda.device(dev):
            _lib.check(_lib.load().ts_pixelcnn_stream_open(net.handle(), _lib.dptr(label), self.B, self.max_rows, C.byref(h)))
        self._h = h

    @property
    def rows(self):
        return int(_lib.load().ts_pixelcnn_stream_rows(self._h))

    def step(self, aud_rows, mode=_lib.TS_SAMPLE_PHILOX, uniforms=None, seed=0, clip_index0=0):
        """aud_rows (B,Hc,aud_dim) device -> codes (B,Hc,2) int64 of the next Hc code rows."""
        dev = self.net._dev()
        aud_rows = _dev_f32(aud_rows, dev)
        B, Hc, _ = aud_rows.shape
        if B != self.B:
            raise ValueError(f"session was opened for B={self.B}, got {B}")
        codes = torch.empty((B, Hc, 2), dtype=torch.int64, device=dev)
        if uniforms is not None:
            uniforms = _dev_f32(uniforms, dev)
            if tuple(uniforms.shape) != (B, Hc, 2):       # the library strides them by the chunk's Hc * 2: a wrong shape would misalign clips b > 0
                raise ValueError(f"step(): uniforms must have shape (B={B}, Hc={Hc}, 2), got {tuple(uniforms.shape)}")
        _lib.check(_lib.load().ts_pixelcnn_stream_step(self._h, _lib.dptr(aud_rows), Hc, mode, _lib.dptr(uniforms),
                                                       int(seed) & (2 ** 64 - 1), int(clip_index0), _lib.dptr(codes),
                                                       _lib.stream_ptr()))
        return codes

    def close(self):
        if self._h is not None:
            torch.cuda.synchronize()
            _lib.load().ts_pixelcnn_stream_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FaceGenerator(NativeModule):
    """`s2g_face.Generator(n_poses, each_dim, dim_list, training, device, identity, num_classes)` (`s2g_face.py:142-224`)."""

    def __init__(self, n_poses=88, each_dim=None, dim_list=None, training=False, device=None, identity=True,
                 num_classes=4, n_layers=12):
        # identity=False is what the reference wrapper builds when convert_to_6d is set (`smplx_face.py:37-45`): no id channels,
        # jaw head each_dim[0] wide (6)
        self.identity, self.num_classes, self.n_layers = bool(identity), num_classes, n_layers
        self.jaw_dim = int(each_dim[0]) if each_dim else (3 if identity else 6)
        if self.jaw_dim != (3 if identity else 6):
            raise NotImplementedError(f"jaw head of width {self.jaw_dim} with identity={identity}: the reference pairs 3 with True, 6 with False")
        self.out_dim = self.jaw_dim + 100
        self.device = device
        super().__init__(synth.face_state_dict(0, n_layers=n_layers, num_classes=num_classes, identity=self.identity, jaw_dim=self.jaw_dim))

    def load_state_dict(self, sd, strict=True):
        # checkpoints written with transformers 4.22 (the reference's pin) spell the weight-normed positional conv
        # `weight_g` / `weight_v`
        ren = {"audio_encoder.encoder.pos_conv_embed.conv.weight_g":
               "audio_encoder.encoder.pos_conv_embed.conv.parametrizations.weight.original0",
               "audio_encoder.encoder.pos_conv_embed.conv.weight_v":
               "audio_encoder.encoder.pos_conv_embed.conv.parametrizations.weight.original1"}
        sd = OrderedDict((ren.get(k.replace("module.", ""), k.replace("module.", "")), v) for k, v in sd.items())
        return super().load_state_dict(sd, strict)

    def _create(self, ctx):
        arr, n, keep = _lib.pack_state_dict(self._sd)
        h = C.c_void_p()
        _lib.check(_lib.load().ts_face_create(ctx, arr, n, self.n_layers, self.num_classes if self.identity else 0, C.byref(h)))
        return h

    def _destroy(self, h):
        _lib.load().ts_face_destroy(h)

    def set_arith(self, bf16_products=0):
        """OPT-IN arithmetic plan of the generator's GEMMs (`ts_face_set_arith`): 0 = fp32 MFMA (default; the parity path),
        3 / 6 = split-bf16 with three / six exact bf16 products per fp32 product.  Returns self."""
        _lib.check(_lib.load().ts_face_set_arith(self.handle(), int(bf16_products)))
        return self

    def run(self, wav, id_vec, frames, want_hidden=False):
        """wav (B,N) device fp32, id_vec (B,num_classes) -> (B,frames,103) [, hidden (B,frames,768)]; (B,frames,106) for identity=False."""
        dev = self._dev()
        wav = _dev_f32(wav, dev)
        B, N = wav.shape
        if self.identity:
            id_vec = _dev_f32(id_vec, dev).reshape(-1, self.num_classes)
            if id_vec.shape[0] == 1 and B > 1:
                id_vec = id_vec.repeat(B, 1).contiguous()
        else:
            id_vec = None                                    # Generator(identity=False) never looks at it
        out = torch.empty((B, frames, self.out_dim), dtype=torch.float32, device=dev)
        hid = torch.empty((B, frames, 768), dtype=torch.float32, device=dev) if want_hidden else None
        _lib.check(_lib.load().ts_face_generate(self.handle(), _lib.dptr(wav), B, N, int(frames), _lib.dptr(id_vec),
                                                _lib.dptr(out), _lib.dptr(hid), _lib.stream_ptr()))
        return (out, hid) if want_hidden else out

    def __call__(self, in_spec, gt_poses=None, id=None, pre_state=None, time_steps=None):
        """reference call shape (`s2g_face.py:196`): in_spec (B,1,N) -> (out (B,time_steps,103), None)."""
        wav = _dev_f32(in_spec, self._dev())
        wav = wav.reshape(wav.shape[0], -1)
        return self.run(wav, id, time_steps), None


class MFCC:
    """Device front-end: `get_mfcc_ta` = torchaudio Resample(sr_in -> sr_out) + MFCC(64) (`data_utils/utils.py:148-231`)."""

    def __init__(self, sr_in, sr_out=22000, fps=30, device=None):
        self.sr_in, self.sr_out, self.fps = int(sr_in), int(sr_out), int(fps)
        idx = torch.cuda.current_device() if device is None else torch.device(device).index
        self._dev = torch.device("cuda", idx if idx is not None else torch.cuda.current_device())
        h = C.c_void_p()
        _lib.check(_lib.load().ts_mfcc_create(_lib.context(self._dev.index), self.sr_in, self.sr_out, self.fps, C.byref(h)))
        self._h = h

    def __del__(self):
        try:
            _lib.load().ts_mfcc_destroy(self._h)
        except Exception:
            pass

    def resample(self, wav):
        """stage 1 alone: wav (B,N) or (N,) at sr_in -> (B,N') at sr_out (torchaudio sinc-Hann polyphase), device tensor."""
        wav = _dev_f32(wav, self._dev)
        if wav.ndim == 1:
            wav = wav[None]
        B, N = wav.shape
        out = torch.empty((B, _lib.load().ts_mfcc_resampled_len(self._h, N)), dtype=torch.float32, device=self._dev)
        _lib.check(_lib.load().ts_mfcc_resample(self._h, _lib.dptr(wav), B, N, _lib.dptr(out), _lib.stream_ptr()))
        return out

    def __call__(self, wav):
        """wav (B,N) or (N,) mono samples at sr_in -> (B,T,64) device tensor."""
        wav = _dev_f32(wav, self._dev)
        if wav.ndim == 1:
            wav = wav[None]
        B, N = wav.shape
        T = _lib.load().ts_mfcc_num_frames(self._h, N)
        out = torch.empty((B, T, 64), dtype=torch.float32, device=self._dev)
        _lib.check(_lib.load().ts_mfcc_forward(self._h, _lib.dptr(wav), B, N, _lib.dptr(out), _lib.stream_ptr()))
        return out


def resample_kaiser_device(wav, sr_in, sr_out, device=None):
    """`librosa.resample(..., res_type='kaiser_best')` on the GPU (`ts_resample_kaiser`): wav (B,N) -> (B, ceil(N*sr_out/sr_in))."""
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    dev = torch.device("cuda", idx if idx is not None else torch.cuda.current_device())
    wav = _dev_f32(wav, dev)
    if wav.ndim == 1:
        wav = wav[None]
    B, N = wav.shape
    lib = _lib.load()
    out = torch.empty((B, lib.ts_resample_kaiser_len(N, int(sr_in), int(sr_out))), dtype=torch.float32, device=dev)
    _lib.check(lib.ts_resample_kaiser(_lib.context(dev.index), _lib.dptr(wav), B, N, int(sr_in), int(sr_out), _lib.dptr(out),
                                      _lib.stream_ptr()))
    return out
