"""Drop-in for `nets/smplx_body_pixel.py` of the reference (speech -> body/hand code indices -> SMPL-X poses).

Same constructor, attributes and method signatures as the reference `TrainWrapper` (`smplx_body_pixel.py:26-326`);
the audio encoder, the autoregressive PixelCNN and both VQ decoders run in libtalkshow_hip.so.  Inference only.
"""
import numpy as np
import torch

from nets.base import TrainWrapperBaseClass, resolve_device
from talkshow_amd import _lib
from talkshow_amd.frontend import get_mfcc_sepa, get_mfcc_ta
from talkshow_amd.modules import AudioEncoder, GatedPixelCNN as pixelcnn, VQVAE as s2g_body, _index_tensor
from talkshow_amd.pose_index import c_index_3d, c_index_6d


def _fresh_seed():
    """A Philox key drawn from torch's generator (so `torch.manual_seed` makes sampling reproducible, as in the reference)."""
    return int(torch.randint(0, 2 ** 62, (1,)).item())


class TrainWrapper(TrainWrapperBaseClass):
    '''
    a wrapper receiving audio features and generating motion (reference: a wrapper receving a batch from data_utils
    and calculate loss)
    '''

    def __init__(self, args, config):
        model_cfg, pose_cfg = config.Model, config.Data.pose
        self.args, self.config = args, config
        self.device = resolve_device(args.gpu)
        self.global_step = self.epoch = 0
        self.convert_to_6d, self.expression = pose_cfg.convert_to_6d, pose_cfg.expression
        self.init_params()
        self.num_classes = 4
        self.audio = True
        self.composition, self.bh_model = model_cfg.composition, model_cfg.bh_model

        # the four networks of `smplx_body_pixel.py:46-57`, same hyper-parameters
        self.audioencoder = AudioEncoder(in_dim=64, num_hiddens=256, num_residual_layers=2,
                                         num_residual_hiddens=256).to(self.device)
        # 6-D rotations double every pose width (39 / 90 -> 78 / 180 modelled dims) and the reference then trades depth for
        # width in the code predictor (`smplx_body_pixel.py:48-52`)
        dim, layer = (512, 10) if self.convert_to_6d else (256, 15)
        self.generator = pixelcnn(2048, dim, layer, self.num_classes, self.audio, self.bh_model).to(self.device)
        vq_kw = dict(embedding_dim=64, num_embeddings=model_cfg.code_num, num_hiddens=1024, num_residual_layers=2,
                     num_residual_hiddens=512)
        self.g_body = s2g_body(self.each_dim[1], **vq_kw).to(self.device)
        self.g_hand = s2g_body(self.each_dim[2], **vq_kw).to(self.device)
        # ... and, as in the reference (`:59-62`), the VQ-VAE weights come from `Model.vq_path` right here
        vq_ckpt = torch.load(model_cfg.vq_path, map_location='cpu')['generator']
        self.g_body.load_state_dict(vq_ckpt['g_body'])
        self.g_hand.load_state_dict(vq_ckpt['g_hand'])

        self.discriminator = None
        self.c_index = c_index_6d if self.convert_to_6d else c_index_3d
        super().__init__(args, config)

    def init_optimizer(self):
        self.generator_optimizer = self.audioencoder_optimizer = self.discriminator_optimizer = None

    def state_dict(self):
        # the six entries of a reference checkpoint (`smplx_body_pixel.py:104-113`); optimiser slots stay empty here
        out = dict.fromkeys(('generator', 'generator_optim', 'audioencoder', 'audioencoder_optim', 'discriminator',
                             'discriminator_optim'))
        out['generator'] = self.generator.state_dict()
        out['audioencoder'] = self.audioencoder.state_dict() if self.audio else None
        return out

    def load_state_dict(self, state_dict):
        """Accepts what the reference accepts (`:115-142`): a checkpoint dict with 'generator' (+ 'audioencoder', optimiser
        entries ignored) or a bare generator state_dict; DataParallel's `module.` prefixes are dropped."""
        def unprefixed(sd):
            return {(k.replace('module.', '') if isinstance(k, str) else k): v for k, v in sd.items()}
        nested = {k: unprefixed(v) for k, v in state_dict.items() if hasattr(v, 'items')}
        self.generator.load_state_dict(nested['generator'] if 'generator' in nested else unprefixed(state_dict))
        if 'audioencoder' in nested and self.audioencoder is not None:
            self.audioencoder.load_state_dict(nested['audioencoder'])

    # ---------------------------------------------------------------------------------------------------------
    def _decode_pair(self, latents):
        """g_body.decode / g_hand.decode + torch.cat (smplx_body_pixel.py:279-285), written into one NLC buffer."""
        B, H, _ = latents.shape
        out = torch.empty((B, 4 * H, self.each_dim[1] + self.each_dim[2]), dtype=torch.float32, device=latents.device)
        self.g_body.decode_nlc(latents[..., 0].contiguous(), out=out, col0=0)
        self.g_hand.decode_nlc(latents[..., 1].contiguous(), out=out, col0=self.each_dim[1])
        return out

    def generate_batch(self, mfcc, ids, mode=_lib.TS_SAMPLE_PHILOX, uniforms=None, seed=None, clip_index0=0, _ids_checked=False):
        """Batched device entry (one call into the C ABI): mfcc (B,T,64), ids (B,) -> codes (B,H,2), poses (B,4H,129).

        This is what `infer_on_audio` runs after the front-end, for B different clips; bench.py and the multi-GPU
        driver use it directly.
        """
        dev = self.generator._dev()
        mfcc = torch.as_tensor(mfcc, dtype=torch.float32, device=dev).contiguous()
        if _ids_checked:   # generate_batches range-checked every batch's ids before stacking them (no sync on the stacked tensor)
            ids = torch.as_tensor(ids, dtype=torch.int64, device=dev).reshape(-1).contiguous()
        else:
            ids = _index_tensor(ids, self.num_classes, 'speaker id', dev)   # IndexError like nn.Embedding; host inputs are checked without a sync
        B, T, _ = mfcc.shape
        H = T // 2 // 2
        # nn.Embedding of a single label broadcasts over the batch in the reference (gated_pixelcnn_v2.py:65-66)
        if ids.numel() == 1 and B > 1:
            ids = ids.repeat(B)
        if ids.numel() != B:
            raise ValueError(f"ids must hold 1 or B={B} speaker indices, got {ids.numel()}")
        if seed is None:   # like the reference, which draws from torch's generator on every call: repeated calls differ
            seed = _fresh_seed() if mode == _lib.TS_SAMPLE_PHILOX else 0     # greedy / injected uniforms never read it
        codes = torch.zeros((B, H, 2), dtype=torch.int64, device=dev)
        poses = torch.empty((B, 4 * H, self.each_dim[1] + self.each_dim[2]), dtype=torch.float32, device=dev)
        if uniforms is not None:
            uniforms = torch.as_tensor(uniforms, dtype=torch.float32, device=dev).contiguous()
        _lib.check(_lib.load().ts_body_pixel_infer(
            self.audioencoder.handle(), self.generator.handle(), self.g_body.handle(), self.g_hand.handle(),
            _lib.dptr(mfcc), _lib.dptr(ids), B, T, mode, _lib.dptr(uniforms), int(seed) & (2 ** 64 - 1), int(clip_index0),
            _lib.dptr(codes), _lib.dptr(poses), _lib.stream_ptr()))
        return codes, poses

    def generate_batches(self, mfccs, ids_list, mode=_lib.TS_SAMPLE_PHILOX, seed=None, clip_index0=0):
        """Coalesced execution of several independent batches (the serving path): the batches' clips are stacked and
        the autoregressive chain runs ONCE over all of them, so a stage's weights are streamed once for all batches
        in flight instead of once per batch (a chain stage is latency-bound below ~64 clips).  Results are bit-identical
        to running each batch alone (tests/test_gpu_parity.py::test_golden_clips_inside_baseline_batches); global clip
        indices (the Philox subsequences) run on from `clip_index0` in list order.

        mfccs: list of (B_i,T,64) device tensors, ids_list: list of (B_i,) -> list of (codes_i, poses_i) views."""
        dev = self.generator._dev()
        sizes = [int(m.shape[0]) for m in mfccs]
        # nn.Embedding's IndexError per submitted batch, BEFORE stacking: host ids are checked on the host, a caller-owned device
        # tensor once per tensor version (modules._check_index_range) — the stacked tensor built below is new on every call, and
        # checking it would cost the serving path one blocking device -> host read per pass
        from talkshow_amd.modules import _check_index_range
        for i in ids_list:
            _check_index_range(i, self.num_classes, 'speaker id')
        mf = torch.cat([torch.as_tensor(m, dtype=torch.float32, device=dev) for m in mfccs], 0)
        ids = torch.cat([torch.as_tensor(i, dtype=torch.int64, device=dev).reshape(-1).expand(b) if
                         torch.as_tensor(i).numel() == 1 else torch.as_tensor(i, dtype=torch.int64, device=dev).reshape(-1)
                         for i, b in zip(ids_list, sizes)], 0)
        codes, poses = self.generate_batch(mf, ids, mode=mode, seed=seed, clip_index0=clip_index0, _ids_checked=True)
        return list(zip(codes.split(sizes, 0), poses.split(sizes, 0)))

    def infer_on_audio(self, aud_fn, initial_pose=None, norm_stats=None, exp=None, var=None, w_pre=False, rand=None,
                       continuity=False, id=None, fps=15, sr=22000, B=1, am=None, am_sr=None, frame=0, **kwargs):
        '''
        (aud_fn, txgfile) -> generated motion (B, T, C)      [smplx_body_pixel.py:232-289]

        Extra keyword arguments understood here (ignored by the reference through **kwargs):
        `greedy=True` (argmax decode), `seed=int`, `uniforms=(B,H,2)`.
        '''
        assert self.args.infer, "train mode"
        self.generator.eval()
        self.g_body.eval()
        self.g_hand.eval()

        if continuity:
            aud_feat, gap = get_mfcc_sepa(aud_fn, sr=sr, fps=fps)
        else:
            aud_feat = get_mfcc_ta(aud_fn, sr=sr, fps=fps, smlpx=True, type='mfcc', am=am)
        aud_feat = aud_feat.transpose(1, 0)
        aud_feat = aud_feat[np.newaxis, ...].repeat(B, axis=0)
        aud_feat = torch.tensor(aud_feat, dtype=torch.float32).to(self.device)

        if id is None:
            id = torch.tensor([0]).to(self.device)
        else:
            lo, hi = int(id.min()), int(id.max())                 # nn.Embedding's IndexError, before anything is launched
            if lo < 0 or hi >= self.num_classes:
                raise IndexError(f"speaker id out of range: [{lo}, {hi}] not within [0, {self.num_classes})")
            id = id.repeat(B)

        mode = _lib.TS_SAMPLE_GREEDY if kwargs.get('greedy', False) else _lib.TS_SAMPLE_PHILOX
        uniforms = kwargs.get('uniforms', None)
        if uniforms is not None:
            mode = _lib.TS_SAMPLE_UNIFORMS
        seed = kwargs.get('seed', None)
        if seed is None:
            seed = _fresh_seed() if mode == _lib.TS_SAMPLE_PHILOX else 0

        with torch.no_grad():
            aud_feat = aud_feat.permute(0, 2, 1)                      # (B, T, 64)
            if continuity:
                # the reference generates the first 2 s, then the rest behind the first part's codes and audio rows as a
                # prefix it re-runs in full (`:260-269`); here the second part continues the first part's row cache
                self.audioencoder.eval()
                session = self.open_stream(id, B, max_frames=max(gap, aud_feat.shape[1] - gap))
                u0 = u1 = None
                if uniforms is not None:                                 # (B, H, 2) for the whole clip: split at the seam's code row
                    uniforms = torch.as_tensor(uniforms, dtype=torch.float32)
                    h0, h1 = gap // 4, (aud_feat.shape[1] - gap) // 4    # code rows of the two parts (each loses its own remainder)
                    if tuple(uniforms.shape) != (B, h0 + h1, 2):
                        raise ValueError(f"continuity: uniforms must have shape (B={B}, {h0} + {h1} code rows, 2), got {tuple(uniforms.shape)}")
                    u0, u1 = uniforms[:, :h0].contiguous(), uniforms[:, h0:h0 + h1].contiguous()
                part0 = session.push(aud_feat[:, :gap], mode=mode, seed=seed, uniforms=u0)
                part1 = session.push(aud_feat[:, gap:], mode=mode, seed=seed, uniforms=u1)
                session.close()
                pred_poses = torch.cat([part0, part1], dim=1).cpu().numpy()
            else:
                self.audioencoder.eval()
                _, poses = self.generate_batch(aud_feat, id, mode=mode, uniforms=uniforms, seed=seed)
                pred_poses = poses.cpu().numpy()

        output = pred_poses
        return output

    def open_stream(self, id, B, max_frames):
        """Long-audio / continuity generation: a session that is fed MFCC chunks and returns the poses of
        each chunk, every chunk continuing the PixelCNN row cache of the ones before it (`ts_pixelcnn_stream_*`) — the
        reference's `infer(..., pre_latents, pre_audio)` chain (`:291-304`) for any number of chunks, at a cost per chunk
        that does not grow with the history.  As in the reference, each chunk goes through the audio encoder and the VQ
        decoders on its own (their receptive fields are not carried across chunks)."""
        return BodyStream(self, id, B, max_frames)

    def infer(self, aud_feat, frame, id, B, pre_latents=None, pre_audio=None, pre_pose=None, mode=None, seed=None):
        """smplx_body_pixel.py:291-304 (continuity helper, reference call shape): returns latents, audio (B,256,H,2),
        body, hand (B,C,T); `pre_latents` / `pre_audio` are re-run as a prefix exactly like the reference does."""
        if mode is None:
            mode = _lib.TS_SAMPLE_PHILOX
        if seed is None:
            seed = _fresh_seed()
        rows = self.audioencoder.forward_nlc(aud_feat)                                    # (B,H,256)
        audio = rows.transpose(1, 2).unsqueeze(dim=-1).repeat(1, 1, 1, 2)
        pre_rows = pre_audio[..., 0].transpose(1, 2).contiguous() if pre_audio is not None else None
        latents, _ = self.generator.run(id, rows, mode=mode, seed=seed, pre_codes=pre_latents, pre_aud=pre_rows)
        out = self._decode_pair(latents)                                                   # Decoder ignores pre_state
        body = out[..., :self.each_dim[1]].transpose(1, 2)
        hand = out[..., self.each_dim[1]:].transpose(1, 2)
        return latents, audio, body, hand

    def generate(self, aud, id, frame_num=0):
        """smplx_body_pixel.py:306-325: aud (B,64,T), id (B,) -> (B,T,129).  (The reference version raises a
        TypeError because `decode` returns tuples; this one returns the tensor it meant to.)"""
        _, poses = self.generate_batch(torch.as_tensor(aud).permute(0, 2, 1), id)
        return poses


class BodyStream:
    """See `TrainWrapper.open_stream`.  `push(mfcc_chunk (B,T,64)) -> poses (B, 4*(T//4), 129)` device tensor."""

    def __init__(self, wrapper, id, B, max_frames):
        self.w, self.B = wrapper, int(B)
        if id is None:
            id = torch.tensor([0])
        self.pix = wrapper.generator.open_stream(id, self.B, max(1, int(max_frames) // 4))
        self.seed = _fresh_seed()          # one stream of random numbers per session unless push() is given a seed

    @property
    def frames(self):
        return 4 * self.pix.rows

    def push(self, mfcc, mode=_lib.TS_SAMPLE_PHILOX, seed=None, clip_index0=0, uniforms=None):
        rows = self.w.audioencoder.forward_nlc(mfcc)                      # (B, T//4, 256), this chunk alone
        codes = self.pix.step(rows, mode=mode, uniforms=uniforms, seed=self.seed if seed is None else seed, clip_index0=clip_index0)
        return self.w._decode_pair(codes)

    def close(self):
        self.pix.close()
