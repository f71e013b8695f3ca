"""Drop-in for `nets/smplx_body_pixel.py` of the reference (speech -> body/hand code indices -> SMPL-X poses).

Same constructor, attributes and method signatures as the reference `TrainWrapper` (`smplx_body_pixel.py:26-326`);
the audio encoder, the autoregressive PixelCNN and both VQ decoders run in libtalkshow_hip.so.  Inference only.
"""
import numpy as np
import torch

from nets.base import TrainWrapperBaseClass, resolve_device
from talkshow_amd import _lib
from talkshow_amd.frontend import get_mfcc_sepa, get_mfcc_ta
from talkshow_amd.modules import AudioEncoder, GatedPixelCNN as pixelcnn, VQVAE as s2g_body, _check_index_range
from talkshow_amd.pose_index import c_index_3d


class TrainWrapper(TrainWrapperBaseClass):
    '''
    a wrapper receiving audio features and generating motion (reference: a wrapper receving a batch from data_utils
    and calculate loss)
    '''

    def __init__(self, args, config):
        self.args = args
        self.config = config
        self.device = resolve_device(self.args.gpu)
        self.global_step = 0

        self.convert_to_6d = self.config.Data.pose.convert_to_6d
        self.expression = self.config.Data.pose.expression
        self.epoch = 0
        self.init_params()
        self.num_classes = 4
        self.audio = True
        self.composition = self.config.Model.composition
        self.bh_model = self.config.Model.bh_model

        self.audioencoder = AudioEncoder(in_dim=64, num_hiddens=256, num_residual_layers=2,
                                         num_residual_hiddens=256).to(self.device)
        if self.convert_to_6d:
            dim, layer = 512, 10
        else:
            dim, layer = 256, 15
        self.generator = pixelcnn(2048, dim, layer, self.num_classes, self.audio, self.bh_model).to(self.device)
        self.g_body = s2g_body(self.each_dim[1], embedding_dim=64, num_embeddings=config.Model.code_num,
                               num_hiddens=1024, num_residual_layers=2, num_residual_hiddens=512).to(self.device)
        self.g_hand = s2g_body(self.each_dim[2], embedding_dim=64, num_embeddings=config.Model.code_num,
                               num_hiddens=1024, num_residual_layers=2, num_residual_hiddens=512).to(self.device)

        # smplx_body_pixel.py:59-62 — the VQ checkpoint is loaded in the constructor
        model_path = self.config.Model.vq_path
        model_ckpt = torch.load(model_path, map_location=torch.device('cpu'))
        self.g_body.load_state_dict(model_ckpt['generator']['g_body'])
        self.g_hand.load_state_dict(model_ckpt['generator']['g_hand'])

        self.discriminator = None
        if self.convert_to_6d:
            raise NotImplementedError("convert_to_6d=true is not used by any shipped config (SURVEY.md §2)")
        self.c_index = c_index_3d

        super().__init__(args, config)

    def init_optimizer(self):
        self.generator_optimizer = None
        self.audioencoder_optimizer = None
        self.discriminator_optimizer = None

    def state_dict(self):
        model_state = {
            'generator': self.generator.state_dict(),
            'generator_optim': None,
            'audioencoder': self.audioencoder.state_dict() if self.audio else None,
            'audioencoder_optim': None,
            'discriminator': None,
            'discriminator_optim': None,
        }
        return model_state

    def load_state_dict(self, state_dict):
        from collections import OrderedDict
        new_state_dict = OrderedDict()  # strip `module.` (smplx_body_pixel.py:117-127)
        for k, v in state_dict.items():
            sub_dict = OrderedDict()
            if v is not None and hasattr(v, 'items'):
                for k1, v1 in v.items():
                    name = k1.replace('module.', '') if isinstance(k1, str) else k1
                    sub_dict[name] = v1
                new_state_dict[k] = sub_dict
            else:
                new_state_dict[k] = v
        state_dict = new_state_dict
        if 'generator' in state_dict:
            self.generator.load_state_dict(state_dict['generator'])
        else:
            self.generator.load_state_dict(state_dict)
        if 'audioencoder' in state_dict and self.audioencoder is not None:
            self.audioencoder.load_state_dict(state_dict['audioencoder'])

    # ---------------------------------------------------------------------------------------------------------
    def _decode_pair(self, latents):
        """g_body.decode / g_hand.decode + torch.cat (smplx_body_pixel.py:279-285), written into one NLC buffer."""
        B, H, _ = latents.shape
        out = torch.empty((B, 4 * H, self.each_dim[1] + self.each_dim[2]), dtype=torch.float32, device=latents.device)
        self.g_body.decode_nlc(latents[..., 0].contiguous(), out=out, col0=0)
        self.g_hand.decode_nlc(latents[..., 1].contiguous(), out=out, col0=self.each_dim[1])
        return out

    def generate_batch(self, mfcc, ids, mode=_lib.TS_SAMPLE_PHILOX, uniforms=None, seed=0, clip_index0=0):
        """Batched device entry (one call into the C ABI): mfcc (B,T,64), ids (B,) -> codes (B,H,2), poses (B,4H,129).

        This is what `infer_on_audio` runs after the front-end, for B different clips; bench.py and the multi-GPU
        driver use it directly.
        """
        dev = self.generator._dev()
        mfcc = torch.as_tensor(mfcc, dtype=torch.float32, device=dev).contiguous()
        ids = torch.as_tensor(ids, dtype=torch.int64, device=dev).reshape(-1).contiguous()
        B, T, _ = mfcc.shape
        H = T // 2 // 2
        # nn.Embedding of a single label broadcasts over the batch in the reference (gated_pixelcnn_v2.py:65-66)
        if ids.numel() == 1 and B > 1:
            ids = ids.repeat(B)
        if ids.numel() != B:
            raise ValueError(f"ids must hold 1 or B={B} speaker indices, got {ids.numel()}")
        _check_index_range(ids, self.num_classes, 'speaker id')   # IndexError like nn.Embedding; sync-free once seen
        codes = torch.zeros((B, H, 2), dtype=torch.int64, device=dev)
        poses = torch.empty((B, 4 * H, self.each_dim[1] + self.each_dim[2]), dtype=torch.float32, device=dev)
        if uniforms is not None:
            uniforms = torch.as_tensor(uniforms, dtype=torch.float32, device=dev).contiguous()
        _lib.check(_lib.load().ts_body_pixel_infer(
            self.audioencoder.handle(), self.generator.handle(), self.g_body.handle(), self.g_hand.handle(),
            _lib.dptr(mfcc), _lib.dptr(ids), B, T, mode, _lib.dptr(uniforms), int(seed) & (2 ** 64 - 1), int(clip_index0),
            _lib.dptr(codes), _lib.dptr(poses), _lib.stream_ptr()))
        return codes, poses

    def generate_batches(self, mfccs, ids_list, mode=_lib.TS_SAMPLE_PHILOX, seed=0, clip_index0=0):
        """Coalesced execution of several independent batches (the serving path): the batches' clips are stacked and
        the autoregressive chain runs ONCE over all of them, so a stage's weights are streamed once for all batches
        in flight instead of once per batch (a chain stage is latency-bound below ~64 clips).  Results are bit-identical
        to running each batch alone (tests/test_gpu_parity.py::test_golden_clips_inside_baseline_batches); global clip
        indices (the Philox subsequences) run on from `clip_index0` in list order.

        mfccs: list of (B_i,T,64) device tensors, ids_list: list of (B_i,) -> list of (codes_i, poses_i) views."""
        dev = self.generator._dev()
        sizes = [int(m.shape[0]) for m in mfccs]
        mf = torch.cat([torch.as_tensor(m, dtype=torch.float32, device=dev) for m in mfccs], 0)
        ids = torch.cat([torch.as_tensor(i, dtype=torch.int64, device=dev).reshape(-1).expand(b) if
                         torch.as_tensor(i).numel() == 1 else torch.as_tensor(i, dtype=torch.int64, device=dev).reshape(-1)
                         for i, b in zip(ids_list, sizes)], 0)
        codes, poses = self.generate_batch(mf, ids, mode=mode, seed=seed, clip_index0=clip_index0)
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
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())

        with torch.no_grad():
            aud_feat = aud_feat.permute(0, 2, 1)                      # (B, T, 64)
            if continuity:
                self.audioencoder.eval()
                pre_pose = {'b': None, 'h': None}
                pre_latents, pre_audio, body_0, hand_0 = self.infer(aud_feat[:, :gap], frame, id, B, pre_pose=pre_pose,
                                                                    mode=mode, seed=seed)
                pre_pose['b'] = body_0[:, :, -4:].transpose(1, 2)
                pre_pose['h'] = hand_0[:, :, -4:].transpose(1, 2)
                _, _, body_1, hand_1 = self.infer(aud_feat[:, gap:], frame, id, B, pre_latents, pre_audio, pre_pose,
                                                  mode=mode, seed=seed + 1)
                body = torch.cat([body_0, body_1], dim=2)
                hand = torch.cat([hand_0, hand_1], dim=2)
                pred_poses = torch.cat([body, hand], dim=1).transpose(1, 2).cpu().numpy()
            else:
                self.audioencoder.eval()
                _, poses = self.generate_batch(aud_feat, id, mode=mode, uniforms=uniforms, seed=seed)
                pred_poses = poses.cpu().numpy()

        output = pred_poses
        return output

    def infer(self, aud_feat, frame, id, B, pre_latents=None, pre_audio=None, pre_pose=None, mode=None, seed=0):
        """smplx_body_pixel.py:291-304 (continuity helper): returns latents, audio (B,256,H,2), body, hand (B,C,T)."""
        if mode is None:
            mode = _lib.TS_SAMPLE_PHILOX
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
