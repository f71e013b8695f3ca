"""Drop-in for `nets/smplx_body_vq.py` of the reference (VQ-VAE encode -> quantise -> decode of GT poses)."""
import numpy as np
import torch
import torch.nn.functional as F

from nets.base import TrainWrapperBaseClass, resolve_device
from nets.utils import denormalize
from talkshow_amd import _lib
from talkshow_amd.modules import VQVAE as s2g_body
from talkshow_amd.pose_index import c_index_3d, c_index_6d


class TrainWrapper(TrainWrapperBaseClass):
    def __init__(self, args, config):
        model_cfg, pose_cfg = config.Model, config.Data.pose
        self.args, self.config = args, config
        self.device = resolve_device(args.gpu)
        self.global_step = self.epoch = 0
        self.convert_to_6d, self.expression = pose_cfg.convert_to_6d, pose_cfg.expression
        self.init_params()
        self.num_classes = 4
        self.composition = model_cfg.composition
        vq_kw = dict(embedding_dim=64, num_embeddings=model_cfg.code_num, num_hiddens=1024, num_residual_layers=2,
                     num_residual_hiddens=512)
        if self.composition:     # separate body (39-d) and hand (90-d) VQ-VAEs, `smplx_body_vq.py:39-43`
            self.g_body = s2g_body(self.each_dim[1], **vq_kw).to(self.device)
            self.g_hand = s2g_body(self.each_dim[2], **vq_kw).to(self.device)
        else:                    # one VQ-VAE over all 129 dims, `:45-46`
            self.g = s2g_body(self.each_dim[1] + self.each_dim[2], **vq_kw).to(self.device)
        self.discriminator = None
        self.c_index = c_index_6d if self.convert_to_6d else c_index_3d   # `smplx_body_vq.py:50-53`: 78 + 180 modelled dims in the 6-D form
        super().__init__(args, config)

    def init_optimizer(self):
        self.g_body_optimizer = self.g_hand_optimizer = self.g_optimizer = None
        self.generator_optimizer = self.discriminator_optimizer = None

    def _nets(self):
        return {'g_body': self.g_body, 'g_hand': self.g_hand} if self.composition else {'g': self.g}

    def state_dict(self):
        # checkpoint layout of `smplx_body_vq.py:77-94`: one entry per network + empty optimiser / discriminator slots
        out = {'discriminator': None, 'discriminator_optim': None}
        for name, net in self._nets().items():
            out[name] = net.state_dict()
            out[name + '_optim'] = None
        return out

    def load_state_dict(self, state_dict):
        for name, net in self._nets().items():
            net.load_state_dict(state_dict[name])

    def parameters(self):
        return self.g_body.parameters() if self.composition else self.g.parameters()

    def reconstruct_batch(self, poses129):
        """Batched device entry: poses (B,T,129) in c_index order -> (codes (B,H,2), recon (B,T',129))."""
        dev = self.g_body._dev()
        poses129 = torch.as_tensor(poses129, dtype=torch.float32, device=dev).contiguous()
        B, T, _ = poses129.shape
        H = T // 2 // 2
        codes = torch.empty((B, H, 2), dtype=torch.int64, device=dev)
        recon = torch.empty((B, 4 * H, poses129.shape[-1]), dtype=torch.float32, device=dev)
        _lib.check(_lib.load().ts_body_vq_infer(self.g_body.handle(), self.g_hand.handle(), _lib.dptr(poses129), B, T,
                                                _lib.dptr(codes), _lib.dptr(recon), _lib.stream_ptr()))
        return codes, recon

    def infer_on_audio(self, aud_fn, initial_pose=None, norm_stats=None, exp=None, var=None, w_pre=False,
                       continuity=False, id=None, fps=15, sr=22000, smooth=False, **kwargs):
        '''
        initial_pose: (B, C, T)  -> reconstruction, np.concatenate'd over the batch: (T, B*129)
        [smplx_body_vq.py:208-295]
        '''
        assert self.args.infer, "train mode"
        if self.config.Data.pose.normalization:
            assert norm_stats is not None
            data_mean = norm_stats[0]
            data_std = norm_stats[1]

        # the reference dereferences gt unconditionally (smplx_body_vq.py:255); say so instead of a TypeError
        if initial_pose is None:
            raise ValueError("s2g_body_vq.infer_on_audio needs initial_pose=(B,165,T): it reconstructs given poses")
        gt = initial_pose[:, :, :].to(self.device).to(torch.float32)
        if id is None:
            id = F.one_hot(torch.tensor([[0]]), self.num_classes).to(self.device)

        with torch.no_grad():
            gt_poses = gt[:, self.c_index].permute(0, 2, 1).contiguous()          # (B, T, 129)
            if self.composition:
                if continuity:
                    chunks = []
                    for i in range(5):                                            # smplx_body_vq.py:258-268
                        _, r = self.reconstruct_batch(gt_poses[:, i * 60:(i + 1) * 60])
                        chunks.append(r)
                    pred_poses = torch.cat(chunks, dim=1)
                else:
                    _, pred_poses = self.reconstruct_batch(gt_poses)
            else:
                _, pred_poses = self.g.forward_nlc(gt_poses)
            pred_poses = pred_poses.cpu().numpy()                                  # already (B, T, C)
        output = pred_poses

        if self.config.Data.pose.normalization:
            output = denormalize(output, data_mean, data_std)

        if smooth:
            # `smplx_body_vq.py:284-291`: frames 149..158 of clip 0 are blended towards their predecessor, the weight of
            # the frame itself growing from 0.08 to 0.8 (a chunk seam at 150 frames)
            first, count, lam = 149, 10, 0.8
            for i in range(count):
                t = first + i
                keep = lam * (i + 1) / count
                output[0, t] = (1 - keep) * output[0, t - 1] + keep * output[0, t]

        output = np.concatenate(output, axis=1)
        return output
