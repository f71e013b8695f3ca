"""Drop-in for `nets/body_ae.py` of the reference: the pose auto-encoder whose latent space the FGD metric is measured in
(`scripts/test_body.py:127-135` -> `evaluation/FGD.py:39-48` -> `TrainWrapper.extract`).  Inference side only.

Surface kept (`body_ae.py:22-152`): `TrainWrapper(args, config)`, `.g` (the `vqvae_1d.AE`, 129 -> 64 x T/4 -> 129),
`load_state_dict({'g': ...})`, `state_dict()`, `extract(x) -> (feat (B, T//4, 64), x (B, T, 129))`, `c_index`, `each_dim`.
The encoder runs in libtalkshow_hip.so (`ts_vqvae_encode` on a quantiser-free handle).
"""
import torch

from nets.base import TrainWrapperBaseClass, resolve_device
from talkshow_amd.modules import AE as s2g_body
from talkshow_amd.pose_index import c_index_3d, c_index_6d


class TrainWrapper(TrainWrapperBaseClass):
    def __init__(self, args, config):
        pose_cfg = config.Data.pose
        self.args, self.config = args, config
        self.device = resolve_device(args.gpu)
        self.global_step = self.epoch = 0
        self.gan = False
        self.convert_to_6d, self.expression = pose_cfg.convert_to_6d, pose_cfg.expression
        self.preleng = getattr(pose_cfg, 'pre_pose_length', 0)
        self.init_params()
        self.num_classes = 4
        self.g = s2g_body(self.each_dim[1] + self.each_dim[2], embedding_dim=64, num_embeddings=0, num_hiddens=1024,
                          num_residual_layers=2, num_residual_hiddens=512).to(self.device)
        self.discriminator = None
        self.c_index = c_index_6d if self.convert_to_6d else c_index_3d   # `body_ae.py:50-53`
        super().__init__(args, config)

    def init_optimizer(self):
        self.g_optimizer = self.generator_optimizer = self.discriminator_optimizer = None

    def state_dict(self):
        return {'g': self.g.state_dict(), 'g_optim': None, 'discriminator': None, 'discriminator_optim': None}

    def load_state_dict(self, state_dict):
        self.g.load_state_dict(state_dict['g'])

    def parameters(self):
        return self.g.parameters()

    def extract(self, x):
        """`body_ae.py:145-152`: x (B,T,D) poses; rows wider than the 129 modelled dims are first cut down the way the
        callers' layouts need it (239 = 102 leading face values + 137..., then the `c_index` gather), then encoded.
        Returns (feat (B, T//4, 64), the 129-d poses that were encoded), both device tensors."""
        self.g.eval()
        x = torch.as_tensor(x, dtype=torch.float32)
        if x.shape[2] > self.full_dim:
            if x.shape[2] == 239:
                x = x[:, :, 102:]
            x = x[:, :, self.c_index]
        x = x.to(self.device).contiguous()
        feat = self.g.encode(x)
        return feat.transpose(1, 2), x
