"""Drop-in for `nets/smplx_face.py` of the reference (speech -> jaw pose + 100 expression parameters).

Same constructor / `infer_on_audio` / `generate` surface as the reference `TrainWrapper` (`smplx_face.py:20-238`); the
wav2vec2 encoder and the LayerNorm conv heads run in libtalkshow_hip.so (`ts_face_generate`).  Inference only.
"""
import numpy as np
import torch
import torch.nn.functional as F

from nets.base import TrainWrapperBaseClass, resolve_device
from nets.utils import denormalize
from talkshow_amd.frontend import get_wav16
from talkshow_amd.modules import FaceGenerator as s2g_face


class TrainWrapper(TrainWrapperBaseClass):
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

        self.generator = s2g_face(
            n_poses=self.config.Data.pose.generate_length,
            each_dim=self.each_dim,
            dim_list=self.dim_list,
            training=not self.args.infer,
            device=self.device,
            identity=False if self.convert_to_6d else True,
            num_classes=self.num_classes,
        ).to(self.device)
        self.discriminator = None
        self.am = None
        super().__init__(args, config)

    def init_optimizer(self):
        self.generator_optimizer = None
        self.discriminator_optimizer = None

    def init_params(self):
        # smplx_face.py:63-93: the face wrapper counts jaw (3), eyes, global orient and the full body
        scale = 2 if self.convert_to_6d else 1
        global_orient = round(3 * scale)
        leye_pose = reye_pose = round(3 * scale)
        jaw_pose = round(3 * scale)
        body_pose = round(63 * scale)
        left_hand_pose = right_hand_pose = round(45 * scale)
        expression = 100 if self.expression else 0

        b_j = 0
        jaw_dim = jaw_pose
        b_e = b_j + jaw_dim
        eye_dim = leye_pose + reye_pose
        b_b = b_e + eye_dim
        body_dim = global_orient + body_pose
        b_h = b_b + body_dim
        hand_dim = left_hand_pose + right_hand_pose
        b_f = b_h + hand_dim
        face_dim = expression

        self.dim_list = [b_j, b_e, b_b, b_h, b_f]
        self.full_dim = jaw_dim + eye_dim + body_dim + hand_dim + face_dim
        self.pose = int(self.full_dim / round(3 * scale))
        self.each_dim = [jaw_dim, eye_dim + body_dim, hand_dim, face_dim]

    def infer_on_audio(self, aud_fn, id=None, initial_pose=None, norm_stats=None, w_pre=False, frame=None, am=None,
                       am_sr=16000, **kwargs):
        '''
        (aud_fn) -> generated face parameters (B, T, 103)        [smplx_face.py:169-218]
        aud_fn: a (B,1,N) tensor of 16 kHz samples (as the reference accepts), or a 16 kHz .wav / .npy path / array.
        '''
        self.generator.eval()

        if self.config.Data.pose.normalization:
            assert norm_stats is not None
            data_mean = norm_stats[0]
            data_std = norm_stats[1]

        if initial_pose is not None:
            B = initial_pose.shape[0]
        else:
            B = 1

        if type(aud_fn) == torch.Tensor:
            aud_feat = aud_fn.to(torch.float32)
        else:
            aud_feat = get_wav16(aud_fn)                                           # (N, 1), librosa.load(sr=16000) semantics
            aud_feat = aud_feat[np.newaxis, ...].repeat(B, axis=0)
            aud_feat = torch.tensor(aud_feat, dtype=torch.float32).transpose(1, 2)  # (B, 1, N)
        if frame is None:
            frame = aud_feat.shape[2] * 30 // 16000
        if id is None:
            id = torch.tensor([[0, 0, 0, 0]], dtype=torch.float32)
        else:
            id = F.one_hot(torch.as_tensor(id).cpu().long(), self.num_classes).to(torch.float32)

        with torch.no_grad():
            pred_poses = self.generator(aud_feat, None, id, time_steps=frame)[0]
            pred_poses = pred_poses.cpu().numpy()
        output = pred_poses

        if self.config.Data.pose.normalization:
            output = denormalize(output, data_mean, data_std)
        return output

    def generate(self, wv2_feat, frame):
        '''smplx_face.py:221-238: wv2_feat (B,1,N) -> tensor (B,frame,103), all-zero id.'''
        self.generator.eval()
        id = torch.zeros((wv2_feat.shape[0], 4), dtype=torch.float32)
        with torch.no_grad():
            pred_poses = self.generator(wv2_feat, None, id, time_steps=frame)[0]
        return pred_poses
