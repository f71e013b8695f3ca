"""Drop-in for `nets/smplx_face.py` of the reference (speech -> jaw pose + 100 expression parameters).

Surface kept from the reference `TrainWrapper` (`smplx_face.py:20-238`): constructor `(args, config)`, attributes
`generator / device / each_dim / dim_list / full_dim / pose / num_classes`, `infer_on_audio(...) -> np (B,T,103)`,
`generate(wv2_feat, frame) -> tensor (B,frame,103)`.  The wav2vec2 encoder and the LayerNorm conv heads run in
libtalkshow_hip.so (`ts_face_generate`).  Inference only: no optimiser, no loss.
"""
import numpy as np
import torch

from nets.base import TrainWrapperBaseClass, resolve_device
from nets.utils import denormalize
from talkshow_amd.frontend import get_wav16
from talkshow_amd.modules import FaceGenerator as s2g_face

# SMPL-X parameter groups in the order the face wrapper lays them out (`smplx_face.py:63-93`): name -> joints
_FACE_LAYOUT = (("jaw", 1), ("eyes", 2), ("body", 1 + 21), ("hands", 2 * 15))


class TrainWrapper(TrainWrapperBaseClass):
    def __init__(self, args, config):
        pose_cfg = config.Data.pose
        self.args, self.config = args, config
        self.device = resolve_device(args.gpu)
        self.global_step = self.epoch = 0
        self.convert_to_6d, self.expression = pose_cfg.convert_to_6d, pose_cfg.expression
        self.num_classes = 4                      # speakers: oliver, chemistry, seth, conan (trainer/options.py:26)
        self.init_params()
        self.generator = s2g_face(n_poses=pose_cfg.generate_length, each_dim=self.each_dim, dim_list=self.dim_list,
                                  training=not args.infer, device=self.device, identity=not self.convert_to_6d,
                                  num_classes=self.num_classes).to(self.device)
        self.discriminator = self.am = None
        super().__init__(args, config)

    def init_optimizer(self):
        self.generator_optimizer = self.discriminator_optimizer = None      # training is out of scope

    def init_params(self):
        """Unlike the body wrappers (`nets/base.py`), the face wrapper counts every SMPL-X group: jaw, both eyes, global
        orientation + 21 body joints, 2 x 15 hand joints (3 values per joint, 6 in the 6-D rotation form) and the
        expression coefficients.  -> each_dim == [3, 72, 90, 100] for the shipped configs."""
        per_joint = 6 if self.convert_to_6d else 3
        width = {name: joints * per_joint for name, joints in _FACE_LAYOUT}
        width["face"] = 100 if self.expression else 0
        starts, at = [], 0
        for name in ("jaw", "eyes", "body", "hands", "face"):
            starts.append(at)
            at += width[name]
        self.dim_list = starts
        self.full_dim = at
        self.pose = int(self.full_dim / per_joint)
        self.each_dim = [width["jaw"], width["eyes"] + width["body"], width["hands"], width["face"]]

    def _identity(self, id):
        # `smplx_face.py:205-208`: no id -> an all-zero 4-vector (NOT class 0), else the one-hot row of the class index
        if id is None:
            return torch.zeros((1, self.num_classes), dtype=torch.float32)
        return torch.eye(self.num_classes, dtype=torch.float32)[torch.as_tensor(id).cpu().long().reshape(-1)]

    def _run(self, samples, id_rows, frame):
        self.generator.eval()
        with torch.no_grad():
            return self.generator(samples, None, id_rows, time_steps=frame)[0]

    def infer_on_audio(self, aud_fn, id=None, initial_pose=None, norm_stats=None, w_pre=False, frame=None, am=None,
                       am_sr=16000, **kwargs):
        '''
        (aud_fn) -> generated face parameters (B, T, 103) as float32 numpy        [smplx_face.py:169-218]
        aud_fn: a (B,1,N) tensor of 16 kHz samples (as the reference accepts), or a .wav / .npy path / array of samples
        (resampled to 16 kHz like `librosa.load(sr=16000)`); B = initial_pose.shape[0] when given, else 1.
        '''
        if self.config.Data.pose.normalization and norm_stats is None:
            raise AssertionError("config.Data.pose.normalization is set: norm_stats=(mean, std) is required")
        if torch.is_tensor(aud_fn):
            samples = aud_fn.to(torch.float32)
        else:
            copies = 1 if initial_pose is None else initial_pose.shape[0]
            mono = torch.from_numpy(get_wav16(aud_fn)[:, 0])                        # (N,)
            samples = mono.reshape(1, 1, -1).repeat(copies, 1, 1)                   # (B, 1, N)
        if frame is None:
            frame = samples.shape[2] * 30 // 16000                                  # 30 fps out of 16 kHz in
        out = self._run(samples, self._identity(id), frame).cpu().numpy()
        if self.config.Data.pose.normalization:
            out = denormalize(out, norm_stats[0], norm_stats[1])
        return out

    def generate(self, wv2_feat, frame):
        '''wv2_feat (B,1,N) samples -> tensor (B,frame,103) with the all-zero identity vector    [smplx_face.py:221-238]'''
        return self._run(wv2_feat, self._identity(None).repeat(wv2_feat.shape[0], 1), frame)
