"""`nets/base.py:5-88` of the reference, inference side.

The reference base class also builds Adam optimisers; training is out of scope here, so
`generator_optimizer` / `discriminator_optimizer` exist but are None, and `state_dict()` reports their entries as
None — checkpoints written by the reference load unchanged (their optimiser entries are ignored).
"""


class TrainWrapperBaseClass():
    def __init__(self, args, config) -> None:
        self.init_optimizer()

    def init_optimizer(self) -> None:
        self.generator_optimizer = None
        self.discriminator_optimizer = None

    def __call__(self, bat):
        raise NotImplementedError("training step: out of scope of the MI355X inference path")

    def get_loss(self, **kwargs):
        raise NotImplementedError

    def state_dict(self):
        # same four entries as a reference checkpoint; only the generator carries data on the inference path
        out = dict.fromkeys(('generator', 'generator_optim', 'discriminator', 'discriminator_optim'))
        out['generator'] = self.generator.state_dict()
        return out

    def parameters(self):
        return self.generator.parameters()

    def load_state_dict(self, state_dict):
        # a whole checkpoint ({'generator': ..., optimiser entries ...}) or the generator's own state_dict
        self.generator.load_state_dict(state_dict.get('generator', state_dict))

    def infer_on_audio(self, aud_fn, initial_pose=None, norm_stats=None, **kwargs):
        raise NotImplementedError

    def init_params(self):
        """Layout of the modelled pose vector (`nets/base.py:58-88` of the reference).

        The body wrappers model no jaw / eye / global-orientation dims (they are fixed or come from the face generator):
        13 body joints (21 SMPL-X body joints minus the 8 lower-body ones) and 2 x 15 hand joints, 3 values each in
        axis-angle or 6 in the 6-D rotation form, plus 100 expression coefficients when `expression` is set.
        Sets `dim_list` (start offsets of jaw, eyes, body, hands, face), `full_dim`, `pose` (joint count) and
        `each_dim` = [jaw, eyes + body, hands, face] widths.
        """
        per_joint = 6 if self.config.Data.pose.convert_to_6d else 3
        widths = {
            'jaw': 0,
            'eyes': 0,
            'body': (21 - 8) * per_joint,
            'hands': 2 * 15 * per_joint,
            'face': 100 if self.expression else 0,
        }
        offsets, at = [], 0
        for part in ('jaw', 'eyes', 'body', 'hands', 'face'):
            offsets.append(at)
            at += widths[part]
        self.dim_list = offsets
        self.full_dim = widths['jaw'] + widths['eyes'] + widths['body'] + widths['hands']
        self.pose = self.full_dim // per_joint
        self.each_dim = [widths['jaw'], widths['eyes'] + widths['body'], widths['hands'], widths['face']]


def resolve_device(gpu):
    """`torch.device(self.args.gpu)` of the reference wrappers: an int is a HIP device index on ROCm."""
    import torch
    if isinstance(gpu, int):
        return torch.device("cuda", gpu)
    return torch.device(gpu)
