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
        model_state = {
            'generator': self.generator.state_dict(),
            'generator_optim': None,
            'discriminator': None,
            'discriminator_optim': None,
        }
        return model_state

    def parameters(self):
        return self.generator.parameters()

    def load_state_dict(self, state_dict):
        if 'generator' in state_dict:
            self.generator.load_state_dict(state_dict['generator'])
        else:
            self.generator.load_state_dict(state_dict)

    def infer_on_audio(self, aud_fn, initial_pose=None, norm_stats=None, **kwargs):
        raise NotImplementedError

    def init_params(self):
        # nets/base.py:58-88: body wrappers model 39 body + 90 hand dims (jaw / eyes / global orient excluded)
        scale = 2 if self.config.Data.pose.convert_to_6d else 1
        global_orient = round(0 * scale)
        leye_pose = reye_pose = round(0 * scale)
        jaw_pose = round(0 * scale)
        body_pose = round((63 - 24) * scale)
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
        self.full_dim = jaw_dim + eye_dim + body_dim + hand_dim
        self.pose = int(self.full_dim / round(3 * scale))
        self.each_dim = [jaw_dim, eye_dim + body_dim, hand_dim, face_dim]


def resolve_device(gpu):
    """`torch.device(self.args.gpu)` of the reference wrappers: an int is a HIP device index on ROCm."""
    import torch
    if isinstance(gpu, int):
        return torch.device("cuda", gpu)
    return torch.device(gpu)
