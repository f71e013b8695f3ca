"""Index bookkeeping between the 165-d SMPL-X axis-angle vector and the 129 modelled dims.

Restates `data_utils/lower_body.py:44-56` (`c_index_3d`): of the 165 pose dims, the jaw/eyes (0-8), global
orientation + lower body joints (9-17, 21-26, 30-35) and dims 45-50 are held fixed; the remaining 129 are what the
body (first 39) and hand (last 90) VQ-VAEs model.  The reference's off-by-six between this list and `part2full`
 is caller-side behaviour and is deliberately not "fixed" here.
"""
import numpy as np

FIX_INDEX_3D = list(range(0, 18)) + list(range(21, 27)) + list(range(30, 36)) + list(range(45, 51))
_all = np.ones(165)
_all[FIX_INDEX_3D] = 0
c_index_3d = np.asarray([i for i, v in enumerate(_all) if v == 1])
assert c_index_3d.shape == (129,)
# the same selection on 6-D rotation rows (`lower_body.py:58-65`): dim i of the axis-angle layout owns dims 2 i, 2 i + 1
c_index_6d = np.asarray([j for i in c_index_3d for j in (2 * i, 2 * i + 1)])
assert c_index_6d.shape == (258,)


# The fixed lower-body block `part2full` inserts (data_utils/lower_body.py:4-8, values quoted from there; 33 = 15 + 6 + 6 + 6)
LOWER_POSE = np.asarray(
    [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 3.0747, -0.0158, -0.0152, -1.1826512813568115, 0.23866955935955048,
     0.15146760642528534, -1.2604516744613647, -0.3160211145877838, -0.1603458970785141, 1.1654603481292725, 0.0, 0.0,
     1.2521806955337524, 0.041598282754421234, -0.06312154978513718] + [0.0] * 12, dtype=np.float32)
assert LOWER_POSE.shape == (33,)


def lower_pose_block(stand=False):
    """lower_body.py:69-75: `stand=True` zeroes the block except the global orientation at [6:9]."""
    if not stand:
        return LOWER_POSE.copy()
    lp = np.zeros(33, np.float32)
    lp[6:9] = LOWER_POSE[6:9]
    return lp


def assemble_full(body_poses, face_params, stand=False):
    """(B,Tb,129) body+hand poses and (B,Tf,103) jaw+expression -> (B,Tf,265) SMPL-X parameter rows, on the GPU.

    The caller-side tail of scripts/demo.py:207-229 (align the body to the face length, concat jaw | body | expression,
    `part2full`) as one HIP launch (`ts_assemble_full`); inputs may be numpy arrays or tensors, the result is a CUDA tensor.
    """
    import torch
    from . import _lib
    lib = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    body = torch.as_tensor(body_poses, dtype=torch.float32, device=dev).contiguous()
    face = torch.as_tensor(face_params, dtype=torch.float32, device=dev).contiguous()
    if body.dim() != 3 or face.dim() != 3 or body.shape[2] != 129 or face.shape[2] != 103 or body.shape[0] != face.shape[0]:
        raise ValueError(f"assemble_full: expected (B,Tb,129) and (B,Tf,103), got {tuple(body.shape)} and {tuple(face.shape)}")
    B, Tb, _ = body.shape
    Tf = face.shape[1]
    out = torch.empty((B, Tf, 265), dtype=torch.float32, device=dev)
    lp = lower_pose_block(stand)
    _lib.check(lib.ts_assemble_full(_lib.context(dev.index), _lib.dptr(body), Tb, _lib.dptr(face), Tf, B, _lib.fptr(lp),
                                    _lib.dptr(out), _lib.stream_ptr()))
    return out
