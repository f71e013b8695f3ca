"""Batched SMPL-X joints / vertices on the GPU: the device counterpart of `get_joints`
(`data_utils/get_j.py:33-50`) and `get_vertices` (`scripts/demo.py:122-152`), which call the third-party `smplx` model one
frame (or 4 sequences) at a time on the CPU in float64.

`SMPLXLayer(model)` takes the model's arrays — a dict (e.g. `np.load('SMPLX_NEUTRAL_2020.npz')` plus the landmark / extra-joint
tables) or a constructed `smplx.SMPLX` module (`from_smplx_module`) — and evaluates whole `(B, T, 265)` TalkSHOW sequences in
one call: `joints(betas, rows) -> (…, 127, 3)`, `vertices(betas, rows) -> (…, V, 3)`.  fp32 on the device; the `smplx`
arithmetic is third-party and the licensed model file is not available here: PARITY UNPINNED (see oracle/smplx_oracle.py).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

# column of a 265-d TalkSHOW row where the axis-angle of SMPL-X joint j starts (get_j.py:21-30): global_orient 9:12,
# body 12:75, jaw 0:3, eyes 3:6 / 6:9, left hand 75:120, right hand 120:165; expression coefficients at 165:265
TALKSHOW_POSE_OFFSETS = np.asarray([9] + [12 + 3 * k for k in range(21)] + [0, 3, 6] + [75 + 3 * k for k in range(15)] +
                                   [120 + 3 * k for k in range(15)], dtype=np.int32)
TALKSHOW_EXPR_OFFSET = 165


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def from_smplx_module(m):
    """Arrays of a constructed `smplx.SMPLX` (use_pca=False) in the layout `SMPLXLayer` takes."""
    g = lambda t: t.detach().cpu().numpy()                                           # noqa: E731
    faces = g(m.faces_tensor)
    return dict(v_template=g(m.v_template), shapedirs=np.concatenate([g(m.shapedirs), g(m.expr_dirs)], -1),
                posedirs=g(m.posedirs), J_regressor=g(m.J_regressor), parents=g(m.parents), lbs_weights=g(m.lbs_weights),
                pose_mean=g(m.pose_mean).reshape(-1), extra_idx=g(m.vertex_joint_selector.extra_joints_idxs),
                lmk_faces=faces[g(m.lmk_faces_idx)], lmk_bary=g(m.lmk_bary_coords), n_betas=int(m.num_betas),
                n_expr=int(m.num_expression_coeffs))


class SMPLXLayer:
    def __init__(self, model, with_vertices=False, device=None, pose_offsets=TALKSHOW_POSE_OFFSETS,
                 expr_offset=TALKSHOW_EXPR_OFFSET):
        if not isinstance(model, dict):
            model = from_smplx_module(model)
        idx = torch.cuda.current_device() if device is None else torch.device(device).index
        self.device = torch.device("cuda", idx if idx is not None else torch.cuda.current_device())
        self.V, self.J = int(model["v_template"].shape[0]), int(np.asarray(model["parents"]).shape[0])
        self.n_betas, self.n_expr = int(model["n_betas"]), int(model["n_expr"])
        self.expr_offset = int(expr_offset)
        S = self.n_betas + self.n_expr
        sd = np.asarray(model["shapedirs"]).reshape(self.V, 3, -1)
        if sd.shape[2] != S:
            raise ValueError(f"shapedirs has {sd.shape[2]} components, n_betas + n_expr = {S}")
        parents = _i32(model["parents"]).copy()
        parents[0] = -1                                                                # the package stores -1 / 2**32-1 for the root
        extra, lmk_f, bary = _i32(model["extra_idx"]), _i32(model["lmk_faces"]).reshape(-1, 3), _f32(model["lmk_bary"]).reshape(-1, 3)
        keep = [_f32(model["v_template"]), _f32(sd), _f32(np.asarray(model["posedirs"]).reshape((self.J - 1) * 9, self.V * 3)),
                _f32(model["J_regressor"]), parents, _f32(model["lbs_weights"]), _f32(np.asarray(model["pose_mean"]).reshape(-1)),
                _i32(pose_offsets), extra, lmk_f, bary]
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))                          # noqa: E731
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().ts_smplx_create(
                _lib.context(self.device.index), self.V, self.J, self.n_betas, self.n_expr, _lib.fptr(keep[0]), _lib.fptr(keep[1]),
                _lib.fptr(keep[2]), _lib.fptr(keep[3]), ip(keep[4]), _lib.fptr(keep[5]), _lib.fptr(keep[6]), ip(keep[7]),
                int(extra.shape[0]), ip(keep[8]), int(lmk_f.shape[0]), ip(keep[9]), _lib.fptr(keep[10]), int(bool(with_vertices)),
                C.byref(h)))
        self._h = h
        self.with_vertices = bool(with_vertices)
        self.num_joints = int(_lib.load().ts_smplx_num_joints(h))

    def __del__(self):
        try:
            _lib.load().ts_smplx_destroy(self._h)
        except Exception:
            pass

    def _run(self, betas, rows, want_verts):
        rows = torch.as_tensor(rows, dtype=torch.float32, device=self.device)
        lead = rows.shape[:-1]
        flat = rows.reshape(-1, rows.shape[-1]).contiguous()
        N = flat.shape[0]
        betas = torch.as_tensor(betas, dtype=torch.float32, device=self.device).reshape(-1, self.n_betas).contiguous()
        if betas.shape[0] not in (1, N):
            raise ValueError(f"betas must be ({self.n_betas},) or one row per pose row, got {tuple(betas.shape)}")
        joints = torch.empty((N, self.num_joints, 3), dtype=torch.float32, device=self.device)
        verts = torch.empty((N, self.V, 3), dtype=torch.float32, device=self.device) if want_verts else None
        _lib.check(_lib.load().ts_smplx_forward(self._h, _lib.dptr(betas), int(betas.shape[0] == N and N > 1), _lib.dptr(flat),
                                                flat.shape[1], self.expr_offset, N, _lib.dptr(joints), _lib.dptr(verts),
                                                _lib.stream_ptr()))
        joints = joints.reshape(*lead, self.num_joints, 3)
        return (joints, verts.reshape(*lead, self.V, 3)) if want_verts else joints

    def joints(self, betas, rows):
        """`get_joints(smplx_model, betas, pred)` (`get_j.py:33-50`): rows (..., 265) -> (..., num_joints, 3)."""
        return self._run(betas, rows, False)

    def vertices(self, betas, rows):
        """`get_vertices` (`demo.py:122-152`): rows (..., 265) -> (joints, vertices (..., V, 3))."""
        return self._run(betas, rows, True)
