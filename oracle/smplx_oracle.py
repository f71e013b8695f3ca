"""CPU oracle of the SMPL-X forward used behind the hot path (`scripts/demo.py:122-152`, `data_utils/get_j.py:20-50`).

TEST INFRASTRUCTURE ONLY (same rule as talkshow_oracle.py).

The arithmetic is THIRD-PARTY: package `smplx`, pinned `smplx~=0.1.28` in the reference's requirements.txt:5, absent from
/root/reference and not installed here; the licensed model file (`SMPLX_NEUTRAL_2020.npz`) is absent too.
PARITY UNPINNED: this file restates the package's published algorithm — `smplx.lbs.lbs` (blend shapes -> joint regression ->
Rodrigues -> pose blend shapes -> rigid transform chain -> linear blend skinning), `vertex_joint_selector`,
`vertices2landmarks` and the parameter layout of `SMPLX.forward` for the way the reference constructs the model
(`scripts/test_body.py:225-246`: use_pca=False, flat_hand_mean=False, num_betas=300, num_expression_coeffs=100, no transl,
no face contour, float64) — in float64 numpy, on SYNTHETIC model parameters of the real model's shapes.  The reference's
call sites anchor the input layout: a 265-d row is [jaw 0:3 | leye 3:6 | reye 6:9 | global_orient 9:12 | body 12:75 |
left hand 75:120 | right hand 120:165 | expression 165:265] (`get_j.py:21-30`).
"""
import numpy as np

# kinematic tree of SMPL-X (55 joints): 22 body joints, jaw, two eyes, 2 x 15 finger joints
SMPLX_PARENTS = np.asarray(
    [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 15, 15, 15,
     20, 25, 26, 20, 28, 29, 20, 31, 32, 20, 34, 35, 20, 37, 38,
     21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50, 21, 52, 53], dtype=np.int64)


def synthetic_model(seed=0, V=512, n_betas=300, n_expr=100, n_extra=21, n_lmk=51, max_bones=4):
    """Random model parameters with the real SMPL-X's structure and value ranges (metres): a point cloud around a stick
    figure, small shape / pose blend shapes, a sparse non-negative joint regressor with rows summing to 1, skinning weights
    with <= `max_bones` bones per vertex summing to 1, a hand-pose mean, extra-joint vertex ids and landmark triangles."""
    rng = np.random.default_rng([seed, V])
    J = SMPLX_PARENTS.shape[0]
    # rest joints: children offset from their parents
    jpos = np.zeros((J, 3))
    for j in range(1, J):
        jpos[j] = jpos[SMPLX_PARENTS[j]] + rng.normal(0, 0.08 if j < 25 else 0.02, 3)
    owner = rng.integers(0, J, V)
    v_template = jpos[owner] + rng.normal(0, 0.03, (V, 3))
    shapedirs = rng.normal(0, 0.004, (V, 3, n_betas + n_expr))
    posedirs = rng.normal(0, 0.002, ((J - 1) * 9, V * 3))
    J_regressor = np.zeros((J, V))
    for j in range(J):
        idx = rng.choice(V, 12, replace=False)
        w = rng.random(12)
        J_regressor[j, idx] = w / w.sum()
    lbs_weights = np.zeros((V, J))
    for v in range(V):
        k = rng.integers(1, max_bones + 1)
        bones = np.unique(np.concatenate([[owner[v]], rng.integers(0, J, k - 1)]))
        w = rng.random(bones.size) + 0.1
        lbs_weights[v, bones] = w / w.sum()
    pose_mean = np.zeros(J * 3)
    pose_mean[25 * 3:] = rng.normal(0, 0.15, 30 * 3)                      # left / right hand means (flat_hand_mean=False)
    extra_idx = rng.choice(V, n_extra, replace=False)
    lmk_faces = np.stack([rng.choice(V, 3, replace=False) for _ in range(n_lmk)])   # faces_tensor[lmk_faces_idx]
    bary = rng.random((n_lmk, 3))
    bary /= bary.sum(1, keepdims=True)
    return dict(v_template=v_template, shapedirs=shapedirs, posedirs=posedirs, J_regressor=J_regressor,
                parents=SMPLX_PARENTS.copy(), lbs_weights=lbs_weights, pose_mean=pose_mean, extra_idx=extra_idx,
                lmk_faces=lmk_faces, lmk_bary=bary, n_betas=n_betas, n_expr=n_expr)


def batch_rodrigues(rot_vecs, epsilon=1e-8):
    """smplx.lbs.batch_rodrigues: (N,3) axis-angle -> (N,3,3); the angle is |r + 1e-8| as the package computes it."""
    angle = np.linalg.norm(rot_vecs + epsilon, axis=1, keepdims=True)
    d = rot_vecs / angle
    c, s = np.cos(angle)[:, None], np.sin(angle)[:, None]
    K = np.zeros((rot_vecs.shape[0], 3, 3))
    K[:, 0, 1], K[:, 0, 2] = -d[:, 2], d[:, 1]
    K[:, 1, 0], K[:, 1, 2] = d[:, 2], -d[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -d[:, 1], d[:, 0]
    return np.eye(3)[None] + s * K + (1 - c) * (K @ K)


def full_pose_from_rows(rows):
    """TalkSHOW 265-d rows -> SMPL-X full pose (N,165) in the package's joint order [global_orient | body 21 | jaw | leye |
    reye | left hand 15 | right hand 15] (`get_j.py:21-30` + the concatenation in `SMPLX.forward`) and expression (N,100)."""
    full = np.concatenate([rows[:, 9:12], rows[:, 12:75], rows[:, 0:3], rows[:, 3:6], rows[:, 6:9], rows[:, 75:120],
                           rows[:, 120:165]], axis=1)
    return full, rows[:, 165:265]


def smplx_forward(model, betas, rows):
    """SMPLX.forward(...)['joints'] and ['vertices'] for TalkSHOW rows: betas (n_betas,) or (N,n_betas), rows (N,265) ->
    joints (N, 55 + n_extra + n_lmk, 3), vertices (N,V,3), float64."""
    rows = np.asarray(rows, np.float64)
    N = rows.shape[0]
    full_pose, expr = full_pose_from_rows(rows)
    full_pose = full_pose + model["pose_mean"][None]
    betas = np.broadcast_to(np.asarray(betas, np.float64).reshape(-1, model["n_betas"]), (N, model["n_betas"]))
    shape = np.concatenate([betas, expr[:, :model["n_expr"]]], axis=1)                       # (N,S)
    v_shaped = model["v_template"][None] + np.einsum("bl,mkl->bmk", shape, model["shapedirs"])
    J = np.einsum("bik,ji->bjk", v_shaped, model["J_regressor"])                            # (N,55,3)
    nj = J.shape[1]
    R = batch_rodrigues(full_pose.reshape(-1, 3)).reshape(N, nj, 3, 3)
    pose_feature = (R[:, 1:] - np.eye(3)).reshape(N, -1)
    v_posed = v_shaped + (pose_feature @ model["posedirs"]).reshape(N, -1, 3)
    # batch_rigid_transform
    parents = model["parents"]
    rel = J.copy()
    rel[:, 1:] -= J[:, parents[1:]]
    T = np.zeros((N, nj, 4, 4))
    T[:, :, :3, :3] = R
    T[:, :, :3, 3] = rel
    T[:, :, 3, 3] = 1
    G = [T[:, 0]]
    for j in range(1, nj):
        G.append(G[parents[j]] @ T[:, j])
    G = np.stack(G, 1)
    posed_joints = G[:, :, :3, 3]
    Jh = np.concatenate([J, np.zeros((N, nj, 1))], -1)[..., None]                             # F.pad(joints, [0,0,0,1])
    A = G.copy()
    A[:, :, :, 3:] -= G @ Jh                                                                  # rel_transforms
    Tv = np.einsum("vj,bjrc->bvrc", model["lbs_weights"], A)
    vh = np.concatenate([v_posed, np.ones((N, v_posed.shape[1], 1))], -1)
    verts = np.einsum("bvrc,bvc->bvr", Tv, vh)[..., :3]
    extra = verts[:, model["extra_idx"]]                                                      # vertex_joint_selector
    tri = verts[:, model["lmk_faces"]]                                                        # (N,L,3,3)
    lmk = np.einsum("blfi,lf->bli", tri, model["lmk_bary"])                                   # vertices2landmarks
    return np.concatenate([posed_joints, extra, lmk], axis=1), verts
