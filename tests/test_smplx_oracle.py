"""CPU checks of oracle/smplx_oracle.py (the float64 restatement of smplx's published LBS; the package and the licensed model
are absent, so the oracle is UNPINNED against them).  What can be checked here: its building blocks against independent
implementations and invariants of linear blend skinning that hold for any correct implementation."""
import numpy as np
import pytest

from oracle import smplx_oracle as SO


def test_rodrigues_against_scipy():
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(0)
    r = rng.standard_normal((200, 3)) * np.array([0.01, 1.0, 3.0])[rng.integers(0, 3, 200)][:, None]
    np.testing.assert_allclose(SO.batch_rodrigues(r), Rotation.from_rotvec(r).as_matrix(), atol=1e-7)
    z = SO.batch_rodrigues(np.zeros((1, 3)))                       # the package's |r + 1e-8| keeps the zero vector finite
    np.testing.assert_allclose(z[0], np.eye(3), atol=1e-7)


def test_rest_pose_and_rigid_motion_invariants():
    m = SO.synthetic_model(seed=1, V=256)
    m["pose_mean"][:] = 0.0                                        # flat hands: the all-zero row is the rest pose
    betas = np.zeros(m["n_betas"])
    rest = np.zeros((1, 265))
    j0, v0 = SO.smplx_forward(m, betas, rest)
    # rest pose: vertices are the template, the first 55 joints the regressed joints, extras / landmarks follow the mesh
    np.testing.assert_allclose(v0[0], m["v_template"], atol=1e-12)
    np.testing.assert_allclose(j0[0, :55], m["J_regressor"] @ m["v_template"], atol=1e-12)
    np.testing.assert_allclose(j0[0, 55:76], m["v_template"][m["extra_idx"]], atol=1e-12)
    lm = np.einsum("lfi,lf->li", m["v_template"][m["lmk_faces"]], m["lmk_bary"])
    np.testing.assert_allclose(j0[0, 76:], lm, atol=1e-12)
    # a pure global rotation (columns 9:12 of a TalkSHOW row) rotates everything rigidly about the root joint
    from scipy.spatial.transform import Rotation
    rv = np.array([0.3, -1.1, 0.6])
    row = np.zeros((1, 265)); row[0, 9:12] = rv
    j1, v1 = SO.smplx_forward(m, betas, row)
    R, root = Rotation.from_rotvec(rv).as_matrix(), j0[0, 0]
    np.testing.assert_allclose(v1[0], (v0[0] - root) @ R.T + root, atol=1e-7)
    np.testing.assert_allclose(j1[0], (j0[0] - root) @ R.T + root, atol=1e-7)


def test_row_layout_follows_the_reference_call_site():
    """get_j.py:21-30: jaw 0:3, eyes 3:9, global orient 9:12, body 12:75, hands 75:165, expression 165:265."""
    rows = np.arange(265, dtype=np.float64)[None]
    full, expr = SO.full_pose_from_rows(rows)
    assert full.shape == (1, 165) and expr.shape == (1, 100)
    assert list(full[0, :3]) == [9, 10, 11] and list(full[0, 3:6]) == [12, 13, 14]            # global orient, first body joint
    assert list(full[0, 66:69]) == [0, 1, 2] and list(full[0, 69:75]) == [3, 4, 5, 6, 7, 8]     # jaw, eyes
    assert full[0, 75] == 75 and full[0, 164] == 164 and expr[0, 0] == 165
