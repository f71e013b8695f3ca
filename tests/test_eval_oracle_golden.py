"""Pin the evaluation oracle (oracle/eval_oracle.py) and the host-side pieces of the evaluation drop-in (Frechet formula,
beat metrics) to values produced by the reference's own evaluation code (tests/golden/make_golden.py, `eval_metrics`).  CPU."""
import numpy as np
import torch

from oracle import eval_oracle as EO


def _rows(g):
    real, gen, at_r, at_g = [], [], 0, 0
    for H in g["clip_rows"]:
        H = int(H)
        real.append(g["real"][at_r:at_r + H * 64].reshape(H, 64)); at_r += H * 64
        gen.append(g["gen"][at_g:at_g + 2 * H * 64].reshape(2 * H, 64)); at_g += 2 * H * 64
    return np.vstack(real), np.vstack(gen)


def test_frechet_and_feature_distance(golden):
    g = golden("eval_metrics")
    real, gen = _rows(g)
    fgd, fd = EO.frechet_scores(gen, real)
    np.testing.assert_allclose(fgd, g["fgd"], rtol=1e-9)
    np.testing.assert_allclose(fd, g["feat_dist"], rtol=1e-6)
    # the drop-in's Frechet formula on float64 moments (what the device statistics feed)
    from talkshow_amd.evaluation import frechet_distance
    x, y = gen.astype(np.float64), real.astype(np.float64)
    got = frechet_distance(x.mean(0), np.cov(x, rowvar=False), y.mean(0), np.cov(y, rowvar=False))
    np.testing.assert_allclose(got, g["fgd"], rtol=1e-5)           # reference: float32 mean, float64 covariance


def test_body_loss_and_diversity(golden):
    g = golden("eval_metrics")
    bl = EO.body_loss(g["gt_joints"], g["pr_joints"])
    np.testing.assert_allclose(bl["LVD"], g["lvd"], rtol=2e-5)      # reference runs in float32
    np.testing.assert_allclose(bl["error"], g["error"], rtol=2e-5)
    np.testing.assert_allclose(bl["diverse"], g["diverse"], rtol=2e-5)
    np.testing.assert_allclose(EO.body_loss(g["gt_joints"], g["pr_joints"][:1], lvd_joints=55)["LVD"], g["lvd_single"], rtol=2e-5)
    np.testing.assert_allclose(EO.diversity(g["kps"]), g["diversity"], rtol=1e-5)


def test_beat_metrics_of_the_drop_in(golden):
    """evaluation.FGD.EmbeddingSpaceEvaluator.get_MAAC / get_BCscore (host side, vectorised) vs the reference's loops."""
    from evaluation.FGD import EmbeddingSpaceEvaluator
    g = golden("eval_metrics")
    ev = EmbeddingSpaceEvaluator(ae=None)
    for k in range(g["joints_real"].shape[0]):
        ev.push_joints(torch.from_numpy(g["joints_gen"][k][None]), torch.from_numpy(g["joints_real"][k]))
        ev.push_aud(torch.from_numpy(g["beats"][k]))
    np.testing.assert_allclose(ev.get_MAAC().numpy(), g["maac"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(ev.get_BCscore(), g["bc"], rtol=1e-6)


def test_symmetric_lvd_formula_vs_reference_value():
    """`metrics.LVD(symmetrical=True)` — the pure-torch branch of this repo's evaluation package against the value the
    reference's own function returned (tests/golden/make_golden.py --only lvd_symmetric), its `~mask.long()` quirk included."""
    import os
    import torch
    from talkshow_amd.evaluation import lvd_symmetric
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "lvd_symmetric.npz"))
    got = float(lvd_symmetric(torch.from_numpy(g["gt_joints"]), torch.from_numpy(g["pr_joints"])))
    np.testing.assert_allclose(got, float(g["lvd_sym"]), rtol=2e-6)
    assert abs(float(g["lvd_sym"]) - float(g["lvd_plain"])) > 1e-3          # the branch does something
    # the reference gathers joints 0..21 (`rearrange`): rows that carry more joints give the same value, fewer an IndexError
    gt55 = np.concatenate([g["gt_joints"], np.ones((g["gt_joints"].shape[0], 33, 3), np.float32)], 1)
    pr55 = np.concatenate([g["pr_joints"], 7 * np.ones(g["pr_joints"].shape[:2] + (33, 3), np.float32)], 2)
    assert float(lvd_symmetric(torch.from_numpy(gt55), torch.from_numpy(pr55))) == got
    import pytest
    with pytest.raises(IndexError):
        lvd_symmetric(torch.from_numpy(g["gt_joints"][:, :20]), torch.from_numpy(g["pr_joints"][:, :, :20]))
