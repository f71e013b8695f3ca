"""CPU oracle of the evaluation metrics: numpy restatement of the reference's formulae.

TEST INFRASTRUCTURE ONLY (same rule as talkshow_oracle.py): only tests/ imports this.  Pinned by
tests/test_eval_oracle_golden.py against values computed by the reference's own `evaluation/FGD.py`,
`evaluation/metrics.py` and the `body_loss` function of `scripts/test_body.py` (tests/golden/make_golden.py, case
`eval_metrics`).
"""
import numpy as np
from scipy import linalg


def frechet_scores(gen_rows, real_rows):
    """`EmbeddingSpaceEvaluator.get_scores` (`evaluation/FGD.py:131-160`): (N_g, D), (N_r, D) stacked feature rows ->
    (frechet_dist, feat_dist)."""
    mu_g, mu_r = np.mean(gen_rows, axis=0), np.mean(real_rows, axis=0)
    s_g, s_r = np.cov(gen_rows, rowvar=False), np.cov(real_rows, rowvar=False)
    diff = mu_g - mu_r
    covmean = linalg.sqrtm(s_g.dot(s_r))
    if isinstance(covmean, tuple):
        covmean = covmean[0]
    if np.iscomplexobj(covmean):
        covmean = covmean.real
    fgd = diff.dot(diff) + np.trace(s_g) + np.trace(s_r) - 2 * np.trace(covmean)
    feat_dist = np.mean([np.sum(np.abs(real_rows[i] - gen_rows[i])) for i in range(real_rows.shape[0])])
    return float(fgd), float(feat_dist)


def body_loss(gt, prs, lvd_joints=22):
    """`body_loss` (`scripts/test_body.py:98-110`): gt (T,J,3), prs (B,T,J,3) float -> dict LVD / error / diverse."""
    gt, prs = gt.astype(np.float64), prs.astype(np.float64)
    g, p = gt[:, :lvd_joints], prs[:, :, :lvd_joints]
    L = min(g.shape[0], p.shape[1])
    gv = np.linalg.norm(g[1:L] - g[:L - 1], axis=-1)                      # (T-1, J)
    pv = np.linalg.norm(p[:, 1:L] - p[:, :L - 1], axis=-1)                # (B, T-1, J)
    lvd = (np.abs(pv - gv).sum(-1) / gv.shape[0]).sum(-1).mean()          # metrics.py:79-84, weight=False
    err = np.linalg.norm(gt[None] - prs, axis=-1).sum(-1).mean()
    div = np.linalg.norm(prs.var(axis=0, ddof=1), axis=-1).sum(-1).mean()
    return {"LVD": float(lvd), "error": float(err), "diverse": float(div)}


def diversity(kps):
    """`metrics.diversity` (`evaluation/metrics.py:96-109`)."""
    d = [np.mean(np.abs(kps[i].astype(np.float64) - kps[j])) for i in range(kps.shape[0]) for j in range(i + 1, kps.shape[0])]
    return float(np.mean(d))
