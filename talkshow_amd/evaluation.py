"""Evaluation on the device: the reductions behind the reference's metrics, through the C ABI.

The reference computes FGD / feature distance (`evaluation/FGD.py:131-160`), LVD / L2 error / variance
(`scripts/test_body.py:98-110`, `evaluation/metrics.py:27-84`) and the pairwise diversity (`metrics.py:96-109`) on the CPU
in numpy / torch after the poses have been copied back.  Here the sums over samples, frames and joints run on the GPU
in float64 (`csrc/eval.hip`), next to the generated poses; what is left on the host is O(D^2) work on a 64 x 64 matrix
(the matrix square root of the Frechet distance, scipy) and scalar divisions.  No CPU path: these raise without a HIP device.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _dev(x, device=None):
    if device is None:
        device = x.device if torch.is_tensor(x) and x.is_cuda else torch.device("cuda", torch.cuda.current_device())
    return torch.as_tensor(x, dtype=torch.float32, device=device).contiguous()


def _ctx(t):
    return _lib.context(t.device.index)


class FeatureStats:
    """Running count / sum / sum of outer products of feature rows (float64 on the device): what `np.mean(axis=0)` and
    `np.cov(rowvar=False)` of `evaluation/FGD.py:133-136` need, accumulated batch by batch without keeping the rows."""

    def __init__(self, dim=64, device=None):
        self.dim = int(dim)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.n = 0
        self.acc = torch.zeros(self.dim + self.dim * self.dim, dtype=torch.float64, device=self.device)
        self._tmp = torch.empty_like(self.acc)

    def push(self, feats):
        f = _dev(feats, self.device).reshape(-1, self.dim)
        if f.shape[0] == 0:
            return
        _lib.check(_lib.load().ts_eval_feat_stats(_ctx(f), _lib.dptr(f), f.shape[0], self.dim, _lib.dptr(self._tmp),
                                                  _lib.stream_ptr()))
        self.acc += self._tmp
        self.n += int(f.shape[0])

    def mean_cov(self):
        """(mu (D,), sigma (D,D)) as float64 numpy: sigma is the unbiased sample covariance like np.cov."""
        if self.n < 2:
            raise ValueError("covariance needs at least two feature rows")
        a = self.acc.cpu().numpy()
        mu = a[:self.dim] / self.n
        outer = a[self.dim:].reshape(self.dim, self.dim)
        sigma = (outer - self.n * np.outer(mu, mu)) / (self.n - 1)
        return mu, sigma


def frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    """d^2 = |mu1 - mu2|^2 + Tr(S1 + S2 - 2 sqrt(S1 S2)) between two Gaussians — the published pytorch-fid formula the
    reference uses (`evaluation/FGD.py:162-211`), incl. its handling of a singular product (eps on the diagonals) and of a
    numerically complex square root (imaginary part dropped if negligible, else ValueError)."""
    from scipy import linalg
    mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
    sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
    if mu1.shape != mu2.shape or sigma1.shape != sigma2.shape:
        raise AssertionError("the two Gaussians must have the same dimension")
    root = linalg.sqrtm(sigma1.dot(sigma2))
    if isinstance(root, tuple):
        root = root[0]
    if not np.isfinite(root).all():
        jitter = np.eye(sigma1.shape[0]) * eps
        root = linalg.sqrtm((sigma1 + jitter).dot(sigma2 + jitter))
    if np.iscomplexobj(root):
        if not np.allclose(np.diagonal(root).imag, 0, atol=1e-3):
            raise ValueError('Imaginary component {}'.format(np.max(np.abs(root.imag))))
        root = root.real
    d = mu1 - mu2
    return d.dot(d) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(root)


def l1_mean_per_row(a, b, rows=None):
    """`feat_dist` of `evaluation/FGD.py:153-158`: mean over rows of sum_d |a - b| for (n, D) arrays."""
    a, b = _dev(a), _dev(b)
    a, b = a.reshape(-1, a.shape[-1]), b.reshape(-1, b.shape[-1])
    n = a.shape[0] if rows is None else rows
    a, b = a[:n].contiguous(), b[:n].contiguous()
    out = torch.empty(1, dtype=torch.float64, device=a.device)
    _lib.check(_lib.load().ts_eval_l1_total(_ctx(a), _lib.dptr(a), _lib.dptr(b), a.numel(), _lib.dptr(out), _lib.stream_ptr()))
    return float(out.item()) / n


def body_loss(gt, prs, lvd_joints=22):
    """`body_loss` of `scripts/test_body.py:98-110`: gt (T,J,3) joints, prs (B,T,J,3) -> {'LVD', 'error', 'diverse'}.

    LVD = `LVD(gt[:, :22], prs[:, :, :22], symmetrical=False, weight=False)` (`metrics.py:27-36,73-84`): the sequences are cut
    to the shorter length, velocity magnitudes compared, summed over joints, averaged over time and samples."""
    gt, prs = _dev(gt), _dev(prs)
    if prs.ndim == 3:
        prs = prs[None]
    B, T, J, _ = prs.shape
    Tl = min(int(gt.shape[0]), T)
    if gt.shape[0] < T:
        raise ValueError("gt must cover the generated frames for the L2 error (test_body.py:103 broadcasts gt over prs)")
    gt = gt[:T].contiguous()
    out = torch.empty(3, dtype=torch.float64, device=prs.device)
    _lib.check(_lib.load().ts_eval_body_loss(_ctx(prs), _lib.dptr(gt), _lib.dptr(prs), B, T, J, min(lvd_joints, J), Tl,
                                             _lib.dptr(out), _lib.stream_ptr()))
    lvd, err, var = out.cpu().tolist()
    return {'LVD': lvd / (B * (Tl - 1)), 'error': err / (B * T), 'diverse': var / T}


def lvd(gt_kps, pr_kps):
    """`evaluation.metrics.LVD(gt, pr)` for the non-symmetrical, unweighted case (`metrics.py:79-94`): gt (T,J,3),
    pr (T,J,3) or (B,T,J,3)."""
    gt, pr = _dev(gt_kps).squeeze(), _dev(pr_kps).squeeze()
    if pr.ndim == 3:
        pr = pr[None]
    B, T, J, _ = pr.shape
    Tl = min(int(gt.shape[0]), T)
    gt = gt[:Tl].contiguous()
    pr = pr[:, :Tl].contiguous()
    out = torch.empty(3, dtype=torch.float64, device=pr.device)
    _lib.check(_lib.load().ts_eval_body_loss(_ctx(pr), _lib.dptr(gt), _lib.dptr(pr), B, Tl, J, J, Tl, _lib.dptr(out),
                                             _lib.stream_ptr()))
    return float(out[0].item()) / (B * (Tl - 1))


def diversity(kps):
    """`evaluation.metrics.diversity` (`metrics.py:96-109`): kps (bs, seq, dim) -> mean over pairs of mean |seq_i - seq_j|."""
    k = _dev(kps)
    bs = k.shape[0]
    if bs < 2:
        return float("nan")                     # np.mean of an empty list
    k = k.reshape(bs, -1).contiguous()
    out = torch.empty(1, dtype=torch.float64, device=k.device)
    _lib.check(_lib.load().ts_eval_diversity(_ctx(k), _lib.dptr(k), bs, k.shape[1], _lib.dptr(out), _lib.stream_ptr()))
    return float(out.item()) / (k.shape[1] * (bs * (bs - 1) // 2))


def motion_angle_series(joints):
    """Elbow / wrist bend-angle series the beat metrics are built on (`evaluation/FGD.py:67-76,88-94`): joints (T,>=22,3).
    The reference overwrites joints 15..20 with 16..21 in place, forms the bone vectors j[15:21] - j[13:19], and takes
    acos of the clamped inner product of bones two apart (NOT normalised — as written), divided by pi."""
    j = torch.as_tensor(joints, dtype=torch.float32).clone()
    j[:, 15:21] = j[:, 16:22].clone()
    vec = j[:, 15:21] - j[:, 13:19]
    inner = torch.clamp((vec[:, 2:] * vec[:, :-2]).sum(-1), -1, 1)
    return torch.acos(inner) / np.pi                                             # (T, 4)
