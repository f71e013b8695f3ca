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
    if a.shape[0] < n or b.shape[0] < n or a.shape[1] != b.shape[1]:
        raise IndexError(f"feat_dist needs {n} rows of equal width in both arrays, got {tuple(a.shape)} and {tuple(b.shape)}")
    out = torch.empty(1, dtype=torch.float64, device=a.device)
    _lib.check(_lib.load().ts_eval_l1_total(_ctx(a), _lib.dptr(a), _lib.dptr(b), a.numel(), _lib.dptr(out), _lib.stream_ptr()))
    return np.float64(out.item() / n)            # numpy scalar like the reference's np.mean: callers do feat_dist.item()


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


# SMPL-X body joints 0..21: the six on the spine (pelvis, spine1..3, neck, head = 0, 3, 6, 9, 12, 15) have no mirror image, the
# others come as left / right pairs (1, 2), (4, 5), (7, 8), (10, 11), (13, 14), (16, 17), (18, 19), (20, 21)
# (`data_utils/lower_body.py:136-141`: `rearrange` is the identity on these 22, `symmetry` flags the paired ones)
_CENTRAL = [0, 3, 6, 9, 12, 15]
_LEFT = [1, 4, 7, 10, 13, 16, 18, 20]
_RIGHT = [2, 5, 8, 11, 14, 17, 19, 21]


def lvd_symmetric(gt, pr):
    """The symmetrical=True branch of `Batch_LVD` (`metrics.py:36-65`) on tensors of any device: gt (T,J,3), pr (B,T,J,3), equal T,
    J >= 22: the reference gathers joints 0..21 (`rearrange`), whatever else the rows carry."""
    if gt.shape[1] < 22 or pr.shape[2] < 22:
        raise IndexError("symmetrical LVD gathers the 22 SMPL-X body joints (indices 0..21)")
    gt, pr = gt[:, :22], pr[:, :, :22]
    gv = (gt[1:] - gt[:-1]).norm(p=2, dim=-1)                       # (T-1, 22)
    pv = (pr[:, 1:] - pr[:, :-1]).norm(p=2, dim=-1)                 # (B, T-1, 22)
    g_side = (gv[:, _LEFT].sum(-1) > gv[:, _RIGHT].sum(-1)).to(gv.dtype)[:, None]
    g_vel = torch.cat([gv[:, _CENTRAL], gv[:, _LEFT] * g_side + gv[:, _RIGHT] * (1 - g_side)], dim=1)
    p_side = (pv[..., _LEFT].sum(-1) > pv[..., _RIGHT].sum(-1)).to(torch.int64)[..., None]
    p_vel = torch.cat([pv[..., _CENTRAL], pv[..., _LEFT] * p_side + pv[..., _RIGHT] * (~p_side)], dim=2)   # ~ on an integer mask, as written
    return ((p_vel - g_vel).abs().sum(-1) / g_vel.shape[0]).sum(-1).mean().to(torch.float32)


def lvd(gt_kps, pr_kps, symmetrical=False):
    """`evaluation.metrics.LVD(gt, pr, symmetrical, weight=False)` (`metrics.py:27-94`): gt (T,J,3), pr (T,J,3) or (B,T,J,3)
    -> 0-d float32 tensor on pr's device (the reference returns a tensor: its callers accumulate it and call `.item()`).

    symmetrical=False: sum over joints of |velocity magnitude difference|, averaged over frames and samples — on the device
    (`ts_eval_body_loss`).  symmetrical=True (only for a 4-D `pr`, as in the reference; joints 0..21, `metrics.py:36-65`): of every mirrored joint pair only one side
    counts per frame — for gt the side whose pairs moved more in that frame; for pr the reference combines the sides as
    `left * m + right * ~m.long()` with m in {0, 1}, i.e. `~` on an INTEGER mask (-1 / -2 instead of 1 / 0): reproduced as
    written, so that numbers stay comparable with the reference's."""
    gt, pr = _dev(gt_kps).squeeze(), _dev(pr_kps).squeeze()
    if pr.ndim == 3:
        # one sample (`metrics.py:80-94`): the reference ignores `symmetrical` here and subtracts the two velocity tracks as they
        # are — unequal lengths are its broadcasting error, not a truncation
        if gt.shape != pr.shape:
            raise RuntimeError(f"LVD of one sample needs gt and pr of the same shape, got {tuple(gt.shape)} and {tuple(pr.shape)}")
        symmetrical = False
        pr = pr[None]
    B, T, J, _ = pr.shape
    Tl = min(int(gt.shape[0]), T)
    gt = gt[:Tl].contiguous()
    pr = pr[:, :Tl].contiguous()
    if symmetrical:
        return lvd_symmetric(gt, pr)
    out = torch.empty(3, dtype=torch.float64, device=pr.device)
    _lib.check(_lib.load().ts_eval_body_loss(_ctx(pr), _lib.dptr(gt), _lib.dptr(pr), B, Tl, J, J, Tl, _lib.dptr(out),
                                             _lib.stream_ptr()))
    return (out[0] / (B * (Tl - 1))).to(torch.float32)


def diversity(kps):
    """`evaluation.metrics.diversity` (`metrics.py:96-109`): kps (bs, seq, dim) -> mean over pairs of mean |seq_i - seq_j|."""
    k = _dev(kps)
    bs = k.shape[0]
    if bs < 2:
        return np.float64("nan")                # np.mean of an empty list
    k = k.reshape(bs, -1).contiguous()
    out = torch.empty(1, dtype=torch.float64, device=k.device)
    _lib.check(_lib.load().ts_eval_diversity(_ctx(k), _lib.dptr(k), bs, k.shape[1], _lib.dptr(out), _lib.stream_ptr()))
    return np.float64(out.item() / (k.shape[1] * (bs * (bs - 1) // 2)))      # numpy scalar like the reference's np.mean


def motion_angle_series(joints):
    """Elbow / wrist bend-angle series the beat metrics are built on (`evaluation/FGD.py:67-76,88-94`): joints (T,>=22,3).
    The reference overwrites joints 15..20 with 16..21 in place, forms the bone vectors j[15:21] - j[13:19], and takes
    acos of the clamped inner product of bones two apart (NOT normalised — as written), divided by pi."""
    j = torch.as_tensor(joints, dtype=torch.float32).clone()
    j[:, 15:21] = j[:, 16:22].clone()
    vec = j[:, 15:21] - j[:, 13:19]
    inner = torch.clamp((vec[:, 2:] * vec[:, :-2]).sum(-1), -1, 1)
    return torch.acos(inner) / np.pi                                             # (T, 4)
