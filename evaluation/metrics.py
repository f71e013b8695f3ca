"""Drop-in for `evaluation/metrics.py` of the reference: `LVD`, `diversity`, `data_driven_baselines` (same signatures);
the sums over frames / joints / pairs run on the GPU (talkshow_amd/evaluation.py)."""
import numpy as np

from talkshow_amd import evaluation as E


def data_driven_baselines(gt_kps):
    '''gt_kps: (T, D) numpy -> (last_step, mean) baselines of `metrics.py:13-25` (host: O(T*D) on one clip)'''
    vel = np.abs(gt_kps[1:] - gt_kps[:-1])
    mean = np.mean(np.abs(vel - np.mean(vel, axis=0)[np.newaxis]))
    last_step = np.mean(np.abs(vel - (gt_kps[1] - gt_kps[0])[np.newaxis]))
    return last_step, mean


def LVD(gt_kps, pr_kps, symmetrical=False, weight=False):
    '''-> 0-d tensor like the reference (`metrics.py:27-94`).  weight=True is not offered: the reference draws its frame weights
    with `.normal_()` on the velocity sums — fresh random numbers on every call, not a function of the inputs (`metrics.py:67-68`)'''
    if weight:
        raise NotImplementedError("weight=True draws random frame weights in the reference (metrics.py:67-68): no defined value")
    return E.lvd(gt_kps, pr_kps, symmetrical=symmetrical)


def diversity(kps):
    '''kps: (bs, seq, dim)'''
    return E.diversity(kps)
