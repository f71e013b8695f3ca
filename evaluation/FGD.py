"""Drop-in for `evaluation/FGD.py` of the reference: `EmbeddingSpaceEvaluator(ae, vae, device)`.

Same methods as the reference class (`FGD.py:15-160`): `push_samples(generated_poses, real_poses)` runs both through
`ae.extract` (the HIP feature extractor, nets/body_ae.py); the feature rows stay on the device, mean / covariance are
accumulated there in float64 as they arrive, and only 64 + 64 x 64 numbers per set ever travel to the host;
`get_scores()` -> `(frechet_dist, feat_dist)`; the beat metrics
(`push_joints`, `push_aud`, `get_MAAC`, `get_BCscore`) follow the reference formulae on the (tiny) angle series.
"""
import math

import numpy as np
import torch

from talkshow_amd import evaluation as E

change_angle = torch.tensor([6.0181e-05, 5.1597e-05, 2.1344e-04, 2.1899e-04])        # FGD.py:14


class EmbeddingSpaceEvaluator:
    def __init__(self, ae, vae=None, device=None):
        self.ae = ae
        self.device = device
        self.real_joints_list, self.generated_joints_list, self.audio_beat_list = [], [], []
        self.reset()

    def reset(self):
        self._real = self._gen = None
        self._pairs = []

    def get_no_of_samples(self):
        return len(self._pairs)

    def push_samples(self, generated_poses, real_poses):
        real_feat, _ = self.ae.extract(real_poses)                    # (1, H, 64) for one ground-truth clip
        gen_feat, _ = self.ae.extract(generated_poses)                # (B, H, 64) for B samples of it
        real_rows = real_feat.reshape(-1, real_feat.shape[-1])
        gen_rows = gen_feat.reshape(-1, gen_feat.shape[-1])
        if self._real is None:
            self._real = E.FeatureStats(real_rows.shape[1], real_rows.device)
            self._gen = E.FeatureStats(gen_rows.shape[1], gen_rows.device)
        self._real.push(real_rows)
        self._gen.push(gen_rows)
        # feat_dist pairs row i of ALL real rows stacked with row i of ALL generated rows stacked (FGD.py:153-158) — with B > 1
        # samples per clip that pairing drifts across clips; it is reproduced as written, which needs the rows themselves
        # (kept on the device)
        self._pairs.append((real_rows, gen_rows))

    def push_joints(self, generated_poses, real_poses):
        self.real_joints_list.append(torch.as_tensor(real_poses).detach().cpu())
        self.generated_joints_list.append(torch.as_tensor(generated_poses).squeeze().detach().cpu())

    def push_aud(self, aud):
        self.audio_beat_list.append(torch.as_tensor(aud).squeeze().detach().cpu())

    def get_scores(self):
        mu_g, sig_g = self._gen.mean_cov()
        mu_r, sig_r = self._real.mean_cov()
        try:
            fgd = E.frechet_distance(mu_g, sig_g, mu_r, sig_r)
        except ValueError:
            fgd = 1e+10
        # distance between the i-th real and the i-th generated row of the stacked feature arrays
        real_all = torch.cat([r for r, _ in self._pairs], 0)
        gen_all = torch.cat([g for _, g in self._pairs], 0)
        feat_dist = E.l1_mean_per_row(real_all, gen_all[:real_all.shape[0]])
        return fgd, feat_dist

    # ---- beat metrics on the bend-angle series (small; host) ------------------------------------------------------------
    def get_MAAC(self):
        rows = []
        for joints in self.real_joints_list:
            angle = E.motion_angle_series(joints)
            rows.append((angle[1:] - angle[:-1]).abs().mean(dim=0, keepdim=True))
        return torch.cat(rows, 0).mean(dim=0)

    def get_BCscore(self):
        thres, sigma = 0.01, 0.1
        total, total_beat = 0.0, 0
        for joints, audio_beat_time in zip(self.generated_joints_list, self.audio_beat_list):
            if joints.dim() == 4:
                joints = joints[0]
            angle = E.motion_angle_series(joints)
            ang_vel = (angle[1:] - angle[:-1]).abs() / change_angle / len(change_angle)
            diff = torch.cat((torch.zeros(1, 4), ang_vel), dim=0).numpy()
            T = joints.shape[0]
            mid, prev, nxt = diff[1:T - 1], diff[0:T - 2], diff[2:T]
            is_beat = (mid < prev) & (mid < nxt) & ((prev - mid >= thres) | (nxt - mid >= thres))     # local minima of the series
            beats = np.atleast_1d(np.asarray(audio_beat_time, dtype=np.float64))
            for i in range(diff.shape[1]):
                times = (np.nonzero(is_beat[:, i])[0] + 1) / 30.0
                if times.size == 0:
                    continue
                gaps = ((beats[:, None] - times[None, :]) ** 2).min(axis=1)
                total += float(np.exp(-gaps / (2 * sigma * sigma)).sum())
                total_beat += len(beats)
        return total / total_beat
