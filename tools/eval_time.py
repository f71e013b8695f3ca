#!/usr/bin/env python3
"""Evaluation reductions (csrc/eval.hip) at evaluation-run sizes: time per call and the HBM rate it implies (they are
streaming reductions: algorithmic bytes = the inputs, read once).
    python tools/eval_time.py > gpurun_out/eval_time.json"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from talkshow_amd import evaluation as E

def timed(f, reps=7):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))

g = torch.Generator(device="cuda").manual_seed(0)
out = {}
n = 1 << 20                                                   # FGD feature rows of a large test set (64-d)
feats = torch.randn(n, 64, device="cuda", generator=g)
fs = E.FeatureStats(64)
ms = timed(lambda: fs.push(feats))
out["feat_stats_1Mx64"] = {"ms": ms, "GBps": feats.numel() * 4 / ms / 1e6, "rows_per_s": n / ms * 1e3}
a, b = torch.randn(n, 64, device="cuda", generator=g), torch.randn(n, 64, device="cuda", generator=g)
ms = timed(lambda: E.l1_mean_per_row(a, b))
out["l1_1Mx64"] = {"ms": ms, "GBps": 2 * a.numel() * 4 / ms / 1e6}
B, T, J = 64, 300, 127                                        # test_body.py: 64 samples of a 10 s clip, SMPL-X joints
gt, prs = torch.randn(T, J, 3, device="cuda", generator=g), torch.randn(B, T, J, 3, device="cuda", generator=g)
ms = timed(lambda: E.body_loss(gt, prs))
out["body_loss_64x300x127"] = {"ms": ms, "GBps": prs.numel() * 4 * 2 / ms / 1e6}   # prs is read twice (error/LVD pass, variance pass)
kps = torch.randn(64, 300, 165, device="cuda", generator=g)
ms = timed(lambda: E.diversity(kps))
out["diversity_64x300x165"] = {"ms": ms, "GBps": 64 * 63 / 2 * 2 * 300 * 165 * 4 / ms / 1e6, "pairs": 2016}
print(json.dumps(out))
