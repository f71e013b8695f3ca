#!/usr/bin/env python3
"""Batched SMPL-X forward (csrc/smplx.*) at the real mesh size on random model parameters: time per call and the rates it
implies.  rows = 32 clips x 300 frames (BASELINE configs[1]'s output), joints only and joints + vertices.
    python tools/smplx_time.py > gpurun_out/smplx_time.json"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from talkshow_amd.smplx_lbs import SMPLXLayer

V, J, NB, NE, NX, NL = 10475, 55, 300, 100, 21, 51
rng = np.random.default_rng(0)
parents = np.asarray([-1] + [max(0, j - 1 - (j % 3)) for j in range(1, J)], np.int32)     # any tree with parent < child
lbs = np.zeros((V, J), np.float32)
own = rng.integers(0, J, (V, 4))
np.put_along_axis(lbs, own, rng.random((V, 4)).astype(np.float32) + 0.1, 1)
lbs /= lbs.sum(1, keepdims=True)
Jr = np.zeros((J, V), np.float32)
for j in range(J):
    idx = rng.choice(V, 12, replace=False); Jr[j, idx] = 1 / 12
model = dict(v_template=rng.standard_normal((V, 3)).astype(np.float32) * 0.3, shapedirs=rng.standard_normal((V, 3, NB + NE)).astype(np.float32) * 4e-3,
             posedirs=rng.standard_normal(((J - 1) * 9, V * 3)).astype(np.float32) * 2e-3, J_regressor=Jr, parents=parents, lbs_weights=lbs,
             pose_mean=np.zeros(J * 3, np.float32), extra_idx=rng.choice(V, NX, replace=False), lmk_faces=rng.integers(0, V, (NL, 3)),
             lmk_bary=np.full((NL, 3), 1 / 3, np.float32), n_betas=NB, n_expr=NE)
N = int(os.environ.get("TS_FRAMES", str(32 * 300)))
rows = torch.from_numpy((rng.standard_normal((N, 265)) * 0.3).astype(np.float32)).cuda()
betas = torch.from_numpy((rng.standard_normal(NB) * 0.8).astype(np.float32)).cuda()
out = {"frames": N, "V": V}
for name, wv in (("joints_only", False), ("joints_and_vertices", True)):
    layer = SMPLXLayer(model, with_vertices=wv)
    f = (lambda: layer.vertices(betas, rows)) if wv else (lambda: layer.joints(betas, rows))
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    ms = float(np.median(ts))
    K = NB + NE + (J - 1) * 9
    r = {"ms": ms, "frames_per_s": N / ms * 1e3}
    if wv:   # blend shapes for every vertex coordinate: N x K x 3V MACs; then one read of the posed mesh + one write of the skinned one
        r["blend_shape_TFLOPs"] = 2.0 * N * K * 3 * V / (ms * 1e-3) / 1e12
        r["vertex_bytes_GB"] = N * V * 3 * 4 * 2 / 1e9
    out[name] = r
    del layer
print(json.dumps(out))
