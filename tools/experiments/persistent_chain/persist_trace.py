#!/usr/bin/env python3
"""Per-stage timeline of the persistent chain kernel (TS_CHAIN_TRACE=1): workgroup 0 of XCD 0 stamps the 100 MHz wall clock at stage
start, first operands issued, first tile done, all tiles done, prefetch issued, barrier passed.  Prints the median phase lengths per
stage kind (sampler / skinny batch by tile count).

    TS_B=256 python tools/persist_trace.py
"""
import os, sys
os.environ["TS_CHAIN_TRACE"] = "1"
os.environ.setdefault("TS_CHAIN_PERSIST", "1")
import ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from talkshow_amd import _lib, synth
lib = _lib.load(); w, _ = bench.build_models(0)
B, T = int(os.environ.get("TS_B", "256")), 300
mf = torch.from_numpy(synth.mfcc_features(1, B, T)).cuda(); ids = torch.from_numpy(synth.speaker_ids(B)).cuda()
feat = w.audioencoder.forward_nlc(mf)
for _ in range(3): w.generator.run(ids, feat, mode=_lib.TS_SAMPLE_GREEDY)
torch.cuda.synchronize()
N = 4096
buf = (C.c_uint64 * (N * 8))()
n = lib.ts_debug_chain_trace(buf, N)
r = np.frombuffer(buf, dtype=np.uint64)[: n * 8].reshape(n, 8).astype(np.int64)
t = r[:, :6] * 0.01
kind = r[:, 6] >> 32; tiles = r[:, 6] & 0xffffffff
print(f"{n} stages, kernel body {t[-1, 5] - t[0, 0]:.1f} us")
rows = {}
for i in range(n):
    key = ("sampler", 0) if kind[i] == 1 else ("gemm", int(tiles[i]))
    if kind[i] == 1:
        ph = [0, 0, t[i, 3] - t[i, 0], t[i, 4] - t[i, 3], t[i, 5] - t[i, 4]]
    else:
        ph = [t[i, 1] - t[i, 0], t[i, 2] - t[i, 1], t[i, 3] - t[i, 2], t[i, 4] - t[i, 3], t[i, 5] - t[i, 4]]
    rows.setdefault(key, []).append(ph + [t[i, 5] - t[i, 0]])
print("kind   tiles/XCD  stages | issue A  first tile  other tiles  prefetch  barrier | stage total (median us) | share of the kernel")
tot = t[-1, 5] - t[0, 0]
for key, v in sorted(rows.items(), key=lambda kv: -len(kv[1]) * np.median(np.array(kv[1])[:, 5])):
    a = np.array(v); m = np.median(a, axis=0)
    print(f"{key[0]:8s} {key[1]:6d} {len(v):7d} | {m[0]:6.2f} {m[1]:9.2f} {m[2]:11.2f} {m[3]:9.2f} {m[4]:8.2f} | {m[5]:8.2f} | {100 * a[:, 5].sum() / tot:5.1f} %")
