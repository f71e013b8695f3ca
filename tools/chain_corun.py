#!/usr/bin/env python3
"""How well do the PixelCNN chains of DIFFERENT passes overlap?  K chains of TS_B clips (default 256) on K streams, K = 1, 2, 3:
wall time per round of K chains, against K x the single-chain time.  Run once per kernel selection, e.g.

    python tools/chain_corun.py                       # default: wide kernel for launches of >= 160 workgroups
    TS_SKINNY_WIDE_MIN=0 python tools/chain_corun.py  # split-K kernels only (several workgroups per CU can be resident)

The 3-stream bench gains from chain || chain and conv || conv, not from conv under chain: this isolates the first.
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from talkshow_amd import _lib, synth

lib = _lib.load(); w, _ = bench.build_models(0)
B, T, N = int(os.environ.get("TS_B", "256")), 300, int(os.environ.get("TS_N", "4"))
dev = torch.device("cuda", 0)
streams = _lib.create_streams(3, 0)
feats, ids = [], torch.from_numpy(synth.speaker_ids(B)).to(dev)
for k in range(3):
    mf = torch.from_numpy(synth.mfcc_features(1 + k, B, T)).to(dev)
    feats.append(w.audioencoder.forward_nlc(mf))
torch.cuda.synchronize()

def chain(k):
    with torch.cuda.stream(streams[k]):
        w.generator.run(ids, feats[k], mode=_lib.TS_SAMPLE_GREEDY)

def t(K):
    for k in range(K): chain(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        for k in range(K): chain(k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e3

one = t(1)
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("TS_SKINNY") or k == "TS_B") or "default"
for K in (1, 2, 3):
    r = t(K)
    print(f"[{tag}] {K} chain(s) of {B} clips in flight: {r:.2f} ms per round = {r / K:.2f} ms per chain ({K * one / r:.2f}x the serial rate)")
