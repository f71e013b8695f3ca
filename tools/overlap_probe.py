#!/usr/bin/env python3
"""How well do the latency-bound PixelCNN chain and the MFMA-bound conv stacks of different batches overlap?

Times, for S = 1, 2, 4 streams: (a) chain only, (b) conv only (VQ encode + decode), (c) both back to back per stream.
"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from talkshow_amd import _lib, synth

lib = _lib.load()
w, _ = bench.build_models(0)
B, T, H = 32, 300, 75
dev = torch.device("cuda", 0)
mfcc = torch.from_numpy(synth.mfcc_features(1000, B, T)).to(dev)
gt = torch.from_numpy(synth.gt_poses(2000, B, T)).to(dev)
ids = torch.from_numpy(synth.speaker_ids(B)).to(dev)
feat = w.audioencoder.forward_nlc(mfcc)
torch.cuda.synchronize()
SMAX = int(os.environ.get("SMAX", "4"))
streams = _lib.create_streams(SMAX, 0)
codes = [torch.empty((B, H, 2), dtype=torch.int64, device=dev) for _ in range(SMAX)]
recon = [torch.empty((B, T, 129), dtype=torch.float32, device=dev) for _ in range(SMAX)]

def chain(k, S):
    with torch.cuda.stream(streams[k % S]):
        w.generator.run(ids, feat, mode=_lib.TS_SAMPLE_GREEDY)

def conv(k, S):
    with torch.cuda.stream(streams[k % S]):
        _lib.check(lib.ts_body_vq_infer(w.g_body.handle(), w.g_hand.handle(), _lib.dptr(gt), B, T,
                                        _lib.dptr(codes[k % S]), _lib.dptr(recon[k % S]), _lib.stream_ptr()))

def both(k, S):
    conv(k, S); chain(k, S)

def timeit(fn, S, n=16):
    for k in range(SMAX): fn(k, S)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n): fn(k, S)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for S in (1, 2, 4) if SMAX <= 4 else (1, 2, 4, SMAX):
    a, b, c = timeit(chain, S), timeit(conv, S), timeit(both, S)
    print(f"S={S}: chain {a:.2f} ms/batch  conv {b:.2f} ms/batch  both {c:.2f} ms/batch  (sum {a+b:.2f})", flush=True)
