#!/usr/bin/env python3
"""CU partitioning probe: conv stacks (VQ encode + decode of a batch) confined to CUs [NC, 256) on SV streams, PixelCNN
chains of other batches on SC unrestricted streams.  Independent work on both sides; SC + SV <= 4 (more than four busy
streams collapse on this stack)."""
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
ctx = _lib.context(0)

def run(NC, SC, SV, n=24):
    cs = _lib.create_streams(SC, 0)
    vs = _lib.create_streams(SV, 0, cus=(NC, 256 - NC) if NC else None)
    codes = [torch.empty((B, H, 2), dtype=torch.int64, device=dev) for _ in range(SV)]
    recon = [torch.empty((B, T, 129), dtype=torch.float32, device=dev) for _ in range(SV)]
    def chain(k):
        with torch.cuda.stream(cs[k % SC]):
            w.generator.run(ids, feat, mode=_lib.TS_SAMPLE_GREEDY)
    def conv(k):
        with torch.cuda.stream(vs[k % SV]):
            _lib.check(lib.ts_body_vq_infer(w.g_body.handle(), w.g_hand.handle(), _lib.dptr(gt), B, T,
                                            _lib.dptr(codes[k % SV]), _lib.dptr(recon[k % SV]), _lib.stream_ptr()))
    def t(fns):
        for k in range(max(SC, SV)):
            for f in fns: f(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(n):
            for f in fns: f(k)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    a, b, c = t([chain]), t([conv]), t([conv, chain])
    print(f"conv on CUs [{NC},256) x{SV} streams, {SC} chain streams: chain {a:.2f}  conv {b:.2f}  both {c:.2f} ms/batch", flush=True)
    torch.cuda.synchronize()
    for s_ in cs + vs:
        lib.ts_stream_destroy(ctx, s_.cuda_stream)

for cfg in [(0, 3, 1), (64, 3, 1), (96, 3, 1), (32, 3, 1), (128, 3, 1)]:
    run(*cfg)
