#!/usr/bin/env python3
"""Do the conv stacks of one pass run UNDER the PixelCNN chain of another?  Stream A: N chains of 256 clips; stream B: N VQ encode +
decode passes of 256 clips; timed alone and together, then the start / end of each on a common clock.  Run with TS_CHAIN_PERSIST=0
(launch graph) and with TS_CHAIN_PERSIST=3 (the lean persistent chain kernel: 168 VGPRs and 35 KB of LDS leave room for a conv_gemm
workgroup on every CU).

    TS_CHAIN_PERSIST=3 python tools/corun_probe.py
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from talkshow_amd import _lib, synth
lib = _lib.load(); w, _ = bench.build_models(0)
B, T, N = int(os.environ.get("TS_B", "256")), 300, int(os.environ.get("TS_N", "6"))
dev = torch.device("cuda", 0)
mf = torch.from_numpy(synth.mfcc_features(1, B, T)).to(dev); ids = torch.from_numpy(synth.speaker_ids(B)).to(dev)
gt = torch.from_numpy(synth.gt_poses(2, B, T)).to(dev)
codes = torch.empty((B, 75, 2), dtype=torch.int64, device=dev); recon = torch.empty((B, T, 129), device=dev)
sa, sb = _lib.create_streams(2, 0)
feat = w.audioencoder.forward_nlc(mf)
torch.cuda.synchronize()
def chain():
    with torch.cuda.stream(sa): w.generator.run(ids, feat, mode=_lib.TS_SAMPLE_GREEDY)
def conv():
    _lib.check(lib.ts_body_vq_infer(w.g_body.handle(), w.g_hand.handle(), _lib.dptr(gt), B, T, _lib.dptr(codes), _lib.dptr(recon), sb.cuda_stream))
def t(fns, n=N):
    for f in fns: f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for f in fns: f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
a, b, c = t([chain]), t([conv]), t([chain, conv])
print(f"TS_CHAIN_PERSIST={os.environ.get('TS_CHAIN_PERSIST', '0')}: chain alone {a:.2f} ms, conv stacks alone {b:.2f} ms, "
      f"one of each together {c:.2f} ms (serial {a + b:.2f}, perfect overlap {max(a, b):.2f})")
# timeline of the together case: start / end of every chain (stream A) and conv pass (stream B) against a common origin
ev = lambda s: (e := torch.cuda.Event(enable_timing=True), e.record(s))[0]
torch.cuda.synchronize()
origin = ev(sa); sb.wait_event(origin); o2 = ev(sb)
marks = []
for k in range(3):
    a0 = ev(sa); chain(); a1 = ev(sa)
    b0 = ev(sb); conv(); b1 = ev(sb)
    marks.append((a0, a1, b0, b1))
torch.cuda.synchronize()
for k, (a0, a1, b0, b1) in enumerate(marks):
    print(f"   iteration {k}: chain {origin.elapsed_time(a0):7.2f} -> {origin.elapsed_time(a1):7.2f} ms   conv {origin.elapsed_time(b0):7.2f} -> {origin.elapsed_time(b1):7.2f} ms")
