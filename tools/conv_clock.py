#!/usr/bin/env python3
"""What shader clock does the chip hold under the conv stacks / the chain?  A one-wave sampler kernel on a side stream records
(wall ticks, shader cycles) every 200 us while the main stream runs VQ encode + decode passes (conv_gemm_f32) and then PixelCNN
chains; prints the clock per phase and the conv stacks' TFLOP/s against the peak AT THAT CLOCK.

    TS_B=256 python tools/conv_clock.py
"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from talkshow_amd import _lib, synth

lib = _lib.load(); w, _ = bench.build_models(0)
B, T = int(os.environ.get("TS_B", "256")), 300
dev = torch.device("cuda", 0)
mfcc = torch.from_numpy(synth.mfcc_features(1000, B, T)).to(dev); gt = torch.from_numpy(synth.gt_poses(2000, B, T)).to(dev)
ids = torch.from_numpy(synth.speaker_ids(B)).to(dev)
codes = torch.empty((B, 75, 2), dtype=torch.int64, device=dev); recon = torch.empty((B, T, 129), device=dev)
main, side = _lib.create_streams(2, 0)
ctx = _lib.context(0)
def conv_pass():
    _lib.check(lib.ts_body_vq_infer(w.g_body.handle(), w.g_hand.handle(), _lib.dptr(gt), B, T, _lib.dptr(codes), _lib.dptr(recon), main.cuda_stream))
feat = w.audioencoder.forward_nlc(mfcc)
def chain_pass():
    with torch.cuda.stream(main):
        w.generator.run(ids, feat, mode=_lib.TS_SAMPLE_GREEDY)
conv_pass(); chain_pass(); torch.cuda.synchronize()

WIN, N = 200, 1500          # 200 us windows, 300 ms
buf = torch.zeros(3 * N, dtype=torch.int64, device=dev)
_lib.check(lib.ts_debug_clock_sample(_lib.dptr(buf), N, WIN, side.cuda_stream))
marks = []
ev = lambda: (e := torch.cuda.Event(enable_timing=True), e.record(main))[0]
time.sleep(0.02)            # idle head
t_host0 = time.perf_counter()
e0 = ev()
phases = []
for name, fn, reps in (("conv", conv_pass, 3), ("chain", chain_pass, 2), ("conv after chain", conv_pass, 2)):
    a = ev()
    for _ in range(reps): fn()
    b = ev()
    phases.append((name, a, b, reps))
torch.cuda.synchronize()
r = buf.cpu().numpy().reshape(N, 3).astype(np.float64)
t_ms = r[:, 0] / 1e5; ghz = r[:, 2] / (r[:, 1] * 10.0)       # cycles per (ticks * 10 ns) = GHz
print(f"sampler: {N} windows of {WIN} us; shader clock min {ghz.min():.3f} median {np.median(ghz):.3f} max {ghz.max():.3f} GHz")
# the sampler's time origin is its own start; find the load's start as the first drop of the clock below 98 % of the idle head
idle = np.median(ghz[:50])
print(f"idle head: {idle:.3f} GHz")
# phases by event times relative to e0; e0's offset on the sampler's axis = first window where the clock leaves the idle band
dropped = np.nonzero(np.abs(ghz - idle) > 0.02 * idle)[0]
off = t_ms[dropped[0]] if len(dropped) else 0.0
print(f"load seen from t = {off:.2f} ms on the sampler's axis")
for name, a, b, reps in phases:
    ta, tb = e0.elapsed_time(a) + off, e0.elapsed_time(b) + off
    sel = (t_ms >= ta + 0.3) & (t_ms <= tb - 0.1)
    g = ghz[sel]
    line = f"{name:18s} {tb - ta:7.2f} ms ({(tb - ta) / reps:.2f} per pass): clock median {np.median(g):.3f} GHz  p10 {np.percentile(g, 10):.3f}  p90 {np.percentile(g, 90):.3f}"
    print(line)
if os.environ.get("TS_CLOCK_DUMP"):
    for i in range(0, N, 5): print(f"{t_ms[i]:8.2f} ms {ghz[i]:.3f}")
