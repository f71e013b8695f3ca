#!/usr/bin/env python3
"""PCIe-inclusive rate of the body path: every step copies its (32,300,64) features and (32,300,129) GT poses from pinned
host memory and brings the generated poses back to the host, 4 batches in flight (same steps as bench.py otherwise)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from talkshow_amd import _lib, synth
lib = _lib.load()
w, _ = bench.build_models(0)
B, T, H, S = 32, 300, 75, 4
dev = torch.device("cuda", 0)
h_mfcc = torch.from_numpy(synth.mfcc_features(1000, B, T)).pin_memory()
h_gt = torch.from_numpy(synth.gt_poses(2000, B, T)).pin_memory()
ids = torch.from_numpy(synth.speaker_ids(B)).to(dev)
streams = _lib.create_streams(S, 0)
h_out = [torch.empty((B, 4 * H, 129), dtype=torch.float32).pin_memory() for _ in range(S)]
codes = [torch.empty((B, H, 2), dtype=torch.int64, device=dev) for _ in range(S)]
def step(k, pcie):
    with torch.cuda.stream(streams[k % S]):
        mf = h_mfcc.to(dev, non_blocking=True) if pcie else d_mfcc
        gt = h_gt.to(dev, non_blocking=True) if pcie else d_gt
        _lib.check(lib.ts_body_vq_infer(w.g_body.handle(), w.g_hand.handle(), _lib.dptr(gt), B, T, _lib.dptr(codes[k % S]), None, _lib.stream_ptr()))
        _, poses = w.generate_batch(mf, ids, mode=_lib.TS_SAMPLE_GREEDY)
        if pcie:
            h_out[k % S].copy_(poses, non_blocking=True)
d_mfcc, d_gt = h_mfcc.to(dev), h_gt.to(dev)
for pcie in (False, True):
    for k in range(S): step(k, pcie)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 32
    for k in range(n): step(k, pcie)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{'PCIe-inclusive' if pcie else 'resident inputs'}: {dt / n * 1e3:.2f} ms/step  {n * B * 300 / dt:.0f} frames/s")
