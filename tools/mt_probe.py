#!/usr/bin/env python3
"""Does multi-threaded submission lift the 4-stream ceiling?  T host threads, one HIP stream each, each thread enqueues
whole bench steps (VQ encode + generate) on its stream; ctypes releases the GIL inside the C calls."""
import os, sys, time, threading
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
SMAX = 8
streams = _lib.create_streams(SMAX, 0)
codes = [torch.empty((B, H, 2), dtype=torch.int64, device=dev) for _ in range(SMAX)]

def step(k):
    with torch.cuda.stream(streams[k]):
        _lib.check(lib.ts_body_vq_infer(w.g_body.handle(), w.g_hand.handle(), _lib.dptr(gt), B, T, _lib.dptr(codes[k]), None, _lib.stream_ptr()))
        w.generate_batch(mfcc, ids, mode=_lib.TS_SAMPLE_GREEDY)

for k in range(SMAX): step(k)
torch.cuda.synchronize()

def run(nthreads, per_thread):
    def worker(k):
        torch.cuda.set_device(0)
        for _ in range(per_thread): step(k)
        streams[k].synchronize()
    ths = [threading.Thread(target=worker, args=(k,)) for k in range(nthreads)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = nthreads * per_thread
    print(f"{nthreads} threads x 1 stream: {dt / n * 1e3:.2f} ms/step  {n * B * 300 / dt:.0f} frames/s", flush=True)

for nt in (1, 2, 4, 5, 6, 8):
    run(nt, 6)
# single thread feeding 4 streams round-robin, for reference
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(24): step(i % 4)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"1 thread x 4 streams: {dt / 24 * 1e3:.2f} ms/step  {24 * B * 300 / dt:.0f} frames/s")
